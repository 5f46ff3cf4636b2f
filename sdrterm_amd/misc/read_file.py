"""Reader for the GPU processors: the producer side of the sdrterm pipeline.

Counterpart of the reference's src/misc/read_file.py (readFile :31-174, generateDomain :177-196) with the same call
signature, so ``readFile(swapEndianness=..., **{**ioArgs.strct, **ioArgs.strct['fileInfo']})`` (sdrterm.py:145-146) keeps
working.  The difference is WHAT travels through the Queues: the reference unpacks every 131072-byte read to complex128,
normalises and IQ-corrects it on the CPU (feedBuffers :100-103) and ships 16 bytes per sample; this reader ships the raw
bytes plus a description (``RawChunk``) and the GPU processors do those per-sample steps themselves
(include/sdrterm_hip.h: sdrt_process_*), carrying the IQ offset on the device.

Kept from the reference because outputs depend on it: reads go into ONE reusable ``readSize`` buffer and every read,
however short, is handed on as a full buffer, so a short final read is followed by the stale tail of the previous one
(read_file.py:119-125); EOF is signalled with ``b''`` on every Queue (:164-166); consumers whose process ended or whose
Queue is closed are dropped (:104-117); the byte order flips with swapEndianness (:48-49).
"""
import socket
import sys
from io import BufferedReader

import numpy as np

from .general_util import applyIgnoreException, eprint, shutdownSocket, tprint, vprint

_DOMAINS = {'B': (0, 255), 'h': (-32768, 32767), 'b': (-128, 127), 'i': (-2147483648, 2147483647), 'H': (0, 65536),
            'I': (0, 4294967295), 'L': (0, 18446744073709551615), 'l': (-9223372036854775808, 9223372036854775807)}


def generateDomain(dataType: str):
    """(xmin, 1/(xmax - xmin)) of an integer encoding, None for the others (read_file.py:177-196; 'H' really is 65536 there)."""
    if dataType not in _DOMAINS:
        return None
    lo, hi = _DOMAINS[dataType]
    return lo, 1 / (-lo + hi)


class RawChunk:
    """One read of the input as the reader saw it: bytes + what they mean.  Picklable (it crosses a multiprocessing Queue)."""
    __slots__ = ('data', 'code', 'bigEndian', 'normalize', 'correctIq', 'impedance')

    def __init__(self, data: bytes, code: str, bigEndian: bool, normalize: bool, correctIq: bool, impedance: float):
        self.data, self.code, self.bigEndian = data, code, bigEndian
        self.normalize, self.correctIq, self.impedance = normalize, correctIq, impedance

    @property
    def itemsize(self) -> int:
        return np.dtype(self.code).itemsize

    @property
    def size(self) -> int:
        """Complex samples in the chunk (what ``z.size`` is for the reference's Queue items)."""
        return len(self.data) // (2 * self.itemsize)

    def __getstate__(self):
        return tuple(getattr(self, k) for k in self.__slots__)

    def __setstate__(self, state):
        for k, v in zip(self.__slots__, state):
            setattr(self, k, v)


class _Consumers:
    """The (process, queue) pairs still being fed."""

    def __init__(self, processes, queues):
        self.pairs = list(zip(list(processes), list(queues)))

    def offer(self, item) -> None:
        for pair in list(self.pairs):
            proc, q = pair
            if proc.exitcode is not None:
                tprint(f'Process : {proc.name} ended; removing {q} from queue')
                q.close()
                self.pairs.remove(pair)
                continue
            try:
                q.put_nowait(item)
            except ValueError:
                tprint(f'Client : {q} closed; removing {proc.name} from queue')
                q.close()
                self.pairs.remove(pair)


def readFile(bitsPerSample: np.dtype = None, dataOffset: int = 0, fs: int = None, buffers=None, processes=None, isDead=None,
             inFile: str = None, readSize: int = 131072, swapEndianness: bool = False, correctIq: bool = False,
             normalize: bool = False, isSocket: bool = False, impedance: int = 50, **_) -> None:
    if fs is None:
        raise ValueError('fs is not specified')
    sample = np.dtype(bitsPerSample)
    if swapEndianness:
        sample = sample.newbyteorder('<' if sample.byteorder == '>' else '>')
    order = sample.byteorder
    big = order == '>' or (order == '=' and sys.byteorder == 'big')
    do_normalize = bool(normalize) and generateDomain(sample.char) is not None
    if do_normalize:
        tprint('Exact input being normalized to interval [-0.8, 0.8]')
    queues = list(buffers)
    consumers = _Consumers(processes, queues)
    scratch = bytearray(readSize)

    def pump(reader) -> None:
        while not isDead.value:
            if not reader.readinto(scratch):
                break
            consumers.offer(RawChunk(bytes(scratch), sample.char, big, do_normalize, bool(correctIq), impedance))

    def from_descriptor() -> None:
        named = inFile is not None
        with open(inFile if named else sys.stdin.fileno(), 'rb', closefd=named) as stream:
            reader = BufferedReader(stream if named else sys.stdin.buffer)
            if dataOffset and stream.seekable():
                stream.seek(dataOffset)
            pump(reader)

    def from_socket() -> None:
        attempts, limit = 0, 5
        while attempts < limit and not isDead.value:
            conn = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                conn.setsockopt(socket.SOL_SOCKET, socket.SO_KEEPALIVE, 1)
                conn.setsockopt(socket.SOL_SOCKET, getattr(socket, 'SO_REUSEPORT', socket.SO_REUSEADDR), 1)
                conn.settimeout(5)
                host, port = inFile.split(':')
                conn.connect((host, int(port)))
                tprint(f'Connected to {conn.getpeername()}')
                attempts = 0
                with conn.makefile('rb') as reader:
                    pump(reader)
            except (TimeoutError, ConnectionError, socket.gaierror) as e:
                attempts += 1
                eprint(f'Connection failed: {e}. Retrying {attempts} of {limit} times')
            finally:
                shutdownSocket(conn)
                conn.close()

    (from_socket if isSocket else from_descriptor)()
    for q in queues:
        applyIgnoreException(q.put_nowait, b'')
        q.close()
    vprint('File reader halted')
