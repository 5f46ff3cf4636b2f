"""What the input is made of: WAV (RIFF / RIFX) header or headerless PCM.

Counterpart of the reference's src/misc/file_util.py (checkWavHeader :118-195, parseRawType :92-104, DataType :44-89): the
same entry points, result keys and error types, so IOArgs-style callers work unchanged.  The parser itself is written as
a little-endian/big-endian field cursor walking the RIFF chunks.

One deliberate replication: the reference reports ``dataOffset`` as the position right AFTER the ``data`` tag, i.e. at the
4-byte chunk-size field (file_util.py:185-194), so its reader interprets those four bytes as the first IQ record(s).
``checkWavHeader`` keeps that by default so that outputs stay identical; ``strictDataOffset=True`` skips the size field.
"""
import struct
from enum import Enum

import numpy as np


class MappableEnum(Enum):
    @classmethod
    def dict(cls):
        return {member.name: member.value for member in cls}

    @classmethod
    def tuples(cls):
        return tuple(cls.dict().items())


class WaveFormat(Enum):
    WAVE_FORMAT_PCM = 0x0001
    WAVE_FORMAT_IEEE_FLOAT = 0x0003
    WAVE_FORMAT_ALAW = 0x0006
    WAVE_FORMAT_MULAW = 0x0007
    WAVE_FORMAT_EXTENSIBLE = 0xFFFE


class ExWaveFormat(Enum):
    PCM_S_LE = 0x01
    PCM_S_BE = 0x02
    PCM_U_LE = 0x05
    PCM_U_BE = 0x06


class DataType(MappableEnum):
    """--encoding names -> numpy dtype of one component (file_util.py:44-56)."""
    b = np.dtype('|b')
    B = np.dtype('|B')
    h = np.dtype('=h')
    H = np.dtype('=H')
    i = np.dtype('=i4')
    I = np.dtype('=u4')  # noqa: E741
    f = np.dtype('=f4')
    d = np.dtype('=f8')

    def __str__(self):
        return self.name

    @classmethod
    def fromWav(cls, bits: int, aFormat: Enum, bFormat: Enum, isRifx: bool) -> np.dtype:
        """Sample dtype of a WAV stream (file_util.py:58-89): float formats by width, PCM by width and the extensible
        sub-format's signedness (8-bit plain PCM is unsigned, 16/32-bit signed); byte order from RIFX or a *_BE sub-format."""
        signed_by_width = {8: {'S': cls.b, 'U': cls.B, None: cls.B},
                           16: {'S': cls.h, 'U': cls.H, None: cls.h},
                           32: {'S': cls.i, 'U': cls.I, None: cls.i}}
        big = bool(isRifx)
        found = None
        if aFormat == WaveFormat.WAVE_FORMAT_IEEE_FLOAT:
            found = {32: cls.f, 64: cls.d}.get(bits)
        elif aFormat in (WaveFormat.WAVE_FORMAT_PCM, WaveFormat.WAVE_FORMAT_EXTENSIBLE):
            sign = None
            if isinstance(bFormat, ExWaveFormat):
                _, sign, order = bFormat.name.split('_')
                big = big or order == 'BE'
            found = signed_by_width.get(bits, {}).get(sign)
        if found is None:
            raise ValueError(f'Unsupported format: {aFormat} @ {bits} bits')
        return found.value.newbyteorder('>' if big else '<')


_FIELDS = ('subchunk1Size', 'audioFormat', 'numChannels', 'sampRate', 'byteRate', 'blockAlign', 'bitsPerSample')


def zipRet(x) -> dict:
    return dict(zip(_FIELDS, x))


def parseRawType(file, fs, enc, isSocket: bool = False) -> dict:
    """Headerless PCM: everything comes from --fs and --encoding (file_util.py:92-104); host:port inputs are network
    byte order."""
    if fs is None or fs < 1 or enc is None:
        raise ValueError('Valid sampling rate, encoding type and bit-size are all required for raw pcm input')
    fs = int(fs)
    sample = DataType[enc].value
    if isinstance(file, str) and ':' in file:
        sample = sample.newbyteorder('>')
    info = zipRet((0, 0, 0, fs, fs, 0, sample))
    info['dataOffset'] = 0
    info['isSocket'] = isSocket
    return info


class _Cursor:
    """Sequential typed reads from a binary stream in one byte order."""

    def __init__(self, stream, order: str = '<'):
        self.stream = stream
        self.order = order

    def take(self, count: int, what: str) -> bytes:
        got = self.stream.read(count)
        if len(got) != count:
            raise ValueError(f'Invalid: truncated header while reading {what}')
        return got

    def fields(self, fmt: str, what: str):
        return struct.unpack(self.order + fmt, self.take(struct.calcsize(self.order + fmt), what))


_GUID_TAIL = bytes.fromhex('000000001000800000aa00389b71')      # the fixed part of a KSDATAFORMAT_SUBTYPE_* GUID


def _is_network_address(name: str) -> bool:
    from socket import getaddrinfo, gaierror
    host, _, port = name.partition(':')
    try:
        return bool(getaddrinfo(host, port))
    except (gaierror, UnicodeError, OverflowError):
        return False


def checkWavHeader(f, fs, enc, strictDataOffset: bool = False) -> dict:
    """Describes the input ``f`` (file_util.py:118-195): None -> raw PCM on stdin; 'host:port' that resolves -> raw PCM from
    a socket; a file starting with RIFF/RIFX -> its fmt chunk; any other file -> raw PCM (ValueError if it is named *.wav)."""
    if f is None:
        return parseRawType(f, fs, enc)
    if isinstance(f, str) and ':' in f and _is_network_address(f):
        return parseRawType(f, fs, enc, True)

    with open(f, 'rb') as stream:
        magic = stream.read(4)
        if magic[:3] != b'RIF':
            if isinstance(f, str) and '.wav' in f:
                raise ValueError('Invalid: Expected raw pcm file, but got malformed RIFF header')
            return parseRawType(f if isinstance(f, str) else None, fs, enc)
        if magic not in (b'RIFF', b'RIFX'):
            raise ValueError('Invalid: Malformed RIFF/X header')
        cur = _Cursor(stream, '>' if magic == b'RIFX' else '<')
        cur.fields('I', 'the RIFF chunk size')
        if cur.take(4, 'the form type') != b'WAVE':
            raise ValueError('Invalid: Expected a wave file')
        if cur.take(4, 'the first chunk tag') != b'fmt ':
            raise ValueError('Invalid: Format section not found')
        fmt = cur.fields('IHHIIHH', 'the fmt chunk')
        info = zipRet(fmt)
        info['isSocket'] = False
        bits, byte_rate, block_align = info['bitsPerSample'], info['byteRate'], info['blockAlign']
        consumed = 16                                     # bytes of the fmt chunk body read so far
        sub = None
        if info['audioFormat'] == WaveFormat.WAVE_FORMAT_EXTENSIBLE.value:
            ext_size, = cur.fields('H', 'the extension size')
            cur.take(max(ext_size - 16, 0), 'the extension fields')
            sub_code, = cur.fields('H', 'the sub-format code')
            if cur.take(len(_GUID_TAIL), 'the sub-format GUID') != _GUID_TAIL:
                raise ValueError('Invalid: SubFormat GUID malformed')
            sub = ExWaveFormat(sub_code)
            consumed += 2 + max(ext_size - 16, 0) + 2 + len(_GUID_TAIL)
        info['bitsPerSample'] = DataType.fromWav(bits, WaveFormat(info['audioFormat']), sub, cur.order == '>')
        info['bitRate'] = (bits * byte_rate * block_align) >> 3
        # the rest of the fmt body (e.g. the cbSize of a non-PCM format), then chunk by chunk up to 'data'
        stream.read(max(info['subchunk1Size'] - consumed, 0))
        while True:
            tag = stream.read(4)
            if len(tag) < 4:
                raise ValueError('Invalid: no data chunk found')
            if tag == b'data':
                break
            size, = cur.fields('I', f'the size of chunk {tag!r}')
            stream.seek(size + (size & 1), 1)
        info['dataOffset'] = stream.tell() + (4 if strictDataOffset else 0)
    return info
