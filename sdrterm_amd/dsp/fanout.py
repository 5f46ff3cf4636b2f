"""TCP fan-out used by ``--simo``: one listening socket, one connected client per output row.

Wire behaviour follows the reference's server (src/dsp/vfo_processor.py:86-126): processing does not start until
as many clients as there are rows have connected; the k-th client to be registered receives row k of every chunk
as big-endian float64; when the processor stops, every connection is shut down and closed.

This is an independent implementation on plain sockets: an acceptor thread registers connections until all rows
are taken, writers use ``sendall`` on the registered sockets.

Several GPUs (one process per GPU, rows split contiguously): the reference serves every row on ONE port
(vfo_processor.py:99-110, example_simo.sh:161), and so does this build.  Rank 0 listens on that port; the k-th connection is
row k of the reference's order, and when that row belongs to another rank its socket is handed to that process -- the file
descriptor travels over a Unix-domain socket (``SCM_RIGHTS``: ``socket.send_fds`` / ``recv_fds``), rank 0 closes its copy and
the owning rank writes to the client directly.  The hand-off socket lives in the abstract namespace under a name derived from
the TCP port (one job per port on a host), only peers of the same user are accepted.  A client script written for the
reference works unchanged against ``gpus=N``.
"""
import os
import select
import socket
import struct
import threading
import time

from ..misc.general_util import eprint, shutdownSocket


def handoff_name(port: int) -> bytes:
    """Abstract-namespace address of the hand-off socket of the --simo job that serves TCP ``port`` on this host."""
    return b'\0sdrterm_amd.simo.%d' % int(port)


def _same_user(conn) -> bool:
    try:
        _pid, uid, _gid = struct.unpack('3i', conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize('3i')))
        return uid == os.getuid()
    except OSError:
        return False


class RowFanOut:
    """The listening side.  ``bounds``: None (one process serves every row) or the [lo, hi) row range of every rank in rank
    order -- this process is rank 0, keeps the connections of its own range and hands the others on."""

    def __init__(self, host: str, port: int, rows: int, bounds=None):
        self.rows = rows
        self._bounds = list(bounds) if bounds and len(bounds) > 1 else None
        self._total = self._bounds[-1][1] if self._bounds else rows
        self._accepted = 0               # connections taken so far == the next row of the reference's order
        self._sinks = []                 # registered client sockets, index == row (of this rank's range)
        self._lock = threading.Lock()
        self._full = threading.Event()   # set once every row has a client
        self._stop = threading.Event()
        self.failed = None               # why the fan-out can no longer be completed (a client socket could not reach its rank)
        self._hello = {}                 # hand-off connections that have not said which rank they are yet: socket -> deadline
        self._listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._listener.bind((host, port))
        self._listener.listen(max(8, self._total))
        self.address = self._listener.getsockname()
        self._peers, self._waiting, self._unix = {}, {}, None      # rank -> hand-off connection / sockets held for a rank not yet here
        if self._bounds:
            self._unix = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            self._unix.bind(handoff_name(self.address[1]))
            self._unix.listen(len(self._bounds))
        self._thread = threading.Thread(target=self._accept_loop, name='simo-acceptor', daemon=True)

    def start(self) -> None:
        eprint(f'\nAccepting connections on {self.address}\n')
        self._thread.start()

    def _owner(self, k: int) -> int:
        if not self._bounds:
            return 0
        for r, (lo, hi) in enumerate(self._bounds):
            if lo <= k < hi:
                return r
        return -1

    def _check_full(self) -> None:
        # the reference starts processing once EVERY row has its client (vfo_processor.py:77): rank 0 keeps accepting until the
        # rows of all ranks are connected and every socket has reached its owner
        if self._accepted >= self._total and not self._waiting:
            self._full.set()

    def _fail(self, why: str) -> None:
        """A row's client cannot reach the rank that computes it: the job as a whole cannot start (the reference starts once
        EVERY row has its client, vfo_processor.py:77).  wait_for_clients returns False and the processor goes down -- instead
        of rank 0 entering the first collective while a helper still waits for a socket that will never come."""
        if self.failed is None:
            self.failed = why
            eprint(f'--simo fan-out failed: {why}')
        self._stop.set()

    def _hand_on(self, rank: int, conn) -> bool:
        """The client socket to the process of ``rank``; False (and the fan-out marked failed) when that cannot be done."""
        peer = self._peers.get(rank)
        if peer is None:
            self._waiting.setdefault(rank, []).append(conn)       # that rank has not knocked yet: in order, as soon as it does
            return True
        try:
            socket.send_fds(peer, [b'c'], [conn.fileno()])
        except OSError as e:
            self._fail(f'hand-off of a client socket to rank {rank} failed: {e}')
            return False
        finally:
            conn.close()                                          # the owning process holds the connection now (or nobody will)
        return True

    def _take_hello(self, conn) -> None:
        """A hand-off connection has become readable: its four bytes say which rank it is.  (Read here, off the accept path: a
        local process that connects and stays silent only ever costs its own slot, never the other accepts.)"""
        self._hello.pop(conn, None)
        try:
            conn.setblocking(False)
            hello = conn.recv(4)
            conn.setblocking(True)
            rank = struct.unpack('<i', hello)[0] if len(hello) == 4 else -1
        except OSError:
            rank = -1
        if not _same_user(conn) or not (0 < rank < len(self._bounds)) or rank in self._peers:
            conn.close()
            return
        with self._lock:
            self._peers[rank] = conn
            for held in self._waiting.pop(rank, []):
                if not self._hand_on(rank, held):
                    break
            self._check_full()

    def _accept_loop(self) -> None:
        socks = [self._listener] + ([self._unix] if self._unix is not None else [])
        while not self._stop.is_set():
            now = time.monotonic()
            for c in [c for c, t in self._hello.items() if t < now]:       # silent for five seconds: not one of ours
                self._hello.pop(c)
                c.close()
            try:
                ready, _, _ = select.select(socks + list(self._hello), [], [], 0.2)
            except (OSError, ValueError):
                break
            for s in ready:
                if s in self._hello:
                    self._take_hello(s)
                    continue
                try:
                    conn, peer = s.accept()
                except OSError:
                    return
                if s is self._unix:
                    # a rank of this job; it announces itself with its number (read when it arrives: _take_hello)
                    self._hello[conn] = time.monotonic() + 5.0
                    continue
                eprint(f'Connection request from {peer}')
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                with self._lock:
                    k = self._accepted
                    owner = self._owner(k)
                    if k >= self._total or owner < 0:
                        # more clients than rows: the reference leaves such a handler blocked forever; refuse instead
                        conn.close()
                        continue
                    self._accepted += 1
                    if owner == 0:
                        self._sinks.append(conn)
                    elif not self._hand_on(owner, conn):
                        return                                    # the row has no client and never will: the job is marked failed
                    self._check_full()

    def wait_for_clients(self, is_dead=None, poll: float = 0.25) -> bool:
        """Blocks until every row has a client (or ``is_dead()`` turns true).  Returns True when complete."""
        while not self._full.wait(poll):
            if self.failed or self._stop.is_set() or (is_dead is not None and is_dead()):
                return False
        if self.failed:
            return False
        eprint('Connection(s) established')
        return True

    @property
    def clients(self):
        with self._lock:
            return list(self._sinks)

    def send_rows(self, z) -> None:
        """z: (rows, m) float64 -> row k to client k as big-endian doubles."""
        for sock, row in zip(self._sinks, z):
            sock.sendall(row.astype('>f8', copy=False).tobytes())

    def close(self) -> None:
        self._stop.set()
        for s in (self._listener, self._unix):
            try:
                if s is not None:
                    s.close()
            except OSError:
                pass
        if self._thread.is_alive():
            self._thread.join(timeout=2)
        with self._lock:
            for s in self._sinks:
                try:
                    eprint(f'Client disconnected: {s.getsockname()}')
                except OSError:
                    pass
                shutdownSocket(s)
                s.close()
            self._sinks.clear()
            for held in self._waiting.values():
                for s in held:
                    s.close()
            self._waiting.clear()
            for s in list(self._peers.values()) + list(self._hello):
                s.close()
            self._peers.clear()
            self._hello.clear()


class HandedRowFanOut(RowFanOut):
    """Ranks 1 .. N-1: no listener of their own; their rows' client sockets arrive from rank 0 (see the module text), in the
    order of the rows."""

    def __init__(self, host: str, port: int, rows: int, rank: int, connect_timeout: float = 600.0):
        self.rows = rows
        self._rank, self._port, self._connect_timeout = rank, port, connect_timeout
        self._sinks = []
        self._lock = threading.Lock()
        self._full = threading.Event()
        self._stop = threading.Event()
        self._listener = self._unix = None
        self._peers, self._waiting, self._hello = {}, {}, {}
        self.failed = None
        self._conn = None
        self.address = (host, port)
        self._thread = threading.Thread(target=self._receive_loop, name='simo-handoff', daemon=True)

    def start(self) -> None:
        eprint(f'\nRank {self._rank}: clients arrive through rank 0 on {self.address}\n')
        self._thread.start()

    def _receive_loop(self) -> None:
        deadline = time.monotonic() + self._connect_timeout
        conn = None
        while not self._stop.is_set():                      # rank 0 may be later than this rank
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                c.connect(handoff_name(self._port))
                if not _same_user(c):
                    # the name is derivable from the port: whoever bound it must be this user's rank 0, or the descriptors it
                    # sends are not this job's clients
                    c.close()
                    self.failed = f'the hand-off socket of port {self._port} belongs to another user'
                    eprint(f'rank {self._rank}: {self.failed}')
                    self._stop.set()
                    return
                conn = c
                break
            except OSError:
                c.close()
                if time.monotonic() > deadline:
                    eprint(f'rank {self._rank}: no hand-off socket for port {self._port}')
                    return
                time.sleep(0.05)
        if conn is None:
            return
        self._conn = conn
        try:
            conn.sendall(struct.pack('<i', self._rank))
            conn.settimeout(0.2)
            while not self._stop.is_set() and len(self._sinks) < self.rows:
                try:
                    msg, fds, _flags, _addr = socket.recv_fds(conn, 1, 1)
                except socket.timeout:
                    continue
                if not msg and not fds:
                    return                                   # rank 0 has gone
                for fd in fds:
                    s = socket.socket(fileno=fd)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    with self._lock:
                        self._sinks.append(s)
                        if len(self._sinks) == self.rows:
                            self._full.set()
        except OSError as e:
            if not self._stop.is_set():
                eprint(f'rank {self._rank}: hand-off failed: {e}')

    def close(self) -> None:
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=2)
        if self._conn is not None:
            try:
                self._conn.close()
            except OSError:
                pass
        with self._lock:
            for s in self._sinks:
                shutdownSocket(s)
                s.close()
            self._sinks.clear()


def make_fanout(host: str, port: int, rank: int, bounds):
    """The fan-out of rank ``rank`` of a job whose ranks own the row ranges ``bounds`` (one entry: a single process)."""
    lo, hi = bounds[rank]
    if rank == 0:
        return RowFanOut(host, port, hi - lo, bounds)
    return HandedRowFanOut(host, port, hi - lo, rank)
