"""TCP fan-out used by ``--simo``: one listening socket, one connected client per output row.

Wire behaviour follows the reference's server (src/dsp/vfo_processor.py:86-126): processing does not start until
as many clients as there are rows have connected; the k-th client to be registered receives row k of every chunk
as big-endian float64; when the processor stops, every connection is shut down and closed.

This is an independent implementation on plain sockets: an acceptor thread registers connections until all rows
are taken, writers use ``sendall`` on the registered sockets.
"""
import socket
import threading

from ..misc.general_util import eprint, shutdownSocket


class RowFanOut:
    def __init__(self, host: str, port: int, rows: int):
        self.rows = rows
        self._sinks = []                 # registered client sockets, index == row
        self._lock = threading.Lock()
        self._full = threading.Event()   # set once every row has a client
        self._stop = threading.Event()
        self._listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self._listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._listener.bind((host, port))
        self._listener.listen(max(8, rows))
        self._listener.settimeout(0.2)
        self.address = self._listener.getsockname()
        self._thread = threading.Thread(target=self._accept_loop, name='simo-acceptor', daemon=True)

    def start(self) -> None:
        eprint(f'\nAccepting connections on {self.address}\n')
        self._thread.start()

    def _accept_loop(self) -> None:
        while not self._stop.is_set():
            try:
                conn, peer = self._listener.accept()
            except socket.timeout:
                continue
            except OSError:
                break
            eprint(f'Connection request from {peer}')
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            with self._lock:
                if len(self._sinks) >= self.rows:
                    # more clients than rows: the reference leaves such a handler blocked forever; refuse instead
                    conn.close()
                    continue
                self._sinks.append(conn)
                if len(self._sinks) == self.rows:
                    self._full.set()

    def wait_for_clients(self, is_dead=None, poll: float = 0.25) -> bool:
        """Blocks until every row has a client (or ``is_dead()`` turns true).  Returns True when complete."""
        while not self._full.wait(poll):
            if self._stop.is_set() or (is_dead is not None and is_dead()):
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
        try:
            self._listener.close()
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
