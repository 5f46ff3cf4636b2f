"""Small host utilities the processors need (logging to stderr, port discovery, socket shutdown).

Counterparts of the helpers the reference keeps in src/misc/general_util.py:29-101; same names and
behaviour so the processor mirrors read like the reference.
"""
import sys
from contextlib import closing
from socket import socket, AF_INET, SOCK_STREAM, SOL_SOCKET, SO_REUSEADDR, SHUT_RDWR

_level = 0


def verboseOn() -> None:
    global _level
    _level = max(_level, 1)


def traceOn() -> None:
    global _level
    _level = 2


def eprint(*args, **kwargs) -> None:
    print(*args, file=sys.stderr, **kwargs)


def vprint(*args, **kwargs) -> None:
    if _level >= 1:
        eprint(*args, **kwargs)


def tprint(*args, **kwargs) -> None:
    if _level >= 2:
        eprint(*args, **kwargs)


def applyIgnoreException(func, *args):
    """Calls ``func(*args)``; an exception is returned instead of raised (general_util.py:63-67)."""
    try:
        return func(*args)
    except Exception as e:      # noqa: BLE001 -- that is the contract
        return e


def findPort(host='localhost') -> int:
    """Ask the OS for a free TCP port on ``host`` (general_util.py:93-101)."""
    with closing(socket(AF_INET, SOCK_STREAM)) as s:
        s.setsockopt(SOL_SOCKET, SO_REUSEADDR, 1)
        s.bind((host, 0))
        return s.getsockname()[1]


def shutdownSocket(*socks) -> None:
    for sock in socks:
        for fn, arg in ((sock.send, b''), (sock.shutdown, SHUT_RDWR)):
            try:
                fn(arg)
            except Exception:
                pass
