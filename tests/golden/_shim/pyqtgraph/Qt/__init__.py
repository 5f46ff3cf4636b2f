"""Qt names the reference's plots import from pyqtgraph.Qt (see ../__init__.py)."""
from .. import Anything
from . import QtCore, QtGui, QtWidgets  # noqa: F401
