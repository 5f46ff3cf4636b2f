from .. import Anything


def __getattr__(name):          # QTimer, QCoreApplication, QMainWindow, QApplication, QGridLayout, QTransform, ...
    if name.startswith('__'):
        raise AttributeError(name)
    return Anything
