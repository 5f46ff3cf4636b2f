"""Stand-in for ``pyqtgraph`` used ONLY by tests/golden/make_golden.py in the build container.

The reference's plots (plots/spectrum_analyzer_plot.py, plots/waterfall_plot.py) build Qt windows in their constructors and
compute what they draw in ``update()``.  pyqtgraph / Qt are not installed here; these objects accept every call a
constructor or ``update()`` makes and do nothing, so that make_golden.py can run the reference's own ``update()`` bodies
and capture the arrays they compute (``amp``, ``image``).  Test tooling: nothing in the product package imports it.
"""


class _AnyType(type):
    def __getattr__(cls, name):         # QTransform.fromScale(...), QApplication.instance(), ...
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return Anything()


class Anything(metaclass=_AnyType):
    """Accepts any construction, attribute, call, item or iteration; returns more of the same."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return Anything()

    def __call__(self, *a, **k):
        return Anything()

    def __getitem__(self, k):
        return Anything()

    def __iter__(self):
        return iter(())


class PlotWidget(Anything):
    pass


class GraphicsLayoutWidget(Anything):
    pass


class ImageItem(Anything):
    pass


colormap = Anything()
