"""Stand-in for ``numba`` used ONLY by tests/golden/make_golden.py in the build container.

The reference (peads/sdrterm) decorates six tiny kernels with ``numba.guvectorize``; numba is not
installed in this image, so importing the reference needs something with that name.  This module
provides a pure-numpy ``guvectorize`` that loops the gufunc core dimensions exactly as the numba
layout string says, which is all the reference relies on.  It is test tooling: nothing in the
product package imports it and it never travels as part of the hot path.
"""
import re

import numpy as np

class _Sig:
    """numba type object stand-in: ``complex128[:]`` etc. only need to remember the scalar dtype."""

    def __init__(self, dt):
        self.dtype = np.dtype(dt)

    def __getitem__(self, item):
        return self


complex128 = _Sig(np.complex128)
float64 = _Sig(np.float64)


def _parse(layout):
    ins, outs = layout.split('->')
    pat = re.compile(r'\(([^)]*)\)')
    split = lambda s: [tuple(x.strip() for x in g.split(',') if x.strip()) for g in pat.findall(s)]
    return split(ins), split(outs)


def guvectorize(sigs, layout, **_kw):
    ins, outs = _parse(layout)
    nin = len(ins)
    in_dtypes = [t.dtype for t in sigs[0][:nin]]

    def deco(fn):
        def wrapper(*args):
            # numba casts each INPUT to the declared signature type (e.g. complex64 -> complex128) before
            # the kernel runs and casts on store into the caller's output array; mirror that.
            args = [np.asarray(a, dtype=in_dtypes[i]) if i < nin else a for i, a in enumerate(args)]
            core = [len(d) for d in ins + outs]
            # loop dimensions = leading dims left after removing each argument's core dims
            loops = [a.shape[:a.ndim - c] if c else a.shape for a, c in zip(args, core)]
            lshape = np.broadcast_shapes(*loops)
            if lshape == ():
                fn(*args)
                return args[-1]
            for idx in np.ndindex(*lshape):
                sub = []
                for a, lp in zip(args, loops):
                    k = len(lp)
                    ix = tuple(i if s != 1 else 0 for i, s in zip(idx[len(idx) - k:], lp)) if k else ()
                    if not k:
                        sub.append(a)
                    elif a.ndim == k:
                        # '()' core signature: numba hands the kernel a one-element view it can write through
                        sub.append(a[ix[:-1] + (slice(ix[-1], ix[-1] + 1),)])
                    else:
                        sub.append(a[ix])
                fn(*sub)
            return args[-1]

        wrapper.__name__ = fn.__name__
        wrapper.__wrapped__ = fn
        return wrapper

    return deco
