"""ORACLE -- CPU restatements of the reference algorithm, used ONLY as the checker.

Test infrastructure.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything from this package; the product package ``sdrterm_amd`` never does.

* ``ref_chain``  numpy/scipy restatement (calls scipy exactly as the reference does)
* ``c_chain``    ctypes front-end of ``chain.c`` (plain C fp64 restatement, also the timed CPU baseline)
"""
