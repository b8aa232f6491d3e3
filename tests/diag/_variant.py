"""Measurement scaffolding (never imported by the product, tests/ or bench.py): point groma_amd at another BUILD of the library
for a one-box A/B.  The product itself reads no environment switch; a diag script calls use_env() before its first op:

    GROMA_HIP_LIB=tests/diag/myvariant.so python tests/diag/<script>.py
    python tests/diag/bench_variant.py tests/diag/myvariant.so --steps 10 --warmup 3        (bench.py on that build)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def use(path):
    from groma_amd import _lib
    if _lib._lib is not None:
        raise RuntimeError("the library is already loaded")
    _lib.LIB_PATH = os.path.abspath(path)
    return _lib.LIB_PATH


def use_env():
    p = os.environ.get("GROMA_HIP_LIB")
    return use(p) if p else None
