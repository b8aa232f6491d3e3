"""bench.py on another build of the library: python tests/diag/bench_variant.py <lib.so> [ENGINE_FLAG=0|1 ...] [bench args]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant
_variant.use(sys.argv[1])
args = sys.argv[2:]
import groma_amd.engine as e
while args and "=" in args[0] and not args[0].startswith("-"):
    name, val = args.pop(0).split("=")
    if name.startswith("groma."):   # e.g. groma.SPIN_SYNC=0
        import groma_amd.groma as gm
        setattr(gm, name[6:], bool(int(val)))
    else:
        setattr(e, name, bool(int(val)))
sys.argv = [os.path.join(_variant.ROOT, "bench.py")] + args
runpy.run_path(sys.argv[0], run_name="__main__")
