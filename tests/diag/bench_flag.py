"""A/B a python-side engine switch inside one gpurun call: python tests/diag/bench_flag.py Q_IN_PLACE=0 [bench args]"""
import os, runpy, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root)
import groma_amd.engine as e
name, val = sys.argv[1].split("=")
setattr(e, name, bool(int(val)))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
