"""bench.py with groma.SPECULATIVE_EXTRACT forced:  python tests/diag/bench_spec_extract.py 0|1 [bench.py args]   (the product reads
no environment switch; this is how the A/B in r05_spec_extract.sh flips it)"""
import os, runpy, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import groma_amd.groma as G
G.SPECULATIVE_EXTRACT = sys.argv[1] == "1"
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
