"""The sampling leg of bench.py alone, 3 steps (for PMC passes: tools/pmc_passes.sh <tag> tools/bench_short.py)."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-roofline', '--no-train', '--no-y-shape', '--no-other-configs']
runpy.run_path(sys.argv[0], run_name='__main__')
