"""dev tool: bench.py against a development build of libkmx (tools/dev_variant.py):  python tools/bench_variant.py NAME [bench.py args]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib

name = sys.argv[1]
if name != "default":
    devlib.use(name)
sys.argv = [os.path.join(devlib.ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
