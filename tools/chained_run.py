"""The chained tracker-vs-twin run of tests/test_gpu_native_tracker.py at length: frames, streams and the mbMapUpdated period from the command line;
prints per frame the largest NavState / cost deviation and the first frame (if any) at which a discrete result differs.
usage: python tools/chained_run.py [frames=96] [map_updated_every=5] [lens=0|1]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from test_gpu_native_tracker import _chained_run
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
mu = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lens = len(sys.argv) > 3 and sys.argv[3] == "1"
dist = None
if lens:
    from viorb_amd.synth import EUROC_DIST
    dist = EUROC_DIST
first, log = _chained_run(n, seeds=[301, 302, 303, 304], map_updated_every=mu, dist=dist)
for w in log:
    print("frame %3d  max |ns - twin| %.3e  max rel chi2 %.3e  states %s" % (w["frame"], w["ns"], w["chi"], w["states"]))
print("first discrete difference (frame, stream):", first)
