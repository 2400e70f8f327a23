"""Where the CPU baseline spends its time (host cores of the GPU box): ORC_TIMING stage times of one 1440p TraceFrame.  python tools/oracle_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # run from anywhere: the package lives in the repo root
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["ORC_TIMING"] = "1"
import time
from helpers import oracle_from
from lumenrenderer_amd.scenes import sponza_standin
o = oracle_from(sponza_standin(), 2560, 1440, 6, blend=True)
o.trace_frame()
print("---- second frame", file=sys.stderr)
t0 = time.time(); o.trace_frame(); print("TraceFrame %.2f s on %d threads" % (time.time() - t0, os.cpu_count()))
