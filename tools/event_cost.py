"""Developer aid: what an event record between two kernels of one stream costs on the device timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
A = be.fill_random((n, n), 1); Bm = be.fill_random((n, n), 2)
s = be.default_stream
def run(k_events, reps=40):
    be.synchronize(); t0 = time.time()
    evs = []
    for i in range(reps):
        C = be.gemm(A, Bm, stream=s)          # (gemm itself records one event for its output)
        for _ in range(k_events):
            evs.append(be.record_new(s))
    be.synchronize(); dt = (time.time() - t0) / reps
    for e in evs: be.recycle_event(e)
    return dt * 1e6
for k in (0, 8, 32, 0, 8, 32):
    print("extra events per kernel %d: %.1f us per iteration" % (k, run(k)))
