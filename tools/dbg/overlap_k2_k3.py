"""Probe: do K1 + K2 of a later chunk make progress on a second (high-priority) stream while K3 fills the chip?"""
import sys, time
import numpy as np, torch
from hyperscanning_signal_analysis_amd.engine import Engine
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad
eng = Engine()
dev = eng.device
m, p, n, hop, F = 64, 8, 1000, 500, 256
x = eng.to_device(synthetic_var_dyad(0, T=300_000))
tw = eng.twiddles(0.5 * np.arange(1, F + 1), 500.0, p)
R = eng.lagcov_regular(x, 0, hop, 599, n, p)
ar, V, _, info = eng.yw_solve(R, m)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def k12():
    Rb = eng.lagcov_regular(x, 0, hop, NB, n, p)
    return eng.yw_solve(Rb, m)
def k3():
    return eng.transfer(ar, m, tw)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); o = fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1)); del o
    return best
print("K3 alone            %.3f ms" % timed(k3))
print("K1+K2 (%d) alone   %.3f ms" % (NB, timed(k12)))
for prio in (0, -1):
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=prio)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True); b1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sA):
            oa = k3(); a1.record()
        with torch.cuda.stream(sB):
            ob = k12(); b1.record()
        torch.cuda.synchronize()
        print("prio %2d: K3 done at %.3f ms, K1+K2 done at %.3f ms" % (prio, t0.elapsed_time(a1), t0.elapsed_time(b1)))
        del oa, ob
    # the other order: K1 + K2 of the NEXT recording queued first, K3 of the current one behind it on the other stream
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True); b1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sB):
            ob = k12(); b1.record()
        with torch.cuda.stream(sA):
            oa = k3(); a1.record()
        torch.cuda.synchronize()
        print("prio %2d, K1+K2 first: K3 done at %.3f ms, K1+K2 done at %.3f ms" % (prio, t0.elapsed_time(a1), t0.elapsed_time(b1)))
        del oa, ob
