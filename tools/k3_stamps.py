"""Diagnostic: per-wave phase shares of the K3 kernel from the -DHMV_STAMP build (make -C csrc stamp).
Phases: 0 A(f) build | 1 panel factorisation (owner wave) | 2 barrier wait | 3 row interchanges |
        4 operand reads + MFMA update | 5 panel write-back + loop glue | 6 outputs."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad
lib = ctypes.CDLL(os.path.join(ROOT, "hyperscanning_signal_analysis_amd", "libhypermvar_stamp.so"))
from hyperscanning_signal_analysis_amd import _lib
for name, (res, args) in _lib.SIGNATURES.items():
    getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
lib.hmv_debug_set_tf_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
n_items, m, p = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64, 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
# realistic AR coefficients: fit dyad 0 windows with the production library
from hyperscanning_signal_analysis_amd.engine import Engine
eng = Engine()
x = eng.to_device(synthetic_var_dyad(0, T=500 * (n_items + 1))[None])
rec = torch.zeros(n_items, dtype=torch.int64, device=dev); st = 500 * torch.arange(n_items, dtype=torch.int64, device=dev)
R = eng.lagcov(x, rec, st, 1000, p)
ar, V, _, info = eng.yw_solve(R, m)
tw = eng.twiddles(0.5 * np.arange(1, F + 1), 500.0, p)
P = torch.empty(n_items, F, 64, 64, dtype=torch.float64, device=dev); rs = torch.empty(n_items, F, 64, dtype=torch.float64, device=dev)
inf = torch.zeros(n_items * F, dtype=torch.int32, device=dev)
wsx = torch.empty(int(lib.hmv_tf_workspace_doubles(n_items, m, p)), dtype=torch.float64, device=dev)
stamps = torch.zeros(n_items * F * 4, 8, dtype=torch.int64, device=dev)
lib.hmv_debug_set_tf_stamps(stamps.data_ptr())
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.hmv_tf_f64(ar.data_ptr(), n_items, m, p, tw.data_ptr(), F, P.data_ptr(), rs.data_ptr(), 0, 0, inf.data_ptr(), 1.0, wsx.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
print("stamped kernel ms:", e0.elapsed_time(e1), "(do not quote: stamps serialise)")
s = stamps.cpu().numpy().astype(np.float64)
tot = s.sum(axis=1)
names = ["A(f) build", "panel: extract/search/recip", "barrier wait", "interchanges", "MFMA update", "panel: pivot row via LDS", "outputs", "panel: elimination"]
print(f"cycles per wave life: median {np.median(tot):.0f}  mean {tot.mean():.0f}")
for k in range(8):
    print(f"  {names[k]:28s} median {np.median(s[:, k]):9.0f} cyc  share {s[:, k].sum() / tot.sum() * 100:5.1f} %")
