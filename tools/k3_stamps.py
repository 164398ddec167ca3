"""Diagnostic: per-wave phase shares of K3 from the -DHMV_STAMP build (make -C csrc stamp), through the FUSED entry
(hmv_tf_ffdtf_f64: publish + in-kernel normalisation, as in the bench).  Usage: k3_stamps.py [windows] [freqs] [form]
(form 1 = compiler-scheduled body, 2 = hand-scheduled body).
Phases: 0 A(f) build | 1 panel factorisation | 2 barrier wait | 3 rest of the loop (operand reads, MFMA updates) |
        4 |H|^2 + LDS transposition + stores issued | 5 store drain + window count | 6 denominators / flag | 7 the row."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad
lib = ctypes.CDLL(os.environ.get("STAMP_LIB", os.path.join(ROOT, "hyperscanning_signal_analysis_amd", "libhypermvar_stamp.so")))
from hyperscanning_signal_analysis_amd import _lib
for name, (res, args) in _lib.SIGNATURES.items():
    getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
lib.hmv_debug_set_tf_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
n_items, m, p = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64, 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
form = int(sys.argv[3]) if len(sys.argv) > 3 else 0
assert lib.hmv_set_tuning(_lib.TUNE_K3_FORM, form) == 0
pad = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # dynamic LDS padding: fewer workgroups per CU
assert lib.hmv_set_tuning(_lib.TUNE_K3_LDS_PAD, pad) == 0
# realistic AR coefficients: fit dyad 0 windows with the production library
from hyperscanning_signal_analysis_amd.engine import Engine
eng = Engine()
x = eng.to_device(synthetic_var_dyad(0, T=500 * (n_items + 1))[None])
rec = torch.zeros(n_items, dtype=torch.int64, device=dev); st = 500 * torch.arange(n_items, dtype=torch.int64, device=dev)
R = eng.lagcov(x, rec, st, 1000, p)
ar, V, _, info = eng.yw_solve(R, m)
tw = eng.twiddles(0.5 * np.arange(1, F + 1), 500.0, p)
ff = torch.empty(n_items, m, m, F, dtype=torch.float64, device=dev)
den = torch.empty(n_items, 64, dtype=torch.float64, device=dev)
inf = torch.zeros(n_items * F, dtype=torch.int32, device=dev)
nws = int(lib.hmv_tf_ffdtf_workspace_bytes(n_items, m, p, F))
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
stamps = torch.zeros(2 * n_items * F * 4, 8, dtype=torch.int64, device=dev)
lib.hmv_debug_set_tf_stamps(stamps.data_ptr())
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.hmv_tf_ffdtf_f64(ar.data_ptr(), n_items, m, p, tw.data_ptr(), F, ff.data_ptr(), den.data_ptr(), 0, inf.data_ptr(),
                              1.0, ws.data_ptr(), nws, 0, 0, 0, 0)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, rc
print(f"form {form} lds pad {pad}: stamped kernel ms:", e0.elapsed_time(e1), "(do not quote: stamps serialise)")
allst = stamps.cpu().numpy().astype(np.float64)
s, det = allst[:n_items * F * 4], allst[n_items * F * 4:]
tot = s.sum(axis=1)
names = ["A(f) build", "panel factorisation", "barrier wait", "loop: operands + MFMA", "outputs: |H|^2, LDS, stores issued",
         "store drain + count", "denominators / flag", "the row"]
print(f"cycles per wave life: median {np.median(tot):.0f}  mean {tot.mean():.0f}")
for k in range(8):
    print(f"  {names[k]:36s} median {np.median(s[:, k]):9.0f} cyc  mean {s[:, k].mean():9.0f}  share {s[:, k].sum() / tot.sum() * 100:5.1f} %")
if det.sum() > 0:
    dn = ["A(f) build", "barrier wait", "loop (waves that do not factor)", "chain: barrier -> N / flag read", "chain: B operands + 16 MFMAs",
          "factor: panel through LDS", "factor: search, reciprocal, pivot row via LDS", "factor: elimination, N write"]
    print("hand-scheduled body, detail (cycles per wave life; a wave factors 4 of the 16 panels):")
    for k in range(8):
        print(f"  {dn[k]:46s} median {np.median(det[:, k]):9.0f}  mean {det[:, k].mean():9.0f}")
