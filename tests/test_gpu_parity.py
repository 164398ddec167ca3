"""Parity of the HIP path (through the C ABI) with the reference: golden vectors generated from the
reference itself (tests/golden/*.npz) and the NumPy oracle on seeded inputs.  All @pytest.mark.gpu.

Tolerance (BASELINE.json north_star): 1e-5 relative, float64.  Written here as
  (i)  max|d| / max|ref| <= 1e-9 (observed ~1e-12; 1e-5 is the contract, 1e-9 is the regression guard) and
  (ii) np.allclose(out, ref, rtol=1e-5, atol=1e-5 * smallest row maximum)   [SURVEY.md 8(d) parity metric]
  (iii) |sum_{j,f} ffDTF[i] - 1| <= 1e-12.
"""
import numpy as np
import pytest
import torch

from oracle import mvar_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hyperscanning_signal_analysis_amd import mtmvar as M
    from hyperscanning_signal_analysis_amd.engine import default_engine
    from hyperscanning_signal_analysis_amd.sliding import sliding_ffdtf, sliding_ffdtf_device
    from hyperscanning_signal_analysis_amd.synthetic import northstar_freqs, synthetic_var_dyad

GUARD = 1e-9


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def assert_parity(out, ref, guard=GUARD):
    assert out.shape == ref.shape
    assert rel(out, ref) <= guard, rel(out, ref)
    if np.isrealobj(ref):
        row_max = np.abs(ref).reshape(ref.shape[0], -1).max(axis=1).min()
    else:
        row_max = np.abs(ref).max()
    assert np.allclose(out, ref, rtol=1e-5, atol=1e-5 * row_max)


def test_native_library_is_what_runs():
    import ctypes
    from hyperscanning_signal_analysis_amd import _lib
    assert isinstance(default_engine().lib, ctypes.CDLL)
    maps = open("/proc/self/maps").read()
    assert "libhypermvar.so" in maps


# ----------------------------------------------------------------------------- golden vectors
def test_g1_config1_all_functions(golden, capsys):
    g = golden("g1_config1.npz")
    x, fs, freqs = g["x"], float(g["fs"]), g["freqs"]
    rl, rr, r0 = M.count_corr(x[:, :, None], 4, 1)
    assert_parity(rl, g["r_left"]); assert_parity(rr, g["r_right"]); assert_parity(r0, g["r"])
    ar, V = M.ar_coeff(x, 4)
    assert ar.shape == (3, 3, 4) and V.shape == (3, 3)
    assert_parity(ar, g["ar"]); assert_parity(V, g["V"])
    H, A = M.mvar_transfer_function(g["ar"], freqs, fs)
    assert H.dtype == np.complex128 and H.shape == (3, 3, len(freqs))
    assert_parity(H, g["H"]); assert_parity(A, g["A"])
    dtf = M.dtf_multivariate(x, freqs, fs, optimal_model_order=4)
    assert "Using provided model order: p = 4" in capsys.readouterr().out     # side effect kept
    assert_parity(dtf, g["dtf"])
    ff = M.full_freq_dtf(x, freqs, fs, optimal_model_order=4)
    assert_parity(ff, g["ffdtf"])
    assert np.abs(ff.sum(axis=(1, 2)) - 1).max() < 1e-12
    S = M.multivariate_spectra(x, freqs, fs, optimal_model_order=4)
    assert "Using provided model order: p =  4" in capsys.readouterr().out
    assert_parity(S, g["spectra"])
    assert np.abs(S - S.transpose(1, 0, 2)).max() < 1e-9 * np.abs(S).max()     # complex-symmetric (Q3)
    for c in ("AIC", "HQ", "SC"):
        crit, rng, popt = M.mvar_criterion(x, 10, c)
        assert np.allclose(crit, g[f"crit_{c}"], rtol=1e-9, atol=1e-10)
        assert list(rng) == list(range(1, 11)) and int(popt) == int(g[f"crit_{c}_popt"])
    ff_auto = M.full_freq_dtf(x, freqs, fs, max_model_order=10, optimal_model_order=None, crit_type="AIC")
    assert "Optimal model order for all channels: p = " in capsys.readouterr().out
    assert_parity(ff_auto, g["ffdtf_auto"])


def test_g2_northstar_window(golden):
    g = golden("g2_northstar.npz")
    x, fs, freqs, p = g["x"], float(g["fs"]), g["freqs"], int(g["p"])
    assert_parity(M.lag_covariances(x, p), g["R"])
    ar, V = M.ar_coeff(x, p)
    assert_parity(ar, g["ar"]); assert_parity(V, g["V"])
    ff = M.full_freq_dtf(x, freqs, fs, optimal_model_order=p)
    assert ff.shape == (64, 64, 256)
    assert_parity(ff[:, :, ::16], g["ffdtf_sub"])
    assert_parity(ff.sum(axis=2), g["ffdtf_fsum"])
    assert abs(ff.sum() - float(g["ffdtf_sum"])) < 1e-10
    assert abs((ff ** 2).sum() - float(g["ffdtf_sqsum"])) < 1e-12
    assert np.abs(ff.sum(axis=(1, 2)) - 1).max() < 1e-12
    S = M.multivariate_spectra(x, freqs, fs, optimal_model_order=p)
    assert_parity(S[:, :, ::64], g["spectra_sub"])


def test_g3_overlapping_windows_batched(golden):
    g = golden("g3_overlap.npz")
    ff = sliding_ffdtf(g["x"], 1000, 3, int(g["p"]), g["freqs"], float(g["fs"]))
    assert ff.shape == (3, 64, 64, 256)
    for i in range(3):
        assert_parity(ff[i][:, :, ::32], g[f"ffdtf_sub{i}"])
        assert_parity(ff[i].sum(axis=2), g[f"ffdtf_fsum{i}"])


def test_g4_config4_small_windows(golden):
    g = golden("g4_config4.npz")
    x, fs, freqs = g["x"], float(g["fs"]), g["freqs"]
    for tag, (nw, ws) in {"a": (3, None), "b": (5, 160)}.items():
        ff = sliding_ffdtf(x, ws, nw, 5, freqs, fs)
        assert_parity(ff, g[f"ff_{tag}"])
    res = M.mvar_analysis(x, freqs, fs, 5, want=("ffdtf", "spectra"))
    assert_parity(res["ffdtf"], g["ff_global"]); assert_parity(res["spectra"], g["sp_global"])
    _, _, popt = M.mvar_criterion(x, 20, "AIC")
    assert int(popt) == int(g["p_opt_auto"])


def test_g5_multitrial(golden):
    g = golden("g5_multitrial.npz")
    rl, rr, r0 = M.count_corr(g["x"], 3, 1)
    assert_parity(rl, g["r_left"]); assert_parity(rr, g["r_right"]); assert_parity(r0, g["r"])
    ar, V = M.ar_coeff(g["x"], 3)
    assert_parity(ar, g["ar"]); assert_parity(V, g["V"])
    H, _ = M.mvar_transfer_function(ar, g["freqs"], float(g["fs"]))
    assert_parity(H, g["H"])


def test_g6_error_behaviour(golden):
    g = golden("g6_errors.npz")
    with pytest.raises(np.linalg.LinAlgError, match="Singular matrix"):      # dead channel, like dgesv
        M.ar_coeff(g["xz"], 3)
    with pytest.raises(ValueError, match="Invalid criterion type"):
        M.mvar_criterion(g["xs"], 3, "BIC")
    with pytest.raises(np.linalg.LinAlgError, match="Singular matrix"):      # A(f) exactly singular
        M.mvar_transfer_function(np.ones((2, 2, 1)) * 0.5, np.array([0.0]), 10.0)
    with pytest.raises(ValueError):
        M.ar_coeff(np.zeros((70, 500)), 2)                                  # > 64 channels: refused loudly


def test_g6_rank_deficient_and_nearly_collinear_windows(golden):
    """What happens where the normal equations are (nearly) singular -- VERDICT r1 weak #1, ADVICE r1.
    * exactly rank-deficient window (channel 3 = channel 0 + channel 1): the reference's dgesv does NOT raise and
      returns ONE of the infinitely many solutions, picked by LAPACK's rounding (fixture xs_ar: |ar| <= 0.3).  K2
      is an unpivoted LDL^T of the (semi-definite) Gram matrix: it meets a non-positive pivot and the drop-in
      raises LinAlgError("Singular matrix") naming the window -- a documented deviation:
      no unique answer exists to be matched.
    * nearly collinear windows (channel 3 = ch0 + ch1 + eps * noise, cond(r_left) 2e5 .. 2e13): both sides return,
      and they agree to the accuracy the conditioning allows, 1e2 * cond * eps (observed ~3e-12 at cond 2e5,
      2e-8 at 2e9); the 1e-5 contract of BASELINE.json therefore holds up to cond ~ 1e9."""
    g = golden("g6_errors.npz")
    assert str(g["singular_raises"]) == "no" and np.abs(g["xs_ar"]).max() < 1.0 and float(g["xs_cond"]) > 1e14
    with pytest.raises(np.linalg.LinAlgError, match="Singular matrix") as ei:
        M.ar_coeff(g["xs"], 3)
    assert list(ei.value.items) == [0] and int(ei.value.info[0]) >= 1
    eps64 = np.finfo(np.float64).eps
    for k in range(3):
        cond = float(g[f"nc{k}_cond"])
        tol = 1e2 * cond * eps64
        try:
            ar, V = M.ar_coeff(g[f"nc{k}_x"], 3)
        except np.linalg.LinAlgError:
            assert cond > 1e12, f"nearly collinear window (cond {cond:.1e}) must not be refused"
            continue
        assert rel(ar, g[f"nc{k}_ar"]) <= tol, (k, cond, rel(ar, g[f"nc{k}_ar"]))
        assert rel(V, g[f"nc{k}_V"]) <= tol
    # one bad window in a batch: the default raises and names it; check="nan" keeps the others
    eng = default_engine()
    x3 = np.stack([g["nc0_x"], g["xs"], g["nc0_x"][::-1].copy()])
    xd = eng.to_device(x3)
    rec = torch.arange(3, dtype=torch.int64, device=eng.device)
    st = torch.zeros(3, dtype=torch.int64, device=eng.device)
    freqs = np.linspace(1.0, 30.0, 16)
    with pytest.raises(np.linalg.LinAlgError) as ei:
        eng.sliding_ffdtf(xd, rec, st, 300, 3, freqs, 64.0)
    assert list(ei.value.items) == [1] and "item 1" in ei.value.args[1]
    out = eng.sliding_ffdtf(xd, rec, st, 300, 3, freqs, 64.0, check="nan").cpu().numpy()
    assert np.isnan(out[1]).all() and np.isfinite(out[0]).all() and np.isfinite(out[2]).all()
    assert_parity(out[0], O.full_freq_dtf(x3[0], freqs, 64.0, 3), 1e-6)      # cond 2e5: ~1e-11 expected


def test_band_sums_and_window_range_shards():
    """hmv_band_sums_f64 against a NumPy reduction, and BASELINE config 2 sharded by window range: every rank
    computes its windows from ITS slice of the recording (rebased starts) and the concatenation equals the
    unsharded result bit for bit (VERDICT r1 missing #4)."""
    from hyperscanning_signal_analysis_amd import distributed as hd
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    T, w, p = 20_000, 1000, 8
    x = synthetic_var_dyad(2, T=T)
    freqs = northstar_freqs(64)
    pos, w = window_positions(T, 2 * T // w - 1, w)
    xd = eng.to_device(x[None])
    rec, st = window_items(1, pos, eng.device)
    full = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            s_lo, s_hi, starts, (w_lo, w_hi) = hd.shard_window_items(pos, w, world, r)
            xs = eng.to_device(x[None, :, s_lo:s_hi])                      # only this rank's samples travel
            rec_r, st_r = window_items(1, starts, eng.device)
            parts.append(eng.sliding_ffdtf(xs, rec_r, st_r, w, p, freqs, 500.0))
        assert torch.equal(torch.cat(parts), full)
    bands = hd.band_integrate(full, freqs)
    ff = full.cpu().numpy()
    assert bands.shape == (len(pos), 64, 64, len(hd.DEFAULT_BANDS))
    for b, (lo, hi) in enumerate(hd.DEFAULT_BANDS):
        sel = (freqs >= lo) & (freqs < hi)
        ref = ff[..., sel].sum(-1) if sel.any() else np.zeros(ff.shape[:-1])
        assert np.allclose(bands[..., b].cpu().numpy(), ref, rtol=1e-13, atol=1e-300)
    odd = eng.band_sums(full[:2, :3, :5, :], [0, 7, 63, 10], [64, 8, 64, 10]).cpu().numpy()      # non-contiguous view
    sub = ff[:2, :3, :5, :]
    assert np.allclose(odd[..., 0], sub.sum(-1), rtol=1e-13) and np.array_equal(odd[..., 1], sub[..., 7])
    assert np.array_equal(odd[..., 2], sub[..., 63]) and not odd[..., 3].any()


def test_g7_partial_coherence_ddtf_gpdc(golden, capsys):
    """the three measures of src/mtmvar.py:287-468 against the reference's own outputs."""
    g = golden("g7_connectivity.npz")
    for tag in "abc":
        x, fs, freqs, p = g[f"x_{tag}"], float(g[f"fs_{tag}"]), g[f"freqs_{tag}"], int(g[f"p_{tag}"])
        kap = M.partial_coherence(g[f"spectra_{tag}"])
        assert kap.dtype == np.complex128
        assert_parity(kap, g[f"pcoh_{tag}"])
        assert_parity(M.direct_dtf(x, freqs, fs, optimal_model_order=p), g[f"ddtf_{tag}"])
        assert_parity(M.gen_partial_directed_coherence(x, freqs, fs, optimal_model_order=p), g[f"gpdc_{tag}"])
    assert "Using provided model order" in capsys.readouterr().out
    assert_parity(M.partial_coherence(g["Z"]), g["pcoh_Z"])              # arbitrary (non-MVAR) complex matrices
    assert np.array_equal(M.partial_coherence(np.full((1, 1, 3), 2.0 + 1.0j)), g["pcoh_1x1"])


@pytest.mark.parametrize("m,p,n,F", [(5, 2, 400, 6), (19, 3, 900, 5), (33, 2, 1200, 3)])
def test_connectivity_measures_vs_oracle(m, p, n, F):
    rng = np.random.default_rng(7 * m + p)
    x = rng.standard_normal((m, n))
    x[:, 1:] += 0.5 * x[:, :-1]
    x[1:] += 0.3 * x[:-1]
    freqs = np.linspace(2.0, 40.0, F)
    res = M.mvar_analysis(x, freqs, 100.0, p, want=("pcoh", "ddtf", "gpdc", "spectra"))
    assert_parity(res["gpdc"], O.gen_partial_directed_coherence(x, freqs, 100.0, p), 1e-8)
    assert np.abs((res["gpdc"] ** 2).sum(axis=0) - 1.0).max() < 1e-12           # columns of GPDC^2 sum to one
    if m <= 19:        # minors by determinant: O(m^5 F) on the host
        kap = O.partial_coherence(O.multivariate_spectra(x, freqs, 100.0, p))
        assert_parity(np.abs(res["pcoh"]), np.abs(kap), 1e-7)
        assert_parity(res["ddtf"], O.direct_dtf(x, freqs, 100.0, p), 1e-7)
    else:              # size-independent properties: unit diagonal, |kappa| symmetric for a complex-symmetric S
        k = res["pcoh"]
        assert np.allclose(k[np.arange(m), np.arange(m)], 1.0)
        assert np.abs(np.abs(k) - np.abs(k).transpose(1, 0, 2)).max() < 1e-8


# ----------------------------------------------------------------------------- oracle on seeded inputs
@pytest.mark.parametrize("m,p,n,F", [(1, 1, 50, 3), (2, 3, 120, 5), (7, 2, 300, 9), (16, 5, 400, 17),
                                     (17, 4, 500, 8), (19, 6, 700, 33), (32, 3, 600, 16), (33, 2, 500, 7),
                                     (48, 4, 900, 12), (50, 7, 1100, 5), (64, 9, 1500, 10)])
def test_shapes_vs_oracle(m, p, n, F):
    rng = np.random.default_rng(100 * m + p)
    x = rng.standard_normal((m, n))
    x[:, 1:] += 0.6 * x[:, :-1]                     # some autocorrelation
    freqs = np.linspace(0.0, 50.0, F)               # includes f = 0 and Nyquist
    fs = 100.0
    res = M.mvar_analysis(x, freqs, fs, p, want=("ar", "V", "H", "A", "dtf", "ffdtf", "spectra"))
    ar, V = O.ar_coeff(x, p)
    H, A = O.mvar_transfer_function(ar, freqs, fs)
    assert_parity(res["ar"], ar, 1e-8); assert_parity(res["V"], V, 1e-8)
    assert_parity(res["A"], A, 1e-8); assert_parity(res["H"], H, 1e-8)
    assert_parity(res["dtf"], np.abs(H) ** 2, 1e-8)
    assert_parity(res["ffdtf"], O.full_freq_dtf(x, freqs, fs, p), 1e-8)
    assert_parity(res["spectra"], O.multivariate_spectra(x, freqs, fs, p), 1e-8)


@pytest.mark.parametrize("m", [5, 16, 19, 33, 48, 64])
def test_transfer_inverse_needs_pivoting(m):
    """Random (non diagonally dominant) coefficient sets force row interchanges in the Gauss-Jordan."""
    rng = np.random.default_rng(m)
    ar = rng.standard_normal((m, m, 3))
    freqs = np.linspace(1, 60, 9)
    H, A = M.mvar_transfer_function(ar, freqs, 128.0)
    Ho, Ao = O.mvar_transfer_function(ar, freqs, 128.0)
    assert_parity(A, Ao, 1e-12); assert_parity(H, Ho, 1e-10)
    eye = np.einsum("ijf,jkf->ikf", H, A)
    assert np.abs(eye - np.eye(m)[:, :, None]).max() < 1e-9


@pytest.mark.parametrize("m,shift", [(64, 1), (64, 16), (64, 17), (64, 32), (64, 47), (64, 63), (48, 31), (20, 7)])
def test_transfer_inverse_scaled_permutation(m, shift):
    """At f = 0 an order-1 model with A_1 = I - P D has A(0) = P D, a scaled cyclic permutation: every pivot
    column holds exactly one non-zero, `shift` rows below the diagonal (crossing the 16-lane DPP rows of the
    wave-wide arg-max), so any error in the pivot search shows up as a singular flag or a wrong inverse."""
    rng = np.random.default_rng(1000 + shift)
    d = rng.uniform(0.5, 2.0, m) * rng.choice([-1.0, 1.0], m)
    PD = np.zeros((m, m))
    PD[(np.arange(m) + shift) % m, np.arange(m)] = d
    ar = (np.eye(m) - PD)[:, :, None]
    freqs = np.array([0.0, 64.0])                    # z = 1 and z = -1 at fs = 128
    H, A = M.mvar_transfer_function(ar, freqs, 128.0)
    assert_parity(A[:, :, 0], PD.astype(complex), 1e-15)
    assert_parity(H[:, :, 0], np.linalg.inv(PD).astype(complex), 1e-14)
    Ho, _ = O.mvar_transfer_function(ar, freqs, 128.0)
    assert_parity(H, Ho, 1e-12)


def test_pivot_threshold_variants_agree():
    from hyperscanning_signal_analysis_amd.engine import Engine
    rng = np.random.default_rng(9)
    ar = 0.3 * rng.standard_normal((64, 64, 4))
    freqs = np.linspace(1, 60, 6)
    e1, e2 = Engine(pivot_tau=1.0), Engine(pivot_tau=0.1)
    arp = e1.to_device(ar[None])
    tw = e1.twiddles(freqs, 128.0, 4)
    H1 = e1.to_mmf_complex(e1.transfer(arp, 64, tw, want_P=False, want_H=True)["H"], 64).cpu().numpy()
    H2 = e2.to_mmf_complex(e2.transfer(arp, 64, tw, want_P=False, want_H=True)["H"], 64).cpu().numpy()
    assert rel(H1, H2) < 1e-10


def test_analytic_properties():
    # diagonal VAR => off-diagonal ffDTF ~ 0; permutation equivariance
    rng = np.random.default_rng(3)
    n, m = 4000, 6
    x = np.zeros((m, n))
    e = rng.standard_normal((m, n))
    for t in range(2, n):
        x[:, t] = 0.5 * x[:, t - 1] - 0.3 * x[:, t - 2] + e[:, t]
    freqs = np.linspace(1, 60, 20)
    ff = M.full_freq_dtf(x, freqs, 128.0, optimal_model_order=2)
    off = ff[~np.eye(m, dtype=bool)]
    assert off.max() < 5e-3 * ff[np.eye(m, dtype=bool)].max()
    perm = rng.permutation(m)
    ffp = M.full_freq_dtf(x[perm], freqs, 128.0, optimal_model_order=2)
    assert rel(ffp, ff[perm][:, perm]) < 1e-9


def test_ar_coefficient_recovery_on_long_var():
    """Known-answer: a long realisation of a stable VAR(2) gives back its coefficients and innovation variance."""
    rng = np.random.default_rng(21)
    m, n = 5, 200_000
    A1 = 0.4 * np.eye(m) + 0.08 * rng.standard_normal((m, m))
    A2 = -0.25 * np.eye(m) + 0.05 * rng.standard_normal((m, m))
    e = rng.standard_normal((n, m)) * np.array([1.0, 0.5, 2.0, 1.5, 0.8])
    x = np.zeros((n, m))
    for t in range(2, n):
        x[t] = A1 @ x[t - 1] + A2 @ x[t - 2] + e[t]
    ar, V = M.ar_coeff(np.ascontiguousarray(x.T), 2)
    assert np.abs(ar[:, :, 0] - A1).max() < 2e-2 and np.abs(ar[:, :, 1] - A2).max() < 2e-2
    assert np.abs(np.diag(V) - np.array([1.0, 0.25, 4.0, 2.25, 0.64])).max() < 5e-2
    aro, Vo = O.ar_coeff(np.ascontiguousarray(x.T), 2)
    assert_parity(ar, aro, 1e-9); assert_parity(V, Vo, 1e-9)


# ----------------------------------------------------------------------------- full-size properties
def test_northstar_full_size_properties():
    """BASELINE.json config 1 sizes (m=64, p=8, F=256, 2 s windows, 50 % overlap) on 60 s of dyad 0:
    size-independent properties on every window + oracle spot checks on a few windows."""
    eng = default_engine()
    fs, w, p = 500.0, 1000, 8
    T = 30_000
    x = synthetic_var_dyad(0, T=T)
    freqs = northstar_freqs(256)
    nw = 2 * T // w - 1
    xd = eng.to_device(x[None])
    ff = sliding_ffdtf_device(xd, w, nw, p, freqs, fs, eng)[0]
    assert ff.shape == (nw, 64, 64, 256)
    assert float((ff.sum(dim=(2, 3)) - 1).abs().max()) < 1e-12          # every row of every window sums to 1
    assert float(ff.min()) >= 0.0 and bool(torch.isfinite(ff).all())
    # determinism: a second run is bit-identical (fixed reduction orders everywhere)
    ff2 = sliding_ffdtf_device(xd, w, nw, p, freqs, fs, eng)[0]
    assert torch.equal(ff, ff2)
    # batching invariance: window k computed alone equals window k of the batch, bit for bit, when K1 sums every
    # window from its own samples (the default shares the 50 % overlap between windows: equal to rounding)
    k = 17
    alone = sliding_ffdtf_device(xd[:, :, 500 * k:500 * k + w].contiguous(), w, 1, p, freqs, fs, eng)[0, 0]
    direct = sliding_ffdtf_device(xd, w, nw, p, freqs, fs, eng, share_overlap=False)[0]
    assert torch.equal(alone, direct[k])
    assert float((direct - ff).abs().max() / direct.abs().max()) < 1e-11
    for k in (0, 29, nw - 1):
        ref = O.full_freq_dtf(x[:, 500 * k:500 * k + w], freqs, fs, p)
        assert_parity(ff[k].cpu().numpy(), ref)


def test_config3_share_of_one_gpu_at_full_size():
    """BASELINE config 3 (64 dyads x 10 min, dyad-sharded over 8 GPUs) as seen by ONE GPU: 8 ten-minute dyads = 4 792
    windows of 64 channels, p = 8, 256 frequencies in one call, the reduced product (what is gathered over xGMI).
    Size-independent checks on every window: the band that spans the whole grid sums to 1 over the columns of every row
    (|.| <= 1e-12); a dyad gives the same bits wherever it sits in the batch and however the batch is chunked (one dyad
    per chunk, four dyads per chunk); one window against the oracle."""
    from hyperscanning_signal_analysis_amd import distributed as hdist
    from hyperscanning_signal_analysis_amd.sliding import regular_grid, window_items, window_positions
    eng = default_engine()
    fs, w, p, T, D = 500.0, 1000, 8, 300_000, 8
    freqs = northstar_freqs(256)
    distinct = [synthetic_var_dyad(60 + d, T=T) for d in range(4)]
    xd = eng.to_device(np.stack([distinct[d % 4] for d in range(D)]))           # dyads d and d + 4 are the same recording
    pos, w = window_positions(T, 2 * T // w - 1, w)
    nw = len(pos)
    rec, st = window_items(D, pos, eng.device)
    grid = regular_grid(pos, w, p)
    lo, hi = hdist.band_bins(freqs)
    lo, hi = list(lo) + [0], list(hi) + [256]                                     # + the whole grid as a sixth band
    a = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, fs, grid=grid, bands=(lo, hi), chunk=nw)
    assert a.shape == (D * nw, 64, 64, 6) and bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0
    assert float((a[..., 5].sum(dim=2) - 1).abs().max()) < 1e-12
    av = a.view(D, nw, 64, 64, 6)
    assert torch.equal(av[:4], av[4:])
    b = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, fs, grid=grid, bands=(lo, hi), chunk=4 * nw)
    assert torch.equal(a, b)
    del b
    k = 311
    ref = O.full_freq_dtf(distinct[2][:, pos[k]:pos[k] + w], freqs, fs, p)
    want = np.stack([ref[..., l0:h0].sum(axis=-1) for l0, h0 in zip(lo, hi)], axis=-1)
    assert_parity(av[6, k].cpu().numpy(), want)


def test_two_stream_overlap_is_bit_identical():
    """K2 split into two half-batches on two HIP streams (fork/join inside the fused C call), and chunking of
    the windows, give the same bits as one stream / one chunk."""
    eng = default_engine()
    x = synthetic_var_dyad(5, T=12_000)
    freqs = northstar_freqs(64)
    xd = eng.to_device(x[None])
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    pos, w = window_positions(12_000, 23, 1000)
    rec, st = window_items(1, pos, eng.device)
    from hyperscanning_signal_analysis_amd import _lib
    T = _lib.FLAG_YW_TILED          # the launch-chain form of K2 (the one that is split over the two streams)
    a = eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0, chunk=23, overlap=False, flags=T)
    b = eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0, chunk=23, overlap=True, flags=T)     # 12 + 11 windows
    c = eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0, chunk=5, overlap=False, flags=T)
    d = eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0, chunk=17, overlap=True, flags=T)     # 9 + 8, then 6 unsplit
    e = eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0, chunk=23, flags=_lib.FLAG_YW_ONE_LAUNCH)   # block LDL^T in one launch
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d) and torch.equal(a, e)


@pytest.mark.parametrize("m,n,p,F,nw", [(64, 1000, 8, 256, 59), (64, 1000, 8, 48, 75), (4, 160, 5, 32, 500),
                                          (19, 400, 3, 32, 200), (33, 300, 2, 16, 400), (48, 600, 4, 16, 400),
                                          (48, 600, 4, 7, 400)])
def test_normalisation_inside_k3_is_bit_identical_to_separate_pass(m, n, p, F, nw):
    """ffDTF normalised inside K3 by the workgroup that completes a window (device-scope arrival counter,
    release / acquire) == |H|^2 written out and normalised by the separate K4 pass, bit for bit, for every
    window -- the ones inside the batch (fused) and the last few (always left to K4).  With the built-in lag rule
    (6 x resident windows) only the 64- and 48-channel cases with enough windows actually fuse here (cases 1, 5, 6);
    the others, and a grid that is not a multiple of 16 frequencies (the last case), are normalised by K4 throughout --
    `test_normalisation_inside_k3_with_a_short_lag` below forces the fused path for the small shapes."""
    from hyperscanning_signal_analysis_amd import _lib
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(11, m=m, p=min(p, 4), T=T, burn=300)
    freqs = np.linspace(0.5, 120.0, F)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    fused = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
    again = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
    plain = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, flags=_lib.FLAG_UNFUSED_NORM)
    chunked = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, chunk=max(1, nw // 3 + 1))
    torch.cuda.synchronize()
    assert torch.equal(fused, plain) and torch.equal(fused, again) and torch.equal(fused, chunked)
    assert float((fused.sum(dim=(2, 3)) - 1).abs().max()) < 1e-12
    k = nw // 2
    assert_parity(fused[k].cpu().numpy(), O.full_freq_dtf(x[:, pos[k]:pos[k] + w], freqs, 500.0, p), 1e-8)


@pytest.mark.parametrize("m,n,p,F,nw,lag", [(4, 160, 5, 32, 300, 8), (19, 400, 3, 32, 150, 8), (33, 300, 2, 16, 200, 8),
                                              (64, 1000, 8, 32, 60, 8), (64, 1000, 8, 256, 24, 9), (48, 600, 4, 16, 200, 8)])
def test_normalisation_inside_k3_with_a_short_lag(m, n, p, F, nw, lag):
    """The built-in lag rule (6 x the windows the chip holds) leaves small matrices and short grids entirely to K4, so the
    16- and 32-channel instantiations of the publish / denominator / row-worker path and `norm_missed_kernel` would never
    run under the other tests.  With the lag forced to 8 (hmv_set_tuning) every shape fuses all but its last 8 windows, and
    rows regularly come up before their window is complete (they go through the missed-row list): still the same bits
    as the separate K4 pass, for every window, twice in a row (the workspace is reused)."""
    from hyperscanning_signal_analysis_amd import _lib
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(13, m=m, p=min(p, 4), T=T, burn=300)
    freqs = np.linspace(0.5, 120.0, F)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    plain = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, flags=_lib.FLAG_UNFUSED_NORM)
    assert eng.lib.hmv_set_tuning(_lib.TUNE_NORM_LAG, lag) == 0 and eng.lib.hmv_get_tuning(_lib.TUNE_NORM_LAG) == lag
    try:
        fused = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
        again = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
        torch.cuda.synchronize()
    finally:
        assert eng.lib.hmv_set_tuning(_lib.TUNE_NORM_LAG, 0) == 0
    assert torch.equal(fused, plain) and torch.equal(again, plain)
    assert float((fused.sum(dim=(2, 3)) - 1).abs().max()) < 1e-12


@pytest.mark.parametrize("m,n,p,nw", [(64, 1000, 8, 6), (33, 300, 2, 5), (19, 400, 5, 4), (4, 160, 7, 9), (64, 700, 1, 3)])
def test_pipelined_levinson_whittle_form_equals_the_first_form(m, n, p, nw):
    """HMV_TUNE_YW_FORM = 3 (csrc/yw_lwr2.hip: the block Levinson-Whittle recursion as a software pipeline, register X
    operands, accumulators that start from the subtrahend) against the default form and the oracle: ar, V and every
    order's log det agree to rounding; every padded size, orders 1 .. 8."""
    from hyperscanning_signal_analysis_amd import _lib
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(23, m=m, p=min(p, 4), T=T, burn=300)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    R = eng.lagcov(xd, rec, st, w, p)
    ar0, V0, ld0, info0 = eng.yw_solve(R, m, want_logdet=True)
    assert eng.lib.hmv_set_tuning(_lib.TUNE_YW_FORM, 3) == 0
    try:
        ar3, V3, ld3, info3 = eng.yw_solve(R, m, want_logdet=True)
        ar3b, V3b, _, _ = eng.yw_solve(R, m)
        torch.cuda.synchronize()
    finally:
        assert eng.lib.hmv_set_tuning(_lib.TUNE_YW_FORM, 0) == 0
    assert not bool(info0.any()) and not bool(info3.any())
    assert torch.equal(ar3, ar3b) and torch.equal(V3, V3b)
    for a, b in ((ar3, ar0), (V3, V0), (ld3, ld0)):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-9
    for k in (0, nw - 1):
        aro, Vo = O.ar_coeff(x[:, pos[k]:pos[k] + w], p)
        assert_parity(ar3[k, :m, :m].cpu().numpy(), aro, 1e-8)
        assert_parity(V3[k, :m, :m].cpu().numpy(), Vo, 1e-8)


@pytest.mark.parametrize("m,n,p,F,nw,lag,bins", [
    (64, 1000, 8, 256, 40, 9, [(0, 8), (8, 16), (16, 26), (26, 60), (60, 90)]),
    (64, 1000, 8, 32, 60, 8, [(0, 32), (3, 4), (5, 5), (30, 32), (0, 1), (7, 21), (16, 17)]),       # 7 bands: two passes
    (33, 300, 2, 64, 200, 8, [(0, 64), (10, 11)]),
    (19, 400, 3, 32, 150, 8, [(1, 31), (0, 2), (31, 32)]),
    (4, 160, 5, 640, 300, 8, [(0, 640), (100, 101), (320, 640)]),     # 16 padded channels: ONE band's weights per pass
    (48, 600, 4, 96, 200, 8, [(0, 96), (95, 96), (40, 41), (0, 0), (1, 2), (3, 96)]),
    (2, 200, 3, 32, 3, 8, [(0, 32)]),                                 # fewer windows than the lag: every row behind K3
    (64, 1000, 8, 64, 1, 8, [(k, k + 5) for k in range(0, 60, 5)])])  # one window, twelve bands (three passes)
def test_band_sums_inside_k3_equal_band_sums_of_the_full_array(m, n, p, F, nw, lag, bins):
    """The reduced product (`sliding_ffdtf(bands=...)`, hmv_sliding_ffdtf_bands_f64): K3's row workers add the frequency
    bands up from the published |H|^2 rows and the (items, m, m, F) array is never written -- the same BITS as
    `band_sums(sliding_ffdtf(...))`, for every window: the ones whose rows are reduced inside K3, the rows that come up
    early (missed-row list) and the last `lag` windows (K4 into scratch + the band-sum kernel).  Overlapping, empty,
    one-bin and full-grid bands; more bands than one pass holds; every padded size; the workspace reused twice; chunks."""
    from hyperscanning_signal_analysis_amd import _lib
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    assert eng.bands_in_kernel(m, F)
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(17, m=m, p=min(p, 4), T=T, burn=300)
    freqs = np.linspace(0.5, 120.0, F)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    lo, hi = [b[0] for b in bins], [b[1] for b in bins]
    full = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
    want = eng.band_sums(full, lo, hi)
    ref = full.cpu().numpy()
    for b, (l0, h0) in enumerate(bins):        # the band-sum kernel itself against NumPy
        assert np.allclose(want[..., b].cpu().numpy(), ref[..., l0:h0].sum(axis=-1), rtol=1e-12, atol=1e-300)
    assert eng.lib.hmv_set_tuning(_lib.TUNE_NORM_LAG, lag) == 0
    try:
        got = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, bands=(lo, hi))
        again = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, bands=(lo, hi))
        chunked = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, bands=(lo, hi), chunk=max(lag + 3, nw // 3 + 1))
        torch.cuda.synchronize()
    finally:
        assert eng.lib.hmv_set_tuning(_lib.TUNE_NORM_LAG, 0) == 0
    default_lag = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, bands=(lo, hi))
    assert got.shape == (nw, m, m, len(bins))
    assert torch.equal(got, want) and torch.equal(again, want) and torch.equal(chunked, want) and torch.equal(default_lag, want)


def test_band_sums_fall_back_to_two_calls_on_other_grids():
    """A grid that is not a multiple of 32 frequencies (or too long for the row worker's LDS block) cannot be reduced
    inside K3: `sliding_ffdtf(bands=...)` then computes the full array and sums it -- same result, and the C entry point
    itself refuses (-10) instead of computing something else."""
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    m, n, p, F, nw = 19, 400, 3, 48, 12
    assert not eng.bands_in_kernel(m, F) and not eng.bands_in_kernel(4, 672) and eng.bands_in_kernel(4, 640)
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(18, m=m, p=3, T=T, burn=300)
    freqs = np.linspace(0.5, 120.0, F)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    lo, hi = [0, 10], [48, 20]
    got = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, bands=(lo, hi))
    want = eng.band_sums(eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0), lo, hi)
    assert torch.equal(got, want)
    ar = eng.empty(2, 32, 32, p)
    tw = eng.twiddles(freqs, 500.0, p)
    b_lo, b_hi = eng.band_tables(lo, hi, F)
    nws = int(eng.lib.hmv_tf_ffdtf_bands_workspace_bytes(2, m, p, F))
    ws = eng.empty(nws // 8 + 1)
    rc = eng.lib.hmv_tf_ffdtf_bands_f64(ar.data_ptr(), 2, m, p, tw.data_ptr(), F, eng.empty(2, m, m, 2).data_ptr(),
                                        b_lo.data_ptr(), b_hi.data_ptr(), 2, eng.empty(2, 32).data_ptr(),
                                        eng.empty(2 * F, dtype=torch.int32).data_ptr(), 0.25, ws.data_ptr(), nws, 0, 0, 0,
                                        eng.stream())
    assert rc == -10 and b"F % 32" in eng.lib.hmv_last_error()


@pytest.mark.parametrize("m,p,F,scale", [(64, 8, 32, 0.05), (64, 2, 16, 3.0), (50, 5, 16, 2.0), (64, 16, 16, 0.3),
                                          (64, 1, 16, 5.0), (57, 3, 48, 1.0), (64, 7, 16, 0.5)])
def test_hand_scheduled_k3_body_equals_compiler_body(m, p, F, scale):
    """The 64-channel body of K3 with hand-allocated registers (csrc/gen/k3gen.py, five workgroups per CU) against the
    compiler-scheduled body (four per CU): H, |H|^2, the row sums and info are the same BITS -- same A(f) build (every
    order: full chunks of eight lags, single lag pairs, the zero padding lag of an odd order), same pivot choices, same
    summation order.  Large coefficients (scale >= 1) make nearly every pivot column interchange rows, which takes the
    stream through its search path, the displaced-row path and the other waves' row swaps; and both agree with NumPy."""
    from hyperscanning_signal_analysis_amd import _lib
    eng = default_engine()
    mp = eng.pad(m)
    assert mp == 64
    rng = np.random.default_rng(100 * p + F)
    items = 5
    ar = np.zeros((items, mp, mp, p))
    ar[:, :m, :m, :] = scale * rng.standard_normal((items, m, m, p)) / np.sqrt(m)
    ar[:, np.arange(m), np.arange(m), 0] += 0.4
    freqs = np.linspace(1.0, 200.0, F)
    tw = eng.twiddles(freqs, 500.0, p)
    ard = eng.to_device(ar)
    outs = {}
    try:
        for form in (1, 2):
            assert eng.lib.hmv_set_tuning(_lib.TUNE_K3_FORM, form) == 0
            outs[form] = eng.transfer(ard, m, tw, want_P=True, want_H=True)
            torch.cuda.synchronize()
    finally:
        assert eng.lib.hmv_set_tuning(_lib.TUNE_K3_FORM, 0) == 0
    a, b = outs[1], outs[2]
    assert not bool(a["info"].any()) and not bool(b["info"].any())
    for k in ("H", "P", "rowsum", "info"):
        assert torch.equal(a[k], b[k]), k
    H = torch.view_as_complex(b["H"])[2, :, :m, :m].cpu().numpy()
    z = np.exp(-(np.arange(p) + 1) * 2 * np.pi * 1j * freqs[:, None] / 500.0)          # (F, p)
    A = np.eye(m)[None] - np.einsum("ijk,fk->fij", ar[2, :m, :m, :], z)
    want = np.linalg.inv(A)
    assert np.abs(H - want).max() / np.abs(want).max() < 1e-9
    if scale >= 1.0:                    # the interchange paths really ran: the inverse of a non-dominant matrix
        assert (np.abs(A[0]).argmax(axis=0) != np.arange(m)).any()


def test_hand_scheduled_k3_body_in_the_fused_path():
    """the whole sliding-window call (ffDTF normalised inside K3) with either body: same bits"""
    from hyperscanning_signal_analysis_amd import _lib
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    x = synthetic_var_dyad(17, m=64, p=4, T=16_000, burn=300)
    freqs = northstar_freqs()
    xd = eng.to_device(x[None])
    pos, w = window_positions(16_000, 31, 1000)
    rec, st = window_items(1, pos, eng.device)
    outs = {}
    try:
        for form in (1, 2):
            assert eng.lib.hmv_set_tuning(_lib.TUNE_K3_FORM, form) == 0
            outs[form] = eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0)
            torch.cuda.synchronize()
    finally:
        assert eng.lib.hmv_set_tuning(_lib.TUNE_K3_FORM, 0) == 0
    assert torch.equal(outs[1], outs[2])
    assert float((outs[2].sum(dim=(2, 3)) - 1).abs().max()) < 1e-12


@pytest.mark.parametrize("m,n,p", [(64, 1000, 8), (64, 700, 3), (50, 900, 5), (33, 500, 2), (19, 400, 6), (16, 300, 1),
                                   (5, 200, 9)])
def test_yule_walker_one_launch_equals_tiled_launch_chain(m, n, p):
    """K2 as one workgroup per window (one launch, operands staged in k-halves) walks the same tile products in
    the same order as the round-1 chain of ~50 tile launches: coefficients, residual covariance and the log
    determinants of the lower orders are the same bits; and both agree with the oracle."""
    from hyperscanning_signal_analysis_amd import _lib
    eng = default_engine()
    W = 37
    x = synthetic_var_dyad(21, m=m, p=min(p, 4), T=n + 10 * (W - 1), burn=300)
    xd = eng.to_device(x[None])
    rec = torch.zeros(W, dtype=torch.int64, device=eng.device)
    st = 10 * torch.arange(W, dtype=torch.int64, device=eng.device)
    R = eng.lagcov(xd, rec, st, n, p)
    for want_logdet in (False, True):
        a1, v1, l1, i1 = eng.yw_solve(R, m, want_logdet, flags=_lib.FLAG_YW_ONE_LAUNCH)
        a2, v2, l2, i2 = eng.yw_solve(R, m, want_logdet, flags=_lib.FLAG_YW_TILED)
        torch.cuda.synchronize()
        assert torch.equal(a1, a2) and torch.equal(v1, v2) and not bool(i1.any()) and not bool(i2.any())
        if want_logdet:
            assert torch.equal(l1, l2)
    aro, Vo = O.ar_coeff(x[:, 50:50 + n], p)
    assert_parity(a1[5, :m, :m].cpu().numpy(), aro, 1e-8); assert_parity(v1[5, :m, :m].cpu().numpy(), Vo, 1e-8)


@pytest.mark.parametrize("m,n,p", [(64, 1000, 8), (64, 700, 3), (50, 900, 5), (33, 500, 2), (19, 400, 6), (16, 300, 1),
                                   (5, 200, 9), (64, 1200, 16)])
def test_levinson_whittle_solver_against_the_block_ldlt_and_the_oracle(m, n, p):
    """K2's default form -- the block Levinson-Whittle recursion on the p + 1 lag blocks (csrc/yw_lwr.hip) -- against the
    block LDL^T of the augmented matrix (csrc/yw_solve.hip) and against the oracle's dense solve: coefficients, residual
    covariance and the log determinants of every lower order (the criterion's terms, here straight from the recursion's
    own forward error covariances) agree to ~1e-12 on well-conditioned windows; no window trips the guard."""
    from hyperscanning_signal_analysis_amd import _lib
    eng = default_engine()
    W = 29
    x = synthetic_var_dyad(41, m=m, p=min(p, 4), T=n + 10 * (W - 1), burn=300)
    xd = eng.to_device(x[None])
    rec = torch.zeros(W, dtype=torch.int64, device=eng.device)
    st = 10 * torch.arange(W, dtype=torch.int64, device=eng.device)
    R = eng.lagcov(xd, rec, st, n, p)
    a0, v0, l0, i0 = eng.yw_solve(R, m, True)                                   # default: the recursion
    a1, v1, l1, i1 = eng.yw_solve(R, m, True, flags=_lib.FLAG_YW_ONE_LAUNCH)      # block LDL^T
    a2, v2, _, i2 = eng.yw_solve(R, m, False)
    torch.cuda.synchronize()
    assert not bool(i0.any()) and not bool(i1.any()) and not bool(i2.any())
    assert torch.equal(a0, a2) and torch.equal(v0, v2)                          # with / without the log determinants
    assert not torch.equal(a0, a1)                                              # really two different algorithms
    # the block LDL^T (explicit tile inverses, p back-substitution sweeps) loses accuracy with the order -- 5e-11 at
    # p = 9, 2e-9 at p = 12, 6e-7 at p = 16 against the oracle -- where the recursion stays at ~5e-12
    tol = 1e-10 if p <= 9 else 1e-5
    for got, want in ((a0, a1), (v0, v1), (l0, l1)):
        assert float((got - want).abs().max() / want.abs().max()) < tol
    for k in (5, W - 1):
        aro, Vo = O.ar_coeff(x[:, 10 * k:10 * k + n], p)
        assert_parity(a0[k, :m, :m].cpu().numpy(), aro, 1e-9); assert_parity(v0[k, :m, :m].cpu().numpy(), Vo, 1e-9)
    crit = O.mvar_criterion(x[:, 50:50 + n], p, "AIC")[0]
    pen = 2.0 * np.arange(1, p + 1) * m * m / n
    assert np.allclose(l0[5].cpu().numpy() + pen, crit, rtol=1e-9, atol=1e-9)


def test_levinson_whittle_guard_re_solves_ill_conditioned_windows(golden):
    """Levinson-type recursions lose accuracy with the conditioning of the error covariances they invert.  Every tile
    inverse reports its extreme pivots; a window in which their ratio falls below 1e-7 is re-solved by the block LDL^T in
    the same call.  The nearly collinear fixtures (cond 2e9, 2e13) come out with the LDL^T's bits, the cond-2e5 fixture and
    an ordinary window with the recursion's; the exactly rank-deficient window is still reported singular."""
    from hyperscanning_signal_analysis_amd import _lib
    g = golden("g6_errors.npz")
    eng = default_engine()
    m, n = g["nc0_x"].shape
    p = g["nc0_ar"].shape[2]
    ordinary = synthetic_var_dyad(43, m=m, p=min(p, 4), T=n, burn=300)
    batch = np.stack([g["nc0_x"], ordinary, g["nc1_x"], g["nc2_x"], g["xs"]])
    xd = eng.to_device(batch)
    W = batch.shape[0]
    rec = torch.arange(W, dtype=torch.int64, device=eng.device)
    st = torch.zeros(W, dtype=torch.int64, device=eng.device)
    R = eng.lagcov(xd, rec, st, n, p)
    a0, v0, _, i0 = eng.yw_solve(R, m)
    a1, v1, _, i1 = eng.yw_solve(R, m, flags=_lib.FLAG_YW_ONE_LAUNCH)
    torch.cuda.synchronize()
    # cond 2e9: guarded, the LDL^T's result bit for bit
    assert torch.equal(a0[2], a1[2]) and torch.equal(v0[2], v1[2]) and int(i0[2]) == int(i1[2]) == 0
    for k in (0, 1):                       # not guarded: the recursion's own result, close to the LDL^T's
        assert not torch.equal(a0[k], a1[k]) and int(i0[k]) == 0
    # cond 2e13 and the exactly rank-deficient window: a non-positive pivot in either solver -> singular, same code
    assert int(i0[3]) == int(i1[3]) and int(i0[4]) == int(i1[4]) != 0
    eps = 2.2e-16
    assert rel(a0[0, :m, :m].cpu().numpy(), g["nc0_ar"]) < 1e2 * float(g["nc0_cond"]) * eps
    assert rel(a0[2, :m, :m].cpu().numpy(), g["nc1_ar"]) < 1e2 * float(g["nc1_cond"]) * eps


@pytest.mark.parametrize("m,n,hop,p,T,first", [(64, 1000, 500, 8, 6000, 0), (64, 1000, 250, 8, 4250, 250), (19, 90, 45, 3, 1000, 10),
                                             (5, 66, 33, 2, 400, 4), (33, 512, 64, 6, 2048, 0), (48, 300, 100, 4, 1500, 200)])
def test_lag_covariances_from_shared_hop_blocks(m, n, hop, p, T, first):
    """K1 on a regular grid (hop blocks summed once, windows = k blocks minus the products that reach past the window
    end) against K1 window by window: same estimator (biased 1/n, not demeaned), sums associated differently --
    equal to rounding; the last window may end exactly at the end of the recording; padded channels keep the
    identity block; hops that are not a multiple of the 4-sample MFMA step are masked correctly."""
    eng = default_engine()
    x = synthetic_var_dyad(31, m=m, p=min(p, 4), T=T, burn=200)
    xd = eng.to_device(x[None])
    n_win = (T - first - n) // hop + 1
    assert first + (n_win - 1) * hop + n <= T
    st = first + hop * torch.arange(n_win, dtype=torch.int64, device=eng.device)
    rec = torch.zeros(n_win, dtype=torch.int64, device=eng.device)
    direct = eng.lagcov(xd, rec, st, n, p)
    shared = eng.lagcov_regular(xd[0], first, hop, n_win, n, p)
    assert shared.shape == direct.shape
    assert float((shared - direct).abs().max() / direct.abs().max()) < 1e-13
    mp = direct.shape[-1]
    if mp > m:
        assert torch.equal(shared[:, 0, m:, m:], torch.eye(mp - m, dtype=torch.float64, device=eng.device).expand(n_win, -1, -1))
        assert not bool(shared[:, 1:, m:, :].any()) and not bool(shared[:, :, :m, m:].any())
    assert_parity(shared[n_win - 1, :, :m, :m].cpu().numpy(), O.lag_covariances(x[:, first + (n_win - 1) * hop:][:, :n], p), 1e-12)
    with pytest.raises(ValueError):
        eng.lagcov_regular(xd[0], first, hop, n_win + 1 + (T - first - n) // hop, n, p)       # past the recording


@pytest.mark.parametrize("m,n,p,F,nw,chunk", [(64, 1000, 8, 32, 40, 33), (19, 400, 3, 20, 9, 4), (4, 160, 5, 30, 5, 64),
                                               (64, 1000, 8, 64, 12, None), (48, 600, 4, 64, 20, 7), (33, 300, 2, 128, 12, None)])
def test_ffdtf_and_spectra_of_every_window_from_one_fit(m, n, p, F, nw, chunk):
    """`Engine.sliding_ffdtf_spectra` (hmv_sliding_ffdtf_spectra_f64): what the reference's orchestrators compute per
    window with two separate fits (full_freq_dtf + multivariate_spectra, eeg_alpha_ibi_ffdtf.py:592-604) from ONE fit and
    ONE set of inverses, batched.  ffDTF equals the ffDTF-only path bit for bit (same K1 form, with and without a declared
    grid), spectra equal the oracle's H V H^T (plain transpose); K5 computes the upper triangle only (V is the fit's own
    symmetric estimate) and mirrors it, so S is EXACTLY symmetric where the reference's is symmetric to rounding; every
    padded size, grids that are and are not a multiple of 64 frequencies (the (m, m, F) store map differs)."""
    from hyperscanning_signal_analysis_amd.sliding import regular_grid, window_items, window_positions
    eng = default_engine()
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(17, m=m, p=min(p, 4), T=T, burn=300)
    freqs = np.linspace(1.0, 100.0, F)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    ff, S = eng.sliding_ffdtf_spectra(xd, rec, st, w, p, freqs, 500.0, chunk=chunk)
    only = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)               # no grid declared: same (direct) K1
    torch.cuda.synchronize()
    assert ff.shape == (nw, m, m, F) and S.shape == (nw, m, m, F) and S.dtype == torch.complex128
    assert torch.equal(ff, only)
    assert torch.equal(S, S.transpose(1, 2))
    for k in (0, nw // 2, nw - 1):
        wk = x[:, pos[k]:pos[k] + w]
        assert_parity(S[k].cpu().numpy(), O.multivariate_spectra(wk, freqs, 500.0, p), 1e-8)
        assert_parity(ff[k].cpu().numpy(), O.full_freq_dtf(wk, freqs, 500.0, p), 1e-8)
    grid = regular_grid(pos, w, p)
    if grid is not None:                        # K1 from shared hop blocks: the same path as sliding_ffdtf(grid=...)
        ffg, Sg = eng.sliding_ffdtf_spectra(xd, rec, st, w, p, freqs, 500.0, chunk=chunk, grid=grid)
        assert torch.equal(ffg, eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, grid=grid))
        assert float((Sg - S).abs().max() / S.abs().max()) < 1e-10 and torch.equal(Sg, Sg.transpose(1, 2))


def test_in_kernel_normalisation_with_changing_data_in_the_same_workspace():
    """The row workers of the in-K3 normalisation read |H|^2 that OTHER workgroups wrote earlier in the same launch,
    without an acquire fence (write-through stores, device-scope flag, non-temporal loads of lines read once per
    launch).  A stale cache line from an EARLIER launch or chunk would go unnoticed if that launch had written the
    same values -- so here consecutive launches and consecutive chunks reuse the same workspace addresses with
    DIFFERENT recordings, in both orders, and every result must equal the separate-pass path bit for bit."""
    from hyperscanning_signal_analysis_amd import _lib
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    eng = default_engine()
    T, w, p = 20_000, 1000, 8
    freqs = northstar_freqs(64)
    xs = [synthetic_var_dyad(d, T=T) for d in (40, 41, 42)]
    pos, w = window_positions(T, 2 * T // w - 1, w)                       # 39 windows: 15 normalised in-kernel
    rec, st = window_items(1, pos, eng.device)
    xd = [eng.to_device(x[None]) for x in xs]
    ref = [eng.sliding_ffdtf(d, rec, st, w, p, freqs, 500.0, flags=_lib.FLAG_UNFUSED_NORM).clone() for d in xd]
    for order in ((0, 1, 2), (2, 0, 1), (1, 1, 0), (0, 2, 2)):
        got = [eng.sliding_ffdtf(xd[k], rec, st, w, p, freqs, 500.0) for k in order]     # back to back, one stream
        torch.cuda.synchronize()
        for k, g in zip(order, got):
            assert torch.equal(g, ref[k])
    # three recordings in one call, chunks of one recording each: the workspace is rewritten per chunk
    x3 = eng.to_device(np.stack(xs))
    rec3, st3 = window_items(3, pos, eng.device)
    both = eng.sliding_ffdtf(x3, rec3, st3, w, p, freqs, 500.0, chunk=len(pos))
    torch.cuda.synchronize()
    assert torch.equal(both.view(3, len(pos), 64, 64, 64), torch.stack(ref))
    assert not torch.equal(ref[0], ref[1])


def test_multi_dyad_batch_matches_single_dyad_runs():
    """Config 3 in miniature: several dyads in one batch (dyad x window items, forced into several chunks)
    give bit-identical results to running each dyad alone -- the property dyad-sharding across GPUs relies on."""
    eng = default_engine()
    fs, w, p, T = 500.0, 1000, 8, 10_000
    freqs = northstar_freqs(32)
    xs = np.stack([synthetic_var_dyad(d, T=T) for d in (0, 1, 2)])
    xd = eng.to_device(xs)
    nw = 2 * T // w - 1
    from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
    pos, w = window_positions(T, nw, w)
    rec, st = window_items(3, pos, eng.device)
    batch = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, fs, chunk=7).view(3, nw, 64, 64, 32)
    for d in range(3):
        alone = sliding_ffdtf_device(xd[d:d + 1].contiguous(), w, nw, p, freqs, fs, eng, share_overlap=False)[0]
        assert torch.equal(alone, batch[d])
    # the same batch on the declared regular grid (K1 shares the overlap; chunks of 7 cut through the recordings)
    from hyperscanning_signal_analysis_amd.sliding import regular_grid
    g = regular_grid(pos, w, p)
    assert g == (500, 0, nw)
    shared = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, fs, chunk=7, grid=g).view(3, nw, 64, 64, 32)
    shared1 = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, fs, grid=g).view(3, nw, 64, 64, 32)
    assert torch.equal(shared, shared1)                               # chunking does not change the block sums
    assert float((shared - batch).abs().max() / batch.abs().max()) < 1e-11
    ref = O.full_freq_dtf(xs[2][:, 500 * 5:500 * 5 + w], freqs, fs, p)
    assert_parity(batch[2, 5].cpu().numpy(), ref)


def test_edge_cases_empty_single_and_ragged_batches():
    """Empty batch, one window, a chunk larger than the batch, windows of several recordings with irregular
    starts (what `_create_windows` produces when n_windows does not divide the recording), F = 1."""
    eng = default_engine()
    x = synthetic_var_dyad(3, T=6_000)
    xd = eng.to_device(np.stack([x, x[::-1].copy()]))                  # two recordings
    freqs = northstar_freqs(8)
    empty_rec = torch.zeros(0, dtype=torch.int64, device=eng.device)
    out0 = eng.sliding_ffdtf(xd, empty_rec, empty_rec, 1000, 4, freqs, 500.0)
    assert out0.shape == (0, 64, 64, 8)
    rec = torch.tensor([0, 1, 1, 0, 0], dtype=torch.int64, device=eng.device)
    st = torch.tensor([0, 17, 4999, 3333, 5000], dtype=torch.int64, device=eng.device)     # last ones end at T
    a = eng.sliding_ffdtf(xd, rec, st, 1000, 4, freqs, 500.0, chunk=64)
    b = eng.sliding_ffdtf(xd, rec, st, 1000, 4, freqs, 500.0, chunk=2)
    assert torch.equal(a, b)
    xs = np.stack([x, x[::-1].copy()])
    for k in range(5):
        r, s = int(rec[k]), int(st[k])
        assert_parity(a[k].cpu().numpy(), O.full_freq_dtf(xs[r][:, s:s + 1000], freqs, 500.0, 4), 1e-8)
    one = eng.sliding_ffdtf(xd, rec[:1], st[:1], 1000, 4, freqs[:1], 500.0)                 # one window, F = 1
    assert one.shape == (1, 64, 64, 1)
    assert_parity(one[0].cpu().numpy(), O.full_freq_dtf(xs[0][:, :1000], freqs[:1], 500.0, 4), 1e-8)


def test_connectivity_kernels_batched_and_singular():
    """The rank-4 kernels over a batch of windows equal the per-window results bit for bit, and a singular
    spectral matrix raises like np.linalg (the reference would divide by a zero minor product instead)."""
    eng = default_engine()
    rng = np.random.default_rng(12)
    m, F, W = 6, 5, 4
    freqs = np.linspace(2.0, 30.0, F)
    x = rng.standard_normal((W, m, 500))
    x[:, :, 1:] += 0.5 * x[:, :, :-1]
    xd = eng.to_device(x)
    rec = torch.arange(W, dtype=torch.int64, device=eng.device)
    st = torch.zeros(W, dtype=torch.int64, device=eng.device)
    R = eng.lagcov(xd, rec, st, 500, 3)
    ar, V, _, _ = eng.yw_solve(R, m)
    t = eng.transfer(ar, m, eng.twiddles(freqs, 100.0, 3), want_P=False, want_H=True, want_A=True)
    S = eng.spectra(t["H"], V, m)
    kap, info = eng.partial_coherence(S, m)
    g = eng.gpdc(t["A"], V, m)
    assert kap.shape == (W, m, m, F) and g.shape == (W, m, m, F) and not bool(info.any())
    for k in range(W):
        res = M.mvar_analysis(x[k], freqs, 100.0, 3, want=("pcoh", "gpdc"))
        assert np.array_equal(kap[k].cpu().numpy(), res["pcoh"]) and np.array_equal(g[k].cpu().numpy(), res["gpdc"])
    Z = np.ones((3, 3, 2), dtype=complex)                     # rank one: every 2x2 minor vanishes
    with pytest.raises(np.linalg.LinAlgError):
        M.partial_coherence(Z)
