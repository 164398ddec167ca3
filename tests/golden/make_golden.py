"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (`/root/reference/src/mtmvar.py`, `/root/reference/src/eeg_alpha_ibi_ffdtf.py`) is
imported read-only; only seeded inputs and the reference's OUTPUTS are written (as .npz data).
Nothing of the reference's source travels.  `/root/reference` does not exist on the GPU box, so
this script is never run there -- the tests read the committed .npz files.

Fixture list:
  g1_config1.npz    3-ch AR(4), n=4000, fs=128, freqs=arange(1,41,0.5): ar, V, H, A, dtf, ffdtf,
                    spectra, AIC/HQ/SC curves (pmax=10), count_corr blocks.
  g2_northstar.npz  one north-star window m=64, n=1000, p=8, F=256: R_0..R_8, ar, V,
                    ffdtf[:, :, ::16], row denominators, float64 checksums, spectra[:, :, ::64].
  g3_overlap.npz    three consecutive 50%-overlap windows of the g2 recording: ar + ffdtf[:, :, ::32].
  g4_config4.npz    4x480 z-scored block, `_create_windows(.., 3, None)` and `(.., 5, 160)`,
                    per-window + global ff_dtf / spectra at p=5, freq grid quirk Q9.
  g5_multitrial.npz multi-trial input (5, 400, 4): r_left, r_right, r, ar, V.
  g6_errors.npz     inputs that make the reference raise (dead channel), a rank-deficient window on which it does
                    NOT raise (what it returns is recorded), three nearly collinear windows (cond 1e5 .. 1e13) with
                    the reference's ar / V, plus window-geometry error cases (recorded as message strings).
                    `python tests/golden/make_golden.py g6` writes only this.
  g7_connectivity.npz  partial_coherence / direct_dtf / gen_partial_directed_coherence on
                    the G1 signal (m=3, p=4), a 4x480 block (p=5), a 7-channel VAR(3), and partial_coherence of
                    an arbitrary complex 5x5x6 array.  `python tests/golden/make_golden.py g7` writes only this.
  g8_faa_chain.npz  the pipeline's scalar DSP chain on seeded 19 x 10240 noise @128 Hz + a 4 Hz IBI series:
                    alpha band-pass, FAA (amp and power), downsampling to 8 Hz, crop, z-scored 4 x 480 block and
                    the reference's _compute_ffDTF on it.  `python tests/golden/make_golden.py g8`.
"""
import io
import os
import sys
import types
from contextlib import redirect_stdout

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from src import mtmvar as ref  # noqa: E402  (the reference itself)

# eeg_alpha_ibi_ffdtf imports xarray at module level (absent here); an empty stub lets the
# numeric methods import.  Only _create_windows / _compute_ffDTF are exercised.
sys.modules.setdefault("xarray", types.ModuleType("xarray"))
from src import eeg_alpha_ibi_ffdtf as ref_pipe  # noqa: E402

from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad  # noqa: E402  (workload generator only)


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def small_var(seed, m, p, n, scale=0.4):
    """Seeded stable VAR(p) for the small configs (own generator, not reference code)."""
    rng = np.random.default_rng(seed)
    A = scale * rng.standard_normal((p, m, m)) / (np.arange(1, p + 1)[:, None, None] * np.sqrt(m))
    comp = np.zeros((m * p, m * p))
    comp[:m] = np.concatenate(list(A), axis=1)
    if p > 1:
        comp[m:, :-m] = np.eye(m * (p - 1))
    rho = np.max(np.abs(np.linalg.eigvals(comp)))
    if rho > 0.9:
        A *= ((0.9 / rho) ** np.arange(1, p + 1))[:, None, None]
    burn = 500
    e = rng.standard_normal((n + burn, m))
    x = np.zeros((n + burn, m))
    for t in range(p, n + burn):
        x[t] = e[t] + sum(A[k] @ x[t - k - 1] for k in range(p))
    return np.ascontiguousarray(x[burn:].T)


def make_g7():
    g = {}
    cases = {"a": (small_var(11, 3, 4, 4000), 128.0, np.arange(1, 41, 2.0), 4),
             "b": (small_var(44, 4, 2, 480), 8.0, np.linspace(0.0, 4.0, 30, endpoint=False)[1:], 5),
             "c": (small_var(77, 7, 3, 800), 100.0, np.linspace(1.0, 45.0, 9), 3)}
    for tag, (x, fs, freqs, p) in cases.items():
        S = quiet(ref.multivariate_spectra, x, freqs, fs, optimal_model_order=p)
        g[f"x_{tag}"], g[f"fs_{tag}"], g[f"freqs_{tag}"], g[f"p_{tag}"] = x, fs, freqs, p
        g[f"spectra_{tag}"] = S
        g[f"pcoh_{tag}"] = ref.partial_coherence(S)
        g[f"ddtf_{tag}"] = quiet(ref.direct_dtf, x, freqs, fs, optimal_model_order=p)
        g[f"gpdc_{tag}"] = quiet(ref.gen_partial_directed_coherence, x, freqs, fs, optimal_model_order=p)
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((5, 5, 6)) + 1j * rng.standard_normal((5, 5, 6))
    g["Z"], g["pcoh_Z"] = Z, ref.partial_coherence(Z)
    g["pcoh_1x1"] = ref.partial_coherence(np.full((1, 1, 3), 2.0 + 1.0j))
    np.savez_compressed(os.path.join(HERE, "g7_connectivity.npz"), **g)
    print("g7_connectivity.npz written")


def make_g8():
    P = ref_pipe.EEG_IBI_FFDTF_Pipeline
    pipe = P.__new__(P)                 # skip the folder scan of __init__; set what the helpers read
    pipe.left_chan, pipe.right_chan = "F3", "F4"
    pipe.fs_ds, pipe.freq_min, pipe.freq_step = 8.0, 1.0, 0.1
    pipe.freq_max = (pipe.fs_ds / 2) - pipe.freq_step
    pipe.plot_global_enabled = pipe.save_global_enabled = False
    pipe.plot_windowed_enabled = pipe.save_windowed_enabled = False
    from g8_inputs import FS_EEG, NAMES, g8_inputs
    inp = g8_inputs()
    g, rows = {}, []
    for role in ("ch", "cg"):
        eeg = inp[f"eeg_{role}"]
        filt = quiet(pipe._alpha_bandpass_filter, eeg, FS_EEG)
        faa = quiet(pipe._compute_asymmetry, filt, NAMES, metric="amp")
        faa_pow = quiet(pipe._compute_asymmetry, filt, NAMES, metric="power")
        faa_ds = quiet(pipe._downsample_signal, faa, FS_EEG, 8.0)
        faa_c = quiet(pipe._crop_signal, faa_ds, 8.0, drop_front_sec=10, keep_duration_sec=60)
        g.update({f"filt_{role}": filt[[2, 3]], f"faa_{role}": faa, f"faa_pow_{role}": faa_pow,
                  f"faa_ds_{role}": faa_ds, f"faa_crop_{role}": faa_c})
        rows += [faa_c, inp[f"ibi8_{role}"]]
    block = np.vstack(rows)
    block = (block - np.mean(block, axis=1, keepdims=True)) / np.std(block, axis=1, keepdims=True)
    g["block"] = block
    pipe.ar_p = 5
    ff, sp, p_opt = quiet(pipe._compute_ffDTF, "W_000", block, ["faa_ch", "ibi_ch", "faa_cg", "ibi_cg"], 8.0,
                          plot=False, save_plot=False)
    g["ff_block"], g["sp_block"], g["p_block"] = ff, sp, p_opt
    np.savez_compressed(os.path.join(HERE, "g8_faa_chain.npz"), **g)
    print("g8_faa_chain.npz written")


def make_g6(pipe=None, x4=None):
    """Error behaviour and ill-conditioned windows.  What the reference's dgesv RETURNS on a rank-deficient or nearly
    collinear window is recorded too (xs_ar / xs_V, nc*_ar / nc*_V with the condition number of r_left): on the
    exactly deficient window the numbers are rounding noise amplified by 1e16 and are recorded as evidence only."""
    if pipe is None:
        pipe = ref_pipe.EEG_IBI_FFDTF_Pipeline.__new__(ref_pipe.EEG_IBI_FFDTF_Pipeline)
    if x4 is None:
        x4 = small_var(44, 4, 5, 480)
    g6 = {}
    xs = small_var(66, 4, 2, 300)
    xs[3] = xs[0] + xs[1]                    # exactly rank-deficient window
    try:
        ar_s, V_s = ref.ar_coeff(xs, 3)
        g6["singular_raises"] = "no"
        g6["xs_ar"], g6["xs_V"] = ar_s, V_s
        g6["xs_cond"] = np.linalg.cond(ref.count_corr(xs[:, :, None], 3, 1)[0])
    except np.linalg.LinAlgError as e:
        g6["singular_raises"] = f"LinAlgError:{e}"
    # nearly collinear channels (bridged electrodes / re-referenced montages): channel 3 = ch0 + ch1 + eps * noise
    rng = np.random.default_rng(660)
    noise = rng.standard_normal(300)
    for k, eps in enumerate((1e-2, 1e-4, 1e-6)):
        xn = small_var(66, 4, 2, 300)
        xn[3] = xn[0] + xn[1] + eps * noise
        ar_n, V_n = ref.ar_coeff(xn, 3)
        g6[f"nc{k}_x"], g6[f"nc{k}_ar"], g6[f"nc{k}_V"] = xn, ar_n, V_n
        g6[f"nc{k}_cond"] = np.linalg.cond(ref.count_corr(xn[:, :, None], 3, 1)[0])
        g6[f"nc{k}_eps"] = eps
    xz = small_var(67, 4, 2, 300)
    xz[2] = 0.0                               # dead channel -> exactly singular normal equations
    try:
        ref.ar_coeff(xz, 3)
        g6["deadchan_raises"] = "no"
    except np.linalg.LinAlgError as e:
        g6["deadchan_raises"] = f"LinAlgError:{e}"
    try:
        ref.mvar_criterion(x4, 3, "BIC", False)
        g6["badcrit"] = "no"
    except ValueError as e:
        g6["badcrit"] = f"ValueError:{e}"
    msgs = []
    for (T, nw, ws) in [(481, 3, None), (480, 3, 100), (480, 3, 481), (480, 5, 478)]:
        try:
            pipe._create_windows(np.zeros((2, T)), nw, ws)
            msgs.append("ok")
        except ValueError as e:
            msgs.append(str(e))
    g6["window_errors"] = np.array(msgs)
    np.savez_compressed(os.path.join(HERE, "g6_errors.npz"), xs=xs, xz=xz, **g6)
    print("g6_errors.npz written:", {k: (float(g6[k]) if k.endswith("cond") else None) for k in g6 if k.endswith("cond")})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        make_g7()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        make_g8()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g6":
        make_g6()
        return
    out = {}
    # ------------------------------------------------------------------ G1
    x = small_var(11, 3, 4, 4000)
    fs = 128.0
    freqs = np.arange(1, 41, 0.5)
    ar, V = ref.ar_coeff(x, 4)
    H, A = ref.mvar_transfer_function(ar, freqs, fs)
    dtf = quiet(ref.dtf_multivariate, x, freqs, fs, optimal_model_order=4)
    ff = quiet(ref.full_freq_dtf, x, freqs, fs, optimal_model_order=4)
    S = quiet(ref.multivariate_spectra, x, freqs, fs, optimal_model_order=4)
    rl, rr, r0 = ref.count_corr(x[:, :, None], 4, 1)
    crit = {}
    for c in ("AIC", "HQ", "SC"):
        cv, rng_, popt = ref.mvar_criterion(x, 10, c, False)
        crit[c] = cv
        crit[c + "_popt"] = int(popt)
    ff_auto = quiet(ref.full_freq_dtf, x, freqs, fs, max_model_order=10, optimal_model_order=None, crit_type="AIC")
    np.savez_compressed(os.path.join(HERE, "g1_config1.npz"), x=x, fs=fs, freqs=freqs, ar=ar, V=V, H=H, A=A,
                        dtf=dtf, ffdtf=ff, spectra=S, r_left=rl, r_right=rr, r=r0, ffdtf_auto=ff_auto,
                        **{f"crit_{k}": v for k, v in crit.items()})
    # ------------------------------------------------------------------ G2 / G3
    rec = synthetic_var_dyad(0, T=4000)            # first 4000 samples' worth of dyad 0 (own seed stream)
    fs = 500.0
    freqs = 0.5 * np.arange(1, 257)
    p = 8
    w0 = rec[:, :1000]
    rl, rr, r0 = ref.count_corr(w0[:, :, None], p, 1)
    R = np.concatenate([r0[None], rr.reshape(p, 64, 64)], axis=0)
    ar, V = ref.ar_coeff(w0, p)
    ff = quiet(ref.full_freq_dtf, w0, freqs, fs, optimal_model_order=p)
    dtf = quiet(ref.dtf_multivariate, w0, freqs, fs, optimal_model_order=p)
    S = quiet(ref.multivariate_spectra, w0, freqs, fs, optimal_model_order=p)
    np.savez_compressed(os.path.join(HERE, "g2_northstar.npz"), x=w0, fs=fs, freqs=freqs, p=p, R=R, ar=ar, V=V,
                        ffdtf_sub=ff[:, :, ::16], den=dtf.sum(axis=(1, 2)),
                        ffdtf_sum=ff.sum(), ffdtf_sqsum=(ff ** 2).sum(), ffdtf_fsum=ff.sum(axis=2),
                        spectra_sub=S[:, :, ::64])
    g3 = {}
    for i, s in enumerate((0, 500, 1000)):
        w = rec[:, s:s + 1000]
        a_, _ = ref.ar_coeff(w, p)
        f_ = quiet(ref.full_freq_dtf, w, freqs, fs, optimal_model_order=p)
        g3[f"ar{i}"] = a_
        g3[f"ffdtf_sub{i}"] = f_[:, :, ::32]
        g3[f"ffdtf_fsum{i}"] = f_.sum(axis=2)
    np.savez_compressed(os.path.join(HERE, "g3_overlap.npz"), x=rec[:, :2000], fs=fs, freqs=freqs, p=p, **g3)
    # ------------------------------------------------------------------ G4
    x4 = small_var(44, 4, 3, 480, scale=0.8)
    x4 = (x4 - x4.mean(axis=1, keepdims=True)) / x4.std(axis=1, keepdims=True)
    pipe = ref_pipe.EEG_IBI_FFDTF_Pipeline.__new__(ref_pipe.EEG_IBI_FFDTF_Pipeline)
    pipe.fs_downsampled = 8.0
    pipe.freq_min, pipe.freq_step = 1.0, 0.1
    pipe.freq_max = pipe.fs_downsampled / 2 - 0.1
    pipe.ar_p = 5
    g4 = {"x": x4, "fs": 8.0,
          "freqs": np.arange(pipe.freq_min, pipe.freq_max + pipe.freq_step, pipe.freq_step)}
    for tag, (nw, ws) in {"a": (3, None), "b": (5, 160)}.items():
        wins = pipe._create_windows(x4, nw, ws)
        g4[f"starts_{tag}"] = np.array([
            next(s for s in range(480 - w.shape[1] + 1) if np.array_equal(x4[:, s:s + w.shape[1]], w))
            for w in wins])
        g4[f"wsize_{tag}"] = wins[0].shape[1]
        ffs, sps = [], []
        for w in wins:
            f_, s_, p_ = pipe._compute_ffDTF("D", w, list("abcd"), 8.0, plot=False)
            ffs.append(f_); sps.append(s_)
        g4[f"ff_{tag}"] = np.array(ffs)
        g4[f"sp_{tag}"] = np.array(sps)
    f_, s_, p_ = pipe._compute_ffDTF("D", x4, list("abcd"), 8.0, plot=False)
    g4["ff_global"], g4["sp_global"], g4["p_opt"] = f_, s_, p_
    pipe.ar_p = None
    f_, s_, p_ = pipe._compute_ffDTF("D", x4, list("abcd"), 8.0, max_model_order=20, crit_type="AIC", plot=False)
    g4["ff_global_auto"], g4["p_opt_auto"] = f_, int(p_)
    np.savez_compressed(os.path.join(HERE, "g4_config4.npz"), **g4)
    # ------------------------------------------------------------------ G5
    rng = np.random.default_rng(55)
    x5 = np.stack([small_var(500 + t, 5, 2, 400) for t in range(4)], axis=2)
    rl, rr, r0 = ref.count_corr(x5, 3, 1)
    ar, V = ref.ar_coeff(x5, 3)
    H, A = ref.mvar_transfer_function(ar, np.arange(1, 30.0), 64.0)
    np.savez_compressed(os.path.join(HERE, "g5_multitrial.npz"), x=x5, r_left=rl, r_right=rr, r=r0, ar=ar, V=V,
                        H=H, freqs=np.arange(1, 30.0), fs=64.0)
    make_g6(pipe, x4)
    make_g7()
    make_g8()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
