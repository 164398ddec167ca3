/* hypermvar -- C ABI of the MI355X (gfx950) sliding-window MVAR / ffDTF engine.
 *
 * This is the drop-in boundary for the hot path of the reference project
 * (SYNCC-IN/hyperscanning-signal-analysis, src/mtmvar.py:35-284, 551-601).  The reference is pure
 * Python/NumPy and has no FFI of its own; the "operator API" it exposes is the set of Python function
 * signatures in src/mtmvar.py.  Each entry point below names the reference function whose arithmetic it
 * replaces; `INTEGRATION.md` shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch-ROCm `tensor.data_ptr()`), float64 unless noted;
 *   - buffers are caller-allocated and caller-owned; the library never allocates, frees or retains them;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 = launched, < 0 = bad argument (see hmv_last_error()), > 0 = hipError_t;
 *   - numerical failures are reported per item through `info` arrays (LAPACK style: 0 = ok,
 *     k > 0 = zero / non-positive pivot met at column k); the Python layer turns them into
 *     numpy.linalg.LinAlgError("Singular matrix") like the reference's np.linalg.solve / inv;
 *   - channel counts are padded to MP = hmv_pad(m) = 16*ceil(m/16) <= 64 inside the library's
 *     intermediate buffers ("MP layout"); user-facing outputs are unpadded.
 */
#ifndef HYPERMVAR_H
#define HYPERMVAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMV_VERSION 120            /* 0.1.2 */
#define HMV_MAX_CHANNELS 64
#define HMV_MAX_ORDER 32

int hmv_version(void);
/* Thread-local text of the last argument error reported by this thread ("" if none). */
const char* hmv_last_error(void);
/* Padded channel count used by the MP-layout buffers, or -1 if m is unsupported (m < 1 or m > 64). */
int hmv_pad(int m);
/* Tuning knobs (process-wide; never needed for correct results -- every setting gives the same numbers up to the
 * documented bit-identity classes).  They replace the environment variables that earlier builds read inside the
 * launch path: the environment is looked at ONCE, when the library is loaded (HYPERMVAR_NORM_LAG,
 * HYPERMVAR_LAG_GROUP, HYPERMVAR_K3_FORM, HYPERMVAR_YW_FORM), and hmv_set_tuning overrides it afterwards.
 *   HMV_TUNE_NORM_LAG   windows between a K3 workgroup and the window whose rows it normalises (0 = built-in rule, >= 8)
 *   HMV_TUNE_LAG_GROUP  lags per K1 workgroup (1..3; 0 = default 3)
 *   HMV_TUNE_K3_FORM    0 = default, 1 = compiler-scheduled K3 body, 2 = hand-scheduled 64-channel K3 body
 *   HMV_TUNE_YW_FORM    0 = default, 1 = block LDL^T of the augmented matrix, 2 = block Levinson-Whittle recursion,
 *                       3 = the same recursion as a software pipeline (yw_lwr2.hip; equal in time, measurement form)
 *   HMV_TUNE_K3_LDS_PAD bytes of unused dynamic LDS added to every K3 workgroup (measurement only: fewer resident
 *                       workgroups per CU, to separate latency from throughput; 0 = none)
 * Returns 0, or -1 for an unknown key / value out of range.  hmv_get_tuning returns the value in force (-1: unknown key). */
#define HMV_TUNE_NORM_LAG 1
#define HMV_TUNE_LAG_GROUP 2
#define HMV_TUNE_K3_FORM 3
#define HMV_TUNE_YW_FORM 4
#define HMV_TUNE_K3_LDS_PAD 5
int hmv_set_tuning(int key, int64_t value);
int64_t hmv_get_tuning(int key);
/* Number of doubles of K2 scratch per item. */
int64_t hmv_yw_workspace_doubles(int m, int p);

/* K1.  R[item][l][MP][MP] = (1/n) X[:, :n-l] X[:, l:]^T for l = 0..p  (biased, not demeaned).
 * Replaces count_corr (src/mtmvar.py:35-87; lags :57-59, lag 0 :72-73).
 * x: [n_rec][m][ld-strided samples]; window `it` covers samples item_start[it] .. +n of recording
 * item_rec[it] (int64 device arrays -- arbitrary starts, as produced by
 * EEG_IBI_FFDTF_Pipeline._create_windows, src/eeg_alpha_ibi_ffdtf.py:451-518). */
int hmv_lagcov_f64(const double* x, int64_t rec_stride, int64_t ld,
                   const int64_t* item_rec, const int64_t* item_start, int64_t n_items,
                   int m, int n, int p, double* R, void* stream);

/* K1 for a REGULAR grid of overlapping windows of one recording: window w covers samples first + w*hop .. + n with
 * n a whole number k of hops (50 % overlap: k = 2).  Every product x_i[t] x_j[t+l] then belongs to up to k windows;
 * the hop blocks are summed once and the windows assembled from the block sums (minus the l products per lag that
 * reach past a window's end): half the flops of hmv_lagcov_f64 at 50 % overlap.  Same estimator (count_corr,
 * src/mtmvar.py:57-59, 72-73: biased 1/n, no demeaning) with the sums associated differently -- equal to
 * hmv_lagcov_f64 to rounding (~1e-16 relative), not bitwise.  x: [m][ld] (ONE recording of T samples), R:
 * [n_win][p+1][MP][MP], workspace: hmv_lagcov_regular_workspace_doubles(n_win, m, n, hop, p) doubles. */
#define HMV_MAX_HOPS_PER_WINDOW 8
int64_t hmv_lagcov_regular_workspace_doubles(int64_t n_win, int m, int n, int64_t hop, int p);
int hmv_lagcov_regular_f64(const double* x, int64_t ld, int64_t T, int64_t first, int64_t hop, int64_t n_win,
                           int m, int n, int p, double* R, double* workspace, void* stream);

/* K2.  Yule-Walker solve.  Replaces ar_coeff (src/mtmvar.py:90-123).
 * ar: [item][MP][MP][p] with ar[i][j][k] multiplying x_j(t-k-1) into x_i(t) (lag fastest: the
 * reference's own (m, m, p) layout when m == MP); V: [item][MP][MP] residual covariance.
 * vq_logdet (optional, may be NULL): [item][p] = log det V_q for model orders q = 1..p, all from the one
 * factorisation at order p (what mvar_criterion, src/mtmvar.py:551-601, gets from p separate fits).
 * flags: 0, HMV_FLAG_YW_TILED or HMV_FLAG_YW_ONE_LAUNCH (below). */
int hmv_yw_solve_f64(const double* R, int64_t n_items, int m, int p, double* ws,
                     double* ar, double* V, double* vq_logdet, int32_t* info, int64_t flags, void* stream);

/* tw[f][k] = exp(-(k+1) * 2*pi*1j * freqs[f] / fs) as interleaved (re, im), k = 0..p-1
 * (src/mtmvar.py:151-153, same operation order). */
int hmv_twiddles_f64(const double* freqs, int F, double fs, int p, double* tw, void* stream);

/* K3.  A(f) = I - sum_k ar[:, :, k] tw[f][k];  H(f) = inv(A(f)).
 * Replaces mvar_transfer_function (src/mtmvar.py:126-162) and the |H|^2 of dtf_multivariate (:232).
 * Optional outputs (NULL to skip), kernel-natural layout [item][f][MP][MP]:
 *   P = |H|^2 with rowsum[item][f][MP] = sum_j |H_ij|^2 (required together), H, A (complex128 interleaved).
 * pivot_tau: 1.0 = partial pivoting on |re|+|im| (LAPACK zgetrf's choice); 0 < tau < 1 keeps the diagonal
 * pivot whenever it is within a factor tau of the column maximum.  info: [item*F + f].
 * ws: caller-owned scratch of hmv_tf_workspace_doubles(n_items, m, p) doubles (the coefficients re-ordered
 * once per item so that every per-frequency read is fully coalesced). */
int64_t hmv_tf_workspace_doubles(int64_t n_items, int m, int p);
int hmv_tf_f64(const double* ar, int64_t n_items, int m, int p, const double* tw, int F,
               double* P, double* rowsum, double* H, double* A, int32_t* info,
               double pivot_tau, double* ws, void* stream);

/* K4.  out[item][i][j][f] = P[item][f][i][j] / sum_{j',f'} P[item][f'][i][j']   (normalise = 1)
 * Replaces the normalisation loop of full_freq_dtf (src/mtmvar.py:281-283); normalise = 0 returns the
 * plain |H|^2 of dtf_multivariate in the reference's (m, m, F) layout.  den: [item][MP] scratch/out. */
int hmv_ffdtf_norm_f64(const double* P, const double* rowsum, double* den, double* out,
                       int64_t n_items, int F, int m, int normalise, void* stream);

/* complex128 [item][f][MP][MP] -> [item][m][m][F] (the reference's H / A / spectra array layout). */
int hmv_transpose_c128(const double* in, double* out, int64_t n_items, int F, int m, void* stream);

/* K5.  S[item][f] = H V H^T, plain transpose (src/mtmvar.py:199).  H, S complex128 [item][f][MP][MP]. */
int hmv_spectra_f64(const double* H, const double* V, double* S, int64_t n_items, int m, int F, void* stream);
/* The same product written by the kernel itself in the reference's array layout, complex128 [item][m][m][F] (what
 * multivariate_spectra returns, src/mtmvar.py:165-201) -- no separate hmv_transpose_c128 pass over S. */
int hmv_spectra_mmf_f64(const double* H, const double* V, double* S, int64_t n_items, int m, int F, void* stream);

/* ---- measures on top of the per-frequency matrices ----------------------------
 * hmv_pack_c128: complex (items, m, m, F) -- the reference's array layout -- to the kernel layout
 *   [item][f][MP][MP] with the identity on the padding (inverse of hmv_transpose_c128).
 * hmv_cinv_c128: Zinv[item][f] = inv(Z[item][f]) by K3's blocked Gauss-Jordan (same pivoting rule);
 *   detph (optional): [item*F + f][2] = det / |det| (product of the pivots, sign of the interchanges).
 * hmv_partial_coherence_c128: kappa from Sinv = inverse spectral matrix and its detph; replaces
 *   partial_coherence (src/mtmvar.py:287-338: minors by np.linalg.det) through M_ij = (-1)^(i+j) det (S^-1)_ji;
 *   kappa: complex [item][f][MP][MP] (hmv_transpose_c128 brings it to (m, m, F)).
 * hmv_gpdc_f64: generalised partial directed coherence from A(f) (hmv_tf_f64's A output) and V
 *   (src/mtmvar.py:388-468); G: real [item][f][MP][MP] (hmv_ffdtf_norm_f64 with normalise = 0 transposes it). */
int hmv_pack_c128(const double* in, double* out, int64_t n_items, int F, int m, void* stream);
int hmv_cinv_c128(const double* Z, int64_t n_items, int m, int F, double* Zinv, double* detph, int32_t* info,
                  double pivot_tau, void* stream);
int hmv_partial_coherence_c128(const double* Sinv, const double* detph, double* kappa, int64_t n_items, int m,
                               int F, void* stream);
int hmv_gpdc_f64(const double* A, const double* V, double* G, int64_t n_items, int m, int F, void* stream);

/* Small elementwise companions, so that no arithmetic of the path is left to the host framework:
 * hmv_trial_mean_f64: R_mean[l] = (R[0][l] + ... + R[T-1][l]) / T over the lag covariances of T trials, the
 *   multi-trial average of count_corr (src/mtmvar.py:54-85; totals accumulated trial by trial, then divided).
 *   R_trials: [n_trials][p+1][MP][MP] (hmv_lagcov_f64 with one item per trial), R_mean: [p+1][MP][MP].
 * hmv_ddtf_f64: ddtf = ffdtf * |kappa|, the product of direct_dtf (src/mtmvar.py:341-385); ffdtf real and kappa
 *   complex128, both [item][m][m][F] (the reference's layout).
 * hmv_band_sums_f64: out[row][b] = sum_{bin_lo[b] <= f < bin_hi[b]} ffdtf[row][f] for the n_rows = items*m*m rows
 *   of an [item][m][m][F] array (bin_lo / bin_hi: int32 device arrays): the band-integrated ffDTF that is
 *   gathered across GPUs (the reference's graph plots integrate ffDTF over a band, src/mtmvar.py:984-987). */
int hmv_trial_mean_f64(const double* R_trials, int64_t n_trials, int m, int p, double* R_mean, void* stream);
int hmv_ddtf_f64(const double* ffdtf, const double* kappa, double* ddtf, int64_t n_items, int m, int F, void* stream);
int hmv_band_sums_f64(const double* ffdtf, int64_t n_rows, int F, const int32_t* bin_lo, const int32_t* bin_hi,
                      int n_bands, double* out, void* stream);

/* Multitaper PSD.  Replaces compute_psd_multitaper (src/psd.py:7-33), i.e.
 * mne.time_frequency.psd_array_multitaper(data, sfreq, fmin, fmax, bandwidth) of mne==1.11.0 with its defaults
 * (remove_dc, non-adaptive eigenvalue weights, normalization "length").  mne is NOT available offline: PARITY
 * UNPINNED -- the algorithm is restated from its published description and checked against an independent
 * NumPy restatement only.  x: [n_ch][ld] (n_times samples used), tapers: [n_tapers][n_times] DPSS windows and
 * weights[k] = sqrt(eigenvalue_k) (hmv_dpss_f64, or scipy.signal.windows.dpss on the host), bins bin_lo..bin_hi of the
 * one-sided spectrum (freq = bin * sfreq / n_times); psd: [n_ch][bin_hi - bin_lo + 1].  Transforms by hipFFT;
 * plans are cached per (n_times, batch).  workspace: hmv_psd_workspace_bytes(ch_chunk, n_times, n_tapers). */
/* DPSS (Slepian) tapers and their concentration ratios on the device: what scipy.signal.windows.dpss(n_times, half_nbw,
 * k_max, sym=bool(sym), norm=2, return_ratios=True) returns (mne's taper generator calls it with sym=False: the symmetric
 * window of n_times + 1 points without its last sample), by the same algorithm -- the k_max largest
 * eigenpairs of the commuting symmetric tridiagonal matrix by Sturm-count multisection and inverse iteration (LAPACK
 * dstebz / dstein restated), SciPy's sign convention, ratios through hipFFT.  tapers: [k_max][n_times], ratios (optional):
 * [k_max].  Agrees with SciPy to ~1e-10 (tests/test_psd.py); PARITY UNPINNED like the PSD itself.  The call synchronises
 * `stream` once when ratios are requested (the FFT plan is created and destroyed inside). */
int64_t hmv_dpss_workspace_bytes(int64_t n_times, int k_max, int sym);
int hmv_dpss_f64(int64_t n_times, double half_nbw, int k_max, int sym, double* tapers, double* ratios,
                 void* workspace, int64_t workspace_bytes, void* stream);
int64_t hmv_psd_workspace_bytes(int64_t ch_chunk, int64_t n_times, int n_tapers);
int hmv_psd_multitaper_f64(const double* x, int64_t n_ch, int64_t n_times, int64_t ld, const double* tapers,
                           const double* weights, int n_tapers, int64_t bin_lo, int64_t bin_hi, double* psd,
                           void* workspace, int64_t workspace_bytes, int64_t ch_chunk, void* stream);

/* Option bits of the fused entry points (`flags`).  0 = the fast defaults. */
#define HMV_FLAG_UNFUSED_NORM 1   /* ffDTF normalisation as a separate pass over |H|^2 (K4) instead of inside K3 */
#define HMV_FLAG_DIRECT_LAGCOV 8   /* K1 sums every window from its own samples even on a regular grid (results then do
                                     not depend on how the windows are laid out: bit-identical across grids) */
#define HMV_FLAG_YW_TILED 2       /* K2 as one workgroup per tile in p + 2 launches (tile column by tile column), and */
#define HMV_FLAG_YW_ONE_LAUNCH 4  /* K2 as one workgroup per window in one launch: same tile products in the same
                                     order, same bits.  Neither flag: one launch, except for large 64-channel
                                     batches, where both forms are HBM-bound and the launch chain is faster. */

/* K3 + K4 in one pass.  ffdtf[item][i][j][f] = |H_ij(f)|^2 / sum_{j',f'} |H_ij'(f')|^2, i.e. the arithmetic of
 * mvar_transfer_function (src/mtmvar.py:126-162), |H|^2 (:232) and the normalisation loop of full_freq_dtf
 * (:281-283).  K3 leaves |H|^2 and its row sums in `workspace` (write-through stores); the workgroup whose
 * matrix completes a window (device-scope arrival counter) adds up the row denominators, and the rows of that
 * window are turned into the output array by workgroups of a LATER window of the same launch while the rest of
 * the chip keeps inverting (nobody waits: a row whose window is not complete in time, and the rows of the last few
 * windows of the batch, which no later workgroup exists for, are done by a small kernel right behind K3).  Bit-identical
 * to hmv_tf_f64 + hmv_ffdtf_norm_f64.  Taken when F % 16 == 0 and ffdtf is 16-byte aligned, otherwise the separate K4
 * pass does all windows.  den: [item][MP] out; H (optional, may be NULL): complex128 [item][f][MP][MP] as from hmv_tf_f64 (what
 * hmv_spectra_f64 takes: ffDTF and spectra from one set of inverses); info: [item*F + f].
 * ev_k3_start / ev_k3_stop (optional hipEvent_t) are recorded on `stream` right before and after the K3 kernel itself
 * (not the coefficient-packing kernel in front of it, not the small kernel behind it). */
int64_t hmv_tf_ffdtf_workspace_bytes(int64_t n_items, int m, int p, int F);
int hmv_tf_ffdtf_f64(const double* ar, int64_t n_items, int m, int p, const double* tw, int F,
                     double* ffdtf, double* den, double* H, int32_t* info, double pivot_tau,
                     void* workspace, int64_t workspace_bytes, int64_t flags,
                     void* ev_k3_start, void* ev_k3_stop, void* stream);

/* The same pass with a REDUCED product: band_out[item][i][j][b] = sum_{bin_lo[b] <= f < bin_hi[b]} ffdtf[item][i][j][f]
 * (bin_lo / bin_hi: int32 device arrays of n_bands entries) -- what the reference's graph plots integrate
 * (src/mtmvar.py:984-987) and what crosses PCIe / xGMI per window instead of the 8.4 MB of the full array.  The row
 * workers inside K3 add the bands up from the published |H|^2 rows and the full-resolution array is never written
 * (only the last few windows of a batch pass through a scratch copy of it).  Same bits as hmv_tf_ffdtf_f64 followed by
 * hmv_band_sums_f64.  Needs F % 32 == 0 and F <= ~2700 / 1300 / 640 at 64 / 32 / 16 padded channels (one band's weights
 * must fit the row worker's LDS block); otherwise -10 and the caller takes the two-call route. */
int64_t hmv_tf_ffdtf_bands_workspace_bytes(int64_t n_items, int m, int p, int F);
int hmv_tf_ffdtf_bands_f64(const double* ar, int64_t n_items, int m, int p, const double* tw, int F,
                           double* band_out, const int32_t* bin_lo, const int32_t* bin_hi, int n_bands,
                           double* den, int32_t* info, double pivot_tau,
                           void* workspace, int64_t workspace_bytes, int64_t flags,
                           void* ev_k3_start, void* ev_k3_stop, void* stream);

/* Fused sliding-window path K1 -> K2 -> K3 -> K4 over all items, processed `chunk` items at a time so the
 * scratch stays bounded.  Equivalent to calling full_freq_dtf(window, freqs, fs, optimal_model_order=p)
 * (src/mtmvar.py:237-284) on every window.  ffdtf: [n_items][m][m][F].
 * ar_out / V_out (optional): [n_items][MP][MP][p] / [n_items][MP][MP].
 * info_yw: [n_items], info_tf: [n_items*F].  workspace: hmv_sliding_workspace_bytes(chunk, m, p, F) bytes.
 * grid_hop > 0 declares a REGULAR window grid and lets K1 share the overlap (hmv_lagcov_regular_f64): the caller
 * vouches that item = rec * grid_nwin + w is the window starting at grid_first + w * grid_hop of recording rec
 * (item_rec / item_start must say the same; they are still what every other stage and the direct form read) and
 * that recordings are grid_T samples long.  Taken when n is 2..8 whole hops; grid_hop = 0: arbitrary windows.
 * ev_k3_start / ev_k3_stop (optional hipEvent_t, NULL to skip) are recorded on `stream` right before
 * and after the LAST chunk's K3 launch, so a caller can time the dominant kernel inside its own timed
 * region without an extra synchronisation.
 * aux_stream (optional second hipStream_t, NULL to disable; used by the tiled form of K2 only): a chain of dependent
 * launches that cannot fill the chip -- it runs as two half-batches, one per stream (fork
 * after K1, join before K3), so their launches interleave on the device; the call still behaves as one
 * operation on `stream`. */
int64_t hmv_sliding_workspace_bytes(int64_t chunk, int m, int p, int F);
int hmv_sliding_ffdtf_f64(const double* x, int64_t rec_stride, int64_t ld,
                          const int64_t* item_rec, const int64_t* item_start, int64_t n_items,
                          int m, int n, int p, const double* freqs, int F, double fs,
                          double* ffdtf, double* ar_out, double* V_out,
                          int32_t* info_yw, int32_t* info_tf,
                          void* workspace, int64_t workspace_bytes, int64_t chunk,
                          double pivot_tau, int64_t flags,
                          int64_t grid_hop, int64_t grid_first, int64_t grid_nwin, int64_t grid_T,
                          void* ev_k3_start, void* ev_k3_stop, void* stream, void* aux_stream);

/* hmv_sliding_ffdtf_f64 with the reduced product of hmv_tf_ffdtf_bands_f64: band_out: [n_items][m][m][n_bands].  The
 * per-dyad loop of the reference (load, compute, save: src/eeg_alpha_ibi_ffdtf.py:661-806) streams recordings through
 * this entry: 154 MB in, 98 MB out per 10-minute dyad instead of 5 GB. */
int64_t hmv_sliding_bands_workspace_bytes(int64_t chunk, int m, int p, int F);
int hmv_sliding_ffdtf_bands_f64(const double* x, int64_t rec_stride, int64_t ld,
                                const int64_t* item_rec, const int64_t* item_start, int64_t n_items,
                                int m, int n, int p, const double* freqs, int F, double fs,
                                double* band_out, const int32_t* bin_lo, const int32_t* bin_hi, int n_bands,
                                double* ar_out, double* V_out,
                                int32_t* info_yw, int32_t* info_tf,
                                void* workspace, int64_t workspace_bytes, int64_t chunk,
                                double pivot_tau, int64_t flags,
                                int64_t grid_hop, int64_t grid_first, int64_t grid_nwin, int64_t grid_T,
                                void* ev_k3_start, void* ev_k3_stop, void* stream, void* aux_stream);

/* hmv_sliding_ffdtf_f64 that ALSO returns the multivariate spectra S(f) = H(f) V H(f)^T (plain transpose, as
 * src/mtmvar.py:199) of every window -- the two products the reference's orchestrators always compute together, there
 * from two separate fits (src/eeg_alpha_ibi_ffdtf.py:592-604, src/mtmvar.py:1100-1113), here from ONE fit and ONE set of
 * inverses: K3 leaves H of a chunk in the workspace, K5 turns it into S_out: complex128 [n_items][m][m][F] (the
 * reference's layout).  V is this library's own residual covariance, symmetric up to rounding, so K5 computes the upper
 * triangle of S only and mirrors it (S_ij and S_ji of the reference differ by rounding; here they are equal). */
int64_t hmv_sliding_spectra_workspace_bytes(int64_t chunk, int m, int p, int F);
int hmv_sliding_ffdtf_spectra_f64(const double* x, int64_t rec_stride, int64_t ld,
                                  const int64_t* item_rec, const int64_t* item_start, int64_t n_items,
                                  int m, int n, int p, const double* freqs, int F, double fs,
                                  double* ffdtf, double* S_out, double* ar_out, double* V_out,
                                  int32_t* info_yw, int32_t* info_tf,
                                  void* workspace, int64_t workspace_bytes, int64_t chunk,
                                  double pivot_tau, int64_t flags,
                                  int64_t grid_hop, int64_t grid_first, int64_t grid_nwin, int64_t grid_T,
                                  void* stream, void* aux_stream);

#ifdef __cplusplus
}
#endif
#endif /* HYPERMVAR_H */
