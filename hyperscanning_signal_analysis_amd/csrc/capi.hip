// C ABI of libhypermvar.so -- argument checking and launch sequencing only; see include/hypermvar.h.
#include "../../include/hypermvar.h"
#include "hmv_kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int pad_of(int m) { return (m < 1 || m > HMV_MAX_CHANNELS) ? -1 : ((m + 15) / 16) * 16; }
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

// Tuning knobs: the environment is parsed once, at load time, with range checks; hmv_set_tuning overrides.
constexpr int N_TUNE = 6;
struct TuneRange { long long lo, hi; };
constexpr TuneRange kTuneRange[N_TUNE] = {{0, 0}, {0, 1 << 20}, {0, 3}, {0, 2}, {0, 3}, {0, 140000}};
long long env_knob(const char* name, int key) {
  const char* e = getenv(name);
  if (!e || !*e) return 0;
  char* end = nullptr;
  const long long v = strtoll(e, &end, 10);
  if (end == e || *end != 0 || v < kTuneRange[key].lo || v > kTuneRange[key].hi) {
    fprintf(stderr, "hypermvar: ignoring %s=%s (not an integer in %lld..%lld)\n", name, e, kTuneRange[key].lo,
            kTuneRange[key].hi);
    return 0;
  }
  return v;
}
std::atomic<long long> g_tune[N_TUNE] = {{0}, {env_knob("HYPERMVAR_NORM_LAG", 1)}, {env_knob("HYPERMVAR_LAG_GROUP", 2)},
                                         {env_knob("HYPERMVAR_K3_FORM", 3)}, {env_knob("HYPERMVAR_YW_FORM", 4)},
                                         {env_knob("HYPERMVAR_K3_LDS_PAD", 5)}};
}  // namespace

namespace hmv {
long long tuning(int key) { return (key >= 1 && key < N_TUNE) ? g_tune[key].load(std::memory_order_relaxed) : -1; }
}

#ifdef HMV_STAMP
// Diagnostic build only (never shipped, not in include/hypermvar.h): where K3 writes its phase stamps.
static unsigned long long* g_tf_stamps = nullptr;
extern "C" void hmv_debug_set_tf_stamps(void* p) { g_tf_stamps = static_cast<unsigned long long*>(p); }
#endif

extern "C" {

int hmv_version(void) { return HMV_VERSION; }
const char* hmv_last_error(void) { return g_err; }
int hmv_pad(int m) { return pad_of(m); }

int hmv_set_tuning(int key, int64_t value) {
  if (key < 1 || key >= N_TUNE) return fail(-1, "hmv_set_tuning: unknown key");
  if (value < kTuneRange[key].lo || value > kTuneRange[key].hi) return fail(-1, "hmv_set_tuning: value out of range");
  g_tune[key].store(value, std::memory_order_relaxed);
  return 0;
}
int64_t hmv_get_tuning(int key) { return hmv::tuning(key); }

int64_t hmv_yw_workspace_doubles(int m, int p) {
  const int mp = pad_of(m);
  if (mp < 0 || p < 1 || p > HMV_MAX_ORDER) return -1;
  return hmv::yw_ws_tiles(p) * (int64_t)mp * mp;
}

int hmv_lagcov_f64(const double* x, int64_t rec_stride, int64_t ld, const int64_t* item_rec,
                   const int64_t* item_start, int64_t n_items, int m, int n, int p, double* R, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_lagcov_f64: channel count must be in 1..64");
  if (p < 1 || p > HMV_MAX_ORDER) return fail(-2, "hmv_lagcov_f64: model order must be in 1..32");
  if (n <= p) return fail(-3, "hmv_lagcov_f64: window shorter than the model order");
  if (!x || !item_rec || !item_start || !R || n_items < 0) return fail(-4, "hmv_lagcov_f64: null pointer");
  hmv::LagcovArgs a{};
  a.x = x; a.rec_stride = rec_stride; a.ld = ld;
  a.item_rec = reinterpret_cast<const long long*>(item_rec);
  a.item_start = reinterpret_cast<const long long*>(item_start);
  a.n_items = n_items; a.m = m; a.n = n; a.p = p; a.R = R;
  return hmv::launch_lagcov(a, mp, S(stream));
}

int64_t hmv_lagcov_regular_workspace_doubles(int64_t n_win, int m, int n, int64_t hop, int p) {
  const int mp = pad_of(m);
  if (mp < 0 || n_win < 0 || hop < 1 || n < 1 || n % hop != 0 || p < 0) return -1;
  return (n_win + n / hop - 1) * (int64_t)(p + 1) * mp * mp;
}

int hmv_lagcov_regular_f64(const double* x, int64_t ld, int64_t T, int64_t first, int64_t hop, int64_t n_win, int m,
                           int n, int p, double* R, double* workspace, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_lagcov_regular_f64: channel count must be in 1..64");
  if (p < 1 || p > HMV_MAX_ORDER) return fail(-2, "hmv_lagcov_regular_f64: model order must be in 1..32");
  if (n <= p) return fail(-3, "hmv_lagcov_regular_f64: window shorter than the model order");
  if (!x || !R || !workspace || n_win < 0) return fail(-4, "hmv_lagcov_regular_f64: null pointer");
  if (hop < 1 || n % hop != 0 || hop <= p)
    return fail(-5, "hmv_lagcov_regular_f64: the window must be a whole number of hops and a hop longer than the order");
  if (first < 0 || ld < T || (n_win > 0 && first + (n_win - 1) * hop + n > T))
    return fail(-6, "hmv_lagcov_regular_f64: windows do not lie inside the recording");
  if (n_win == 0) return 0;
  const int k = (int)(n / hop);
  hmv::LagcovArgs a{};
  a.x = x; a.rec_stride = 0; a.ld = ld; a.item_rec = nullptr; a.item_start = nullptr;
  a.n_items = n_win + k - 1; a.m = m; a.n = (int)hop; a.p = p; a.R = workspace;
  a.blocks = 1; a.blk_first = first; a.blk_T = T;
  int rc = hmv::launch_lagcov(a, mp, S(stream));
  if (rc) return rc;
  hmv::LagcombArgs c{};
  c.Q = workspace; c.x = x; c.ld = ld; c.first = first; c.hop = hop; c.T = T; c.n_win = n_win; c.k = k; c.m = m; c.p = p;
  c.R = R;
  return hmv::launch_lagcomb(c, mp, S(stream));
}

int hmv_yw_solve_f64(const double* R, int64_t n_items, int m, int p, double* ws, double* ar, double* V,
                     double* vq_logdet, int32_t* info, int64_t flags, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_yw_solve_f64: channel count must be in 1..64");
  if (p < 1 || p > HMV_MAX_ORDER) return fail(-2, "hmv_yw_solve_f64: model order must be in 1..32");
  if (!R || !ws || !ar || !V || !info || n_items < 0) return fail(-4, "hmv_yw_solve_f64: null pointer");
  hmv::YwArgs a;
  a.R = R; a.n_items = n_items; a.m = m; a.p = p; a.ws = ws; a.ar = ar; a.V = V;
  a.Vq_logdet = vq_logdet; a.info = info; a.tiled = (flags & HMV_FLAG_YW_TILED) ? 1 : ((flags & HMV_FLAG_YW_ONE_LAUNCH) ? 0 : -1);
  return hmv::launch_yw(a, mp, S(stream));
}

int hmv_twiddles_f64(const double* freqs, int F, double fs, int p, double* tw, void* stream) {
  if (!freqs || !tw || F < 0 || p < 1) return fail(-4, "hmv_twiddles_f64: bad argument");
  if (F == 0) return 0;
  return hmv::launch_twiddles(freqs, F, fs, p, tw, S(stream));
}

int64_t hmv_tf_workspace_doubles(int64_t n_items, int m, int p) {
  const int mp = pad_of(m);
  if (mp < 0 || p < 1 || n_items < 0) return -1;
  return (int64_t)hmv::tf_workspace_doubles(n_items, mp, p);
}

int hmv_tf_f64(const double* ar, int64_t n_items, int m, int p, const double* tw, int F, double* P,
               double* rowsum, double* H, double* A, int32_t* info, double pivot_tau, double* ws,
               void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_tf_f64: channel count must be in 1..64");
  if (p < 1) return fail(-2, "hmv_tf_f64: model order must be >= 1");
  if (!ar || !tw || !info || !ws || n_items < 0 || F < 0) return fail(-4, "hmv_tf_f64: null pointer");
  if ((P == nullptr) != (rowsum == nullptr)) return fail(-5, "hmv_tf_f64: P and rowsum go together");
  if (!(pivot_tau > 0.0) || pivot_tau > 1.0) return fail(-6, "hmv_tf_f64: pivot_tau must be in (0, 1]");
  hmv::TfArgs a{};
  a.ar = ar; a.arx = ws; a.tw = tw; a.Zin = nullptr; a.detph = nullptr; a.P = P; a.rowsum = rowsum; a.H = H; a.A = A; a.info = info;
  a.n_items = n_items; a.F = F; a.p = p; a.m = m; a.tau = pivot_tau;
  a.stamps = nullptr;
#ifdef HMV_STAMP
  a.stamps = g_tf_stamps;
#endif
  return hmv::launch_tf_inv(a, mp, S(stream));
}

int hmv_ffdtf_norm_f64(const double* P, const double* rowsum, double* den, double* out, int64_t n_items,
                       int F, int m, int normalise, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_ffdtf_norm_f64: channel count must be in 1..64");
  if (!P || !out || (normalise && (!rowsum || !den))) return fail(-4, "hmv_ffdtf_norm_f64: null pointer");
  hmv::NormArgs a;
  a.P = P; a.rowsum = rowsum; a.den = den; a.out = out; a.n_items = n_items; a.F = F; a.m = m; a.m_pad = mp;
  a.normalise = normalise;
  return hmv::launch_ffdtf_norm(a, S(stream));
}

int hmv_transpose_c128(const double* in, double* out, int64_t n_items, int F, int m, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_transpose_c128: channel count must be in 1..64");
  if (!in || !out) return fail(-4, "hmv_transpose_c128: null pointer");
  return hmv::launch_transpose_c128(in, out, n_items, F, m, mp, S(stream));
}

int hmv_spectra_f64(const double* H, const double* V, double* Sout, int64_t n_items, int m, int F, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_spectra_f64: channel count must be in 1..64");
  if (!H || !V || !Sout) return fail(-4, "hmv_spectra_f64: null pointer");
  hmv::SpecArgs a;
  a.H = H; a.V = V; a.S = Sout; a.S_mmf = nullptr; a.n_items = n_items; a.F = F; a.m = m; a.sym = 0;
  return hmv::launch_spectra(a, mp, S(stream));
}

int hmv_spectra_mmf_f64(const double* H, const double* V, double* Sout, int64_t n_items, int m, int F, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_spectra_mmf_f64: channel count must be in 1..64");
  if (!H || !V || !Sout) return fail(-4, "hmv_spectra_mmf_f64: null pointer");
  hmv::SpecArgs a;
  a.H = H; a.V = V; a.S = nullptr; a.S_mmf = Sout; a.n_items = n_items; a.F = F; a.m = m; a.sym = 0;
  return hmv::launch_spectra(a, mp, S(stream));
}

int hmv_pack_c128(const double* in, double* out, int64_t n_items, int F, int m, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_pack_c128: channel count must be in 1..64");
  if (!in || !out || n_items < 0 || F < 0) return fail(-4, "hmv_pack_c128: null pointer");
  return hmv::launch_pack_c128(in, out, n_items, F, m, mp, S(stream));
}

int hmv_cinv_c128(const double* Z, int64_t n_items, int m, int F, double* Zinv, double* detph, int32_t* info,
                  double pivot_tau, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_cinv_c128: channel count must be in 1..64");
  if (!Z || !Zinv || !info || n_items < 0 || F < 0) return fail(-4, "hmv_cinv_c128: null pointer");
  if (!(pivot_tau > 0.0) || pivot_tau > 1.0) return fail(-6, "hmv_cinv_c128: pivot_tau must be in (0, 1]");
  hmv::TfArgs a{};
  a.ar = nullptr; a.arx = nullptr; a.tw = nullptr; a.Zin = Z; a.detph = detph; a.P = nullptr; a.rowsum = nullptr;
  a.H = Zinv; a.A = nullptr; a.info = info; a.n_items = n_items; a.F = F; a.p = 0; a.m = m; a.tau = pivot_tau;
  a.stamps = nullptr;
  return hmv::launch_cinv(a, mp, S(stream));
}

int hmv_partial_coherence_c128(const double* Sinv, const double* detph, double* kappa, int64_t n_items, int m, int F,
                               void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_partial_coherence_c128: channel count must be in 1..64");
  if (!Sinv || !detph || !kappa || n_items < 0 || F < 0) return fail(-4, "hmv_partial_coherence_c128: null pointer");
  return hmv::launch_pcoh(Sinv, detph, kappa, n_items, F, m, mp, S(stream));
}

int hmv_gpdc_f64(const double* A, const double* V, double* G, int64_t n_items, int m, int F, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_gpdc_f64: channel count must be in 1..64");
  if (!A || !V || !G || n_items < 0 || F < 0) return fail(-4, "hmv_gpdc_f64: null pointer");
  return hmv::launch_gpdc(A, V, G, n_items, F, m, mp, S(stream));
}

int hmv_trial_mean_f64(const double* R_trials, int64_t n_trials, int m, int p, double* R_mean, void* stream) {
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_trial_mean_f64: channel count must be in 1..64");
  if (!R_trials || !R_mean || n_trials < 1 || n_trials > 0x7fffffff || p < 0)
    return fail(-4, "hmv_trial_mean_f64: bad argument");
  return hmv::launch_trial_mean(R_trials, R_mean, (long long)(p + 1) * mp * mp, (int)n_trials, S(stream));
}

int hmv_ddtf_f64(const double* ffdtf, const double* kappa, double* ddtf, int64_t n_items, int m, int F, void* stream) {
  if (!ffdtf || !kappa || !ddtf || n_items < 0 || m < 1 || F < 0) return fail(-4, "hmv_ddtf_f64: bad argument");
  return hmv::launch_ddtf(ffdtf, kappa, ddtf, (long long)n_items * m * m * F, S(stream));
}

int hmv_band_sums_f64(const double* ffdtf, int64_t n_rows, int F, const int32_t* bin_lo, const int32_t* bin_hi,
                      int n_bands, double* out, void* stream) {
  if (!ffdtf || !bin_lo || !bin_hi || !out || n_rows < 0 || F < 1 || n_bands < 0)
    return fail(-4, "hmv_band_sums_f64: bad argument");
  return hmv::launch_band_sums(ffdtf, bin_lo, bin_hi, out, n_rows, F, n_bands, S(stream));
}

int64_t hmv_psd_workspace_bytes(int64_t ch_chunk, int64_t n_times, int n_tapers) {
  if (ch_chunk < 1 || n_times < 2 || n_tapers < 1) return -1;
  return (int64_t)hmv::psd_workspace_bytes(ch_chunk, n_times, n_tapers);
}

int hmv_psd_multitaper_f64(const double* x, int64_t n_ch, int64_t n_times, int64_t ld, const double* tapers,
                           const double* weights, int n_tapers, int64_t bin_lo, int64_t bin_hi, double* psd,
                           void* workspace, int64_t workspace_bytes, int64_t ch_chunk, void* stream) {
  if (!x || !tapers || !weights || !psd || !workspace) return fail(-4, "hmv_psd_multitaper_f64: null pointer");
  if (n_ch < 0 || n_times < 2 || n_times > 0x7fffffff || n_tapers < 1 || ch_chunk < 1 || ld < n_times)
    return fail(-2, "hmv_psd_multitaper_f64: bad size");
  if (bin_lo < 0 || bin_hi < bin_lo || bin_hi > n_times / 2) return fail(-3, "hmv_psd_multitaper_f64: bad frequency bins");
  if (workspace_bytes < hmv::psd_workspace_bytes(ch_chunk, n_times, n_tapers))
    return fail(-7, "hmv_psd_multitaper_f64: workspace too small");
  const int rc = hmv::launch_psd(x, n_ch, n_times, ld, tapers, weights, n_tapers, bin_lo, bin_hi, psd, workspace, ch_chunk,
                                 S(stream));
  if (rc <= -20) return fail(rc, "hmv_psd_multitaper_f64: hipFFT plan / execution failed");
  return rc;
}

int64_t hmv_dpss_workspace_bytes(int64_t n_times, int k_max, int sym) {
  if (n_times < 2 || k_max < 1 || k_max > n_times) return -1;
  return (int64_t)hmv::dpss_workspace_bytes(n_times, k_max, sym != 0);
}

int hmv_dpss_f64(int64_t n_times, double half_nbw, int k_max, int sym, double* tapers, double* ratios, void* workspace,
                 int64_t workspace_bytes, void* stream) {
  if (!tapers || !workspace) return fail(-4, "hmv_dpss_f64: null pointer");
  if (n_times < 2 || n_times > 0x3fffffff || k_max < 1 || k_max > n_times)
    return fail(-2, "hmv_dpss_f64: bad size");
  if (!(half_nbw > 0.0) || half_nbw >= 0.5 * (double)n_times)
    return fail(-3, "hmv_dpss_f64: the time-half-bandwidth product must lie in (0, n_times / 2)");
  if (workspace_bytes < hmv::dpss_workspace_bytes(n_times, k_max, sym != 0)) return fail(-7, "hmv_dpss_f64: workspace too small");
  const int rc = hmv::launch_dpss(n_times, half_nbw, k_max, sym != 0, tapers, ratios, workspace, S(stream));
  if (rc <= -20) return fail(rc, "hmv_dpss_f64: hipFFT plan / execution failed");
  if (rc < 0) return fail(rc, "hmv_dpss_f64: bad argument");
  return rc;
}

// ---- K3 with the ffDTF normalisation folded in ------------------------------------------------------
namespace {
struct TfFfWs {
  size_t off_arx, off_P, off_rowsum, off_cnt, off_tail, total;
};
int64_t norm_lag_items(int mp, int F);
// `bands`: the reduced-product form also needs the full-resolution rows of the last `lag` windows of a batch (the ones the
// separate K4 pass normalises) for a moment, before their band sums are taken
TfFfWs tf_ff_layout(int64_t n, int mp, int p, int F, bool bands = false) {
  TfFfWs w;
  size_t o = 0;
  const size_t t = (size_t)mp * mp;
  w.off_arx = o;    o += align256(sizeof(double) * hmv::tf_workspace_doubles(n, mp, p));
  w.off_P = o;      o += align256(sizeof(double) * n * F * t);
  w.off_rowsum = o; o += align256(sizeof(double) * n * F * mp);
  w.off_cnt = o;    o += align256(sizeof(int) * (2 * n + 1 + n * mp));     // wcount, ready, missed (TfArgs)
  w.off_tail = o;
  if (bands) {
    const int64_t lag = norm_lag_items(mp, F);
    o += align256(sizeof(double) * (size_t)(n < lag ? n : lag) * F * t);
  }
  w.total = o;
  return w;
}
// Rows of window w are normalised inside K3 by workgroups of window w + lag, and the last `lag` windows of a batch
// by the separate K4 pass.  lag = six times the number of windows the chip holds at once (resident workgroups /
// F): at the north-star shape (4 windows resident) lag 8 left a third of the rows unfinished when they came up,
// 16 a tenth, 24 none (profiles/r02_norm_lag_sweep.txt).  A row that comes up too early is not waited for (it goes
// to norm_missed_kernel), so this only tunes speed.
int64_t norm_lag_items(int mp, int F) {
  const int64_t slots = (mp == 64 || mp == 48) ? 1024 : (mp == 32 ? 2048 : 5120);
  int64_t lag = (6 * slots + F - 1) / (F < 1 ? 1 : F);
  if (const int64_t t = hmv::tuning(HMV_TUNE_NORM_LAG)) lag = t;        // hmv_set_tuning: experiments and tests
  return lag < 8 ? 8 : lag;
}
}  // namespace

int64_t hmv_tf_ffdtf_workspace_bytes(int64_t n_items, int m, int p, int F) {
  const int mp = pad_of(m);
  if (mp < 0 || n_items < 0 || p < 1 || F < 1) return -1;
  return (int64_t)tf_ff_layout(n_items, mp, p, F).total;
}

namespace {
// ffdtf != NULL: the full array.  band_out != NULL: its band sums only (see hmv_tf_ffdtf_bands_f64).
int tf_ffdtf_impl(const char* who, const double* ar, int64_t n_items, int m, int p, const double* tw, int F, double* ffdtf,
                  double* band_out, const int32_t* bin_lo, const int32_t* bin_hi, int n_bands, double* den, double* H,
                  int32_t* info, double pivot_tau, void* workspace, int64_t workspace_bytes, int64_t flags,
                  void* ev_k3_start, void* ev_k3_stop, void* stream) {
  auto failw = [&](int code, const char* msg) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s: %s", who, msg);
    return fail(code, buf);
  };
  const int mp = pad_of(m);
  const bool bands = (band_out != nullptr);
  if (mp < 0) return failw(-1, "channel count must be in 1..64");
  if (p < 1) return failw(-2, "model order must be >= 1");
  if (n_items == 0) return 0;
  if (!ar || !tw || (!ffdtf && !bands) || !den || !info || !workspace || n_items < 0 || F < 1)
    return failw(-4, "null pointer / empty grid");
  if (!(pivot_tau > 0.0) || pivot_tau > 1.0) return failw(-6, "pivot_tau must be in (0, 1]");
  if (bands) {
    if (!bin_lo || !bin_hi || n_bands < 1) return failw(-4, "band bins missing");
    if (F % 32 != 0 || F > hmv::tf_band_max_F(mp) || (flags & HMV_FLAG_UNFUSED_NORM))
      return failw(-10, "in-kernel band sums need F % 32 == 0, F within the row worker's LDS block and the fused normalisation");
  }
  const TfFfWs w = tf_ff_layout(n_items, mp, p, F, bands);
  if ((int64_t)w.total > workspace_bytes) return failw(-7, "workspace too small");
  char* base = static_cast<char*>(workspace);
  const size_t t = (size_t)mp * mp;
  hmv::TfArgs a{};
  a.ar = ar; a.arx = reinterpret_cast<double*>(base + w.off_arx); a.tw = tw;
  a.P = reinterpret_cast<double*>(base + w.off_P); a.rowsum = reinterpret_cast<double*>(base + w.off_rowsum);
  a.H = H;
  a.info = info; a.n_items = n_items; a.F = F; a.p = p; a.m = m; a.tau = pivot_tau;
  // the in-kernel normaliser moves 16 bytes per lane: whole 16-frequency lines of a 16-byte aligned output
  const bool fused = bands || (!(flags & HMV_FLAG_UNFUSED_NORM) && (F % 16 == 0) && (reinterpret_cast<uintptr_t>(ffdtf) % 16 == 0));
  // Every window is published and normalised by this launch: row i of window w < n - lag by a workgroup of window w + lag
  // inside K3, the rows of the last `lag` windows (and any row that came up before its window was complete) by
  // norm_missed_kernel right behind it -- the separate K4 pass over a second layout of |H|^2 is for the unfused form only.
  const int64_t lag = norm_lag_items(mp, F);
  const int64_t n_fused = fused ? n_items : 0;
  if (n_fused > 0) {
    a.ff = bands ? nullptr : ffdtf; a.den = den; a.fuse_items = n_fused; a.lag = (int)lag;
    if (bands) {
      a.bands = band_out; a.band_lo = reinterpret_cast<const int*>(bin_lo); a.band_hi = reinterpret_cast<const int*>(bin_hi);
      a.nb = n_bands;
    }
    a.wcount = reinterpret_cast<int*>(base + w.off_cnt);
    a.ready = a.wcount + n_items;
    a.missed = a.ready + n_items;
  }
#ifdef HMV_STAMP
  a.stamps = g_tf_stamps;
#endif
  hipStream_t st = S(stream);
  int rc = hmv::launch_tf_inv(a, mp, st, reinterpret_cast<hipEvent_t>(ev_k3_start), reinterpret_cast<hipEvent_t>(ev_k3_stop));
  if (rc) return rc;
  if (n_fused < n_items) {
    const int64_t n_tail = n_items - n_fused;
    double* tail = bands ? reinterpret_cast<double*>(base + w.off_tail) : ffdtf + (size_t)n_fused * m * m * F;
    rc = hmv_ffdtf_norm_f64(a.P + (size_t)n_fused * F * t, a.rowsum + (size_t)n_fused * F * mp, den + (size_t)n_fused * mp,
                            tail, n_tail, F, m, 1, stream);
    if (!rc && bands)
      rc = hmv::launch_band_sums(tail, reinterpret_cast<const int*>(bin_lo), reinterpret_cast<const int*>(bin_hi),
                                 band_out + (size_t)n_fused * m * m * n_bands, n_tail * (long long)m * m, F, n_bands, st);
  }
  return rc;
}
}  // namespace

int64_t hmv_tf_ffdtf_bands_workspace_bytes(int64_t n_items, int m, int p, int F) {
  const int mp = pad_of(m);
  if (mp < 0 || n_items < 0 || p < 1 || F < 1) return -1;
  return (int64_t)tf_ff_layout(n_items, mp, p, F, true).total;
}

int hmv_tf_ffdtf_f64(const double* ar, int64_t n_items, int m, int p, const double* tw, int F, double* ffdtf,
                     double* den, double* H, int32_t* info, double pivot_tau, void* workspace, int64_t workspace_bytes,
                     int64_t flags, void* ev_k3_start, void* ev_k3_stop, void* stream) {
  return tf_ffdtf_impl("hmv_tf_ffdtf_f64", ar, n_items, m, p, tw, F, ffdtf, nullptr, nullptr, nullptr, 0, den, H, info,
                       pivot_tau, workspace, workspace_bytes, flags, ev_k3_start, ev_k3_stop, stream);
}

int hmv_tf_ffdtf_bands_f64(const double* ar, int64_t n_items, int m, int p, const double* tw, int F, double* band_out,
                           const int32_t* bin_lo, const int32_t* bin_hi, int n_bands, double* den, int32_t* info,
                           double pivot_tau, void* workspace, int64_t workspace_bytes, int64_t flags, void* ev_k3_start,
                           void* ev_k3_stop, void* stream) {
  if (!band_out) return fail(-4, "hmv_tf_ffdtf_bands_f64: null pointer / empty grid");
  return tf_ffdtf_impl("hmv_tf_ffdtf_bands_f64", ar, n_items, m, p, tw, F, nullptr, band_out, bin_lo, bin_hi, n_bands, den,
                       nullptr, info, pivot_tau, workspace, workspace_bytes, flags, ev_k3_start, ev_k3_stop, stream);
}

// ---- fused sliding-window path ----------------------------------------------------------------------
namespace {
struct SlidingWs {
  size_t off_R, off_Q, off_ws, off_ar, off_V, off_tf, off_den, off_tw, off_H, total;
};
SlidingWs sliding_layout(int64_t chunk, int mp, int p, int F, bool bands = false, bool spectra = false) {
  SlidingWs w;
  size_t o = 0;
  const size_t t = (size_t)mp * mp;
  w.off_R = o;      o += align256(sizeof(double) * chunk * (p + 1) * t);
  w.off_Q = o;      o += align256(sizeof(double) * (chunk + HMV_MAX_HOPS_PER_WINDOW - 1) * (p + 1) * t);   // hop-block sums
  w.off_ws = o;     o += align256(sizeof(double) * chunk * hmv::yw_ws_tiles(p) * t);
  w.off_ar = o;     o += align256(sizeof(double) * chunk * t * p);
  w.off_V = o;      o += align256(sizeof(double) * chunk * t);
  w.off_tf = o;     o += tf_ff_layout(chunk, mp, p, F, bands).total;
  w.off_den = o;    o += align256(sizeof(double) * chunk * mp);
  w.off_tw = o;     o += align256(sizeof(double) * F * p * 2);
  w.off_H = o;
  if (spectra) o += align256(sizeof(double) * 2 * chunk * F * t);       // H (complex) of one chunk, between K3 and K5
  w.total = o;
  return w;
}
// Fork / join events of the two-stream K2 split: created once per (host thread, device), not per call.
struct ForkJoin {
  hipEvent_t fork = nullptr, join = nullptr;
};
ForkJoin* fork_join_events() {
  constexpr int MAXDEV = 64;
  thread_local ForkJoin tl[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  ForkJoin& e = tl[dev];
  if (!e.fork) {
    if (hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &e;
}
}  // namespace

int64_t hmv_sliding_workspace_bytes(int64_t chunk, int m, int p, int F) {
  const int mp = pad_of(m);
  if (mp < 0 || chunk < 1 || p < 1 || p > HMV_MAX_ORDER || F < 1) return -1;
  return (int64_t)sliding_layout(chunk, mp, p, F).total;
}

namespace {
int sliding_impl(const char* who, const double* x, int64_t rec_stride, int64_t ld, const int64_t* item_rec,
                 const int64_t* item_start, int64_t n_items, int m, int n, int p, const double* freqs, int F, double fs,
                 double* ffdtf, double* band_out, const int32_t* bin_lo, const int32_t* bin_hi, int n_bands, double* S_out,
                 double* ar_out, double* V_out, int32_t* info_yw, int32_t* info_tf, void* workspace, int64_t workspace_bytes, int64_t chunk,
                 double pivot_tau, int64_t flags, int64_t grid_hop, int64_t grid_first, int64_t grid_nwin, int64_t grid_T,
                 void* ev_k3_start, void* ev_k3_stop, void* stream, void* aux_stream) {
  const bool bands = (band_out != nullptr);
  auto fail = [&](int code, const char* msg) {
    const char* own = strchr(msg, ':');             // messages below are written "hmv_sliding_ffdtf_f64: ..."
    char buf[220];
    snprintf(buf, sizeof(buf), "%s%s", who, own ? own : msg);
    return ::fail(code, buf);
  };
  const int mp = pad_of(m);
  if (mp < 0) return fail(-1, "hmv_sliding_ffdtf_f64: channel count must be in 1..64");
  if (p < 1 || p > HMV_MAX_ORDER) return fail(-2, "hmv_sliding_ffdtf_f64: model order must be in 1..32");
  if (n <= p) return fail(-3, "hmv_sliding_ffdtf_f64: window shorter than the model order");
  if (n_items == 0) return 0;                                    // empty batch: nothing to do, nothing to check
  if (!x || !item_rec || !item_start || !freqs || (!ffdtf && !bands) || !info_yw || !info_tf || !workspace || F < 1 || chunk < 1)
    return fail(-4, "hmv_sliding_ffdtf_f64: null pointer / empty grid");
  if (bands && (!bin_lo || !bin_hi || n_bands < 1)) return fail(-4, "hmv_sliding_ffdtf_f64: band bins missing");
  const SlidingWs w = sliding_layout(chunk, mp, p, F, bands, S_out != nullptr);
  if ((int64_t)w.total > workspace_bytes) return fail(-7, "hmv_sliding_ffdtf_f64: workspace too small");
  // Regular grid (the caller vouches: item = rec * grid_nwin + w starts at grid_first + w * grid_hop of recording rec,
  // recordings are grid_T samples long): K1 sums every hop block once and assembles the windows from the blocks.
  bool regular = grid_hop > 0 && !(flags & HMV_FLAG_DIRECT_LAGCOV);
  if (regular) {
    if (grid_nwin < 1 || grid_first < 0 || n_items % grid_nwin != 0 || grid_first + (grid_nwin - 1) * grid_hop + n > grid_T ||
        ld < grid_T)
      return fail(-9, "hmv_sliding_ffdtf_f64: inconsistent regular window grid");
    regular = (n % grid_hop == 0) && (n / grid_hop >= 2) && (n / grid_hop <= HMV_MAX_HOPS_PER_WINDOW) && grid_hop > p;
  }
  // Second stream (HMV_FLAG_YW_TILED only): the tile-per-workgroup form of K2 is a chain of ~25 launches of at
  // most a few workgroups per window that cannot fill the chip; it runs as two half-batches, one per stream,
  // whose launches interleave on the device (fork after K1, join before K3).  The default one-launch form of K2
  // needs none of this.  Chunk pipelining (K1/K2 of chunk c+1 under K3 of chunk c) was measured and does NOT
  // work: K3 holds every wave slot and starves the other stream.
  hipStream_t st0 = S(stream), st1 = S(aux_stream);
  const bool split = (aux_stream && aux_stream != stream) && !(flags & HMV_FLAG_YW_ONE_LAUNCH);
  ForkJoin* fj = nullptr;
  if (split) {
    fj = fork_join_events();
    if (!fj) return fail(-8, "hmv_sliding_ffdtf_f64: cannot create fork/join events");
  }
  int rc = 0;
  const size_t t = (size_t)mp * mp;
  const int64_t n_chunks = (n_items + chunk - 1) / chunk;
  char* base = static_cast<char*>(workspace);
  double* R = reinterpret_cast<double*>(base + w.off_R);
  double* Qb = reinterpret_cast<double*>(base + w.off_Q);
  double* ws = reinterpret_cast<double*>(base + w.off_ws);
  double* ar = reinterpret_cast<double*>(base + w.off_ar);
  double* V = reinterpret_cast<double*>(base + w.off_V);
  void* tfws = base + w.off_tf;
  double* den = reinterpret_cast<double*>(base + w.off_den);
  double* tw = reinterpret_cast<double*>(base + w.off_tw);
  const int64_t tfws_bytes = (int64_t)tf_ff_layout(chunk, mp, p, F, bands).total;
  rc = hmv_twiddles_f64(freqs, F, fs, p, tw, st0);
  const size_t ws_item = (size_t)hmv_yw_workspace_doubles(m, p);
  for (int64_t ci = 0; ci < n_chunks && rc == 0; ++ci) {
    const int64_t i0 = ci * chunk;
    const int64_t c = (n_items - i0 < chunk) ? (n_items - i0) : chunk;
    double* ar_c = ar_out ? ar_out + (size_t)i0 * t * p : ar;
    double* V_c = V_out ? V_out + (size_t)i0 * t : V;
    if (regular) {
      // items i0 .. i0+c-1 as runs of consecutive windows of one recording each (item = rec * grid_nwin + w)
      for (int64_t it = i0; it < i0 + c && rc == 0;) {
        const int64_t rec = it / grid_nwin, w0 = it - rec * grid_nwin;
        const int64_t run = ((grid_nwin - w0) < (i0 + c - it)) ? (grid_nwin - w0) : (i0 + c - it);
        rc = hmv_lagcov_regular_f64(x + rec * rec_stride, ld, grid_T, grid_first + w0 * grid_hop, grid_hop, run, m, n, p,
                                    R + (size_t)(it - i0) * (p + 1) * t, Qb, st0);
        it += run;
      }
    } else {
      rc = hmv_lagcov_f64(x, rec_stride, ld, item_rec + i0, item_start + i0, c, m, n, p, R, st0);
    }
    if (rc) break;
    // the tiled form of K2 (asked for, or chosen for a large 64-channel chunk) as two half-batches
    // K2: the Levinson-Whittle recursion unless an LDL^T form is asked for (or HMV_TUNE_YW_FORM = 1, which picks the
    // LDL^T form by batch shape as before: the launch chain for large 64-channel chunks)
    const bool ldl = (flags & (HMV_FLAG_YW_TILED | HMV_FLAG_YW_ONE_LAUNCH)) || hmv::tuning(HMV_TUNE_YW_FORM) == 1;
    const bool tiled = (flags & HMV_FLAG_YW_TILED) || (ldl && !(flags & HMV_FLAG_YW_ONE_LAUNCH) && mp == 64 && c >= 128);
    const int64_t yw_flags = (flags & ~(int64_t)(HMV_FLAG_YW_TILED | HMV_FLAG_YW_ONE_LAUNCH)) |
                             (ldl ? (tiled ? HMV_FLAG_YW_TILED : HMV_FLAG_YW_ONE_LAUNCH) : 0);
    const int64_t c0 = (split && tiled && c >= 16) ? (c + 1) / 2 : c, c1 = c - c0;
    if (c1 > 0) {
      // fork: st1 may start once K1 is done.  Whatever happens on st1 afterwards, st0 joins it again before this call
      // returns, so that the caller's stream never runs ahead of work this call put on the second stream.
      int hrc = (int)hipEventRecord(fj->fork, st0);
      if (!hrc) hrc = (int)hipStreamWaitEvent(st1, fj->fork, 0);
      if (hrc) { rc = hrc; break; }                      // nothing was put on st1
      rc = hmv_yw_solve_f64(R + (size_t)c0 * (p + 1) * t, c1, m, p, ws + (size_t)c0 * ws_item,
                            ar_c + (size_t)c0 * t * p, V_c + (size_t)c0 * t, nullptr, info_yw + i0 + c0, yw_flags, st1);
      hrc = (int)hipEventRecord(fj->join, st1);
      if (!hrc) hrc = (int)hipStreamWaitEvent(st0, fj->join, 0);
      if (hrc) {                                         // cannot express the join as an event: join on the host
        (void)hipStreamSynchronize(st1);
        if (!rc) rc = hrc;
      }
      if (rc) break;
    }
    rc = hmv_yw_solve_f64(R, c0, m, p, ws, ar_c, V_c, nullptr, info_yw + i0, yw_flags, st0);
    if (rc) break;
    const bool last = (ci == n_chunks - 1);
    double* Hc = S_out ? reinterpret_cast<double*>(base + w.off_H) : nullptr;
    rc = tf_ffdtf_impl(who, ar_c, c, m, p, tw, F, bands ? nullptr : ffdtf + (size_t)i0 * m * m * F,
                       bands ? band_out + (size_t)i0 * m * m * n_bands : nullptr, bin_lo, bin_hi, n_bands, den, Hc,
                       info_tf + (size_t)i0 * F, pivot_tau, tfws, tfws_bytes, flags, last ? ev_k3_start : nullptr,
                       last ? ev_k3_stop : nullptr, st0);
    if (!rc && S_out) {      // K5 from the same inverses and the same fit; V is this library's own (symmetric) estimate
      hmv::SpecArgs sa;
      sa.H = Hc; sa.V = V_c; sa.S = nullptr; sa.S_mmf = S_out + (size_t)i0 * m * m * F * 2; sa.n_items = c; sa.F = F; sa.m = m;
      sa.sym = 1;
      rc = hmv::launch_spectra(sa, mp, st0);
    }
  }
  return rc;
}
}  // namespace

int64_t hmv_sliding_bands_workspace_bytes(int64_t chunk, int m, int p, int F) {
  const int mp = pad_of(m);
  if (mp < 0 || chunk < 1 || p < 1 || p > HMV_MAX_ORDER || F < 1) return -1;
  return (int64_t)sliding_layout(chunk, mp, p, F, true).total;
}

int hmv_sliding_ffdtf_f64(const double* x, int64_t rec_stride, int64_t ld, const int64_t* item_rec,
                          const int64_t* item_start, int64_t n_items, int m, int n, int p,
                          const double* freqs, int F, double fs, double* ffdtf, double* ar_out, double* V_out,
                          int32_t* info_yw, int32_t* info_tf, void* workspace, int64_t workspace_bytes,
                          int64_t chunk, double pivot_tau, int64_t flags, int64_t grid_hop, int64_t grid_first,
                          int64_t grid_nwin, int64_t grid_T, void* ev_k3_start, void* ev_k3_stop, void* stream,
                          void* aux_stream) {
  return sliding_impl("hmv_sliding_ffdtf_f64", x, rec_stride, ld, item_rec, item_start, n_items, m, n, p, freqs, F, fs, ffdtf,
                      nullptr, nullptr, nullptr, 0, nullptr, ar_out, V_out, info_yw, info_tf, workspace, workspace_bytes, chunk,
                      pivot_tau, flags, grid_hop, grid_first, grid_nwin, grid_T, ev_k3_start, ev_k3_stop, stream, aux_stream);
}

int hmv_sliding_ffdtf_bands_f64(const double* x, int64_t rec_stride, int64_t ld, const int64_t* item_rec,
                                const int64_t* item_start, int64_t n_items, int m, int n, int p,
                                const double* freqs, int F, double fs, double* band_out, const int32_t* bin_lo,
                                const int32_t* bin_hi, int n_bands, double* ar_out, double* V_out,
                                int32_t* info_yw, int32_t* info_tf, void* workspace, int64_t workspace_bytes,
                                int64_t chunk, double pivot_tau, int64_t flags, int64_t grid_hop, int64_t grid_first,
                                int64_t grid_nwin, int64_t grid_T, void* ev_k3_start, void* ev_k3_stop, void* stream,
                                void* aux_stream) {
  if (!band_out && n_items != 0) return fail(-4, "hmv_sliding_ffdtf_bands_f64: null pointer / empty grid");
  return sliding_impl("hmv_sliding_ffdtf_bands_f64", x, rec_stride, ld, item_rec, item_start, n_items, m, n, p, freqs, F, fs,
                      nullptr, band_out, bin_lo, bin_hi, n_bands, nullptr, ar_out, V_out, info_yw, info_tf, workspace,
                      workspace_bytes, chunk, pivot_tau, flags, grid_hop, grid_first, grid_nwin, grid_T, ev_k3_start, ev_k3_stop,
                      stream, aux_stream);
}

int64_t hmv_sliding_spectra_workspace_bytes(int64_t chunk, int m, int p, int F) {
  const int mp = pad_of(m);
  if (mp < 0 || chunk < 1 || p < 1 || p > HMV_MAX_ORDER || F < 1) return -1;
  return (int64_t)sliding_layout(chunk, mp, p, F, false, true).total;
}

int hmv_sliding_ffdtf_spectra_f64(const double* x, int64_t rec_stride, int64_t ld, const int64_t* item_rec,
                                  const int64_t* item_start, int64_t n_items, int m, int n, int p,
                                  const double* freqs, int F, double fs, double* ffdtf, double* S_out, double* ar_out,
                                  double* V_out, int32_t* info_yw, int32_t* info_tf, void* workspace,
                                  int64_t workspace_bytes, int64_t chunk, double pivot_tau, int64_t flags,
                                  int64_t grid_hop, int64_t grid_first, int64_t grid_nwin, int64_t grid_T, void* stream,
                                  void* aux_stream) {
  if (!S_out && n_items != 0) return fail(-4, "hmv_sliding_ffdtf_spectra_f64: null pointer / empty grid");
  return sliding_impl("hmv_sliding_ffdtf_spectra_f64", x, rec_stride, ld, item_rec, item_start, n_items, m, n, p, freqs, F,
                      fs, ffdtf, nullptr, nullptr, nullptr, 0, S_out, ar_out, V_out, info_yw, info_tf, workspace,
                      workspace_bytes, chunk, pivot_tau, flags, grid_hop, grid_first, grid_nwin, grid_T, nullptr, nullptr,
                      stream, aux_stream);
}

}  // extern "C"
