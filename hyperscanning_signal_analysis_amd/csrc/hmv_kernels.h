// Internal launch interface between the C-ABI (capi.hip) and the kernels.  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmv {

// ---- tuning knobs (capi.hip; include/hypermvar.h hmv_set_tuning).  Read by the launchers, never by kernels.
long long tuning(int key);

// ---- K1 lag covariance ------------------------------------------------------------------------
struct LagcovArgs {
  const double* x;          // [n_rec][m][ld]  (channel-major, sample-contiguous)
  long long rec_stride;     // doubles between recordings
  long long ld;             // doubles between channels
  const long long* item_rec;    // [n_items] recording index of each window (device)
  const long long* item_start;  // [n_items] first sample of each window (device)
  long long n_items;
  int m, n, p;              // channels, window length, model order
  double* R;                // [n_items][p+1][MP][MP]
  // hop-block mode (blocks != 0): item b is the block of `n` (= hop) samples starting at blk_first + b * n of ONE
  // recording x (item_rec / item_start unused); lagged partners beyond the block are read up to the end of the
  // recording (blk_T samples), the sums are left unscaled and the padding is left zero: R = partial sums Q_l(b).
  int blocks;
  long long blk_first, blk_T;
};
int launch_lagcov(const LagcovArgs& a, int m_pad, hipStream_t st);
// windows of k consecutive hop blocks from the block sums: R[w][l] = (sum_{j<k} Q[w+j][l] - C_l(w)) / n
struct LagcombArgs {
  const double* Q;          // [n_win + k - 1][p+1][MP][MP]
  const double* x;          // the recording: [m][ld]
  long long ld, first, hop, T;
  long long n_win;
  int k, m, p;
  double* R;                // [n_win][p+1][MP][MP]
};
int launch_lagcomb(const LagcombArgs& a, int m_pad, hipStream_t st);

// ---- K2 Yule-Walker solve (block LDL^T of the block-Toeplitz normal equations) ------------------
struct YwArgs {
  const double* R;          // [n_items][p+1][MP][MP]
  long long n_items;
  int m, p;
  double* ws;               // [n_items][ws_tiles(p)][MP][MP] scratch
  double* ar;               // [n_items][MP][MP][p]   (row, col, lag) -- lag fastest
  double* V;                // [n_items][MP][MP]
  double* Vq_logdet;        // optional [n_items][p]: log det V_q for q = 1..p (model-order criterion)
  int* info;                // [n_items]
  int tiled;                // block LDL^T forms (yw_solve.hip): 0: one workgroup per window, one launch;  1: one workgroup
                            // per tile, p + 2 launches;  -1: no form asked for -- the block Levinson-Whittle recursion
                            // (yw_lwr.hip) unless HMV_TUNE_YW_FORM says otherwise
  int only_guarded;         // internal: the one-launch LDL^T kernel re-solves only the windows the recursion flagged
};
long long yw_ws_tiles(int p);
int launch_yw(const YwArgs& a, int m_pad, hipStream_t st);
int launch_yw_lwr(const YwArgs& a, int m_pad, hipStream_t st);
int launch_yw_lwr2(const YwArgs& a, int m_pad, hipStream_t st);

// ---- K3 transfer matrix inverse ---------------------------------------------------------------
struct TfArgs {
  const double* ar;         // [n_items][MP][MP][p]  (reference layout; read by the packing kernel only)
  const double* arx;        // scratch, tf_workspace_doubles(): the same coefficients in K3's register order
  const double* tw;         // [F][p][2]
  const double* Zin;        // general inverse only: complex [n_items][F][MP][MP] (ar / arx / tw unused)
  double* detph;            // general inverse only, optional: [n_items*F][2] det / |det| incl. interchange sign
  double* P;                // optional [n_items][F][MP][MP]
  double* rowsum;           // required with P: [n_items][F][MP]
  double* H;                // optional complex [n_items][F][MP][MP]
  double* A;                // optional complex [n_items][F][MP][MP]
  int* info;                // [n_items*F]
  long long n_items;
  int F, p, m;
  double tau;               // pivot threshold: 1.0 = LAPACK partial pivoting
  // ffDTF normalisation inside K3 (hot path only; ff == nullptr: off).  Items < fuse_items publish their |H|^2
  // row-major; row i of item w < tail0 is normalised in-kernel by a workgroup of item w + lag, the rows of items
  // tail0 .. fuse_items - 1 (the last `lag` of the batch) by norm_missed_kernel right behind K3; the caller runs K4
  // on the items >= fuse_items (none on the hot path).
  double* ff;               // [n_items][m][m][F]
  double* den;              // [n_items][MP]
  int* wcount;              // [n_items] arrival counters          } one block of 2 * n_items + 1 ints that the
  int* ready;               // [n_items] denominators are in place  } launcher zeroes, followed by the list
  int* missed;              // [1 + fuse_items * MP] count, rows     } of rows left to norm_missed_kernel
  long long fuse_items;
  int lag;                  // (tail0 = max(0, fuse_items - lag): no field of its own -- the hand-scheduled K3 has no SGPR to spare)
  // reduced product (hot path, in-kernel normalisation only; bands == nullptr: off): instead of the ffDTF array the row
  // workers write its band sums, bands[item][i][j][b] = sum_{band_lo[b] <= f < band_hi[b]} ffdtf[item][i][j][f]
  // (the arithmetic of band_sums_stream_kernel, ffdtf_norm.hip: same partial sums, same tree, same bits); ff is unused
  double* bands;            // [n_items][m][m][nb]
  const int* band_lo;       // [nb] device
  const int* band_hi;       // [nb] device
  int nb;
  unsigned long long* stamps;   // diagnostic builds (-DHMV_STAMP) only: [wave][8] phase cycle sums; else null
};
int launch_twiddles(const double* freqs, int F, double fs, int p, double* tw, hipStream_t st);
int launch_tf_inv(const TfArgs& a, int m_pad, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int launch_cinv(const TfArgs& a, int m_pad, hipStream_t st);
long long tf_workspace_doubles(long long n_items, int m_pad, int p);
int tf_band_max_F(int m_pad);

// ---- K4 ffDTF normalisation + layout transposes -------------------------------------------------
struct NormArgs {
  const double* P;          // [n_items][F][MP][MP]
  const double* rowsum;     // [n_items][F][MP]
  double* den;              // [n_items][MP] scratch/out: sum_f rowsum
  double* out;              // [n_items][m][m][F]
  long long n_items;
  int F, m, m_pad;
  int normalise;            // 1: ffDTF, 0: plain |H|^2 (dtf_multivariate)
};
int launch_ffdtf_norm(const NormArgs& a, hipStream_t st);

int launch_trial_mean(const double* in, double* out, long long n, int trials, hipStream_t st);
int launch_ddtf(const double* ff, const double* kappa, double* out, long long n, hipStream_t st);
int launch_band_sums(const double* in, const int* lo, const int* hi, double* out, long long rows, int F, int nb,
                     hipStream_t st);

// complex [n_items][F][MP][MP] -> complex [n_items][m][m][F]
int launch_transpose_c128(const double* in, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st);

// ---- K5 spectra S(f) = H V H^T (plain transpose) ---------------------------------------------------
struct SpecArgs {
  const double* H;          // complex [n_items][F][MP][MP]
  const double* V;          // [n_items][MP][MP]
  double* S;                // complex [n_items][F][MP][MP] (kernel-natural), or
  double* S_mmf;            // complex [n_items][m][m][F], the reference's array layout, written by the kernel itself
  long long n_items;
  int F, m;
  int sym;                  // 1: V is symmetric (this library's own fit) -- with S_mmf only the upper triangle is computed
};
int launch_spectra(const SpecArgs& a, int m_pad, hipStream_t st);

// ---- measures on top of K3 / K5 (connect.hip) -----------------------------------------------------------
int launch_pack_c128(const double* in, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st);
int launch_pcoh(const double* Sinv, const double* detph, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st);
int launch_gpdc(const double* A, const double* V, double* out, long long n_items, int F, int m, int m_pad, hipStream_t st);

// ---- multitaper PSD (psd.hip; hipFFT for the transforms) -------------------------------------------------
long long psd_workspace_bytes(long long ch_chunk, long long n, int K);
int launch_psd(const double* x, long long n_ch, long long n, long long ld, const double* tapers, const double* w, int K,
               long long lo, long long hi, double* psd, void* workspace, long long ch_chunk, hipStream_t st);

// ---- DPSS tapers (dpss.hip; hipFFT for the concentration ratios) -------------------------------------------------
long long dpss_workspace_bytes(long long M, int K, int sym);
int launch_dpss(long long M, double NW, int K, int sym, double* tapers, double* ratios, void* workspace, hipStream_t st);

}  // namespace hmv
