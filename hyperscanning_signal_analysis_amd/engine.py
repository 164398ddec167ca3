"""Device-side engine: stages buffers with PyTorch-ROCm and drives the HIP kernels through the C ABI.

PyTorch is plumbing here (device memory, streams, H2D/D2H copies); all arithmetic of the hot path
runs in libhypermvar.so.  Every method takes / returns torch tensors that live on the engine's device;
`hyperscanning_signal_analysis_amd.mtmvar` wraps them into the reference's NumPy signatures.

Layouts ("MP layout": channels padded to MP = 16*ceil(m/16)):
    R       (items, p+1, MP, MP)      lag covariances                               K1
    ar      (items, MP, MP, p)        AR coefficients, lag fastest                  K2
    V       (items, MP, MP)           residual covariance                           K2
    P/H/A/S (items, F, MP, MP)        kernel-natural per-frequency matrices         K3 / K5
    ffdtf   (items, m, m, F)          the reference's (m, m, F) array per window    K4
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

__all__ = ["Engine", "default_engine", "SingularMatrixError", "validate_items"]


# Pivot threshold of the per-frequency inverses of A(f) (K3): a row interchange happens only when some row's |re| + |im|
# exceeds the diagonal's by more than a factor 1 / tau (threshold partial pivoting: multipliers bounded by 1 / tau).
# tau = 1 is LAPACK's partial pivoting.  A(f) = I - sum_k A_k z_k is close to the identity; with tau = 1 nine per cent of
# its pivot columns interchange two rows whose candidates differ by a few per cent -- an interchange that buys no
# accuracy (max-norm distance to numpy.linalg.inv 2.2e-15 either way, measured on the north-star dyad) and costs every
# wave of the workgroup a trip through LDS: K3 8.19 -> 7.64 ms at tau = 0.25, where 0.2 % of the columns still
# interchange (profiles/r03_k3_ab_notes.md).  `Engine(pivot_tau=1.0)` or HYPERMVAR_PIVOT_TAU=1 restores LAPACK's rule;
# the general complex inverse (partial coherence of arbitrary spectral matrices) always uses tau = 1.
DEFAULT_PIVOT_TAU = 0.25


class SingularMatrixError(np.linalg.LinAlgError):
    """Raised where the reference's np.linalg.solve / np.linalg.inv raise LinAlgError('Singular matrix')."""


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def validate_items(x: torch.Tensor, item_rec: torch.Tensor, item_start: torch.Tensor, n: int, p: int):
    """Refuse window descriptors the kernels would read out of bounds with, BEFORE anything is launched: wrong
    dtype / device, a recording index outside x, a window that does not lie inside its recording (one tiny
    reduction on the tensors' device).  x: (n_rec, m, T).  Pure tensor logic: also runs on CPU tensors."""
    n_rec, m, T = x.shape
    for name, t in (("item_rec", item_rec), ("item_start", item_start)):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.int64 or t.device != x.device or t.dim() != 1:
            raise ValueError(f"{name} must be a 1-D int64 tensor on {x.device}")
    if item_rec.numel() != item_start.numel():
        raise ValueError("item_rec and item_start must have the same length")
    if int(n) <= int(p):
        raise ValueError(f"window length ({n}) must exceed the model order ({p})")
    if int(n) > T:
        raise ValueError(f"window length ({n}) exceeds the recording length ({T})")
    if item_rec.numel() == 0:
        return
    lim = torch.stack([item_rec.min(), item_rec.max(), item_start.min(), item_start.max()]).cpu().tolist()
    if lim[0] < 0 or lim[1] >= n_rec:
        raise ValueError(f"item_rec must lie in [0, {n_rec}), got [{lim[0]}, {lim[1]}]")
    if lim[2] < 0 or lim[3] + int(n) > T:
        raise ValueError(f"windows [start, start + {n}) must lie in [0, {T}), got starts in [{lim[2]}, {lim[3]}]")


class Engine:
    def __init__(self, device=None, pivot_tau: float | None = None, max_workspace_bytes: int = 24 << 30):
        import os
        if pivot_tau is None:
            pivot_tau = float(os.environ.get("HYPERMVAR_PIVOT_TAU", DEFAULT_PIVOT_TAU))
        self.lib = _lib.load()                      # fails loudly when the HIP library is not built
        if not torch.cuda.is_available():
            raise RuntimeError("hypermvar needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.pivot_tau = float(pivot_tau)
        self.max_workspace_bytes = int(max_workspace_bytes)
        self._ws = {}
        self._aux = {}        # second HIP stream for chunk overlap in the fused path

    # ------------------------------------------------------------------ helpers
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def to_device(self, a, dtype=torch.float64):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def pad(self, m: int) -> int:
        mp = self.lib.hmv_pad(int(m))
        if mp < 0:
            raise ValueError(f"hypermvar supports 1..64 channels, got {m}")
        return mp

    def _workspace(self, nbytes: int):
        """Cached scratch of the fused call, one buffer per HIP stream: two calls on different streams never share
        (and so never race on) a workspace; calls on one stream are ordered by the stream."""
        key = self.stream()
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws.pop(key, None)
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    @staticmethod
    def raise_on_info(info: torch.Tensor, what: str, per_item: int = 1):
        """LAPACK-style info array -> the reference's LinAlgError("Singular matrix") (np.linalg.solve / inv).
        The exception additionally carries which items failed (`.items`, `.info`: first bad pivot column, 1-based;
        with `per_item` = F for the per-frequency inverses, `.freqs` too)."""
        bad = torch.nonzero(info != 0).flatten()
        if bad.numel() == 0:
            return
        idx = bad.cpu().numpy()
        err = SingularMatrixError("Singular matrix")
        err.stage = what
        err.items = np.unique(idx // per_item)
        err.freqs = (idx % per_item) if per_item > 1 else None
        err.info = info[bad].cpu().numpy()
        err.args = ("Singular matrix", f"{what}: {len(err.items)} item(s) failed, first: item {int(err.items[0])}"
                    + (f" frequency {int(err.freqs[0])}" if per_item > 1 else "") + f" pivot column {int(err.info[0])}")
        raise err

    def check_items(self, x: torch.Tensor, item_rec: torch.Tensor, item_start: torch.Tensor, n: int, p: int):
        validate_items(x, item_rec, item_start, n, p)

    # ------------------------------------------------------------------ K1
    def lagcov(self, x: torch.Tensor, item_rec: torch.Tensor, item_start: torch.Tensor, n: int, p: int):
        """x: (n_rec, m, T) float64 device tensor -> R (items, p+1, MP, MP)."""
        assert x.dim() == 3 and x.dtype == torch.float64 and x.is_cuda
        x = x if x.stride(2) == 1 else x.contiguous()
        n_rec, m, T = x.shape
        mp = self.pad(m)
        self.check_items(x, item_rec, item_start, n, p)
        n_items = int(item_rec.numel())
        R = self.empty(n_items, p + 1, mp, mp)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_lagcov_f64(x.data_ptr(), x.stride(0), x.stride(1), item_rec.data_ptr(),
                                         item_start.data_ptr(), n_items, m, int(n), int(p), R.data_ptr(),
                                         self.stream())
        _lib.check(rc, "hmv_lagcov_f64")
        return R

    def lagcov_regular(self, x: torch.Tensor, first: int, hop: int, n_win: int, n: int, p: int):
        """x: (m, T) ONE recording; windows first + w*hop .. + n (n = k hops) -> R (n_win, p+1, MP, MP) with the hop
        blocks summed once and shared by the overlapping windows (`hmv_lagcov_regular_f64`)."""
        assert x.dim() == 2 and x.dtype == torch.float64 and x.is_cuda
        x = x if x.stride(1) == 1 else x.contiguous()
        m, T = x.shape
        mp = self.pad(m)
        nws = int(self.lib.hmv_lagcov_regular_workspace_doubles(n_win, m, int(n), int(hop), int(p)))
        if nws < 0:
            raise ValueError("the window must be a whole number of hops")
        ws = self.empty(max(nws, 1))
        R = self.empty(n_win, p + 1, mp, mp)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_lagcov_regular_f64(x.data_ptr(), x.stride(0), T, int(first), int(hop), int(n_win), m, int(n),
                                                 int(p), R.data_ptr(), ws.data_ptr(), self.stream())
        _lib.check(rc, "hmv_lagcov_regular_f64")
        return R

    def trial_mean(self, R: torch.Tensor, m: int):
        """(trials, p+1, MP, MP) -> (1, p+1, MP, MP): count_corr's average over trials (mtmvar.py:78-85)."""
        trials, p1, mp, _ = R.shape
        out = self.empty(1, p1, mp, mp)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_trial_mean_f64(R.data_ptr(), trials, m, p1 - 1, out.data_ptr(), self.stream())
        _lib.check(rc, "hmv_trial_mean_f64")
        return out

    # ------------------------------------------------------------------ K2
    def yw_solve(self, R: torch.Tensor, m: int, want_logdet: bool = False, flags: int = 0):
        n_items, p1, mp, _ = R.shape
        p = p1 - 1
        ws = self.empty(n_items * int(self.lib.hmv_yw_workspace_doubles(m, p)))
        ar = self.empty(n_items, mp, mp, p)
        V = self.empty(n_items, mp, mp)
        info = self.empty(n_items, dtype=torch.int32)
        logdet = self.empty(n_items, p) if want_logdet else None
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_yw_solve_f64(R.data_ptr(), n_items, m, p, ws.data_ptr(), ar.data_ptr(), V.data_ptr(),
                                           _ptr(logdet), info.data_ptr(), int(flags), self.stream())
        _lib.check(rc, "hmv_yw_solve_f64")
        return ar, V, logdet, info

    # ------------------------------------------------------------------ K3 (+K4/K5 layout kernels)
    def twiddles(self, freqs, fs: float, p: int):
        f = self.to_device(np.asarray(freqs, dtype=np.float64))
        tw = self.empty(f.numel(), p, 2)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_twiddles_f64(f.data_ptr(), f.numel(), float(fs), p, tw.data_ptr(), self.stream())
        _lib.check(rc, "hmv_twiddles_f64")
        return tw

    def transfer(self, ar: torch.Tensor, m: int, tw: torch.Tensor, want_P=True, want_H=False, want_A=False):
        """ar (items, MP, MP, p) -> dict of kernel-natural (items, F, MP, MP) tensors + info."""
        n_items, mp, _, p = ar.shape
        F = tw.shape[0]
        out = {}
        P = self.empty(n_items, F, mp, mp) if want_P else None
        rowsum = self.empty(n_items, F, mp) if want_P else None
        H = self.empty(n_items, F, mp, mp, 2) if want_H else None
        A = self.empty(n_items, F, mp, mp, 2) if want_A else None
        info = self.empty(n_items * F, dtype=torch.int32)
        ws = self.empty(max(1, int(self.lib.hmv_tf_workspace_doubles(n_items, m, p))))
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_tf_f64(ar.data_ptr(), n_items, m, p, tw.data_ptr(), F, _ptr(P), _ptr(rowsum),
                                     _ptr(H), _ptr(A), info.data_ptr(), self.pivot_tau, ws.data_ptr(),
                                     self.stream())
        _lib.check(rc, "hmv_tf_f64")
        out.update(P=P, rowsum=rowsum, H=H, A=A, info=info)
        return out

    def normalise(self, P, rowsum, m: int, normalise: bool = True):
        n_items, F, mp, _ = P.shape
        den = self.empty(n_items, mp)
        out = self.empty(n_items, m, m, F)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_ffdtf_norm_f64(P.data_ptr(), _ptr(rowsum), den.data_ptr(), out.data_ptr(), n_items,
                                             F, m, 1 if normalise else 0, self.stream())
        _lib.check(rc, "hmv_ffdtf_norm_f64")
        return out, den

    def to_mmf_complex(self, Z: torch.Tensor, m: int):
        """(items, F, MP, MP, 2) -> complex128 (items, m, m, F), the reference's H / A / spectra layout."""
        n_items, F, mp, _, _ = Z.shape
        out = self.empty(n_items, m, m, F, 2)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_transpose_c128(Z.data_ptr(), out.data_ptr(), n_items, F, m, self.stream())
        _lib.check(rc, "hmv_transpose_c128")
        return torch.view_as_complex(out)

    def spectra(self, H: torch.Tensor, V: torch.Tensor, m: int):
        n_items, F, mp, _, _ = H.shape
        S = self.empty(n_items, F, mp, mp, 2)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_spectra_f64(H.data_ptr(), V.data_ptr(), S.data_ptr(), n_items, m, F, self.stream())
        _lib.check(rc, "hmv_spectra_f64")
        return S

    # ------------------------------------------------------------------ measures on top of K3 / K5
    def pack_complex(self, Z: torch.Tensor):
        """complex128 (items, m, m, F) -> kernel layout (items, F, MP, MP, 2), identity on the padding."""
        n_items, m, _, F = Z.shape
        mp = self.pad(m)
        zin = torch.view_as_real(Z.contiguous())
        out = self.empty(n_items, F, mp, mp, 2)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_pack_c128(zin.data_ptr(), out.data_ptr(), n_items, F, m, self.stream())
        _lib.check(rc, "hmv_pack_c128")
        return out

    def complex_inverse(self, Z: torch.Tensor, m: int):
        """Z (items, F, MP, MP, 2) -> (inverse, det/|det| (items*F, 2), info)."""
        n_items, F, mp, _, _ = Z.shape
        Zi = self.empty(n_items, F, mp, mp, 2)
        detph = self.empty(n_items * F, 2)
        info = self.empty(n_items * F, dtype=torch.int32)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_cinv_c128(Z.data_ptr(), n_items, m, F, Zi.data_ptr(), detph.data_ptr(), info.data_ptr(),
                                        1.0, self.stream())
        _lib.check(rc, "hmv_cinv_c128")
        return Zi, detph, info

    def partial_coherence(self, S: torch.Tensor, m: int):
        """S spectral matrices (items, F, MP, MP, 2) in kernel layout -> kappa complex128 (items, m, m, F)."""
        n_items, F, mp, _, _ = S.shape
        Si, detph, info = self.complex_inverse(S, m)
        kap = self.empty(n_items, F, mp, mp, 2)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_partial_coherence_c128(Si.data_ptr(), detph.data_ptr(), kap.data_ptr(), n_items, m, F,
                                                     self.stream())
        _lib.check(rc, "hmv_partial_coherence_c128")
        return self.to_mmf_complex(kap, m), info

    def ddtf(self, ff: torch.Tensor, kappa: torch.Tensor):
        """ffDTF (items, m, m, F) real x |partial coherence| (items, m, m, F) complex128 -> dDTF (mtmvar.py:341-385)."""
        n_items, m, _, F = ff.shape
        ff = ff.contiguous()
        kr = torch.view_as_real(kappa.contiguous())
        out = self.empty(n_items, m, m, F)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_ddtf_f64(ff.data_ptr(), kr.data_ptr(), out.data_ptr(), n_items, m, F, self.stream())
        _lib.check(rc, "hmv_ddtf_f64")
        return out

    def band_tables(self, bin_lo, bin_hi, F: int):
        """Device copies (int32) of the bin ranges [bin_lo[b], bin_hi[b]) of a band set on an F-point grid, cached."""
        lo_h, hi_h = np.asarray(bin_lo, dtype=np.int32), np.asarray(bin_hi, dtype=np.int32)
        if (lo_h < 0).any() or (hi_h > F).any() or (hi_h < lo_h).any() or lo_h.shape != hi_h.shape or lo_h.ndim != 1:
            raise ValueError("band bin ranges must satisfy 0 <= lo <= hi <= F")
        # the bin tables live on the device per distinct band set: a fresh pageable-memory copy per call is a synchronous
        # copy on the compute stream, i.e. a host wait for everything queued before it (the streamed path's stall)
        key = (lo_h.tobytes(), hi_h.tobytes())
        cache = self.__dict__.setdefault("_band_tables", {})
        if key not in cache:
            if len(cache) >= 16:
                cache.pop(next(iter(cache)))
            cache[key] = (torch.as_tensor(lo_h).to(self.device), torch.as_tensor(hi_h).to(self.device))
        return cache[key]

    def bands_in_kernel(self, m: int, F: int) -> bool:
        """Can K3's row workers add up the bands themselves (`sliding_ffdtf(bands=...)` without the full array)?"""
        cap = {16: 640, 32: 1280, 48: 1984, 64: 2688}[self.pad(m)]     # doubles of LDS the row worker has, whole 32s
        return F % 32 == 0 and F <= cap

    def band_sums(self, ff: torch.Tensor, bin_lo, bin_hi):
        """(..., F) -> (..., n_bands): sums over the bin ranges [bin_lo[b], bin_hi[b])."""
        F = ff.shape[-1]
        ff = ff.contiguous()
        lo, hi = self.band_tables(bin_lo, bin_hi, F)
        nb = int(lo.numel())
        rows = ff.numel() // F if F else 0
        out = self.empty(*ff.shape[:-1], nb)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_band_sums_f64(ff.data_ptr(), rows, F, lo.data_ptr(), hi.data_ptr(), nb, out.data_ptr(),
                                            self.stream())
        _lib.check(rc, "hmv_band_sums_f64")
        return out

    def gpdc(self, A: torch.Tensor, V: torch.Tensor, m: int):
        """A (items, F, MP, MP, 2) from `transfer(want_A=True)`, V (items, MP, MP) -> GPDC (items, m, m, F)."""
        n_items, F, mp, _, _ = A.shape
        G = self.empty(n_items, F, mp, mp)
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_gpdc_f64(A.data_ptr(), V.data_ptr(), G.data_ptr(), n_items, m, F, self.stream())
        _lib.check(rc, "hmv_gpdc_f64")
        out, _ = self.normalise(G, None, m, normalise=False)
        return out

    # ------------------------------------------------------------------ fused sliding-window path
    def sliding_chunk(self, n_items: int, m: int, p: int, F: int, lanes: int = 1) -> int:
        per_item = int(self.lib.hmv_sliding_workspace_bytes(1, m, p, F))
        cap = max(1, self.max_workspace_bytes // max(per_item * lanes, 1))
        want = n_items
        return max(1, min(want, cap))

    def copy_streams(self):
        """The upload and the download stream of `stream_dyads`, created ONCE per engine.  HIP streams share a handful of
        hardware queues (four by default) and two streams on one queue run one after the other: with fresh streams per
        call the download of recording d and the upload of recording d + 1 ended up on the compute stream's queue and
        the whole pipeline ran serially (device timeline, `tools/dbg/e2e_depth.py`: compute -> download -> upload ->
        compute, 12.2 ms per dyad instead of 10.4).  High priority: their own queues, apart from the compute streams'."""
        if "_copy" not in self.__dict__:
            self._copy = (torch.cuda.Stream(self.device, priority=-1), torch.cuda.Stream(self.device, priority=-1))
        return self._copy

    def aux_stream(self):
        """Second stream of the fused call (the tiled form of K2 runs as two half-batches), one per calling stream."""
        key = self.stream()
        if key not in self._aux:
            self._aux[key] = torch.cuda.Stream(device=self.device)
        return self._aux[key]

    def sliding_ffdtf(self, x: torch.Tensor, item_rec: torch.Tensor, item_start: torch.Tensor, n: int, p: int,
                      freqs, fs: float, out: torch.Tensor | None = None, return_ar: bool = False,
                      check: bool = True, chunk: int | None = None, k3_events=None, overlap: bool = True,
                      flags: int = 0, grid=None, validate: bool = True, bands=None):
        """ffDTF of every window: x (n_rec, m, T) -> (items, m, m, F).  One C-ABI call (K1->K2->K3->K4).

        check: True raises numpy.linalg.LinAlgError("Singular matrix") if ANY window failed, like the reference's
        np.linalg.solve / inv (the exception names the windows); "nan" returns every window and NaN-fills the
        failed ones; False skips the check (and its synchronisation); "mask" returns (out, bad) with `bad` the boolean
        device mask of the failed windows and no synchronisation (the caller NaN-fills what it keeps).
        validate=False skips the bounds check of the window descriptors and of a declared grid (two host
        synchronisations): for callers that validated the same descriptors before (`stream_dyads`).
        overlap: give the library a second stream: the Yule-Walker stage (K2), whose launches cannot fill the
        chip, then runs as two half-batches that interleave on the device (see include/hypermvar.h).
        flags: option bits of include/hypermvar.h (`_lib.FLAG_*`); 0 = the fast defaults.
        grid: (hop, first, n_win) when the items are a REGULAR grid -- item = rec * n_win + w is the window starting
        at first + w * hop of recording rec (`sliding.regular_grid` derives it from the start positions): K1 then sums
        every hop block once and shares it between the overlapping windows (half its flops at 50 % overlap; equal to
        the direct form to rounding, not bitwise -- `_lib.FLAG_DIRECT_LAGCOV` keeps the direct form).
        k3_events: optional pair of raw hipEvent_t handles (`torch.cuda.Event.cuda_event` of events that
        have been recorded once) which the library records around the dominant kernel.
        bands: (bin_lo, bin_hi) -- the REDUCED product: instead of the (items, m, m, F) array the call returns its band
        sums (items, m, m, n_bands), band b = sum over the bins bin_lo[b] <= f < bin_hi[b] (`distributed.band_bins`).
        The row workers inside K3 add them up and the full array is never written (`hmv_sliding_ffdtf_bands_f64`); the
        same bits as `band_sums(sliding_ffdtf(...))`, which is also what runs when the grid does not suit the kernel
        (`bands_in_kernel`).  `out`, if given, is the band array.
        """
        assert x.dim() == 3 and x.dtype == torch.float64 and x.is_cuda
        x = x if x.stride(2) == 1 else x.contiguous()
        n_rec, m, T = x.shape
        mp = self.pad(m)
        if validate:
            self.check_items(x, item_rec, item_start, n, p)
        n_items = int(item_rec.numel())
        f = freqs if isinstance(freqs, torch.Tensor) else self.to_device(np.asarray(freqs, dtype=np.float64))
        F = int(f.numel())
        if n_items == 0:                      # empty batch (torch gives empty tensors a null data pointer)
            empty = self.empty(0, m, m, F)
            if return_ar:
                return empty, self.empty(0, mp, mp, p), self.empty(0, mp, mp), (self.empty(0, dtype=torch.int32),) * 2
            return empty
        band_out = None
        if bands is not None:
            b_lo, b_hi = self.band_tables(bands[0], bands[1], F)
            nb = int(b_lo.numel())
            if nb == 0 or not self.bands_in_kernel(m, F) or (flags & _lib.FLAG_UNFUSED_NORM):
                # two calls: the full array (scratch), then its band sums
                res = self.sliding_ffdtf(x, item_rec, item_start, n, p, f, fs, return_ar=return_ar, check=check, chunk=chunk,
                                         k3_events=k3_events, overlap=overlap, flags=flags, grid=grid, validate=validate)
                full = res[0] if isinstance(res, tuple) else res
                red = self.band_sums(full, bands[0], bands[1])
                if out is not None:
                    out.copy_(red)
                    red = out
                return (red,) + tuple(res[1:]) if isinstance(res, tuple) else red
            band_out = self.empty(n_items, m, m, nb) if out is None else out
            assert band_out.is_contiguous() and tuple(band_out.shape) == (n_items, m, m, nb)
        chunk = self.sliding_chunk(n_items, m, p, F) if chunk is None else int(chunk)
        wsf = self.lib.hmv_sliding_workspace_bytes if band_out is None else self.lib.hmv_sliding_bands_workspace_bytes
        nbytes = int(wsf(chunk, m, p, F))
        ws = self._workspace(nbytes)
        aux = self.aux_stream().cuda_stream if overlap else 0
        if out is None and band_out is None:
            out = self.empty(n_items, m, m, F)
        ar = self.empty(n_items, mp, mp, p) if return_ar else None
        V = self.empty(n_items, mp, mp) if return_ar else None
        info_yw = self.empty(n_items, dtype=torch.int32)
        info_tf = self.empty(n_items * F, dtype=torch.int32)
        g_hop, g_first, g_nwin = (int(v) for v in grid) if grid is not None else (0, 0, 0)
        if grid is not None:
            # With a declared grid K1 addresses the windows by (item // n_win, first + (item % n_win) * hop) and never
            # reads item_rec / item_start: they must say the same thing, or the results belong to other windows (and a
            # recording index past n_rec would be read out of bounds).  One device comparison per call.
            if g_nwin < 1 or n_items % g_nwin or g_hop < 1 or n_items // g_nwin > n_rec:
                raise ValueError("grid = (hop, first, n_win) does not match the number of items / recordings")
            k = torch.arange(n_items, dtype=torch.int64, device=self.device)
            same = (not validate) or (torch.equal(item_rec, k // g_nwin) and
                                      torch.equal(item_start, g_first + (k % g_nwin) * g_hop))
            if not same:
                raise ValueError("grid = (hop, first, n_win) contradicts item_rec / item_start "
                                 "(items must be recording-major, window-minor on the declared grid)")
        with torch.cuda.device(self.device):
            if band_out is None:
                rc = self.lib.hmv_sliding_ffdtf_f64(
                    x.data_ptr(), x.stride(0), x.stride(1), item_rec.data_ptr(), item_start.data_ptr(), n_items,
                    m, int(n), int(p), f.data_ptr(), F, float(fs), out.data_ptr(), _ptr(ar), _ptr(V),
                    info_yw.data_ptr(), info_tf.data_ptr(), ws.data_ptr(), nbytes, chunk, self.pivot_tau, int(flags),
                    g_hop, g_first, g_nwin, T, k3_events[0] if k3_events else 0, k3_events[1] if k3_events else 0,
                    self.stream(), aux)
            else:
                rc = self.lib.hmv_sliding_ffdtf_bands_f64(
                    x.data_ptr(), x.stride(0), x.stride(1), item_rec.data_ptr(), item_start.data_ptr(), n_items,
                    m, int(n), int(p), f.data_ptr(), F, float(fs), band_out.data_ptr(), b_lo.data_ptr(), b_hi.data_ptr(), nb,
                    _ptr(ar), _ptr(V), info_yw.data_ptr(), info_tf.data_ptr(), ws.data_ptr(), nbytes, chunk, self.pivot_tau,
                    int(flags), g_hop, g_first, g_nwin, T, k3_events[0] if k3_events else 0,
                    k3_events[1] if k3_events else 0, self.stream(), aux)
                out = band_out
        _lib.check(rc, "hmv_sliding_ffdtf_f64" if band_out is None else "hmv_sliding_ffdtf_bands_f64")
        if check == "nan":          # keep the good windows, NaN-fill the ones whose fit or inverse was singular
            badw = (info_yw != 0) | (info_tf.view(n_items, F) != 0).any(dim=1)
            if bool(badw.any()):
                out[badw] = float("nan")
        elif check == "mask":       # no host synchronisation: the caller decides what to NaN-fill (streamed recordings)
            return out, (info_yw != 0) | (info_tf.view(n_items, F) != 0).any(dim=1)
        elif check:
            self.raise_on_info(info_yw, "ar_coeff (Yule-Walker solve)")
            self.raise_on_info(info_tf, "mvar_transfer_function (inverse of A(f))", per_item=F)
        if return_ar:
            return out, ar, V, (info_yw, info_tf)
        return out


    # ------------------------------------------------------------------ recordings streamed from the host
    def stream_dyads(self, dyads, n: int, positions, p: int, freqs, fs: float, bands=None, reduce=None, depth: int = 3,
                     check="nan", keep_full=None, timeline=None, out=None):
        """The reference's outer loop -- load a recording, compute, save, next (eeg_alpha_ibi_ffdtf.py:661-806) -- as a
        pipeline: while recording d computes, recording d + 1 crosses PCIe on a copy stream and the reduced result of
        recording d - 1 goes back on another one.  `dyads`: an iterable of host arrays (m, T) float64 -- NumPy arrays
        (staged through pinned buffers) or pinned torch tensors (copied as they are).  Every recording gets the windows
        `positions` (length n).  What leaves the device per recording is `reduce(ffdtf)` -- by default the band-integrated
        ffDTF (windows, m, m, n_bands) of `bands` (`distributed.DEFAULT_BANDS`): full-resolution ffDTF is 5 GB per
        10-minute dyad, i.e. >= 80 ms of PCIe against ~10 ms of compute.  `keep_full(d, ffdtf_device)` is called (on the
        compute stream's timeline) for callers that want to consume the full array on the device.
        `depth`: recordings in flight (buffer slots).  With two, "result of d - 2 on the host -> upload of d -> compute of
        d" is a chain as long as a dyad's compute and the compute stream waits for its input (0.84 of the resident rate,
        measured); the third slot takes the copies off the critical path.
        Returns the list of reduced results as NumPy arrays, in order.  `out`: optional pinned host tensor
        (n_recordings, *reduced shape) that receives the results directly (no host-side copy: at 98 MB per 10-minute
        dyad that copy alone takes longer than the dyad's compute); the returned arrays are then views of it.  Same
        bits as the resident path: the arithmetic does not know where its input came from (tests/test_gpu_pipeline.py)."""
        from . import distributed as hdist
        from .sliding import regular_grid, window_items
        dev = self.device
        f = freqs if isinstance(freqs, torch.Tensor) else self.to_device(np.asarray(freqs, dtype=np.float64))
        fhost = f.cpu().numpy()
        lo, hi = hdist.band_bins(fhost, hdist.DEFAULT_BANDS if bands is None else bands)
        # default product without a consumer of the full array: K3's row workers add the bands up themselves and the 5 GB
        # per dyad are never written (`sliding_ffdtf(bands=...)`); same bits as band_sums(full array)
        in_kernel_bands = (reduce is None and keep_full is None)
        if reduce is None:
            def reduce(ff):
                return self.band_sums(ff, lo, hi)
        pos = np.asarray(positions, dtype=np.int64)
        item_rec, item_start = window_items(1, pos, dev)
        grid = regular_grid(pos, n, p)
        validated = False                    # the descriptors are the same for every recording: checked once
        import collections
        import time
        comp = torch.cuda.current_stream(dev)
        s_in, s_out = self.copy_streams()
        slots, results, pending, deferred = [], [], collections.deque(), None
        # timeline: besides the host-side marks, the device-side intervals of every recording (HIP events on the three
        # streams, read after the last recording): when its upload, its kernels and its download started and ended
        tev = [] if timeline is not None else None
        if tev is not None:
            t_base = torch.cuda.Event(enable_timing=True)
            t_base.record(comp)

        def mark(stream):
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            return e

        def collect_one():
            d, k = pending.popleft()
            slots[k]["d2h"].synchronize()
            results.append(out[d].numpy() if out is not None else slots[k]["out_pin"].numpy().copy())
            if timeline is not None:
                timeline.append(("collected", d, time.perf_counter()))

        for d, arr in enumerate(dyads):
            k = d % depth
            while pending and pending[0][0] <= d - depth:      # slot k's previous recording: result on the host first
                if deferred is not None and pending[0][0] == d - 1:     # (depth 1: its download is not even queued yet)
                    deferred()
                    deferred = None
                collect_one()
            pinned_in = isinstance(arr, torch.Tensor) and arr.is_pinned()
            host = arr if isinstance(arr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
            m, T = host.shape
            if len(slots) <= k:
                slots.append({"x": self.empty(1, m, T),
                              "ff": None if in_kernel_bands else self.empty(len(pos), m, m, int(f.numel())), "in_pin": None,
                              "h2d": torch.cuda.Event(), "done": torch.cuda.Event(), "d2h": torch.cuda.Event(),
                              "red": None, "out_pin": None})
            sl = slots[k]
            if tuple(sl["x"].shape[1:]) != (m, T):
                raise ValueError("stream_dyads: every recording must have the same shape")
            if pinned_in:
                src = host
            else:
                if sl["in_pin"] is None:
                    sl["in_pin"] = torch.empty(m, T, dtype=torch.float64).pin_memory()
                elif d >= depth:
                    sl["h2d"].synchronize()                    # the copy out of this staging buffer has finished
                sl["in_pin"].copy_(host)                       # host memcpy, under the GPU's work on earlier recordings
                src = sl["in_pin"]
            with torch.cuda.stream(s_in):
                if d >= depth:
                    s_in.wait_event(sl["done"])                # the kernels that read x[k] have finished
                if tev is not None:
                    tev.append({"d": d, "h0": mark(s_in)})
                sl["x"][0].copy_(src, non_blocking=True)
                sl["h2d"].record(s_in)
                if tev is not None:
                    tev[-1]["h1"] = mark(s_in)
            # The download of the PREVIOUS recording is queued only now, behind this recording's upload: both directions
            # go through one in-order copy queue, and a download queued first -- it waits for its kernels -- keeps the
            # next upload, and with it the next recording's kernels, waiting too (device timeline of the first
            # recordings, tools/dbg/e2e_depth.py: compute -> download -> upload -> compute, nothing overlapped)
            if deferred is not None:
                deferred()
                deferred = None
            comp.wait_event(sl["h2d"])
            if tev is not None:
                tev[-1]["c0"] = mark(comp)
            if in_kernel_bands:
                if sl["red"] is None:
                    sl["red"] = self.empty(len(pos), m, m, len(lo))
                elif d >= depth:
                    comp.wait_event(sl["d2h"])                 # (already collected on the host: a formality)
                red, bad = self.sliding_ffdtf(sl["x"], item_rec, item_start, n, p, f, fs, out=sl["red"], check="mask",
                                              grid=grid, validate=not validated, bands=(lo, hi))
            else:
                ff, bad = self.sliding_ffdtf(sl["x"], item_rec, item_start, n, p, f, fs, out=sl["ff"], check="mask", grid=grid,
                                             validate=not validated)
                if keep_full is not None:
                    if check == "nan":
                        ff.masked_fill_(bad.view(-1, 1, 1, 1), float("nan"))
                    keep_full(d, ff)
                red = reduce(ff)
            validated = True
            if check == "nan" and red.shape[0] == bad.shape[0]:     # NaN-fill what leaves the device (98 MB, not 5 GB)
                red.masked_fill_(bad.view(-1, *([1] * (red.dim() - 1))), float("nan"))
            if d < depth:                                      # first use of this slot: where the result goes on the host
                if out is None:
                    sl["out_pin"] = torch.empty(red.shape, dtype=red.dtype).pin_memory()
                elif not (out.is_pinned() and tuple(out.shape[1:]) == tuple(red.shape) and out.dtype == red.dtype):
                    raise ValueError("stream_dyads: `out` must be a pinned host tensor (recordings, *%s)" % (tuple(red.shape),))
            if not in_kernel_bands:                            # (in-kernel bands: `red` IS the slot's buffer)
                if sl["red"] is None or sl["red"].shape != red.shape:
                    sl["red"] = torch.empty_like(red)
                elif d >= depth:
                    comp.wait_event(sl["d2h"])                 # (already collected on the host: a formality)
                sl["red"].copy_(red)
            sl["done"].record(comp)
            if tev is not None:
                tev[-1]["c1"] = mark(comp)
            def download(sl=sl, d=d, te=(tev[-1] if tev is not None else None)):
                with torch.cuda.stream(s_out):
                    s_out.wait_event(sl["done"])
                    if te is not None:
                        te["o0"] = mark(s_out)
                    (out[d] if out is not None else sl["out_pin"]).copy_(sl["red"], non_blocking=True)
                    sl["d2h"].record(s_out)
                    if te is not None:
                        te["o1"] = mark(s_out)
            deferred = download
            pending.append((d, k))
            if timeline is not None:
                timeline.append(("queued", d, time.perf_counter()))
        if deferred is not None:
            deferred()
        while pending:
            collect_one()
        if tev is not None:
            torch.cuda.synchronize(dev)
            for t in tev:
                timeline.append(("device_ms", t["d"], {k: t_base.elapsed_time(t[k]) for k in ("h0", "h1", "c0", "c1", "o0", "o1")}))
        return results

    # ------------------------------------------------------------------ ffDTF + spectra from ONE fit
    def sliding_ffdtf_spectra(self, x: torch.Tensor, item_rec: torch.Tensor, item_start: torch.Tensor, n: int, p: int,
                              freqs, fs: float, chunk: int | None = None, check: bool = True, out_ff=None, out_S=None,
                              grid=None, flags: int = 0):
        """Both products the reference's orchestrators always compute together (full_freq_dtf + multivariate_spectra,
        /root/reference/src/eeg_alpha_ibi_ffdtf.py:592-604, src/mtmvar.py:1100-1113) from ONE fit and ONE set of
        inverses per window, in ONE C-ABI call (`hmv_sliding_ffdtf_spectra_f64`): K1 -> K2 -> K3 (ffDTF normalised
        in-kernel, H left in the workspace) -> K5 (S written in the reference's (m, m, F) layout), `chunk` windows at a
        time (H is 16.8 MB per window; default: as many as `max_workspace_bytes` allows).  grid / flags as in
        `sliding_ffdtf`.  Returns (ffdtf (items, m, m, F) real, S (items, m, m, F) complex)."""
        assert x.dim() == 3 and x.dtype == torch.float64 and x.is_cuda
        x = x if x.stride(2) == 1 else x.contiguous()
        n_rec, m, T = x.shape
        self.check_items(x, item_rec, item_start, n, p)
        n_items = int(item_rec.numel())
        f = freqs if isinstance(freqs, torch.Tensor) else self.to_device(np.asarray(freqs, dtype=np.float64))
        F = int(f.numel())
        ff = self.empty(n_items, m, m, F) if out_ff is None else out_ff
        S = self.empty(n_items, m, m, F, 2) if out_S is None else out_S
        if n_items == 0:
            return ff, torch.view_as_complex(S)
        if chunk is None:
            per_item = int(self.lib.hmv_sliding_spectra_workspace_bytes(1, m, p, F))
            chunk = max(1, min(n_items, self.max_workspace_bytes // max(per_item, 1)))
        chunk = int(chunk)
        nbytes = int(self.lib.hmv_sliding_spectra_workspace_bytes(chunk, m, p, F))
        ws = self._workspace(nbytes)
        info_yw = self.empty(n_items, dtype=torch.int32)
        info_tf = self.empty(n_items * F, dtype=torch.int32)
        g_hop, g_first, g_nwin = (int(v) for v in grid) if grid is not None else (0, 0, 0)
        if grid is not None:
            if g_nwin < 1 or n_items % g_nwin or g_hop < 1 or n_items // g_nwin > n_rec:
                raise ValueError("grid = (hop, first, n_win) does not match the number of items / recordings")
            k = torch.arange(n_items, dtype=torch.int64, device=self.device)
            if not (torch.equal(item_rec, k // g_nwin) and torch.equal(item_start, g_first + (k % g_nwin) * g_hop)):
                raise ValueError("grid = (hop, first, n_win) contradicts item_rec / item_start "
                                 "(items must be recording-major, window-minor on the declared grid)")
        with torch.cuda.device(self.device):
            rc = self.lib.hmv_sliding_ffdtf_spectra_f64(
                x.data_ptr(), x.stride(0), x.stride(1), item_rec.data_ptr(), item_start.data_ptr(), n_items,
                m, int(n), int(p), f.data_ptr(), F, float(fs), ff.data_ptr(), S.data_ptr(), 0, 0,
                info_yw.data_ptr(), info_tf.data_ptr(), ws.data_ptr(), nbytes, chunk, self.pivot_tau, int(flags),
                g_hop, g_first, g_nwin, T, self.stream(), self.aux_stream().cuda_stream)
        _lib.check(rc, "hmv_sliding_ffdtf_spectra_f64")
        if check:
            self.raise_on_info(info_yw, "ar_coeff (Yule-Walker solve)")
            self.raise_on_info(info_tf, "mvar_transfer_function (inverse of A(f))", per_item=F)
        return ff, torch.view_as_complex(S)


_default = None


def default_engine() -> Engine:
    global _default
    if _default is None:
        _default = Engine()
    return _default
