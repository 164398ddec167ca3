"""Runs in a child process under the ASan runtime (tests/test_cabi_cpu.py): drives every C-ABI entry point of the
host-sanitized build (make -C csrc asan) through its argument checks and its pure host helpers.  No GPU: every
call here is refused before anything is launched, or is a pure host function."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HYPERMVAR_LIB"] = os.path.join(ROOT, "hyperscanning_signal_analysis_amd", "libhypermvar_asan.so")
from hyperscanning_signal_analysis_amd import _lib  # noqa: E402

lib = _lib.load()
D = 0x1000          # a non-null "device pointer" that is never dereferenced on the host
n = 0
for m in (0, 1, 16, 19, 64, 65, -3):
    lib.hmv_pad(m); n += 1
for m, p in ((64, 8), (3, 1), (65, 8), (64, 0), (64, 33)):
    lib.hmv_yw_workspace_doubles(m, p); n += 1
for args in ((1, 64, 8, 256), (599, 64, 8, 256), (0, 64, 8, 256), (5, 65, 8, 256), (5, 64, 40, 256), (5, 4, 5, 0)):
    lib.hmv_sliding_workspace_bytes(*args); lib.hmv_tf_ffdtf_workspace_bytes(*args); n += 2
lib.hmv_tf_workspace_doubles(10, 64, 8); lib.hmv_tf_workspace_doubles(-1, 64, 8); lib.hmv_psd_workspace_bytes(4, 1000, 7)
lib.hmv_lagcov_regular_workspace_doubles(599, 64, 1000, 500, 8); lib.hmv_lagcov_regular_workspace_doubles(5, 64, 1000, 300, 8)
lib.hmv_psd_workspace_bytes(0, 1, 0)
assert lib.hmv_dpss_workspace_bytes(1000, 8, 0) > 0 and lib.hmv_dpss_workspace_bytes(1, 1, 0) < 0 and lib.hmv_dpss_workspace_bytes(10, 11, 1) < 0
bad = [
    lib.hmv_lagcov_f64(D, 0, 0, D, D, 1, 65, 100, 4, D, 0),
    lib.hmv_lagcov_f64(D, 0, 0, D, D, 1, 4, 100, 40, D, 0),
    lib.hmv_lagcov_f64(D, 0, 0, D, D, 1, 4, 3, 4, D, 0),
    lib.hmv_lagcov_f64(0, 0, 0, D, D, 1, 4, 100, 4, D, 0),
    lib.hmv_lagcov_regular_f64(D, 1000, 1000, 0, 500, 1, 99, 1000, 4, D, D, 0),
    lib.hmv_lagcov_regular_f64(D, 1000, 1000, 0, 300, 1, 4, 1000, 4, D, D, 0),
    lib.hmv_lagcov_regular_f64(D, 1000, 1000, 600, 500, 1, 4, 1000, 4, D, D, 0),
    lib.hmv_lagcov_regular_f64(D, 1000, 1000, 0, 500, 1, 4, 1000, 4, D, 0, 0),
    lib.hmv_sliding_ffdtf_f64(D, 0, 1000, D, D, 3, 4, 100, 4, D, 8, 100.0, D, 0, 0, D, D, D, 1 << 30, 3, 1.0, 0, 50, 0, 2, 1000, 0, 0, 0, 0),
    lib.hmv_yw_solve_f64(D, 1, 70, 4, D, D, D, 0, D, 0, 0),
    lib.hmv_yw_solve_f64(D, 1, 4, 0, D, D, D, 0, D, 0, 0),
    lib.hmv_yw_solve_f64(0, 1, 4, 4, D, D, D, 0, D, 0, 0),
    lib.hmv_twiddles_f64(0, 4, 100.0, 2, D, 0),
    lib.hmv_tf_f64(D, 1, 99, 2, D, 4, D, D, 0, 0, D, 1.0, D, 0),
    lib.hmv_tf_f64(D, 1, 4, 2, D, 4, D, 0, 0, 0, D, 1.0, D, 0),
    lib.hmv_tf_f64(D, 1, 4, 2, D, 4, D, D, 0, 0, D, 0.0, D, 0),
    lib.hmv_tf_ffdtf_f64(D, 1, 99, 2, D, 4, D, D, 0, D, 1.0, D, 1 << 20, 0, 0, 0, 0),
    lib.hmv_tf_ffdtf_f64(D, 5, 4, 2, D, 4, D, D, D, D, 1.0, D, 16, 0, 0, 0, 0),
    lib.hmv_tf_ffdtf_f64(D, 5, 4, 2, D, 4, D, D, 0, D, 2.0, D, 1 << 30, 0, 0, 0, 0),
    lib.hmv_ffdtf_norm_f64(D, D, D, D, 1, 4, 99, 1, 0),
    lib.hmv_ffdtf_norm_f64(0, D, D, D, 1, 4, 4, 1, 0),
    lib.hmv_transpose_c128(0, D, 1, 4, 4, 0),
    lib.hmv_spectra_f64(0, D, D, 1, 4, 4, 0),
    lib.hmv_spectra_mmf_f64(D, 0, D, 1, 4, 4, 0),
    lib.hmv_spectra_mmf_f64(D, D, D, 1, 99, 4, 0),
    lib.hmv_pack_c128(D, 0, 1, 4, 4, 0),
    lib.hmv_cinv_c128(D, 1, 4, 4, D, 0, D, 1.5, 0),
    lib.hmv_partial_coherence_c128(D, 0, D, 1, 4, 4, 0),
    lib.hmv_gpdc_f64(D, D, 0, 1, 4, 4, 0),
    lib.hmv_trial_mean_f64(D, 0, 4, 2, D, 0),
    lib.hmv_ddtf_f64(0, D, D, 1, 4, 4, 0),
    lib.hmv_band_sums_f64(D, 4, 0, D, D, 2, D, 0),
    lib.hmv_psd_multitaper_f64(D, 4, 1000, 900, D, D, 3, 1, 100, D, D, 1 << 30, 4, 0),
    lib.hmv_psd_multitaper_f64(D, 4, 1000, 1000, D, D, 3, 1, 600, D, D, 1 << 30, 4, 0),
    lib.hmv_psd_multitaper_f64(D, 4, 1000, 1000, D, D, 3, 1, 100, D, D, 8, 4, 0),
    lib.hmv_dpss_f64(1000, 4.0, 8, 0, 0, D, D, 1 << 40, 0),
    lib.hmv_dpss_f64(1000, 4.0, 1001, 0, D, D, D, 1 << 40, 0),
    lib.hmv_dpss_f64(1000, 500.0, 8, 0, D, D, D, 1 << 40, 0),
    lib.hmv_dpss_f64(1000, 4.0, 8, 1, D, D, D, 64, 0),
    lib.hmv_sliding_ffdtf_f64(D, 0, 0, D, D, 3, 65, 100, 4, D, 8, 100.0, D, 0, 0, D, D, D, 1 << 30, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
    lib.hmv_sliding_ffdtf_f64(D, 0, 0, D, D, 3, 4, 100, 4, D, 8, 100.0, D, 0, 0, D, D, D, 64, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
    lib.hmv_sliding_ffdtf_f64(D, 0, 0, D, D, 3, 4, 100, 4, 0, 8, 100.0, D, 0, 0, D, D, D, 1 << 30, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
    # the reduced-product entries: band bins missing, grids the row workers cannot take, unfused flag, null output
    lib.hmv_tf_ffdtf_bands_f64(D, 5, 4, 2, D, 32, D, 0, D, 2, D, D, 1.0, D, 1 << 30, 0, 0, 0, 0),
    lib.hmv_tf_ffdtf_bands_f64(D, 5, 4, 2, D, 48, D, D, D, 2, D, D, 1.0, D, 1 << 30, 0, 0, 0, 0),
    lib.hmv_tf_ffdtf_bands_f64(D, 5, 4, 2, D, 672, D, D, D, 2, D, D, 1.0, D, 1 << 30, 0, 0, 0, 0),
    lib.hmv_tf_ffdtf_bands_f64(D, 5, 4, 2, D, 32, D, D, D, 2, D, D, 1.0, D, 1 << 30, 1, 0, 0, 0),
    lib.hmv_tf_ffdtf_bands_f64(D, 5, 4, 2, D, 32, 0, D, D, 2, D, D, 1.0, D, 1 << 30, 0, 0, 0, 0),
    lib.hmv_tf_ffdtf_bands_f64(D, 5, 4, 2, D, 32, D, D, D, 2, D, D, 1.0, D, 16, 0, 0, 0, 0),
    lib.hmv_sliding_ffdtf_bands_f64(D, 0, 0, D, D, 3, 4, 100, 4, D, 32, 100.0, D, D, 0, 2, 0, 0, D, D, D, 1 << 30, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
    lib.hmv_sliding_ffdtf_bands_f64(D, 0, 0, D, D, 3, 4, 100, 4, D, 32, 100.0, 0, D, D, 2, 0, 0, D, D, D, 1 << 30, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
    lib.hmv_sliding_ffdtf_bands_f64(D, 0, 0, D, D, 3, 4, 100, 4, D, 32, 100.0, D, D, D, 2, 0, 0, D, D, D, 64, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
]
assert all(rc < 0 for rc in bad), bad
assert lib.hmv_sliding_ffdtf_f64(D, 0, 0, D, D, 0, 4, 100, 4, D, 8, 100.0, D, 0, 0, D, D, D, 0, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == 0
assert lib.hmv_tf_ffdtf_f64(D, 0, 4, 2, D, 4, D, D, 0, D, 1.0, D, 0, 0, 0, 0, 0) == 0        # empty batches: nothing to do
assert lib.hmv_sliding_ffdtf_bands_f64(D, 0, 0, D, D, 0, 4, 100, 4, D, 32, 100.0, D, D, D, 2, 0, 0, D, D, D, 0, 3, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == 0
assert lib.hmv_tf_ffdtf_bands_workspace_bytes(30, 64, 8, 256) > lib.hmv_tf_ffdtf_workspace_bytes(30, 64, 8, 256) > 0
assert lib.hmv_sliding_bands_workspace_bytes(30, 64, 8, 256) > lib.hmv_sliding_workspace_bytes(30, 64, 8, 256) > 0
assert len(lib.hmv_last_error()) > 0
# tuning knobs: range-checked, readable back, process-wide
assert lib.hmv_set_tuning(1, 16) == 0 and lib.hmv_get_tuning(1) == 16 and lib.hmv_set_tuning(1, 0) == 0
assert lib.hmv_set_tuning(0, 1) < 0 and lib.hmv_set_tuning(99, 1) < 0 and lib.hmv_set_tuning(2, 7) < 0 and lib.hmv_set_tuning(3, -1) < 0
assert lib.hmv_get_tuning(99) == -1 and lib.hmv_get_tuning(2) == 0
print(f"asan driver ok: {n + len(bad)} calls")
