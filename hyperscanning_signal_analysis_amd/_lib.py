"""ctypes binding of libhypermvar.so (the C ABI declared in include/hypermvar.h).

The shared library is built in-tree by `__graft_entry__.build()` / `make -C csrc`.  There is NO CPU
fallback: if the library is missing or a symbol is absent, importing the engine fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhypermvar.so")

# name -> (restype, argtypes); mirrors include/hypermvar.h one to one
SIGNATURES = {
    "hmv_version": (c_int, []),
    "hmv_last_error": (c_char_p, []),
    "hmv_pad": (c_int, [c_int]),
    "hmv_set_tuning": (c_int, [c_int, c_int64]),
    "hmv_get_tuning": (c_int64, [c_int]),
    "hmv_yw_workspace_doubles": (c_int64, [c_int, c_int]),
    "hmv_lagcov_f64": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                               c_void_p, c_void_p]),
    "hmv_lagcov_regular_workspace_doubles": (c_int64, [c_int64, c_int, c_int, c_int64, c_int]),
    "hmv_lagcov_regular_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "hmv_yw_solve_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_void_p]),
    "hmv_twiddles_f64": (c_int, [c_void_p, c_int, c_double, c_int, c_void_p, c_void_p]),
    "hmv_tf_workspace_doubles": (c_int64, [c_int64, c_int, c_int]),
    "hmv_tf_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "hmv_ffdtf_norm_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                   c_void_p]),
    "hmv_transpose_c128": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_spectra_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_spectra_mmf_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_pack_c128": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_cinv_c128": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p]),
    "hmv_partial_coherence_c128": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_gpdc_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_trial_mean_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "hmv_ddtf_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "hmv_band_sums_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "hmv_dpss_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "hmv_dpss_f64": (c_int, [c_int64, c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "hmv_psd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "hmv_psd_multitaper_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                       c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "hmv_sliding_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "hmv_tf_ffdtf_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "hmv_tf_ffdtf_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_double, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "hmv_tf_ffdtf_bands_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "hmv_tf_ffdtf_bands_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                       c_void_p]),
    "hmv_sliding_bands_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "hmv_sliding_ffdtf_bands_f64": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                            c_int, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_int,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double,
                                            c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_void_p]),
    "hmv_sliding_spectra_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "hmv_sliding_ffdtf_spectra_f64": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                              c_int, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_int64, c_int64,
                                              c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "hmv_sliding_ffdtf_f64": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                      c_int, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int64, c_int64, c_double, c_int64, c_int64, c_int64, c_int64,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
}

# option bits of the fused entry points (include/hypermvar.h)
FLAG_UNFUSED_NORM = 1
FLAG_YW_TILED = 2
FLAG_YW_ONE_LAUNCH = 4
FLAG_DIRECT_LAGCOV = 8
# hmv_set_tuning keys
TUNE_NORM_LAG = 1
TUNE_LAG_GROUP = 2
TUNE_K3_FORM = 3
TUNE_YW_FORM = 4
TUNE_K3_LDS_PAD = 5


_lib = None


class HypermvarLibraryError(ImportError):
    pass


def load():
    """Load libhypermvar.so and bind every symbol of include/hypermvar.h.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    # A/B measurements of kernel variants on ONE box: HYPERMVAR_LIB names another build of the same library
    LIB_PATH = os.environ.get("HYPERMVAR_LIB", LIB_PATH)
    if not os.path.exists(LIB_PATH):
        raise HypermvarLibraryError(
            f"{LIB_PATH} not found: build the HIP library first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or make -C {_HERE}/csrc). "
            "There is no CPU fallback for the MVAR/ffDTF path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise HypermvarLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc == 0:
        return
    lib = load()
    if rc < 0:
        raise ValueError(f"{what}: {lib.hmv_last_error().decode()} (code {rc})")
    raise RuntimeError(f"{what}: HIP error {rc}")
