"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/hypermvar.h
declares, rejects bad arguments without touching a GPU, and the product never falls back to the oracle."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hyperscanning_signal_analysis_amd")


@pytest.fixture(scope="module")
def lib():
    from hyperscanning_signal_analysis_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j", "8"], check=True)
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hypermvar.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hmv_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(lib):
    from hyperscanning_signal_analysis_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 12
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and include/hypermvar.h disagree"
    for s in syms:
        assert hasattr(lib, s)
    hdr = open(os.path.join(ROOT, "include", "hypermvar.h")).read()
    assert lib.hmv_version() == int(re.search(r"#define HMV_VERSION (\d+)", hdr).group(1))


def test_argument_checks_without_gpu(lib):
    assert lib.hmv_pad(3) == 16 and lib.hmv_pad(16) == 16 and lib.hmv_pad(19) == 32 and lib.hmv_pad(64) == 64
    assert lib.hmv_pad(0) == -1 and lib.hmv_pad(65) == -1
    # K2 scratch: the larger of the two solvers' needs (block LDL^T: 2 T(p+1) + 2p tiles; recursion: 4p + 6) + the guard tile
    assert lib.hmv_yw_workspace_doubles(64, 8) == (2 * 45 + 16 + 1) * 64 * 64
    assert lib.hmv_yw_workspace_doubles(4, 1) == (4 * 1 + 6 + 1) * 16 * 16
    assert lib.hmv_sliding_workspace_bytes(1, 64, 8, 256) > 8 * 64 * 64 * 256
    # bad arguments are refused before any launch
    assert lib.hmv_lagcov_f64(0, 0, 0, 0, 0, 1, 65, 100, 4, 0, 0) == -1
    assert b"channel count" in lib.hmv_last_error()
    assert lib.hmv_lagcov_f64(0, 0, 0, 0, 0, 1, 8, 100, 40, 0, 0) == -2
    assert lib.hmv_lagcov_f64(0, 0, 0, 0, 0, 1, 8, 4, 4, 0, 0) == -3
    assert lib.hmv_tf_f64(1, 1, 8, 2, 1, 4, 0, 0, 0, 0, 1, 1.5, 1, 0) == -6
    assert lib.hmv_tf_workspace_doubles(3, 20, 5) == 3 * 32 * 32 * 6


def test_product_does_not_import_oracle_or_fall_back():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{f} mentions the oracle"
    # engine refuses to run without a GPU instead of computing on the CPU
    import torch
    if not torch.cuda.is_available():
        from hyperscanning_signal_analysis_amd.engine import Engine
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            Engine()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from hyperscanning_signal_analysis_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_bad_window_descriptors_are_refused_before_any_launch():
    """item_rec / item_start go straight into kernel address arithmetic, so dtype, device, recording
    index and window bounds are checked on the host side first (pure tensor logic: runs on CPU tensors here)."""
    import torch
    from hyperscanning_signal_analysis_amd.engine import validate_items
    x = torch.zeros(2, 4, 1000, dtype=torch.float64)
    rec = torch.tensor([0, 1, 1], dtype=torch.int64)
    st = torch.tensor([0, 10, 800], dtype=torch.int64)
    validate_items(x, rec, st, 200, 5)                                   # fine: last window ends at 1000
    validate_items(x, rec[:0], st[:0], 200, 5)                           # empty batch
    for bad_rec, bad_st, n, p, msg in [
        (rec.to(torch.int32), st, 200, 5, "int64"),
        (rec, st.to(torch.float64), 200, 5, "int64"),
        (rec, st[:2], 200, 5, "same length"),
        (torch.tensor([0, 2, 1]), st, 200, 5, "item_rec must lie"),
        (torch.tensor([0, -1, 1]), st, 200, 5, "item_rec must lie"),
        (rec, torch.tensor([0, 10, 801]), 200, 5, "must lie in"),
        (rec, torch.tensor([-1, 10, 800]), 200, 5, "must lie in"),
        (rec, st, 5, 5, "exceed the model order"),
        (rec, st, 1001, 5, "exceeds the recording length"),
    ]:
        with pytest.raises(ValueError, match=msg):
            validate_items(x, bad_rec, bad_st, n, p)


def test_host_sanitizer_build_of_the_c_abi():
    """the host code of the C-ABI shim (argument checks, workspace layouts, launch sequencing) built
    with AddressSanitizer + UBSan (`make -C csrc asan`; device code untouched) and driven through every entry
    point's refusal paths in a child process under the ASan runtime.  CPU only; never on the GPU box."""
    import glob
    asan_lib = os.path.join(PKG, "libhypermvar_asan.so")
    if not os.path.exists(asan_lib):
        r = subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "asan", "-j", "8"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("host-sanitized build not available on this box: " + r.stderr[-300:])
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert rt, "ASan runtime of the ROCm clang not found"
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("HYPERMVAR_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asan driver ok" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
