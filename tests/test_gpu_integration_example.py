"""The ctypes stub printed in INTEGRATION.md section 2 is executed as written (only the library path is made
absolute) and checked against the oracle: the documentation cannot drift from the C ABI."""
import os
import re

import numpy as np
import pytest

from oracle import mvar_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_oracle():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "hmv_sliding_ffdtf_f64.argtypes" in b)
    lib = os.path.join(ROOT, "hyperscanning_signal_analysis_amd", "libhypermvar.so")
    stub = stub.replace('ctypes.CDLL("libhypermvar.so")', f'ctypes.CDLL(r"{lib}")')
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, 3000))
    x[:, 1:] += 0.5 * x[:, :-1]
    freqs = np.linspace(1.0, 40.0, 7)
    starts = [0, 700, 2000]
    ff = ns["full_freq_dtf_windows"](x, starts, 1000, freqs, 100.0, 3)
    assert ff.shape == (3, 6, 6, 7)
    for k, s in enumerate(starts):
        ref = O.full_freq_dtf(x[:, s:s + 1000], freqs, 100.0, 3)
        assert np.abs(ff[k] - ref).max() <= 1e-9 * np.abs(ref).max()


def test_integration_md_reduced_product_stub_runs_and_matches_oracle():
    """The second stub of INTEGRATION.md (section 5): `hmv_sliding_ffdtf_bands_f64` bound with ctypes alone."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "hmv_sliding_ffdtf_bands_f64.argtypes" in b)
    lib = os.path.join(ROOT, "hyperscanning_signal_analysis_amd", "libhypermvar.so")
    stub = stub.replace('ctypes.CDLL("libhypermvar.so")', f'ctypes.CDLL(r"{lib}")')
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((6, 3000))
    x[:, 1:] += 0.5 * x[:, :-1]
    freqs = np.linspace(1.0, 48.0, 32)
    starts = [0, 700, 2000]
    bands = [(1.0, 8.0), (8.0, 30.0), (30.0, 100.0)]
    got = ns["band_ffdtf_windows"](x, starts, 1000, freqs, 100.0, 3, bands)
    assert got.shape == (3, 6, 6, 3)
    for k, s in enumerate(starts):
        ref = O.full_freq_dtf(x[:, s:s + 1000], freqs, 100.0, 3)
        want = np.stack([ref[..., (freqs >= a) & (freqs < b)].sum(axis=-1) for a, b in bands], axis=-1)
        assert np.abs(got[k] - want).max() <= 1e-9 * np.abs(want).max()
    assert np.abs(got.sum(axis=(2, 3)) - 1.0).max() < 1e-12        # the three bands tile the grid: rows sum to one
