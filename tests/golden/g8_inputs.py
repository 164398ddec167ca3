"""Seeded inputs of the g8 fixture (FAA preprocessing chain): regenerated identically by the generator script
and by the test, so that only the reference's OUTPUTS are stored."""
import numpy as np

NAMES = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "P3", "P4", "O1", "O2", "F7", "F8", "T3", "T4", "T5", "T6",
         "Fz", "Cz", "Pz"]
FS_EEG = 128.0


def g8_inputs():
    """{'eeg_ch', 'eeg_cg': (19, 10240) noise + common alpha, 'ibi8_ch', 'ibi8_cg': (480,) 8 Hz series}."""
    rng = np.random.default_rng(2024)
    out = {}
    t = np.arange(int(80 * FS_EEG)) / FS_EEG
    for role in ("ch", "cg"):
        eeg = rng.standard_normal((19, t.size)) + 0.8 * np.sin(2 * np.pi * (9.5 + rng.random()) * t)[None]
        eeg[3] *= 1.5
        out[f"eeg_{role}"] = eeg
        out[f"ibi8_{role}"] = rng.standard_normal(480).cumsum() * 0.05 + rng.standard_normal(480)
    return out
