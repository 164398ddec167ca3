"""hypermvar -- MI355X-native sliding-window MVAR / ffDTF connectivity engine.

Drop-in for the hot path of SYNCC-IN/hyperscanning-signal-analysis:
    from hyperscanning_signal_analysis_amd import mtmvar            # same functions as src/mtmvar.py
    from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline
    from hyperscanning_signal_analysis_amd.sliding import sliding_ffdtf   # batched dyad x window entry point

Importing this package does not touch the GPU; the HIP library is loaded (and required) on first use.
"""
__version__ = "0.1.0"
