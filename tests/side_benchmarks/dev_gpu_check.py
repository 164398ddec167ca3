import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import mvar_oracle as O
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad
from hyperscanning_signal_analysis_amd import mtmvar as M
from hyperscanning_signal_analysis_amd.engine import default_engine
from hyperscanning_signal_analysis_amd.sliding import sliding_ffdtf
def rel(a,b): return np.abs(a-b).max()/np.abs(b).max()
g = np.load(ROOT + '/tests/golden/g1_config1.npz')
x, fs, freqs = g['x'], float(g['fs']), g['freqs']
R = M.lag_covariances(x, 4); print('R m=3', rel(R, O.lag_covariances(x,4)))
ar, V = M.ar_coeff(x, 4); print('ar', rel(ar, g['ar']), 'V', rel(V, g['V']))
H, A = M.mvar_transfer_function(g['ar'], freqs, fs); print('H', rel(H, g['H']), 'A', rel(A, g['A']))
ff = M.full_freq_dtf(x, freqs, fs, optimal_model_order=4); print('ff', rel(ff, g['ffdtf']), np.abs(ff/g['ffdtf']-1).max())
S = M.multivariate_spectra(x, freqs, fs, optimal_model_order=4); print('S', rel(S, g['spectra']))
for c in ('AIC','HQ','SC'):
    crit, rng, popt = M.mvar_criterion(x, 10, c); print(c, np.abs(crit-g['crit_'+c]).max(), popt, g['crit_'+c+'_popt'])
g = np.load(ROOT + '/tests/golden/g2_northstar.npz')
x, fs, freqs, p = g['x'], float(g['fs']), g['freqs'], int(g['p'])
R = M.lag_covariances(x, p); print('R m=64', rel(R, g['R']))
ar, V = M.ar_coeff(x, p); print('ar', rel(ar, g['ar']), 'V', rel(V, g['V']))
ff = M.full_freq_dtf(x, freqs, fs, optimal_model_order=p)
print('ff sub', rel(ff[:,:,::16], g['ffdtf_sub']), np.abs(ff[:,:,::16]/g['ffdtf_sub']-1).max(), 'rowsum', np.abs(ff.sum(axis=(1,2))-1).max())
S = M.multivariate_spectra(x, freqs, fs, optimal_model_order=p); print('S', rel(S[:,:,::64], g['spectra_sub']))
g = np.load(ROOT + '/tests/golden/g3_overlap.npz')
ffs = sliding_ffdtf(g['x'], 1000, 3, 8, g['freqs'], float(g['fs']))
for i in range(3): print('win', i, np.abs(ffs[i][:,:,::32]/g[f'ffdtf_sub{i}']-1).max())
# random complex inverse stress (pivoting): random AR
rng = np.random.default_rng(5)
for m in (5, 16, 19, 33, 48, 64):
    arr = rng.standard_normal((m,m,3))
    fr = np.linspace(1, 60, 7)
    H, A = M.mvar_transfer_function(arr, fr, 128.0)
    Ho, Ao = O.mvar_transfer_function(arr, fr, 128.0)
    print('rand m', m, 'H', rel(H, Ho), 'A', rel(A, Ao))
