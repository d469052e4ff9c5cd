import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, oat_amd
from oat_amd.synth import disc_hsv_window
rows, cols = 1080, 1920
dev = torch.device('cuda:0')
pool = bench.make_pool_device(rows, cols, 1, 48, 0, dev)
hp = oat_amd.HotPath(rows, cols, n_streams=1, adaptation_coeff=0.01, erode=3, dilate=7, area=(20.0, 1e5), ring_depth=8, **disc_hsv_window())
hp.track_dev(pool[0].data_ptr())
for i in range(400):
    hp.track_dev(pool[(i + 1) % 48].data_ptr())
nm = hp.mog_state()[0].reshape(rows, cols)
h = np.bincount(nm.ravel(), minlength=6)
print('nmodes histogram', h, h / h.sum())
w = nm.reshape(-1, 64)           # per wave (64 consecutive px; cols is a multiple of 64)
mx = w.max(1)
print('per-wave max nmodes histogram', np.bincount(mx, minlength=6) / len(mx))
need = (np.arange(1, 5)[None, :] < mx[:, None]).sum()           # planes groups loaded at wave granularity
legit = (np.arange(1, 5)[None, None, :] < w[:, :, None]).sum()
print('extra-mode loads: wave-granular lanes', need * 64, 'legit lanes', legit, 'ratio', need * 64 / max(legit, 1))
# 16-lane sector granularity
s16 = nm.reshape(-1, 16).max(1)
need16 = (np.arange(1, 5)[None, :] < s16[:, None]).sum() * 16
print('sector(16 lanes)-granular lanes', need16, 'ratio', need16 / max(legit, 1), 'extra B/px sector-granular', need16 * 20 / nm.size, 'legit B/px', legit * 20 / nm.size)
