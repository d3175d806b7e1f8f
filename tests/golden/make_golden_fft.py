#!/usr/bin/env python3
"""Generates tests/golden/golden_fft_v1.npz from the genuine reference (run in the build container
only; /root/reference never travels):  data/data_utils.py:13-35 `computeFFT` applied window by window
as data/dataloader_detection.py:57-71 does, then utils.py:393-428 `StandardScaler.transform`.
Only OUTPUTS are stored; the raw signal is the closed form `closed_form.fft_raw_signal()`."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from closed_form import fft_raw_signal  # noqa: E402

for _m in ("h5py", "pyedflib"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
from data.data_utils import computeFFT  # noqa: E402
import utils as ref_utils  # noqa: E402

W, MEAN, STD = 200, 3.924, 1.560
raw = fft_raw_signal()
steps = []
for t in range(raw.shape[1] // W):
    ft, _ = computeFFT(raw[:, t * W:(t + 1) * W], n=W)      # (channels, W/2) log amplitudes
    steps.append(ft)
clip = np.stack(steps, axis=0)                               # (T, channels, W/2), float64
scaler = ref_utils.StandardScaler(mean=MEAN, std=STD)
out = {"fft/logamp": clip.astype(np.float64), "fft/standardized": scaler.transform(clip).astype(np.float32),
       "fft/mean_std": np.array([MEAN, STD])}
np.savez_compressed(os.path.join(HERE, "golden_fft_v1.npz"), **out)
print({k: v.shape for k, v in out.items()}, float(clip.min()), float(clip.max()))
