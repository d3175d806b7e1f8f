"""Closed-form (RNG-free) fills shared by the golden-vector generator and the tests, so the
committed fixtures only need to hold OUTPUTS; every input/weight is rebuilt from a formula."""
import math

import numpy as np


def cf(shape, scale=0.1, freq=0.37, phase=0.0, dtype=np.float32):
    """scale * sin(freq * i + phase) over the flattened index i, reshaped."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return (scale * np.sin(freq * i + phase)).reshape(shape).astype(dtype)


def cf_adjacency(n, phase=0.0):
    """A dense non-symmetric non-negative 'adjacency' with unit diagonal."""
    a = np.abs(cf((n, n), scale=1.0, freq=0.61, phase=phase, dtype=np.float64))
    np.fill_diagonal(a, 1.0)
    return a.astype(np.float32)


def cf_params(shapes, base_phase=0.0):
    """Deterministic parameter dict for a {name: shape} map.  Weight scale follows the
    reference's xavier-normal law (std = 1.414*sqrt(2/(fan_in+fan_out))) so activations stay
    in the same regime as a real model; biases get small non-zero values so they matter."""
    out = {}
    for k, (name, shape) in enumerate(sorted(shapes.items())):
        ph = base_phase + 0.71 * k
        if len(shape) == 2 and name.endswith(".weight") and "dconv" in name:
            std = 1.414 * math.sqrt(2.0 / (shape[0] + shape[1]))
            out[name] = cf(shape, scale=std * math.sqrt(2.0), freq=0.913, phase=ph)
        elif len(shape) == 2:
            out[name] = cf(shape, scale=0.15, freq=0.913, phase=ph)
        else:
            out[name] = cf(shape, scale=0.05, freq=1.31, phase=ph)
    return out


def sample_view(a, step=97):
    """Strided sample + summary stats: a compact fingerprint of a large gradient tensor."""
    flat = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.concatenate([[flat.sum(), np.abs(flat).sum(), np.square(flat).sum()], flat[::step]])


def fft_raw_signal(n_ch=19, n_samples=800):
    """Closed-form 'raw EEG' (float64, microvolt-like scale) for the featurisation goldens: a few
    sinusoids, a chirp and hash noise so that every FFT bin carries energy; channel 5 is silent in the
    second window (exact zeros -> the amp == 0 -> 1e-8 rule of computeFFT)."""
    i = np.arange(n_ch * n_samples, dtype=np.float64).reshape(n_ch, n_samples)
    t = np.arange(n_samples, dtype=np.float64)[None, :]
    ch = np.arange(n_ch, dtype=np.float64)[:, None]
    x = 30.0 * np.sin(0.211 * t + 0.3 * ch) + 12.0 * np.sin(1.37 * t + 0.11 * ch * ch) + 5.0 * np.sin(0.0007 * t * t + ch)
    noise = np.sin(12.9898 * i + 78.233) * 43758.5453
    x = x + 8.0 * (noise - np.floor(noise) - 0.5)
    x[5, 200:400] = 0.0
    return x


def train_task(b=32, t=12, n=19, d=100):
    """Closed-form synthetic detection task for the training-trajectory golden: noise-like clips (hash
    noise, unit variance) whose first ten features carry a per-clip offset; label = 1[offset > 0]
    (the synthetic-label rule of SURVEY.md §8d, made learnable).  Returns x (b,t,n,d) f32, y (b,) f32."""
    i = np.arange(b * t * n * d, dtype=np.float64)
    h = np.sin(12.9898 * i + 78.233) * 43758.5453
    x = ((h - np.floor(h)) - 0.5) * math.sqrt(12.0)
    x = x.reshape(b, t, n, d)
    off = 0.1 * np.sin(2.1 * np.arange(b, dtype=np.float64) + 0.4)
    x[:, :, :, :10] += off[:, None, None, None]
    return x.astype(np.float32), (off > 0).astype(np.float32)


def cf_dropout_mask(shape, p=0.5, phase=0.0):
    """Closed-form stand-in for one nn.Dropout(p) draw in training mode: keep-mask x 1/(1-p) (hash noise thresholded at p)."""
    i = np.arange(int(np.prod(shape)), dtype=np.float64)
    h = np.sin(12.9898 * i + 78.233 + phase) * 43758.5453
    keep = (h - np.floor(h)) >= p
    return (keep.astype(np.float32) / np.float32(1.0 - p)).reshape(shape)
