"""Loader of the C++ operator library `torch.ops.eeg_dcrnn_cpp.*` (csrc/torch_ops.cpp: TORCH_LIBRARY over the C ABI,
SURVEY.md 8(b)).  Optional: the package itself uses `torch.ops.eeg_dcrnn.*` (ops.py), which carries the autograd
formulas; this library is for C++ callers (link libeeg_dcrnn_torch.so) and for eager launches without Python / ctypes
in the call path.  Registered for the CUDA (= HIP) dispatch key only: CPU tensors are refused by the dispatcher."""
import os

import torch

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeeg_dcrnn_torch.so")
_loaded = False


def load(path: str = TORCH_LIB_PATH):
    """torch.ops.load_library(libeeg_dcrnn_torch.so) once per process; returns torch.ops.eeg_dcrnn_cpp."""
    global _loaded
    if not _loaded:
        if not os.path.exists(path):
            raise ImportError(f"{path} not built: run `make -C eeg_gnn_ssl_amd/csrc torch` "
                              "(or __graft_entry__.build())")
        torch.ops.load_library(path)
        _loaded = True
    return torch.ops.eeg_dcrnn_cpp
