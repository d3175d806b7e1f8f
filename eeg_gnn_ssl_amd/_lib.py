"""ctypes binding of libeeg_dcrnn_hip.so (C ABI: include/eeg_dcrnn.h).

This is the stub a maintainer of the reference would add to call the MI355X kernels from
Python.  The product loads ONLY the in-tree HIP library and fails loudly if it
is missing: there is no CPU fallback.  (`EegDcrnnLib(path)` can be constructed on any build of the
same ABI; the module-level `_LIB` slot is what `get_lib()` hands out — tests/ point it at the emulator
build of the same sources to check kernel logic without a GPU, the product never does.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libeeg_dcrnn_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "libeeg_dcrnn_hip_dev.so")     # `make dev`: tools/ and `bench.py --tune` only
ABI_VERSION = 5


class LayerDims(ctypes.Structure):
    """mirror of `eeg_layer_dims` (include/eeg_dcrnn.h)."""
    _fields_ = [("T", c_int32), ("B", c_int32), ("N", c_int32), ("H", c_int32), ("Fin", c_int32),
                ("M", c_int32), ("act", c_int32), ("p_batched", c_int32), ("x_planes_ready", c_int32),
                ("x_batch_major", c_int32), ("x_plane_stride", c_int64), ("pack3", c_void_p), ("spectral", c_void_p),
                ("spack", c_void_p)]


class DecoderDims(ctypes.Structure):
    """mirror of `eeg_decoder_dims` (include/eeg_dcrnn.h)."""
    _fields_ = [("T", c_int32), ("B", c_int32), ("N", c_int32), ("H", c_int32), ("Dout", c_int32),
                ("M", c_int32), ("L", c_int32), ("act", c_int32), ("p_batched", c_int32), ("dropout_p", c_float),
                ("teacher_on_device", c_int32)]


_FP = c_void_p  # device pointers travel as integers (tensor.data_ptr())

_SIGNATURES = {
    "eeg_dcrnn_last_error": (c_char_p, []),
    "eeg_dcrnn_abi_version": (c_int, []),
    "eeg_dcrnn_is_device_build": (c_int, []),
    "eeg_dcrnn_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "eeg_dcrnn_zero": (c_int, [_FP, c_size_t, c_void_p]),
    "eeg_dcrnn_prof_enable": (c_int, [c_int]),
    "eeg_dcrnn_prof_report": (c_int, [ctypes.c_char_p, c_size_t]),
    "eeg_dcrnn_prof_clock_probe": (c_int, [_FP, c_void_p]),
    "eeg_dcrnn_prof_clock_samples": (c_int, [_FP]),
    "eeg_dcrnn_hop_polys": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_fft_features": (c_int, [_FP, c_int, c_int, c_int, c_int, _FP, _FP, c_float, c_float, _FP, _FP, c_void_p]),
    "eeg_dcrnn_corr_graph_ws_floats": (c_size_t, [c_int, c_int]),
    "eeg_dcrnn_corr_graph": (c_int, [_FP, c_int, c_int, c_int, c_int, c_int, _FP, _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_pack_floats": (c_size_t, [c_int, c_int, c_int]),
    "eeg_dcrnn_pack_cell": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_pack3_halves": (c_size_t, [c_int, c_int, c_int]),
    "eeg_dcrnn_pack_cell_bf16x3": (c_int, [_FP, _FP, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_diffuse_fwd": (c_int, [_FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_diffuse_adj": (c_int, [_FP, _FP, c_int, c_int, c_int, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_dconv_fwd_ws_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "eeg_dcrnn_dconv_fwd": (c_int, [_FP, _FP, c_int, c_int, c_int, c_int, c_int, _FP, _FP, c_int, _FP, _FP, c_void_p]),
    "eeg_dcrnn_dconv_bwd_ws_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "eeg_dcrnn_dconv_bwd": (c_int, [_FP, _FP, c_int, c_int, c_int, c_int, c_int, _FP, c_int, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_spectral_basis_floats": (c_size_t, [c_int]),
    "eeg_dcrnn_spectral_basis": (c_int, [_FP, c_int, _FP, c_void_p]),
    "eeg_dcrnn_spectral_pack_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "eeg_dcrnn_pack_cell_spectral": (c_int, [_FP, _FP, _FP, c_int, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_spectral_ok": (c_int, [POINTER(LayerDims), c_int]),
    "eeg_dcrnn_spectral_rows": (c_size_t, [c_size_t]),
    "eeg_dcrnn_layer_fwd_ws_floats": (c_size_t, [POINTER(LayerDims)]),
    "eeg_dcrnn_batch_major_ok": (c_int, [POINTER(LayerDims)]),
    "eeg_dcrnn_layer_fwd": (c_int, [POINTER(LayerDims)] + [_FP] * 14 + [c_void_p]),
    "eeg_dcrnn_layer_bwd_ws_floats": (c_size_t, [POINTER(LayerDims), c_int]),
    "eeg_dcrnn_layer_bwd": (c_int, [POINTER(LayerDims)] + [_FP] * 22 + [c_void_p]),
    "eeg_dcrnn_decoder_saved_floats": (c_size_t, [POINTER(DecoderDims)]),
    "eeg_dcrnn_decoder_fwd_ws_floats": (c_size_t, [POINTER(DecoderDims)]),
    "eeg_dcrnn_decoder_bwd_ws_floats": (c_size_t, [POINTER(DecoderDims)]),
    "eeg_dcrnn_decoder_is_persistent": (c_int, [POINTER(DecoderDims)]),
    # (`teacher` travels as a void*: a host int32[T] array or, with dims.teacher_on_device = 1, a device pointer)
    "eeg_dcrnn_decoder_fwd": (c_int, [POINTER(DecoderDims), _FP, c_void_p, _FP, _FP, POINTER(c_void_p), _FP, _FP,
                                      _FP, _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_decoder_bwd": (c_int, [POINTER(DecoderDims), c_void_p, _FP, POINTER(c_void_p), _FP, _FP, _FP, _FP, _FP,
                                      POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                      _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_teacher_flags": (c_int, [_FP, _FP, c_int64, ctypes.c_double, c_int, _FP, c_void_p]),
    "eeg_dcrnn_augment_draw": (c_int, [_FP, c_int, c_int, _FP, _FP, _FP, _FP, _FP, _FP, c_int, _FP, c_void_p]),
    "eeg_dcrnn_pack_cells": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(ctypes.c_int32),
                             c_int, c_int, POINTER(c_void_p), _FP, c_int, POINTER(c_void_p), c_void_p]),
    "eeg_dcrnn_cls_head_loss_ws_floats": (c_size_t, [c_int, c_int, c_int]),
    "eeg_dcrnn_cls_head_loss": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_int, c_float, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                                c_void_p]),
    "eeg_dcrnn_gather_last": (c_int, [_FP, _FP, c_int, c_int, c_int, _FP, c_void_p]),
    "eeg_dcrnn_rng_take": (c_int, [_FP, ctypes.c_uint64, _FP, c_void_p]),
    "eeg_dcrnn_cls_head_fwd": (c_int, [_FP, _FP, _FP, c_int, c_int, c_int, c_int, c_float, _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_cls_head_bwd": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_float, _FP, _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_dropout_mask": (c_int, [_FP, c_size_t, c_float, _FP, c_void_p]),
    "eeg_dcrnn_bce_logits": (c_int, [_FP, _FP, c_int, _FP, _FP, c_void_p]),
    "eeg_dcrnn_ce_logits": (c_int, [_FP, _FP, c_int, c_int, _FP, _FP, c_void_p]),
    "eeg_dcrnn_masked_loss_ws_floats": (c_size_t, []),
    "eeg_dcrnn_masked_loss": (c_int, [_FP, _FP, c_size_t, c_int, c_float, c_float, c_float, c_int, _FP, _FP, _FP, c_void_p]),
    "eeg_dcrnn_clip_adam_ws_floats": (c_size_t, []),
    "eeg_dcrnn_clip_adam": (c_int, [_FP, _FP, _FP, _FP, c_size_t, c_float, c_float, c_float, c_float, c_float, c_float,
                                    c_int, c_float, _FP, _FP, c_void_p]),
    "eeg_dcrnn_clip_adam_dev": (c_int, [_FP, _FP, _FP, _FP, c_size_t, c_float, _FP, c_float, c_float, c_float, c_float,
                                        _FP, c_float, _FP, _FP, c_void_p]),
}


# include/eeg_dcrnn_dev.h: exported by the dev build (libeeg_dcrnn_hip_dev.so) and the test emulator only
_SIGNATURES_DEV = {
    "eeg_dcrnn_set_seq_probe": (c_int, [_FP]),
    "eeg_dcrnn_set_tuning": (c_int, [c_int, c_int]),
}


class EegDcrnnError(RuntimeError):
    """Raised when a library call reports an error (shape/dtype/launch problems)."""


class EegDcrnnLib:
    def __init__(self, path: str = HIP_LIB_PATH, strict: bool = True):
        """strict=False (development A/B runs against a library built from an older commit, `bench.py --lib`): entry points
        the file does not export are skipped instead of refused."""
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the MI355X HIP library is not built. Run `python -c 'import "
                f"__graft_entry__ as g; g.build()'` (or `make -C eeg_gnn_ssl_amd/csrc`). There is no CPU fallback.")
        self.path = path
        self._dll = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                if not strict:
                    continue
                raise ImportError(f"{path} does not export {name} (declared in include/eeg_dcrnn.h)") from e
            fn.restype = res
            fn.argtypes = args
        self.is_dev_build = hasattr(self._dll, "eeg_dcrnn_set_tuning")
        if self.is_dev_build:
            for name, (res, args) in _SIGNATURES_DEV.items():
                fn = getattr(self._dll, name)
                fn.restype, fn.argtypes = res, args
        # strict=False tolerates MISSING entry points only: a library of another ABI version has entry points with other
        # signatures (shifted arguments = wild pointers), so it is refused in both modes
        if self._dll.eeg_dcrnn_abi_version() != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {self._dll.eeg_dcrnn_abi_version()} != {ABI_VERSION}; rebuild")
        self.is_device_build = bool(self._dll.eeg_dcrnn_is_device_build())

    def last_error(self) -> str:
        return self._dll.eeg_dcrnn_last_error().decode()

    def call(self, name: str, *args):
        """Invoke an int-returning entry point, raising EegDcrnnError on a non-zero status."""
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise EegDcrnnError(f"{name}: {self.last_error()}")

    def query(self, name: str, *args):
        return getattr(self._dll, name)(*args)


_LIB = None


def get_lib() -> EegDcrnnLib:
    """The product library (HIP, gfx950).  Loaded on first use; ImportError if absent."""
    global _LIB
    if _LIB is None:
        lib = EegDcrnnLib(HIP_LIB_PATH)
        if not lib.is_device_build:
            raise ImportError(f"{HIP_LIB_PATH} is not a device build")
        _LIB = lib
    return _LIB
