"""TEST INFRASTRUCTURE: load the emulator build of the kernel sources (tests/emu) and install it
as the library object used by eeg_gnn_ssl_amd.ops, so the C ABI + Python host layer can be
checked against the oracle on a machine without a GPU.  Never used by the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def install_emulator():
    import build_emu
    from eeg_gnn_ssl_amd import _lib
    path = build_emu.build()
    lib = _lib.EegDcrnnLib(path)
    assert not lib.is_device_build
    _lib._LIB = lib
    return lib


def uninstall():
    from eeg_gnn_ssl_amd import _lib
    _lib._LIB = None
