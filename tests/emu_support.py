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
    _register_cpu_key()
    return lib


_cpu_registered = False


def _register_cpu_key():
    """The product registers its operators for the CUDA (= HIP) key only.  The emulator runs on host memory, so the
    tests install the SAME implementations for the CPU dispatch key here (once per process)."""
    global _cpu_registered
    if _cpu_registered:
        return
    from eeg_gnn_ssl_amd import ops
    for name, impl in ops._impls.items():
        ops._libdef.impl(name, impl, "CPU")
    _cpu_registered = True


def uninstall():
    from eeg_gnn_ssl_amd import _lib
    _lib._LIB = None
