"""TEST INFRASTRUCTURE: compile the kernel SOURCES of eeg_gnn_ssl_amd/csrc against the fiber-based
SIMT emulator (simt_emu.h) into tests/_emu_build/libeeg_dcrnn_emu.so so that the kernel logic,
the C ABI orchestration and the Python host layer can be exercised on a machine without a GPU.
The product never loads this library; only tests do (see tests/emu_support.py)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "eeg_gnn_ssl_amd", "csrc")
# EEG_EMU_XFLAGS: extra -D switches (the experiment knobs of development A/B builds, csrc/Makefile XFLAGS) -> their own build directory
XFLAGS = os.environ.get("EEG_EMU_XFLAGS", "").split()
OUT_DIR = os.path.join(ROOT, "tests", "_emu_build" + ("_" + "".join(c if c.isalnum() else "_" for c in "".join(XFLAGS)) if XFLAGS else ""))
OUT = os.path.join(OUT_DIR, "libeeg_dcrnn_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def sources():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cpp"))]
    deps += [os.path.join(HERE, "simt_emu.h"), os.path.join(HERE, "platform_emu.h"), os.path.join(HERE, "emu_impl.cpp"),
             os.path.join(ROOT, "include", "eeg_dcrnn.h"), os.path.join(ROOT, "include", "eeg_dcrnn_dev.h"),
             os.path.join(ROOT, "include", "eeg_dcrnn_prof.h"), os.path.abspath(__file__)]
    return deps


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(sources()):
        return OUT
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    common = [cxx, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-DEEG_PLATFORM_HEADER=\"platform_emu.h\"", "-DEEG_DEV", "-I", HERE, "-I", CSRC,
              "-Wno-unused-function", "-Wno-unknown-attributes"] + XFLAGS
    objs = []
    jobs = []
    units = [("api", os.path.join(CSRC, "api.cpp"), []),
             ("emu_impl", os.path.join(HERE, "emu_impl.cpp"), [])]
    units.append(("gemmq", os.path.join(CSRC, "gemmq_inst.cpp"), []))
    units.append(("spec", os.path.join(CSRC, "spec_inst.cpp"), []))
    units.append(("dec", os.path.join(CSRC, "dec_inst.cpp"), []))
    units.append(("decb", os.path.join(CSRC, "decb_inst.cpp"), []))
    units.append(("seqs", os.path.join(CSRC, "seqs_inst.cpp"), []))
    for h in (16, 32, 64):
        units.append((f"seq_h{h}", os.path.join(CSRC, "seq_inst.cpp"), [f"-DEEG_SEQ_H={h}"]))
    for name, src, extra in units:
        obj = os.path.join(OUT_DIR, name + ".o")
        objs.append(obj)
        jobs.append(subprocess.Popen(common + extra + ["-c", src, "-o", obj]))
    for j in jobs:
        if j.wait() != 0:
            raise RuntimeError("emulator build failed")
    subprocess.check_call([cxx, "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
