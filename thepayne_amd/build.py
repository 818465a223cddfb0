"""Build recipe for libpayne_hip.so (hipcc, gfx950 only, in-tree).

    python -m thepayne_amd.build [--force]

The library is a plain C-ABI shared object (include/payne_hip.h); it is built
in-tree so that it travels with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpayne_hip.so")
SOURCES = ["payne_hip.hip"]
HEADERS = ["post_core.hpp", "post_seq.hpp", "host_tables.hpp", "ns_core.hpp", "dense_kernels.hpp", "post_kernels.hpp",
           "sed_kernel.hpp", "sampler_kernels.hpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc",
               "-DNDEBUG", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "payne_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


DIAG_LIB = os.path.join(HERE, "libpayne_hip_diag.so")


def build_diag(verbose=False):
    """Diagnostic twin with per-phase cycle stamps in the post kernel (-DPAYNE_STAMPS);
    used by tools/post_stamps.py only, never by the product path."""
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-DPAYNE_STAMPS", "-I", os.path.join(ROOT, "include"), "-o", DIAG_LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (res.stdout, res.stderr))
    return DIAG_LIB


def build_lib(force=False, verbose=False):
    """Compile csrc/*.hip -> thepayne_amd/libpayne_hip.so; returns the path."""
    if not force and not _stale():
        return LIB
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (res.stdout, res.stderr))
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
