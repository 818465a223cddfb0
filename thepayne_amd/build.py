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
SOURCES = ["payne_hip.hip", "k_dense.hip", "k_post_lean.hip", "k_post_full_a.hip", "k_post_full_b.hip", "k_post_big.hip", "k_smooth.hip"]
HEADERS = ["post_core.hpp", "post_seq.hpp", "host_tables.hpp", "ns_core.hpp", "dense_kernels.hpp", "post_kernels.hpp",
           "sed_kernel.hpp", "sampler_kernels.hpp", "sampler_core.hpp", "select.hpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-DNDEBUG", "-Wall", "-Wno-unused-function"]
OBJDIR = os.path.join(HERE, "build")


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


def source_hash():
    """sha256[:12] over the library's sources (csrc/*.hip, *.hpp, include/payne_hip.h): identifies the build a
    profile under profiles/ was taken with (bench.py only quotes PMC counters whose hash matches)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(ROOT, "include", "payne_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:12]


def _compile_link(out, extra=(), tag="", verbose=False, force=False):
    """Each unit of SOURCES -> build/<unit><tag>.o (in parallel, only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "payne_hip.h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", tag + ".o"))
        sp = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(sp)):
            jobs.append([hipcc] + HIPCC_FLAGS + list(extra) + ["-I", os.path.join(ROOT, "include"), "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        return cmd, subprocess.run(cmd, capture_output=True, text=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for cmd, res in ex.map(run, jobs):
            if res.returncode != 0:
                raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), res.stdout, res.stderr))
    objs = [os.path.join(OBJDIR, src.replace(".hip", tag + ".o")) for src in SOURCES]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out] + objs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
    return out


def build_diag(verbose=False):
    """Diagnostic twin with per-phase cycle stamps in the post kernel (-DPAYNE_STAMPS);
    used by tools/post_stamps.py only, never by the product path."""
    return _compile_link(DIAG_LIB, extra=["-DPAYNE_STAMPS"], tag="_diag", verbose=verbose)


def build_variant(tag, flags, verbose=False):
    """An experimental twin thepayne_amd/libpayne_hip_<tag>.so built with extra compiler flags (tools/ only: timing
    experiments such as -DPAYNE_EXP_SKIP=<phase mask>; never loaded by the product path)."""
    return _compile_link(os.path.join(HERE, "libpayne_hip_%s.so" % tag), extra=list(flags), tag="_" + tag, verbose=verbose)


def build_lib(force=False, verbose=False):
    """Compile csrc/*.hip -> thepayne_amd/libpayne_hip.so; returns the path."""
    if not force and not _stale():
        return LIB
    return _compile_link(LIB, verbose=verbose, force=force)


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
