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
SOURCES = ["payne_hip.hip", "k_dense.hip", "k_post_lean.hip", "k_post_full_a.hip", "k_post_full_b.hip", "k_post_big.hip", "k_post_chip2.hip", "k_smooth.hip"]
HEADERS = ["post_core.hpp", "post_seq.hpp", "host_tables.hpp", "ns_core.hpp", "dense_kernels.hpp", "post_kernels.hpp",
           "sed_kernel.hpp", "sed_core.hpp", "post_onchip.hpp", "post_onchip2.hpp", "sampler_kernels.hpp", "sampler_core.hpp", "select.hpp"]
# (-amdgpu-kernarg-preload-count: a kernel's leading scalar arguments arrive in registers at wave start instead of by s_load)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-DNDEBUG", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
OBJDIR = os.path.join(HERE, "build")


def variant_dir():
    """Where diagnostic / ablation twins of the library (objects and .so) are built: a scratch directory outside the package
    unless PAYNE_VARIANT_DIR says otherwise, so that they neither ship with the repository snapshot nor sit beside the product."""
    import tempfile
    d = os.environ.get("PAYNE_VARIANT_DIR") or os.path.join(tempfile.gettempdir(), "payne_variants_%d" % os.getuid())
    os.makedirs(d, exist_ok=True)
    return d


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


def diag_lib_path():
    return os.path.join(variant_dir(), "libpayne_hip_diag.so")


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


def _compile_link(out, extra=(), tag="", verbose=False, force=False, objdir=None):
    """Each unit of SOURCES -> <objdir>/<unit><tag>.o (in parallel, only the stale ones), then one link.  Safe against
    concurrent callers (N ranks finding a stale library at once): one holder of the lock file builds, objects and the
    library are written under temporary names and renamed into place."""
    import fcntl
    objdir = objdir or OBJDIR
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock" + tag), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return _compile_link_locked(out, extra, tag, verbose, force, objdir)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _compile_link_locked(out, extra, tag, verbose, force, objdir):
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "payne_hip.h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", tag + ".o"))
        sp = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(sp)):
            jobs.append([hipcc] + HIPCC_FLAGS + list(extra) + ["-I", os.path.join(ROOT, "include"), "-c", sp, "-o", obj + ".tmp.o"])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        return cmd, subprocess.run(cmd, capture_output=True, text=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for cmd, res in ex.map(run, jobs):
            if res.returncode != 0:
                raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), res.stdout, res.stderr))
            os.replace(cmd[-1], cmd[-1][:-len(".tmp.o")])
    objs = [os.path.join(objdir, src.replace(".hip", tag + ".o")) for src in SOURCES]
    if not jobs and os.path.exists(out) and all(os.path.getmtime(o) <= os.path.getmtime(out) for o in objs):
        return out                                   # another holder of the lock built it meanwhile
    tmp_out = out + ".tmp%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", tmp_out] + objs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
    os.replace(tmp_out, out)
    return out


def build_diag(verbose=False):
    """Diagnostic twin with per-phase cycle stamps in the post kernel (-DPAYNE_STAMPS);
    used by tools/post_stamps.py only, never by the product path."""
    return _compile_link(diag_lib_path(), extra=["-DPAYNE_STAMPS"], tag="_diag", verbose=verbose, objdir=variant_dir())


def build_variant(tag, flags, verbose=False):
    """An experimental twin <variant_dir()>/libpayne_hip_<tag>.so built with extra compiler flags (tools/ only: timing
    experiments, stamped twins such as ['-DPAYNE_STAMPS', '-DPAYNE_STAMPS_ENDS_ONLY']; never loaded by the product path, never built
    inside the package)."""
    d = variant_dir()
    return _compile_link(os.path.join(d, "libpayne_hip_%s.so" % tag), extra=list(flags), tag="_" + tag, verbose=verbose, objdir=d)


def build_lib(force=False, verbose=False):
    """Compile csrc/*.hip -> thepayne_amd/libpayne_hip.so; returns the path."""
    if not force and not _stale():
        return LIB
    return _compile_link(LIB, verbose=verbose, force=force)


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
