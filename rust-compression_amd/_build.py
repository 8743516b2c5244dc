"""Builds the HIP shared library (gfx950 only) in-tree: rust-compression_amd/libbz2_mi355x.so."""
import contextlib
import fcntl
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libbz2_mi355x.so")
SOURCES = ["k_rle1.hip", "k_bwt.hip", "k_mtf.hip", "k_huff.hip", "k_emit.hip", "k_dec.hip", "k_deflate.hip", "engine.hip", "dec_engine.hip", "deflate_engine.hip", "capi.hip"]
HEADERS = ["bzgpu.h", "engine_state.h", "copy_pool.h", "bz2_rnums.h", "k_deflate.h", os.path.join("..", "..", "include", "bz2_mi355x.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function",
         "-D__HIP_PLATFORM_AMD__"] + os.environ.get("BZ_EXTRA_FLAGS", "").split()


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per checkout (ranks started together all find a stale library): an fcntl lock on a file
    beside the sources; the others wait, find the library fresh and do nothing."""
    with open(os.path.join(HERE, ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _link(cmd, target):
    """the linker writes a temporary name in the same directory; os.replace puts it in place whole (nobody ever
    dlopens a half-written library)"""
    tmp = "%s.tmp.%d" % (target, os.getpid())
    try:
        subprocess.check_call(cmd + ["-o", tmp])
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    with _build_lock():
        return _build_locked(force, verbose)


def _build_locked(force, verbose):
    deps_h = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + deps_h):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (s, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("HIP build failed")
    if force or procs or _stale(SO, objs):
        _link([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs, SO)
    # (libbz2_mi355x_rccl.so is built on demand by rccl_lib() / build_rccl(): the codec library must load on hosts
    # without the RCCL headers or library)
    return SO


RCCL_SO = os.path.join(HERE, "libbz2_mi355x_rccl.so")


def build_rccl(force=False, verbose=False):
    """The RCCL transport (csrc/rccl_comm.hip) as its own library: only it links librccl."""
    src = os.path.join(CSRC, "rccl_comm.hip")
    hdr = os.path.join(CSRC, "..", "..", "include", "bz2_mi355x.h")
    with _build_lock():
        if not (force or _stale(RCCL_SO, [src, hdr])):
            return RCCL_SO
        cmd = [HIPCC] + FLAGS + ["-shared", src, "-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        _link(cmd, RCCL_SO)
    return RCCL_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
