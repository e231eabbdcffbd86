"""Builds libdsmil_hip.so (all HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

The shared object is git-ignored but travels to the GPU box with the gpurun snapshot.
hipcc cross-compiles without a GPU, so this is also the driver's "does it build" check.
"""
import glob
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libdsmil_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("DSMIL_CFLAGS", "").split()
# DSMIL_CFLAGS: extra compile flags for instrumented builds (e.g. -DDSMIL_TRACE, tools_stamp.py)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=True):
    """Compile csrc/*.hip into libdsmil_hip.so if missing or older than its sources."""
    if not force and not _stale():
        return LIB
    objs = []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *[os.path.getmtime(h) for h in
                                         glob.glob(os.path.join(CSRC, "*.h")) +
                                         glob.glob(os.path.join(ROOT, "include", "*.h"))]):
            cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    print(LIB)
