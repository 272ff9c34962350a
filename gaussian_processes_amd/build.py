"""Build the HIP shared library for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m gaussian_processes_amd.build [--force]

Output: gaussian_processes_amd/lib/libgpfit_mi355x.so (git-ignored; travels to the GPU box
with the gpurun snapshot)."""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBNAME = "libgpfit_mi355x.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-Wno-unused-result", "-I" + os.path.join(os.path.dirname(PKG), "include"), "-I" + CSRC]
# development builds only (e.g. GPFIT_EXTRA_FLAGS=-DGPFIT_DEV: timing experiments that give wrong results by design);
# part of the digest, so the next ordinary build replaces such a library
FLAGS += os.environ.get("GPFIT_EXTRA_FLAGS", "").split()


def _digest():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(os.path.dirname(PKG), "include", "*.h"))):
        h.update(f.encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    out = lib_path()
    stamp = os.path.join(LIBDIR, ".digest")
    dig = _digest()
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp, "w").write(dig)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
