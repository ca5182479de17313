"""Compile wavenet_amd/csrc/*.hip into wavenet_amd/libwavenet_hip.so for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so then travels
in-tree to the GPU box.  Usage: ``python -m wavenet_amd.build [--force]``.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libwavenet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fvisibility=hidden: only what include/wavenet_hip.h declares (inside its `#pragma GCC visibility push(default)`) is exported
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"] + \
    os.environ.get("WAVENET_HIP_EXTRA_FLAGS", "").split()        # e.g. -DWN16_STAMPS for the in-kernel timestamps


def _newest_header() -> float:
    hs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, hdr_mtime: float) -> str:
    obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_mtime):
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdr = _newest_header()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        # the version script drops what hipcc itself adds to the dynamic table (one __hip_cuid_* marker per translation unit)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden",
               "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
