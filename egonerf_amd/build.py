"""Builds libegonerf_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists: the product path
raises if this library is missing."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libegonerf_hip.so")
SOURCES = ["ego_ops.hip", "ego_shade.hip", "ego_render.hip", "ego_reg.hip", "ego_metrics.hip", "ego_wgrad.hip"]
HEADERS = ["ego_device.h", "ego_host.h", "ego_train.inc", os.path.join("..", "..", "include", "egonerf_hip.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libegonerf_hip.so)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-shared", "-fPIC",
           *[os.path.join(CSRC, f) for f in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
