"""Builds libegonerf_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists: the product path
raises if this library is missing."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libegonerf_hip.so")
SOURCES = ["ego_ops.hip", "ego_shade.hip", "ego_render.hip", "ego_reg.hip", "ego_metrics.hip", "ego_wgrad.hip", "ego_generic.hip"]
HEADERS = ["ego_device.h", "ego_host.h", "ego_train.inc", "variants.h", "ego_generic.h", os.path.join("..", "..", "include", "egonerf_hip.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libegonerf_hip.so)")


TESTED_HIPCC = ("7.2.",)  # HIP versions whose code generation passed the determinism soak of DESIGN.md 5.1


def hipcc_version() -> str:
    out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("HIP version:"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def source_hash() -> str:
    """sha256 over the library's sources, headers and compile flags: what `libegonerf_hip.so.hash` records for the binary next
    to it, and what profiles/r*/pmc_traffic.json records for the build its counters were taken from."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(repr((COMMON_FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def is_stale() -> bool:
    """True unless the binary's recorded source hash equals the tree's (mtimes do not survive a snapshot copy)."""
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".hash"):
        return True
    return open(LIB + ".hash").read().strip() != source_hash()


# Per-source extra flags.  ego_shade.hip is built without the SLP vectoriser: it forms {w00, w01}-style pairs of the interpolation
# weights, and the packed fp32 instructions that then broadcast the high half of such a pair are what every non-reproducible build
# of these kernels had in common.  The source also pins the weights to separate registers; either measure alone
# was enough in every soak, neither costs time.
EXTRA_FLAGS = {"ego_shade.hip": ["-fno-slp-vectorize"]}
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC"]


def build_library(force: bool = False, verbose: bool = False, extra: list | None = None, out: str | None = None) -> str:
    """`extra` (or the environment variable EGO_EXTRA_FLAGS) adds compiler flags to every source and `out` names another output
    file: experiment builds (tools/variant_test.sh)."""
    extra = list(extra or []) + os.environ.get("EGO_EXTRA_FLAGS", "").split()
    out = out or os.environ.get("EGO_LIB_OUT") or LIB
    if not force and not extra and out == LIB and not is_stale():
        return LIB
    ver = hipcc_version()
    if not ver.startswith(TESTED_HIPCC) and not os.environ.get("EGO_ALLOW_UNTESTED_HIPCC"):
        # a code-generation-dependent fault was found and fenced off with this compiler; another one must
        # pass tests/test_hip_determinism.py (3000 repetitions per op) before its binaries are trusted
        raise RuntimeError(f"hipcc {ver} is not a tested compiler ({TESTED_HIPCC}); set EGO_ALLOW_UNTESTED_HIPCC=1 to build anyway "
                           "and run tests/test_hip_determinism.py on the result")
    objdir = os.path.join(HERE, "build" + ("_" + os.path.basename(out) if out != LIB else ""))
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for f in SOURCES:  # one hipcc per source, in parallel: the sources have no device-side references to one another
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        objs.append(obj)
        per_file = [] if os.environ.get("EGO_NO_PER_FILE_FLAGS") else EXTRA_FLAGS.get(f, [])  # experiments only
        cmd = [_hipcc(), *COMMON_FLAGS, *per_file, *extra, "-c", os.path.join(CSRC, f), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for f, pr in procs:
        log, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {f}:\n" + log)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + r.stdout + r.stderr)
    os.replace(out + ".tmp", out)
    if out == LIB and not extra:
        open(LIB + ".hash", "w").write(source_hash() + "\n")
    if out != LIB:
        shutil.rmtree(objdir, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
