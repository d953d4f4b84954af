"""Per-phase instruction budget of the SHIPPED shade kernel, from the code object inside libegonerf_hip.so ("a per-phase
ISA budget that proves where the floor is").

    python tools/isa_budget.py [--kernel 'k_shade_h<0, false, false, 2, true>'] > profiles/rNN/shade_isa_budget.txt

The tile loop of k_shade_h is straight-line code between `s_setprio 2` (gather + basis phase) ... `s_setprio 0` (MLP phase) ... the branch
back; instruction classes are counted per phase, priced with the issue costs measured in tools/coissue_probe.hip / fp6_probe.hip
(4 clk per wave64 VALU, 32 clk per v_mfma_f32_32x32x16_f16, 38 clk per v_mfma_scale_f32_32x32x64_f8f6f4) and compared with the PMC
counters of the same kernel (profiles/rNN/pmc_traffic.json)."""
import argparse
import collections
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from egonerf_amd import build as B  # noqa: E402


def disassemble(kernel_substr: str):
    with tempfile.TemporaryDirectory() as tmp:
        objdump, cos = B._code_objects(B.LIB, tmp)
        for co in cos:
            dis = subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout
            dem = subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout
            lines = dem.splitlines()
            for i, l in enumerate(lines):
                if re.match(r"^[0-9a-f]+ <.*>:$", l) and kernel_substr in l:
                    j = i + 1
                    while j < len(lines) and not re.match(r"^[0-9a-f]+ <.*>:$", lines[j]):
                        j += 1
                    return [x for x in lines[i + 1:j] if x.strip()]
    raise SystemExit(f"kernel {kernel_substr!r} not found in {B.LIB}")


def klass(m):
    if m.startswith("v_mfma_scale"): return "MFMA fp6/fp8 (block-scaled, 32x32x64)"
    if m.startswith("v_mfma"): return "MFMA fp16 (32x32x16)"
    if m.startswith("ds_"): return "LDS"
    if m.startswith(("global_", "buffer_", "scratch_")): return "VMEM"
    if m.startswith(("s_waitcnt", "s_nop")): return "wait / nop"
    if m.startswith("s_"): return "SALU"
    if m.startswith("v_"): return "VALU"
    return "other"


def valu_kind(l, m):
    if "dpp" in l or "quad_perm" in l or "row_" in l: return "cross-lane (DPP exchange of the team gather)"
    if re.match(r"v_pk_(fma|mul|add)_f32", m): return "packed fp32 arithmetic (interpolation, layer 3)"
    if re.match(r"v_(fma|mul|add|sub|fmac|mac)_f32", m): return "fp32 arithmetic (interpolation, weights, encodings)"
    if "fma_mix" in m: return "v_fma_mix (residual x - hi of the fp16 split)"
    if "cvt_scalef32" in m: return "fp6 conversions (32 values each)"
    if "cvt" in m: return "conversions (fp32 -> packed fp16, int <-> float)"
    if re.match(r"v_max_i32", m): return "ReLU (integer max on the float bits)"
    if re.match(r"v_(max|min|max3|min3|med3)", m): return "min / max (block-scale search, clamps)"
    if re.match(r"v_(mov|accvgpr)", m): return "register moves"
    if re.match(r"v_(cndmask|cmp)", m): return "select / compare (team halves, zero padding, grid choice)"
    if re.match(r"v_(add_u32|sub_u32|subrev|lshl|lshr|ashr|mad_u|mul_lo|mul_hi|and|or|xor|bfe|add_co|addc|mad_i|add3|lshl_add|add_lshl|bitop|mul_u32|mad_u64|lshl_or)", m):
        return "integer / address arithmetic (tap offsets)"
    if re.match(r"v_(sin|cos|exp|log|rcp|rsq|sqrt|floor|fract|trunc|rndne|ceil|div|ldexp|frexp)", m): return "transcendental / rounding (sincos of the encodings, floor of the taps)"
    if "readfirstlane" in m or "readlane" in m: return "lane -> scalar"
    return "other: " + m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_shade_h<0, false, false, 2, true>")
    a = ap.parse_args()
    lines = disassemble(a.kernel)
    mn = [(l.split()[0], l) for l in lines]
    prio = [i for i, (m, l) in enumerate(mn) if m == "s_setprio"]
    scratch = [(i, l.strip().split("//")[0].strip()) for i, (m, l) in enumerate(mn) if m.startswith("scratch_")]
    if len(prio) < 2:
        raise SystemExit("phase markers (s_setprio) not found")
    g0, g1 = prio[0], prio[1]
    # the MLP phase ends at the last backward branch of the tile loop (first s_cbranch after the phase with a negative / large offset)
    def offset(l):   # branch operand as printed by llvm-objdump: a 16-bit word count, >= 32768 = backwards
        try:
            return int(l.split()[1])
        except (IndexError, ValueError):
            return 0
    back = [i for i, (m, l) in enumerate(mn) if i > g1 and (m.startswith("s_cbranch") or m == "s_branch") and offset(l) >= 32768]
    m1 = back[0] if back else len(mn) - 1
    phases = {"gather + basis (s_setprio 2 ... s_setprio 0)": (g0, g1), "MLP: PE, layers 1-3, sigmoid, store (s_setprio 0 ... loop branch)": (g1, m1)}
    print(f"kernel {a.kernel}: {len(mn)} instructions in the code object; tile loop = instructions {g0} .. {m1}")
    print(f"scratch (spill) instructions: {[(i, s) for i, s in scratch]}")
    print("  -> " + ("none: the kernel does not spill" if not scratch else
                     "all OUTSIDE the tile loop (prologue store / per-64-tile mask batch reload): the spilled VGPRs cost nothing per tile"
                     if all(i < g0 or i > m1 for i, _ in scratch) else "some INSIDE the tile loop"))
    total = collections.Counter()
    for name, (lo, hi) in phases.items():
        c, v = collections.Counter(), collections.Counter()
        for m, l in mn[lo:hi]:
            k = klass(m)
            c[k] += 1
            if k == "VALU":
                v[valu_kind(l, m)] += 1
        total.update(c)
        clk = c["VALU"] * 4 + c["MFMA fp16 (32x32x16)"] * 32 + c["MFMA fp6/fp8 (block-scaled, 32x32x64)"] * 38
        print(f"\n== {name}: {hi - lo} instructions (static; the border-straddling `mixed` path of the gather is counted too)")
        for k, n in c.most_common():
            print(f"   {k:44s} {n:5d}")
        print(f"   issue cost of this phase: VALU x 4 + fp16 MFMA x 32 + scaled MFMA x 38 = {clk} clk per tile and wave")
        print("   VALU by purpose:")
        for k, n in v.most_common():
            print(f"      {k:74s} {n:5d}")
    pm = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "pmc_traffic.json")))
    if pm:
        d = json.load(open(pm[-1])).get("k_shade_h<SHADE,f16f6>")
        if d and "SQ_INSTS_VALU_per_SE" in d:
            tiles = 4096 * 512 / 32
            mfma = d["SQ_INSTS_MFMA_per_SE"] * 32 / tiles
            valu = d["SQ_INSTS_VALU_per_SE"] * 32 / tiles - mfma
            print(f"\ncounters of the same kernel ({os.path.relpath(pm[-1], REPO)}): {valu:.0f} VALU and {mfma:.0f} MFMA instructions EXECUTED per tile "
                  f"(the static count above includes the rare mixed-grid path once more)")
    print(f"\nstatic totals of the tile loop: {dict(total)}")


if __name__ == "__main__":
    main()
