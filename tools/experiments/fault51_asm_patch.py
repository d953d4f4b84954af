"""DESIGN.md 5.1, bounded follow-up: build the KNOWN-FAULTY form of ego_shade.hip (-DEGO_PAIRED_WEIGHTS, SLP vectoriser on), then
re-assemble its own device assembly with ONLY the packed fp32 instructions that broadcast the HIGH half of a register pair
(`op_sel:[1,0,0]` / `[0,1]` / `[1,0]` without op_sel_hi) replaced by their two-instruction scalar equivalents - same registers,
same schedule, same arithmetic (v_fma_f32 / v_mul_f32 round like the packed forms).  If the failing library fails and the patched
one is clean in one GPU session, the instruction FORM is the trigger, not the register allocation or schedule around it.

  python tools/experiments/fault51_asm_patch.py <out_dir> [kernel-substring]
writes <out_dir>/libvariant_f0.so (failing form, unpatched assembly through the same manual pipeline), libvariant_p1.so (replaced)
and p2 / p3 / p4: the instructions kept, each preceded by s_waitcnt vmcnt(0) / s_waitcnt lgkmcnt(0) / two s_nop 7."""
import os, re, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from egonerf_amd import build as B
out = os.path.abspath(sys.argv[1])
only = sys.argv[2] if len(sys.argv) > 2 else None
os.makedirs(out, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
flags = [*B.COMMON_FLAGS, "-DEGO_PAIRED_WEIGHTS"]   # no -fno-slp-vectorize: the faulty form
src = os.path.join(B.CSRC, "ego_shade.hip")
run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)
tmp = os.path.join(out, "tmp"); shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
run([B._hipcc(), *flags, "-c", src, "-o", os.path.join(tmp, "ref.o"), "-save-temps=obj"], cwd=tmp)
asm = os.path.join(tmp, "ego_shade-hip-amdgcn-amd-amdhsa-gfx950.s")
text = open(asm).read()

def patch(text):
    n, cur, outl = 0, None, []
    pair = lambda s: int(re.match(r"v\[(\d+):(\d+)\]", s).group(1))
    for line in text.splitlines():
        m = re.match(r"^(_ZN\S+):", line)
        if m: cur = m.group(1)
        mm = re.match(r"\s*v_pk_(fma|mul)_f32 (v\[\d+:\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\])(?:, (v\[\d+:\d+\]))? op_sel:\[([01,]+)\]\s*$", line)
        if mm and (only is None or only in (cur or "")):
            op, d, a, b, c, sel = mm.group(1), pair(mm.group(2)), pair(mm.group(3)), pair(mm.group(4)), mm.group(5), [int(x) for x in mm.group(6).split(",")]
            # op_sel picks the source half for the LOW result; op_sel_hi (absent = all ones) the HIGH half for the high result
            al, bl = a + sel[0], b + sel[1]
            if op == "fma":
                c = pair(c); cl = c + sel[2]
                outl.append(f"\tv_fma_f32 v{d}, v{al}, v{bl}, v{cl}")
                outl.append(f"\tv_fma_f32 v{d + 1}, v{a + 1}, v{b + 1}, v{c + 1}")
            else:
                outl.append(f"\tv_mul_f32_e64 v{d}, v{al}, v{bl}")
                outl.append(f"\tv_mul_f32_e64 v{d + 1}, v{a + 1}, v{b + 1}")
            # the low result must not clobber a register the high instruction still reads
            assert d not in (a + 1, b + 1) and (op != "fma" or d != c + 1), line
            n += 1
        else:
            outl.append(line)
    return "\n".join(outl) + "\n", n

def assemble(asm_text, name):
    s = os.path.join(tmp, name + ".s"); open(s, "w").write(asm_text)
    o, hs, fb, ho = (os.path.join(tmp, name + e) for e in (".o", ".hsaco", ".hipfb", "_host.o"))
    run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hs, o])
    run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", f"-input={hs}", f"-output={fb}"])
    run([B._hipcc(), *flags, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", ho])
    objs = [ho]
    for f in B.SOURCES:
        if f == "ego_shade.hip": continue
        ob = os.path.join(tmp, f.replace(".hip", ".o"))
        if not os.path.exists(ob):
            run([B._hipcc(), *B.COMMON_FLAGS, "-c", os.path.join(B.CSRC, f), "-o", ob])
        objs.append(ob)
    lib = os.path.join(out, f"libvariant_{name}.so")
    run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    return lib

def pad(text, before):
    """Leave the instructions alone and put `before` in front of each of them (timing / wait experiments)."""
    outl, n = [], 0
    for line in text.splitlines():
        if re.match(r"\s*v_pk_(fma|mul)_f32 .* op_sel:\[[01,]+\]\s*$", line):
            outl.extend(before); n += 1
        outl.append(line)
    return "\n".join(outl) + "\n", n


patched, n = patch(text)
left = len(re.findall(r"v_pk_(?:fma|mul|add)_f32[^\n]*op_sel:\[[01,]+\]\s*\n", patched))
print("high-half broadcast instructions replaced:", n, "| left in the patched assembly:", left)
print(assemble(text, "f0"))
print(assemble(patched, "p1"))
for name, before in (("p2", ["\ts_waitcnt vmcnt(0)"]), ("p3", ["\ts_waitcnt lgkmcnt(0)"]), ("p4", ["\ts_nop 7", "\ts_nop 7"])):
    t, k = pad(text, before)
    print(name, k, "instructions padded with", before, assemble(t, name))
