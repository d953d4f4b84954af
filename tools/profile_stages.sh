#!/bin/bash
# PMC counters for the isolated phases of the shade kernel (tools/stage_timing.py): k_shade_h<1> = gather+basis, <2> = MLP
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_stages
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-50)
  rocprofv3 --kernel-trace --pmc $pmc -d "$OUT/$name" -o p -- python $ROOT/tools/stage_timing.py f16x3 > "$OUT/$name.log" 2>&1
done
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
for p in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    db = sqlite3.connect(p)
    q = "select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%k_shade_h%' group by name, counter_name"
    for name, ctr, val, n in db.execute(q):
        tag = name.split("k_shade_h")[1][:12]
        print(f"{tag:14s} {ctr:28s} {val:16.1f} n={n}")
PY
