#!/bin/bash
# On the GPU box: kernel + memory-copy timeline of volume_renderer(host rays, keep_alpha=True, empty_gpu_cache=True) - the reference's call
# pattern - over a few 4096 x 512 chunks: do the device -> host copies run under the next chunk's kernels, and how long do they take?
cd /tmp && export TMPDIR=/tmp EGO_SKIP_SELFTEST=1
rm -rf /tmp/ht
cat > /tmp/ht_run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from egonerf_amd import synth
from egonerf_amd.renderer import volume_renderer
dev = torch.device("cuda", 0)
cfg = synth.SceneConfig()
model = synth.build_model(cfg, synth.make_weights(cfg, seed=1234), dev)
kw = dict(n_coarse=512, n_fine=0, exp_sampling=True, resampling=False, use_coarse_sample=True, chunk=4096, device=dev)
host = torch.from_numpy(synth.make_rays(4096 * 16, seed=1))
with torch.no_grad():
    for _ in range(3):
        volume_renderer(host, model, keep_alpha=True, empty_gpu_cache=True, **kw)
        torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ht -o ht -- python /tmp/ht_run.py > /tmp/ht.log 2>&1 || tail -5 /tmp/ht.log
python - <<'PY'
import sqlite3, glob
p = glob.glob("/tmp/ht/**/*.db", recursive=True)[0]
db = sqlite3.connect(p)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
mt = [t for t in tabs if "memory_copy" in t]
print("copy tables:", mt)
sym = [t for t in tabs if "kernel_symbol" in t]
names = {}
sc = [r[1] for r in db.execute(f"pragma table_info('{sym[0]}')")]
nm = "display_name" if "display_name" in sc else "kernel_name"
for kid, n in db.execute(f"select id, {nm} from '{sym[0]}'"):
    names[kid] = n.replace("(anonymous namespace)::", "").replace("void ", "")[:50]
ev = [(s, e, "K " + names.get(k, str(k))) for s, e, k in db.execute(f"select start, end, kernel_id from '{kt}'")]
if mt:
    cols = [r[1] for r in db.execute(f"pragma table_info('{mt[0]}')")]
    print(cols)
    szc = "size" if "size" in cols else None
    q = f"select start, end, {szc or 0} from '{mt[0]}'"
    ev += [(s, e, "C copy %d B" % sz) for s, e, sz in db.execute(q)]
ev.sort()
# last call: take the last 16 shade kernels' span
sh = [i for i, x in enumerate(ev) if "k_shade_h" in x[2]]
a = sh[-6]
t0 = ev[a][0]
for s, e, n in ev[a - 3:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {n}")
PY
