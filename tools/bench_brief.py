"""bench.py (no CPU baseline), three times, reduced to ms per step / per kernel — for A/B runs inside one gpurun session."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"], capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print("ms/step", round(d["ms_per_step"], 4), "shade", round(d["roofline"]["ms"], 4),
          {k: round(v, 4) for k, v in d["roofline"]["other_kernels_ms"].items()})
