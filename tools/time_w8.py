"""Rank 0's share of the bunny frame at world sizes 1 and 8 (env knobs are read at init: one process per setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); R.set_tuning(k, int(v))
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
out = []
for world in (1, 8):
    p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
    buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
    s.time_render_device(p, buf.data_ptr(), 1)
    out.append(s.time_render_device(p, buf.data_ptr(), 8))
print(" ".join(sys.argv[1:]), {k: os.environ[k] for k in os.environ if k.startswith("RTM_")}, f"w1 {out[0]:.2f} ms  w8 {out[1]:.2f} ms", flush=True)
