"""Times the bunny frame and rank 0's share at world 8 under the current environment (RTM_* variables)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
out = []
for world in (1, 8):
    p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
    buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
    s.time_render_device(p, buf.data_ptr(), 1)
    out.append(s.time_render_device(p, buf.data_ptr(), 5))
print(" ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RTM_")), f": w1 {out[0]:.2f} ms  w8 {out[1]:.2f} ms", flush=True)
