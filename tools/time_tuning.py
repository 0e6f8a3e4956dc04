"""Times the bunny frame (and rank 0's share at world 8) for values of one rtm_set_tuning key: time_tuning.py key v1 v2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
key = sys.argv[1]
for v in sys.argv[2:]:
    R.set_tuning(key, int(v))
    out = []
    for world in (1, 8):
        p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
        buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
        s.time_render_device(p, buf.data_ptr(), 1)
        out.append(s.time_render_device(p, buf.data_ptr(), 5))
    print(f"{key} {v}: w1 {out[0]:.2f} ms  w8 {out[1]:.2f} ms", flush=True)
