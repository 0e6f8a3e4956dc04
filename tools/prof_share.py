"""Renders rank 0's share of the bunny frame at a world size, a few times (for rocprofv3 --kernel-trace).  usage: prof_share.py world iters [streams]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
world, iters = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3: R.set_tuning("front_streams", int(sys.argv[3]))
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
for _ in range(iters):
    s.render_device(p, buf.data_ptr())
    torch.cuda.synchronize()
print("done")
