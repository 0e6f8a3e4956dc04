"""rocprofv3 driver: rank 0's share of the bunny frame at world size N (row interleave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
for _ in range(2):
    s.render_device(p, buf.data_ptr())
torch.cuda.synchronize()
