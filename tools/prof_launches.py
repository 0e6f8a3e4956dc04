"""Per-launch durations (HIP events, one chunk in flight) of rank 0's share of the bunny frame at a world size."""
import os, sys
os.environ["RTM_PROFILE_ROUNDS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
R.set_tuning("front_streams", 1)
s.time_render_device(p, buf.data_ptr(), 2)
print(s.profile_render_device(p, buf.data_ptr()))
print("frame ms (1 stream):", s.time_render_device(p, buf.data_ptr(), 5))
