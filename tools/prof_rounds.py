"""Per-launch kernel durations of one frame (RTM_PROFILE_ROUNDS=1), one chunk in flight.  usage: prof_rounds.py scene spp [world]"""
import os, sys
os.environ["RTM_PROFILE_ROUNDS"] = "1"; os.environ.setdefault("RTM_FRONT_STREAMS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
name, spp = sys.argv[1], int(sys.argv[2]); world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if name == "soup":
    import bench
    s = bench.soup_scene(R, 1_000_000, 20250725)
else:
    s = R.Scene(path=os.path.join(ROOT, "assets/scene", name + ".scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=spp, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
s.profile_render_device(p, buf.data_ptr())
pr = s.profile_render_device(p, buf.data_ptr())
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in pr.items()})
print("frame ms", s.time_render_device(p, buf.data_ptr(), 5))
