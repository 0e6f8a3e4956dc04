"""Frame times of the four named scenes at 20 spp (default settings)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
for name in ("simple_sphere", "multisphere", "suzanne", "bunny"):
    s = R.Scene(path=os.path.join(ROOT, "assets/scene", name + ".scene"), base_dir=os.path.join(ROOT, "assets"))
    p = R.make_params(s.width, s.height, spp=20, seed=0)
    buf = torch.empty(s.height * s.width * 3, dtype=torch.float64, device="cuda")
    s.time_render_device(p, buf.data_ptr(), 1)
    print(f"{name:14s} {s.time_render_device(p, buf.data_ptr(), 5):8.2f} ms   checksum {float(buf.sum().item()):.9f}", flush=True)
