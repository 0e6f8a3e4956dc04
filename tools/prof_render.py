"""Small driver for rocprofv3 passes: renders <workload> frames through the C ABI (no oracle, no bench extras)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
rtm = load_package()
rtm.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if name == "soup":
    import bench
    s = bench.soup_scene(rtm, 1_000_000, 20250725)
else:
    s = rtm.Scene(path=os.path.join(ROOT, "assets", "scene", name + ".scene"), base_dir=os.path.join(ROOT, "assets"))
p = rtm.make_params(s.width, s.height, spp=spp, seed=0)
buf = torch.empty(s.height * s.width * 3, dtype=torch.float64, device="cuda")
for _ in range(iters):
    s.render_device(p, buf.data_ptr())
torch.cuda.synchronize()
print("done", name, spp, iters)
