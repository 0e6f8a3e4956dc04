"""Config 5 (1M-triangle soup + teapot, 3840x3840): per-kernel split and production work counters at a given spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from rtm_loader import load_package
R = load_package(); R.init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
s = bench.soup_scene(R, 1_000_000, 20250725)
p = R.make_params(size, size, spp=spp, seed=0)
buf = torch.empty(size * size * 3, dtype=torch.float64, device="cuda")
pr = s.profile_render_device(p, buf.data_ptr())
ms = s.time_render_device(p, buf.data_ptr(), 2)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in pr.items()}, flush=True)
print(f"frame {ms:.1f} ms; {pr['rays_total']/ms/1e3:.1f} M issued rays/s; per traced ray: cells {pr['cells_visited']/pr['traced_rays']:.1f}, tri tests {pr['tri_tests']/pr['traced_rays']:.1f}", flush=True)
if os.environ.get("RTM_DIAG"):
    print(s.render_device(p, buf.data_ptr(), stats=True), flush=True)
