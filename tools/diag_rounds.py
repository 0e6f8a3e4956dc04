"""Per-round diagnostics of the ray-front pipeline (list lengths, slowest trace wave) for rank 0's share at a world size."""
import os, sys
os.environ["RTM_FRONT_DEBUG_ROUNDS"] = "1"; os.environ["RTM_DIAG_KEEP_CULL"] = "1"; os.environ["RTM_FRONT_STREAMS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
print(s.render_device(p, buf.data_ptr(), stats=True))
