"""Diagnostics (variant library built with -DRTM_SHADE_TIMES): cycles per phase of the shade kernel's iterations, per round."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTM_LIB_PATH"] = os.path.join(ROOT, "raytracer-rs_amd/lib/variants/lib_st.so")
os.environ["RTM_SHADE_TIMES"] = "1"
os.environ.setdefault("RTM_FRONT_STREAMS", "1")
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
s.render_device(p, buf.data_ptr()); torch.cuda.synchronize()
