import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
name = sys.argv[1]; spp = int(sys.argv[2])
s = R.Scene(path=os.path.join(ROOT, "assets/scene", name + ".scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=spp, seed=0)
buf = torch.empty(s.height * s.width * 3, dtype=torch.float64, device="cuda")
print(os.environ.get("RTM_S_THRESHOLD"), os.environ.get("RTM_T_MIN_LANES"), s.render_device(p, buf.data_ptr(), stats=True), flush=True)
