"""One rank's share of the bunny frame for world sizes 1,2,4,8, for several front_streams settings, one GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
for streams in (1, 2, 3, 4):
    R.set_tuning("front_streams", streams)
    out = []
    for world in (1, 2, 4, 8):
        p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
        rows = R.local_rows(p)
        buf = torch.empty(rows * s.width * 3, dtype=torch.float64, device="cuda")
        s.time_render_device(p, buf.data_ptr(), 1)
        out.append(s.time_render_device(p, buf.data_ptr(), 5))
    print(f"front_streams {streams}: " + "  ".join(f"w{w} {m:.2f} ms" for w, m in zip((1, 2, 4, 8), out)), flush=True)
