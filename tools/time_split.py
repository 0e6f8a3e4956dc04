"""Times one rank's share of the bunny frame for world sizes 1,2,4,8 (row interleave), one GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
for world in (1, 2, 4, 8):
    p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
    rows = R.local_rows(p)
    buf = torch.empty(rows * s.width * 3, dtype=torch.float64, device="cuda")
    s.time_render_device(p, buf.data_ptr(), 1)
    ms = s.time_render_device(p, buf.data_ptr(), 5)
    pw = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world, kernel_wave=True)
    s.time_render_device(pw, buf.data_ptr(), 1)
    msw = s.time_render_device(pw, buf.data_ptr(), 5)
    print(f"world {world}: rank 0 renders {rows} rows in {ms:.2f} ms (front), {msw:.2f} ms (wave) -> ideal scaling x{world}, render-only x{(49.16 if world>1 else ms)/ms:.2f}", flush=True)
