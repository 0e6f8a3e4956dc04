"""One rank's share of BASELINE config 5 (10^6-triangle soup + teapot, 3840^2) for world sizes 1, 2, 4, 8 on one GPU:
rows y = 0 (mod world).  usage: time_split_soup.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from rtm_loader import load_package
R = load_package(); R.init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
s = bench.soup_scene(R, 1_000_000, 20250725)
out = []
for world in (1, 2, 4, 8):
    p = R.make_params(s.width, s.height, spp=spp, seed=0, row_begin=0, row_stride=world)
    buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
    s.time_render_device(p, buf.data_ptr(), 1)
    out.append(s.time_render_device(p, buf.data_ptr(), 2))
    print(f"world {world}: {out[-1]:.1f} ms", flush=True)
print(f"soup {s.width}x{s.height} at {spp} spp: " + "  ".join(f"w{w} {m:.1f} ms" for w, m in zip((1, 2, 4, 8), out)) +
      f"   w1/w8 = {out[0] / out[3]:.2f}", flush=True)
