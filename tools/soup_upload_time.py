"""Host-side cost of the 1M-triangle soup scene: front end (grid build) and rtm_scene_upload (records, masks, bitmaps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from rtm_loader import load_package
R = load_package(); R.init(0)
t = time.time(); s = bench.soup_scene(R, 1_000_000, 20250725); t1 = time.time() - t
t = time.time(); s.device_handle(); torch.cuda.synchronize(); t2 = time.time() - t
p = R.make_params(3840, 3840, spp=1, seed=0)
buf = torch.empty(3840 * 3840 * 3, dtype=torch.float64, device="cuda")
ms = s.time_render_device(p, buf.data_ptr(), 1)
print(f"front end (parse, OBJ, grid of 1M triangles, numpy generation) {t1:.2f} s; rtm_scene_upload {t2:.2f} s; one 3840^2 1-spp frame {ms:.0f} ms", flush=True)
