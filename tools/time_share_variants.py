"""Rank shares (world 1, 8) of the bunny frame for lib variants (separate processes).  usage: time_share_variants.py variant..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
s = R.Scene(path=os.path.join(%r, "assets/scene/bunny.scene"), base_dir=os.path.join(%r, "assets"))
out = []
for world in (1, 2, 4, 8):
    p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
    buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
    s.time_render_device(p, buf.data_ptr(), 1)
    out.append(s.time_render_device(p, buf.data_ptr(), 5))
print("%%-24s " %% os.environ.get("RTM_TAG", "") + "  ".join("w%%d %%.2f ms" %% (w, m) for w, m in zip((1, 2, 4, 8), out)))
''' % (ROOT, ROOT, ROOT)
for v in sys.argv[1:]:
    env = dict(os.environ); tag = v
    if "@" in v:
        v, kv = v.split("@", 1)
        for item in kv.split(","):
            k, val = item.split("="); env[k] = val
    if v != "default":
        env["RTM_LIB_PATH"] = os.path.join(ROOT, "raytracer-rs_amd/lib/variants", f"lib_{v}.so")
    env["RTM_TAG"] = tag
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout if r.returncode == 0 else f"{tag}: FAILED rc={r.returncode}\n{r.stderr[-800:]}\n"); sys.stdout.flush()
