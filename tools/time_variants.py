"""Times render frames for each lib variant (separate processes).  usage: time_variants.py scene spp variant..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
name, spp = sys.argv[1], int(sys.argv[2])
if name == "soup":
    import bench
    s = bench.soup_scene(R, 1_000_000, 20250725)
else:
    s = R.Scene(path=os.path.join(%r, "assets/scene", name + ".scene"), base_dir=os.path.join(%r, "assets"))
p = R.make_params(s.width, s.height, spp=spp, seed=0)
buf = torch.empty(s.height * s.width * 3, dtype=torch.float64, device="cuda")
s.time_render_device(p, buf.data_ptr(), 1)
ms = s.time_render_device(p, buf.data_ptr(), 3)
chk = float(buf.sum().item())
print("%%-28s %%-12s spp=%%d  %%8.2f ms  checksum %%.12f" %% (os.environ.get("RTM_TAG", ""), name, spp, ms, chk))
''' % (ROOT, ROOT, ROOT)
scene, spp = sys.argv[1], sys.argv[2]
for v in sys.argv[3:]:
    env = dict(os.environ)
    tag = v
    if "@" in v:  # variant@ENV=VAL,ENV=VAL
        v, kv = v.split("@", 1)
        for item in kv.split(","):
            k, val = item.split("=")
            env[k] = val
    if v != "default":
        env["RTM_LIB_PATH"] = os.path.join(ROOT, "raytracer-rs_amd/lib/variants", f"lib_{v}.so")
    env["RTM_TAG"] = tag
    r = subprocess.run([sys.executable, "-c", child, scene, spp], env=env, capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout if r.returncode == 0 else f"{tag}: FAILED rc={r.returncode}\n{r.stderr[-800:]}\n")
    sys.stdout.flush()
    if r.returncode != 0:
        sys.exit(1)
