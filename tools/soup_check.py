"""Config-5 check: 1M-triangle soup + teapot; parity vs oracle at 256x256 1 spp, then timing at 3840x3840."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, bench, oracle as O
from rtm_loader import load_package
R = load_package(); R.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
s = bench.soup_scene(R, n, 20250725)
f = s.flat()
txt = ("size 3840 3840\n\ncamera from (4.0, -8, 0) to (0, 0, 7)\n\nplane (0, 0, 1.0) 10 white\nplane (0, 1.0, 0) 2 metal:white\n"
       'model "./models/teapot.obj" white > scale 0.4 > translate (2.5, 1.2, 6)\nlight (1, 0, 2) green\nlight (-1, 0, 2) blue')
o = O.OracleScene(text=txt, base_dir=os.path.join(ROOT, "assets"), allow_missing_normals=True)
rng = np.random.Generator(np.random.Philox(20250725))
c = rng.uniform(-2, 2, (n, 1, 3)) + np.array([0.0, 0.0, 7.0])
pos = (c + rng.uniform(-0.02, 0.02, (n, 3, 3))).astype(np.float32)
g = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]); g /= np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
o.add_triangles(pos.reshape(n, 9), np.repeat(g, 3, axis=0).reshape(n, 9).astype(np.float32))
t = time.time(); cr = o.render(spp=1, seed=1, width=256, height=256); tc = time.time() - t
gr = s.render(spp=1, seed=1, width=256, height=256, stats=True)
same = np.array_equal(gr["obj_id"], cr["obj_id"]) and np.array_equal(gr["tri_id"], cr["tri_id"])
bits = int((gr["rgb"].view(np.uint64) != cr["rgb"].view(np.uint64)).sum())
rays = sum(cr["counters"][k] for k in ("rays_camera", "rays_bounce", "rays_shadow"))
print(f"parity 256^2 1spp: ids equal {same}, differing f64 words {bits}, counters equal {gr['counters'] == cr['counters']}; oracle {tc:.1f}s = {rays/tc/1e6:.2f} Mrays/s", flush=True)
gn = s.render(spp=1, seed=1, width=256, height=256)
gw = s.render(spp=1, seed=1, width=256, height=256, kernel_wave=True)
print("production path (culling, arrival masks) bits equal:", np.array_equal(gn["rgb"], cr["rgb"]), " wave kernel:", np.array_equal(gw["rgb"], cr["rgb"]), flush=True)
print("counters", cr["counters"], flush=True)
for spp in (1, 4, 20):
    p = R.make_params(3840, 3840, spp=spp, seed=0)
    buf = torch.empty(3840 * 3840 * 3, dtype=torch.float64, device="cuda")
    st = s.render_device(p, buf.data_ptr(), stats=True)
    ms = s.time_render_device(p, buf.data_ptr(), 1)
    r = st["rays_camera"] + st["rays_bounce"] + st["rays_shadow"]
    print(f"3840^2 spp={spp}: {ms:.1f} ms, {r/ms/1e3:.1f} Mrays/s, tri tests/ray {st['tri_tests']/r:.1f}, cells/ray {st['cells_visited']/r:.1f}", flush=True)
