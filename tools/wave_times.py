"""Diagnostics (variant library built with -DRTM_WAVE_TIMES): when the trace waves of each launch start, run out of queue and end.
usage: wave_times.py [world]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTM_LIB_PATH"] = os.path.join(ROOT, "raytracer-rs_amd/lib/variants/lib_wt.so")
os.environ["RTM_WAVE_TIMES"] = "/tmp/wave_times.bin"
os.environ.setdefault("RTM_FRONT_STREAMS", "1")
import torch
from rtm_loader import load_package
R = load_package(); R.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = R.Scene(path=os.path.join(ROOT, "assets/scene/bunny.scene"), base_dir=os.path.join(ROOT, "assets"))
p = R.make_params(s.width, s.height, spp=20, seed=0, row_begin=0, row_stride=world)
buf = torch.empty(R.local_rows(p) * s.width * 3, dtype=torch.float64, device="cuda")
for _ in range(2):
    s.render_device(p, buf.data_ptr()); torch.cuda.synchronize()
a = np.fromfile("/tmp/wave_times.bin", dtype=np.uint64).reshape(64, 8192, 4).astype(np.int64)
for r in range(12):
    w = a[r]; w = w[w[:, 2] > 0]
    if not len(w): continue
    t0 = w[:, 0].min(); us = lambda x: (x - t0) / 100.0
    beg, exh, end, rays = us(w[:, 0]), us(np.where(w[:, 1] > 0, w[:, 1], w[:, 2])), us(w[:, 2]), w[:, 3]
    q = lambda v: " ".join(f"{x:7.1f}" for x in np.percentile(v, [0, 10, 50, 90, 99, 100]))
    print(f"launch {r}: waves {len(w)} rays {rays.sum()} (per wave min {rays.min()} median {int(np.median(rays))} max {rays.max()})")
    print(f"   begin     us  p0/10/50/90/99/100: {q(beg)}")
    print(f"   exhausted us  p0/10/50/90/99/100: {q(exh)}")
    print(f"   end       us  p0/10/50/90/99/100: {q(end)}")
    print(f"   end - exhausted                 : {q(end - exh)}", flush=True)
if os.environ.get("WT_DETAIL"):
    r = int(os.environ["WT_DETAIL"]); w = a[r]; ids = np.nonzero(w[:, 2] > 0)[0]; w = w[ids]; t0 = w[:, 0].min()
    order = np.argsort(w[:, 1])
    print("earliest exhausted waves of launch", r, ": wave id, rays, exhausted us, end us")
    for i in list(order[:24]) + list(order[-8:]):
        print(f"   {ids[i]:5d} {w[i, 3]:6d} {(w[i, 1] - t0) / 100.0:8.1f} {(w[i, 2] - t0) / 100.0:8.1f}")
    early = ids[order[: len(order) // 10]]
    print("earliest tenth: wave id mod 4 histogram", np.bincount(early % 4, minlength=4), " id >= 2048:", int((early >= 2048).sum()), "of", len(early),
          " ids mod 256 spread:", np.bincount((early // 4) % 8, minlength=8))
