"""Generates the committed golden fixtures from the CPU oracle (oracle/), in THIS container.

The reference cannot run anywhere in this pipeline (Rust, no toolchain) and ships no golden images
(.MISSING_LARGE_BLOBS), so these vectors pin the ORACLE's behaviour at commit time, not the reference's:
they catch drift of the oracle and let the GPU tests run without trusting a freshly built oracle alone.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

SIZE, SEED = 96, 20250725
for name in ("simple_sphere", "multisphere", "suzanne", "bunny"):
    s = O.OracleScene(path=os.path.join(ROOT, "assets", "scene", name + ".scene"), base_dir=os.path.join(ROOT, "assets"))
    r1 = s.render(spp=1, seed=SEED, width=SIZE, height=SIZE)
    r4 = s.render(spp=4, seed=SEED, width=SIZE, height=SIZE)
    cnt = np.array([r1["counters"][k] for k in O.COUNTER_NAMES], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, f"{name}_{SIZE}.npz"), size=SIZE, seed=SEED, rgb_1spp=r1["rgb"],
                        obj_id=r1["obj_id"], tri_id=r1["tri_id"], counters_1spp=cnt, rgb_4spp=r4["rgb"])
    print(name, "written", r1["rgb"].mean())
