"""A longer run of the differential fuzz of test_parity_gpu.py (not collected by pytest: run by hand on a GPU box).

    python tests/fuzz_long.py [trials] [first seed]

Random scenes of spheres (up to 200: the packed passes and the f32 sphere screen), planes, lights and triangle soups, rendered
at 2 spp by the device path (counting build, product build, wave kernel) and by the oracle: every f64 word, id and counter."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from test_parity_gpu import _random_scene_text, _soup, assert_frames_match  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import oracle as orc
    from rtm_loader import load_package
    rtm = load_package()
    rtm.init(0)
    for trial in range(trials):
        rng = np.random.default_rng(seed0 + trial)
        n_sph = int(rng.choice([0, 3, 5, 9, 31, 33, 64, 130, 200]))
        txt = _random_scene_text(rng, n_sph, int(rng.integers(0, 3)))
        a, b = rtm.Scene(text=txt), orc.OracleScene(text=txt)
        for m in range(int(rng.integers(0, 3))):
            pos, nrm = _soup(rng, int(rng.integers(50, 3000)), (rng.uniform(-1, 1), rng.uniform(-1, 1), 5.0 + m), 1.5, rng.uniform(0.05, 0.6))
            a.add_triangles(pos, nrm, metal=bool(m))
            b.add_triangles(pos, nrm, metal=bool(m))
        spp = int(rng.integers(1, 4))
        cr = b.render(spp=spp, seed=trial)
        gr = a.render(spp=spp, seed=trial, stats=True)
        assert_frames_match(gr, cr, f"fuzz_long seed {seed0 + trial}")
        assert gr["counters"] == cr["counters"], f"seed {seed0 + trial}"
        assert np.array_equal(a.render(spp=spp, seed=trial)["rgb"], cr["rgb"]), f"seed {seed0 + trial} (product build)"
        assert np.array_equal(a.render(spp=spp, seed=trial, kernel_wave=True)["rgb"], cr["rgb"]), f"seed {seed0 + trial} (wave kernel)"
        print(f"seed {seed0 + trial}: {n_sph} spheres, {spp} spp: equal", flush=True)
    print("all equal")


if __name__ == "__main__":
    main()
