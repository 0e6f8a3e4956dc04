"""rtm_grid_build: Model::create_grid (model.rs:147-201) on the GPU against the host front end's
grid -- which tests/test_frontend_abi.py pins bit-equal to the oracle's restatement and to the survey's list counts
(44,343 / 32,420: only right with Box3::include's quirk, box3D.rs:31-45).  Bar: offset_array and triangle_indices
identical, element for element."""
import numpy as np
import pytest

import bench
from conftest import ASSETS

pytestmark = pytest.mark.gpu


def _check_model(rtm, m, what):
    pos9 = m["pos"]                                   # [n_tris, 9] f64: get_triangle(i), model.rs:108-122
    off, idx = rtm.grid_build(pos9, m["bbox_center"], m["bbox_half"], m["cells_per_side"])
    assert np.array_equal(off, m["offsets"]), f"{what}: offset_array differs"
    assert np.array_equal(idx, m["tri_idx"]), f"{what}: triangle_indices differ"
    return off, idx


@pytest.mark.parametrize("name,n_idx", (("bunny", 44343), ("suzanne", 32420)))
def test_named_models(name, n_idx, rtm_gpu, rtm_scene):
    m = rtm_scene(name).flat()["models"][0]
    off, idx = _check_model(rtm_gpu, m, name)
    assert idx.size == n_idx and off[-1] == n_idx    # the counts the include quirk produces


def test_million_triangle_soup_and_teapot(rtm_gpu):
    s = bench.soup_scene(rtm_gpu, 1_000_000, 20250725)
    for m, what in zip(s.flat()["models"], ("teapot", "soup")):
        _check_model(rtm_gpu, m, what)
    s.close()


def test_edge_inputs(rtm_gpu):
    """Triangles outside the box (saturating casts: negative -> cell 0, beyond -> n - 1), degenerate and repeated
    triangles, a box with a zero extent (cell size 0: divisions by zero, NaN -> 0), other grid sizes, no triangles."""
    rng = np.random.default_rng(5)
    for n_cells, n_tris, spread in ((32, 3000, 1.5), (7, 500, 0.7), (1, 10, 1.0), (64, 2000, 1.2), (128, 300, 1.0)):
        c = rng.uniform(-spread, spread, (n_tris, 1, 3))
        pos = (c + rng.uniform(-0.3, 0.3, (n_tris, 3, 3))).reshape(n_tris, 9)
        pos[::17, 3:] = np.tile(pos[::17, :3], 2)            # points
        pos[5::23] = pos[4::23][: pos[5::23].shape[0]]        # repeated triangles
        txt = "size 8 8"
        a = rtm_gpu.Scene(text=txt)
        a.add_triangles(pos.astype(np.float32), np.zeros_like(pos, dtype=np.float32) + 1.0)
        m = a.flat()["models"][0]
        if n_cells == 32:
            _check_model(rtm_gpu, m, f"soup {n_tris}")
        # other grid sizes and boxes smaller than the triangles: against a numpy restatement of model.rs:147-201
        ctr, half = np.array([0.1, -0.2, 0.05]), np.array([1.0, 0.8, 0.0 if n_cells == 7 else 1.1])
        off, idx = rtm_gpu.grid_build(m["pos"], ctr, half, n_cells)
        roff, ridx = _create_grid_numpy(m["pos"], ctr, half, n_cells)
        assert np.array_equal(off, roff) and np.array_equal(idx, ridx), f"n={n_cells}"
    off, idx = rtm_gpu.grid_build(np.zeros((0, 9)), (0, 0, 0), (1, 1, 1), 32)
    assert idx.size == 0 and not off.any()
    with pytest.raises(rtm_gpu.RtmError):
        rtm_gpu.grid_build(np.zeros((1, 9)), (0, 0, 0), (1, 1, 1), 129)


def _create_grid_numpy(pos9, center, half, n):
    """model.rs:147-201 with box3D.rs:17-45, 55-61 in numpy f64 (IEEE, no contraction), vectorised over triangles."""
    with np.errstate(all="ignore"):
        bmin, bmax = center - half, center + half
        cell = (bmax - bmin) / float(n)
        lo_hi = []
        for c in range(3):
            ctr, h = pos9[:, c].copy(), np.zeros(pos9.shape[0])
            for k in (1, 2):
                d = pos9[:, k * 3 + c] - ctr
                grow = np.abs(d) > h
                h2 = (np.abs(d) + h) / 2.0
                ctr2 = ctr + (d - h2 * np.where(np.signbit(d), -1.0, 1.0)) / 2.0
                h, ctr = np.where(grow, h2, h), np.where(grow, ctr2, ctr)

            def cast(v):  # `as usize` saturating, NaN -> 0, then .min(n - 1)
                v = np.where(v > 0, v, 0.0)
                return np.minimum(np.where(v >= n, n - 1, np.floor(v)), n - 1).astype(np.int64)
            lo_hi.append((cast(((ctr - h) - bmin[c]) / cell[c]), cast(((ctr + h) - bmin[c]) / cell[c])))
    lists = [[] for _ in range(n ** 3)]
    (x0, x1), (y0, y1), (z0, z1) = lo_hi
    for t in range(pos9.shape[0]):
        for ix in range(x0[t], x1[t] + 1):
            for iy in range(y0[t], y1[t] + 1):
                for iz in range(z0[t], z1[t] + 1):
                    lists[iz * n * n + iy * n + ix].append(t)
    off = np.cumsum([len(l) for l in lists]).astype(np.uint32)
    idx = np.array([t for l in lists for t in l], dtype=np.uint32)
    return off, idx
