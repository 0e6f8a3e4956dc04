"""Pins for the CPU oracle (the reference has no tests and no golden images: "parity unpinned" by it).

Every expected value here is derived BY HAND from the cited reference lines, not produced by the oracle.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import ROOT

NO_ID = 0xFFFFFFFF


def test_philox_known_answers(orc):
    # Random123 known-answer vectors for philox4x32-10 (counter, key) -> output
    assert [hex(v) for v in orc.philox(0, 0, 0, 0, 0)] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    seed = 0xFFFFFFFF | (0xFFFFFFFF << 32)
    assert [hex(v) for v in orc.philox(seed, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)] == [
        "0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    seed = 0xA4093822 | (0x299F31D0 << 32)
    assert [hex(v) for v in orc.philox(seed, 0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344)] == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_float_mapping(orc):
    # rand 0.8.5 UniformFloat: (u64 >> 12) into the mantissa of [1,2), minus 1, times (hi-lo), plus lo
    j, r = orc.sample_rng(7, 123, 4, 2)
    o = orc.philox(7, 123, 4, 0, 0)
    u = ((int(o[1]) << 32 | int(o[0])) >> 12) / 2.0 ** 52
    assert j[0] == u * 1.0 + -0.5
    assert -0.5 <= j[0] < 0.5 and -0.5 <= j[1] < 0.5
    a = orc.philox(7, 123, 4, 1 + 2 * 2, 0)
    b = orc.philox(7, 123, 4, 2 + 2 * 2, 0)
    assert r[0] == ((int(a[1]) << 32 | int(a[0])) >> 12) / 2.0 ** 52
    assert r[1] == ((int(a[3]) << 32 | int(a[2])) >> 12) / 2.0 ** 52
    assert r[2] == ((int(b[1]) << 32 | int(b[0])) >> 12) / 2.0 ** 52


def test_simple_sphere_centre_pixel(oracle_scene):
    """main.rs:165-168, camera.rs:46-52, core.rs:183-197 at pixel (600,600), zero jitter:
    u = v = 0 -> dir (0,0,1); oc = (0,0,-3), a = 1, b = -6, c = 8, disc = 4, t = (6-2)/2 = 2 -> sphere (object 0),
    P = (0,0,2), n = (0,0,-1)."""
    s = oracle_scene("simple_sphere")
    rays = s.shoot(np.array([[0.0, 0.0]]))
    assert rays.tolist() == [[0, 0, 0, 0, 0, 1]]
    h = s.trace(rays)
    assert h["obj_id"][0] == 0 and h["tri_id"][0] == NO_ID
    assert h["t"][0] == 2.0
    assert h["normal"][0].tolist() == [0.0, 0.0, -1.0]


def test_simple_sphere_corner_pixel(oracle_scene):
    """Pixel (0,0): u = v = -0.5 -> dir = (-.5,-.5,1)/sqrt(1.5) misses the sphere, hits plane z=6 (object 1):
    core.rs:199-205: dv = n.d = 1/sqrt(1.5); t = ((0,0,6)-o).n / dv = 6/dv; normal as written (0,0,1)."""
    s = oracle_scene("simple_sphere")
    rays = s.shoot(np.array([[-0.5, -0.5]]))
    l = math.sqrt(0.25 + 0.25 + 1.0)  # vec3.rs:179,213-215
    assert rays[0, 3:].tolist() == [-0.5 / l, -0.5 / l, 1.0 / l]
    h = s.trace(rays)
    assert h["obj_id"][0] == 1
    assert h["t"][0] == 6.0 / (1.0 / l)
    assert h["normal"][0].tolist() == [0.0, 0.0, 1.0]


def test_simple_sphere_centre_colour(oracle_scene):
    """cast (main.rs:57-132) at the centre pixel with zero jitter, by hand:
    hit 1: sphere at P=(0,0,2), n=(0,0,-1).  Lights sit at z=3 behind the surface point: v.n = -1/sqrt(5) < 0.001
    for all four -> diffuse 0 (occlusion does not matter).  Plastic bounce: normalize(n + random()).
    The first term is therefore 0 and colour only arrives from later bounces, attenuated by 0.1 per bounce
    (main.rs:112-115,125): the pixel must be <= 0.1 * (max light sum + sky) and strictly positive."""
    s = oracle_scene("simple_sphere")
    r = s.render(spp=1, seed=0, zero_jitter=True)
    assert r["obj_id"][600, 600] == 0
    px = r["rgb"][600, 600]
    assert np.all(px > 0.0) and np.all(px < 0.35)


def test_sphere_inside_is_invisible(orc):
    """Quirk Q3 (core.rs:194,75): only the near root is used, so from inside a sphere t < 0 and hit() drops it."""
    s = orc.OracleScene(text="size 8 8\nsphere (0,0,0) 5 white")
    h = s.trace(np.array([[0, 0, 0, 0, 0, 1.0]]))
    assert h["obj_id"][0] == NO_ID


def test_object_tie_goes_to_first(orc):
    """core.rs:79: strict '<' keeps the earlier object when two give the same t."""
    s = orc.OracleScene(text="size 8 8\nplane (0,0,1) 4 red\nplane (0,0,1) 4 blue")
    h = s.trace(np.array([[0, 0, 0, 0, 0, 1.0]]))
    assert h["obj_id"][0] == 0 and h["t"][0] == 4.0


def test_plane_normal_not_normalised(orc):
    """core.rs:198-210: t = ((n*d - o).n)/(n.d) with the normal as written; n=(0,0,2), d=3 -> centre (0,0,6),
    t = (6*2)/(2*1) = 6; normal returned un-normalised."""
    s = orc.OracleScene(text="size 8 8\nplane (0,0,2) 3 white")
    h = s.trace(np.array([[0, 0, 0, 0, 0, 1.0]]))
    assert h["t"][0] == 6.0 and h["normal"][0].tolist() == [0, 0, 2.0]


def test_material_keywords(orc):
    """parser.rs:318-332: `metal:` and `plastic:` both give Metal; bare colour gives Plastic; fuchsia is (0.96,0,96)."""
    s = orc.OracleScene(text="size 8 8\nsphere (0,0,3) 1 metal:red\nsphere (0,0,3) 1 plastic:blue\nsphere (0,0,3) 1 fuchsia")
    assert [s.object(i)[3] for i in range(3)] == [1, 1, 0]
    assert s.object(2)[2].tolist() == [0.96, 0.0, 96.0]


def test_parser_errors(orc):
    """parser.rs:445-469: the loop tests the raw cursor, so trailing blanks after a light raise "unexpected token ''"
    at 0-based line:column (parser.rs:67-75)."""
    with pytest.raises(ValueError, match=r"unexpected token '' at 2:0"):
        orc.OracleScene(text="size 10 10\nlight (0,0,0) red\n")
    with pytest.raises(ValueError, match="expected 'size'"):
        orc.OracleScene(text="sphere (0,0,0) 1 red")
    with pytest.raises(ValueError, match="cannot interp"):
        orc.OracleScene(text="size 10 ten")
    with pytest.raises(ValueError, match="Cannot load model"):
        orc.OracleScene(text='size 4 4\nmodel "./nope.obj" white', base_dir="/nonexistent")


def test_camera_default_and_look_at(orc):
    """camera.rs:13-43: default basis; look_at gives an orthonormal right-handed-by-construction basis."""
    s = orc.OracleScene(text="size 4 4")
    assert s.camera().tolist() == [0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1]
    s = orc.OracleScene(text="size 4 4\ncamera from (4.0, -8, 0) to (0, 0, 7)")
    c = s.camera()
    f = np.array([-4.0, 8.0, 7.0])
    f = f / math.sqrt((16.0 + 64.0) + 49.0)
    assert c[3:6].tolist() == f.tolist()
    assert abs(np.dot(c[3:6], c[6:9])) < 1e-15 and abs(np.dot(c[3:6], c[9:12])) < 1e-15


def test_grid_structural_goldens(oracle_scene):
    """counts that only come out right when Box3::include's quirk (box3D.rs:31-45) and the
    saturating binning (model.rs:159-167) are reproduced (a true-AABB binning gives 43,686 / 31,470)."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "grid_stats.json")))
    for name, oi in (("bunny", 2), ("suzanne", 0)):
        m = oracle_scene(name).model(oi)
        ln = np.diff(np.concatenate([[0], m["offsets"]]).astype(np.int64))
        g = gold[name]
        assert m["n_tris"] == g["n_tris"]
        assert len(m["tri_idx"]) == g["n_tri_idx"]
        assert int((ln > 0).sum()) == g["non_empty_cells"]
        assert int(ln.max()) == g["max_list"]
        assert len(m["offsets"]) == 32 ** 3
        np.testing.assert_allclose(m["bbox_center"] - m["bbox_half"], g["bbox_min"], atol=2e-4)
        np.testing.assert_allclose(m["bbox_center"] + m["bbox_half"], g["bbox_max"], atol=2e-4)
        # lists ascend by triangle index inside every cell (model.rs:152,179)
        st = np.concatenate([[0], m["offsets"]]).astype(np.int64)
        idx = m["tri_idx"].astype(np.int64)
        d = np.diff(idx)
        boundaries = set(st[1:-1].tolist())
        bad = [i for i in np.nonzero(d <= 0)[0] if (i + 1) not in boundaries]
        assert not bad


def test_faithful_equals_pretransformed(oracle_scene):
    """model.rs:108-122 is a pure function of the vertex: hoisting the transform cannot change a bit."""
    s = oracle_scene("suzanne")
    a = s.render(spp=1, seed=3, width=96, height=96, faithful=True)
    b = s.render(spp=1, seed=3, width=96, height=96, faithful=False)
    assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["tri_id"], b["tri_id"])
    assert a["counters"] == b["counters"]


def test_rgb8_rule(orc):
    """main.rs:35-37: (c*255.0) as u8 truncates toward zero and saturates; NaN -> 0."""
    v = np.array([[0.0, 0.5, 1.0], [2.0, -1.0, float("nan")], [0.999, 1e-9, 96.0]])
    assert orc.to_rgb8(v).tolist() == [[0, 127, 255], [255, 0, 0], [254, 0, 255]]


# ---- 3D-DDA known answers, worked out on paper from core.rs:94-167 -------------------------------
# Model: floor triangle (0,0,0)(32,0,0)(0,0,32), ceiling triangle (0,32,0)(32,32,32)(0,32,32), so that the box is [0,32]^3
# and a cell is exactly 1 x 1 x 1 (every quantity below is a binary fraction: the arithmetic is exact), and the target
# T2 = (10.5,8,6.5)(10.5,8.5,6.5)(10.5,8,7).  Box3::include (box3D.rs:31-45) grows T2's box to y in [7.875, 8.375],
# z in [6.375, 6.875] -- not its bounding box -- so T2 is listed in cells (10,7,6) and (10,8,6).
DDA_TRIS = np.array([[0, 0, 0, 32, 0, 0, 0, 0, 32], [0, 32, 0, 32, 32, 32, 0, 32, 32], [10.5, 8, 6.5, 10.5, 8.5, 6.5, 10.5, 8, 7]], dtype=np.float32)
DDA_NRMS = np.tile(np.array([-1, 0, 0], dtype=np.float32), (3, 3))
# Ray A: origin (-1, 5.25, 5.25), direction (1, 0.25, 0.125).  Slab test: tmin = 1, entry point (0, 5.5, 5.375), cell (0,5,5);
# t_max = (1, 2, 5), t_delta = (1, 4, 8).  Steps (core.rs:151-167; x only if tx < ty AND tx < tz, else y if ty < tz, else z):
DDA_CELLS_A = [(0, 5, 5), (1, 5, 5),            # tx 1 -> 2
               (1, 6, 5),                       # tx = ty = 2: not x; ty 2 < tz 5: y, ty -> 6
               (2, 6, 5), (3, 6, 5), (4, 6, 5),  # tx 2, 3, 4 -> 5
               (4, 6, 6),                       # tx = tz = 5: not x (strict <); ty 6 < tz 5 is false: z, tz -> 13
               (5, 6, 6),                       # tx 5 -> 6
               (5, 7, 6),                       # tx = ty = 6: y, ty -> 10
               (6, 7, 6), (7, 7, 6), (8, 7, 6), (9, 7, 6),  # tx 6..9 -> 10
               (9, 8, 6),                       # tx = ty = 10: y, ty -> 14
               (10, 8, 6)]                      # tx 10 -> 11; T2 is listed here: hit at x = 10.5, t = 11.5, u = 0.25, v = 0.375
RAY_A = [-1.0, 5.25, 5.25, 1.0, 0.25, 0.125]
# Ray B (quirk Q6): origin (-2, 5.5, 5.5), direction (1, 0, 0).  step = -1 for the zero components (core.rs:102-104),
# t_max.y = t_max.z = (5 - 5.5)/0 = -inf, so `tx < ty` is false, `ty < tz` is false, and EVERY step goes along -z: the walk
# leaves the grid through z = 0 after six cells and never reaches the triangle the ray geometrically hits at (10.5, 5.5, 5.5).
DDA_TRIS_B = np.array([[0, 0, 0, 32, 0, 0, 0, 0, 32], [0, 32, 0, 32, 32, 32, 0, 32, 32], [10.5, 5, 5, 10.5, 7, 5, 10.5, 5, 7]], dtype=np.float32)
RAY_B = [-2.0, 5.5, 5.5, 1.0, 0.0, 0.0]
DDA_CELLS_B = [(0, 5, 5), (0, 5, 4), (0, 5, 3), (0, 5, 2), (0, 5, 1), (0, 5, 0)]


def _dda_models(orc, tris):
    import py_restatement as P
    o = orc.OracleScene(text="size 8 8")
    o.add_triangles(tris, DDA_NRMS)
    m = o.model(0)
    pm = P.Model(m["bbox_center"].tolist(), m["bbox_half"].tolist(), 32, m["pos"].tolist(), m["nrm"].tolist(),
                 [int(v) for v in m["offsets"]], [int(v) for v in m["tri_idx"]])
    return o, m, pm


def test_dda_known_answer_cell_sequence(orc):
    o, m, pm = _dda_models(orc, DDA_TRIS)
    assert m["bbox_center"].tolist() == [16.0] * 3 and m["bbox_half"].tolist() == [16.0] * 3
    off = np.concatenate([[0], m["offsets"]]).astype(int)
    cell = lambda x, y, z: z * 1024 + y * 32 + x
    for c in ((10, 7, 6), (10, 8, 6)):                                # Box3::include's quirk: two cells, not one
        assert m["tri_idx"][off[cell(*c)]: off[cell(*c) + 1]].tolist() == [2]
    assert off[cell(10, 9, 6) + 1] == off[cell(10, 9, 6)] and int((m["tri_idx"] == 2).sum()) == 2
    visited = []
    t, nn, tri = pm.ray_intersect(tuple(RAY_A[:3]), tuple(RAY_A[3:]), visited)
    assert visited == DDA_CELLS_A
    assert (t, tri, nn) == (11.5, 2, (-1.0, 0.0, 0.0))
    r = o.trace(np.array([RAY_A]))
    assert (r["obj_id"][0], r["tri_id"][0], r["t"][0], r["normal"][0].tolist()) == (0, 2, 11.5, [-1.0, 0.0, 0.0])
    c = o.trace_counters(RAY_A)
    assert (c["bbox_tests"], c["cells_visited"], c["tri_tests"], c["model_hits"]) == (1, len(DDA_CELLS_A), 1, 1)


def test_dda_known_answer_axis_parallel_ray_walks_the_wrong_way(orc):
    o, m, pm = _dda_models(orc, DDA_TRIS_B)
    import py_restatement as P
    assert P.tri_intersect(*pm.tri(2), tuple(RAY_B[:3]), tuple(RAY_B[3:])) == 12.5   # the triangle test itself would accept
    visited = []
    assert pm.ray_intersect(tuple(RAY_B[:3]), tuple(RAY_B[3:]), visited) is None
    assert visited == DDA_CELLS_B
    assert o.trace(np.array([RAY_B]))["obj_id"][0] == 0xFFFFFFFF
    c = o.trace_counters(RAY_B)
    assert (c["bbox_tests"], c["cells_visited"], c["tri_tests"], c["model_hits"]) == (1, 6, 0, 0)
