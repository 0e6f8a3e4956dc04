"""The C++ oracle against the independent pure-Python restatement (oracle/py_restatement.py): object id,
triangle id, t and normal must agree BIT FOR BIT on camera rays, shadow-like rays from surface points and random
rays (axis-parallel ones included) of every named scene."""
import math

import numpy as np
import pytest

from conftest import SCENES

NO_ID = 0xFFFFFFFF
N_RAYS = {"simple_sphere": 40000, "multisphere": 6000, "suzanne": 30000, "bunny": 30000}


def build_py_scene(o):
    import py_restatement as P
    objs = []
    for oi in range(o.n_objects):
        kind, abcd, _, _ = o.object(oi)
        if kind == 0:
            objs.append(("sphere", tuple(abcd[:3].tolist()), float(abcd[3])))
        elif kind == 1:
            objs.append(("plane", tuple(abcd[:3].tolist()), float(abcd[3])))
        else:
            m = o.model(oi)
            objs.append(("model", P.Model(m["bbox_center"].tolist(), m["bbox_half"].tolist(), m["cells_per_side"],
                                          m["pos"].tolist(), m["nrm"].tolist(), [int(v) for v in m["offsets"]],
                                          [int(v) for v in m["tri_idx"]])))
    return objs


@pytest.mark.parametrize("name", SCENES)
def test_cpp_oracle_equals_python_restatement(name, oracle_scene):
    import py_restatement as P
    o = oracle_scene(name)
    objs = build_py_scene(o)
    rng = np.random.default_rng(hash(name) % 1000)
    n = N_RAYS[name]
    cam = o.shoot(rng.uniform(-0.5, 0.5, (n // 2, 2)))
    first = o.trace(cam)
    # secondary rays: from the camera rays' hit points toward random points (shadow-like), incl. self-hit cases (Q12)
    ok = first["obj_id"] != NO_ID
    pts = cam[ok, :3] + cam[ok, 3:] * first["t"][ok, None]
    k = min(len(pts), n // 4)
    sec = np.concatenate([pts[:k], rng.normal(size=(k, 3))], axis=1)
    rnd = np.concatenate([rng.uniform(-3, 3, (n // 4, 3)) + np.array([0, 0, 4.0]), rng.normal(size=(n // 4, 3))], axis=1)
    rnd[:20, 3] = 0.0      # zero direction components (Q6)
    rnd[20:30, 3:5] = 0.0
    rays = np.concatenate([cam, sec, rnd], axis=0)
    c = o.trace(rays)
    mism = 0
    for i, r in enumerate(rays.tolist()):
        h = P.hit(objs, tuple(r[:3]), tuple(r[3:]))
        if h is None:
            assert c["obj_id"][i] == NO_ID, i
            continue
        oi, tri, t, nn = h
        assert c["obj_id"][i] == oi, i
        assert c["tri_id"][i] == (NO_ID if tri is None else tri), i
        assert np.float64(t).view(np.uint64) == c["t"][i].view(np.uint64), (i, t, c["t"][i])
        for a, b in zip(nn, c["normal"][i].tolist()):
            assert (a == b) or (math.isnan(a) and math.isnan(b)), (i, nn, c["normal"][i])
    assert (c["obj_id"] != NO_ID).mean() > 0.05
