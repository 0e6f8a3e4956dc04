"""CPU tests of the host side: the C-ABI library loads and exports what include/rt_mi355x.h declares, and the
front end's flattened buffers equal, bit for bit, what the oracle's faithful object graph yields."""
import ctypes
import re

import numpy as np
import pytest

from conftest import ASSETS, SCENES, scene_path


def test_exports_match_header(rtm):
    hdr = open(rtm.INCLUDE_PATH).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rtm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(rtm.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(rtm.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert "gfx950" in rtm.version()


def test_struct_sizes(rtm):
    # layout contract of the header (x86-64 SysV): checked so a Rust #[repr(C)] mirror can rely on it
    assert ctypes.sizeof(rtm.RtmParams) == 40
    assert ctypes.sizeof(rtm.RtmStats) == 72
    assert ctypes.sizeof(rtm.RtmCamera) == 104
    assert ctypes.sizeof(rtm.RtmModel) == 48 + 8 + 16 + 8 + 72 + 72 + 8
    p = rtm.make_params(1200, 1200, row_begin=3, row_stride=8)
    assert rtm.local_rows(p) == 150
    p = rtm.make_params(10, 10, row_begin=9, row_stride=4)
    assert rtm.local_rows(p) == 1


@pytest.mark.parametrize("name", SCENES)
def test_flattened_scene_equals_oracle(name, rtm_scene, oracle_scene):
    f = rtm_scene(name).flat()
    o = oracle_scene(name)
    cam = np.concatenate([f["camera"][k] for k in ("position", "forward", "up", "right")] + [[f["camera"]["zoom"]]])
    assert np.array_equal(cam, o.camera())
    assert f["n_objects"] == o.n_objects and len(f["lights"]) == o.n_lights
    for li in range(o.n_lights):
        assert np.array_equal(f["lights"][li], o.light(li))
    si = pi = mi = 0
    for oi in range(o.n_objects):
        kind, abcd, col, mt = o.object(oi)
        if kind == 0:
            assert np.array_equal(f["spheres"][si], abcd) and f["sph_obj"][si] == oi
            m = f["sph_mat"][si]
            si += 1
        elif kind == 1:
            assert np.array_equal(f["planes"][pi], abcd) and f["pl_obj"][pi] == oi
            m = f["pl_mat"][pi]
            pi += 1
        else:
            fm, om = f["models"][mi], o.model(oi)
            assert fm["obj_idx"] == oi and fm["cells_per_side"] == 32
            # get_triangle / get_triangle_normals (model.rs:108-139) evaluated per access by the oracle
            assert np.array_equal(fm["pos"], om["pos"])
            assert np.array_equal(fm["nrm"], om["nrm"])
            assert np.array_equal(fm["offsets"].astype(np.uint64), om["offsets"])
            assert np.array_equal(fm["tri_idx"].astype(np.uint64), om["tri_idx"])
            assert np.array_equal(fm["bbox_center"], om["bbox_center"])
            assert np.array_equal(fm["bbox_half"], om["bbox_half"])
            m = fm["mat_idx"]
            mi += 1
        assert np.array_equal(f["materials"][m], col) and f["mat_type"][m] == mt


def test_frontend_error_behaviour(rtm):
    """Same messages and 0-based line:column as parser.rs:43-75, 224-256, 415-420, 464-467."""
    with pytest.raises(rtm.RtmError, match=r"unexpected token '' at 2:0") as e:
        rtm.Scene(text="size 10 10\nlight (0,0,0) red\n")
    assert e.value.code == -3
    with pytest.raises(rtm.RtmError, match="expected 'size', getting 'sphere' instead"):
        rtm.Scene(text="sphere (0,0,0) 1 red")
    with pytest.raises(rtm.RtmError, match="cannot interp 'ten' as a float"):
        rtm.Scene(text="size 10 ten")
    with pytest.raises(rtm.RtmError, match="Cannot load model"):
        rtm.Scene(text='size 4 4\nmodel "./nope.obj" white', base_dir="/nonexistent")
    with pytest.raises(rtm.RtmError, match="unexpected token while parsing trasform"):
        rtm.Scene(text='size 4 4\nmodel "./models/monkey.obj" white > shear 2', base_dir=ASSETS)
    with pytest.raises(rtm.RtmError, match="Cannot load model"):  # quads: "Model should be triangulated first"
        rtm.Scene(text='size 4 4\nmodel "./models/cylinder.obj" white', base_dir=ASSETS)
    with pytest.raises(rtm.RtmError, match="Cannot load model"):  # teapot.obj has no vn lines: reference refuses it
        rtm.Scene(text='size 4 4\nmodel "./models/teapot.obj" white', base_dir=ASSETS)
    with pytest.raises(rtm.RtmError) as e:
        rtm.Scene(path="/nonexistent/x.scene")
    assert e.value.code == -4


def test_frontend_tokens(rtm):
    """parser.rs:126-214: comments, '.5' / '-.5' / '+1' numbers, size truncation `as u32` (parser.rs:472-473)."""
    s = rtm.Scene(text="# c\nsize 12.9 7.2 # trailing\ncamera from (+1, -.5, .5)\nsphere (0,0,3) .5 metal:(0.1,0.2,0.3)")
    assert (s.width, s.height) == (12, 7)
    f = s.flat()
    assert f["camera"]["position"].tolist() == [1.0, -0.5, 0.5]
    assert f["camera"]["forward"].tolist() == [0.0, 0.0, 1.0]  # no `to`: looks at position + z (parser.rs:371-375)
    assert f["spheres"][0].tolist() == [0, 0, 3, 0.5] and f["mat_type"][0] == 1
    assert f["materials"][0].tolist() == [0.1, 0.2, 0.3]


def test_teapot_with_synth_normals_matches_oracle(rtm, orc):
    """teapot.obj has no `vn`: NOT loadable by the reference; both sides synthesise geometric normals (stated)."""
    txt = 'size 64 64\nmodel "./models/teapot.obj" white > scale 0.3 > translate (0, -0.5, 6)'
    f = rtm.Scene(text=txt, base_dir=ASSETS, synth_normals=True).flat()["models"][0]
    o = orc.OracleScene(text=txt, base_dir=ASSETS, allow_missing_normals=True).model(0)
    assert f["n_tris"] == 6320
    assert np.array_equal(f["pos"], o["pos"]) and np.array_equal(f["nrm"], o["nrm"])
    assert np.array_equal(f["tri_idx"].astype(np.uint64), o["tri_idx"])


def test_add_triangles_matches_oracle(rtm, orc):
    rng = np.random.default_rng(5)
    c = rng.uniform(-1, 1, (500, 1, 3)) + np.array([0, 0, 5.0])
    pos = (c + rng.uniform(-0.2, 0.2, (500, 3, 3))).astype(np.float32).reshape(500, 9)
    nrm = rng.normal(size=(500, 9)).astype(np.float32)
    a = rtm.Scene(text="size 32 32")
    a.add_triangles(pos, nrm, color=(0.5, 0.25, 1.0), metal=True)
    b = orc.OracleScene(text="size 32 32")
    b.add_triangles(pos, nrm, color=(0.5, 0.25, 1.0), metal=True)
    fa, ob = a.flat(), b.model(0)
    assert np.array_equal(fa["models"][0]["pos"], ob["pos"]) and np.array_equal(fa["models"][0]["nrm"], ob["nrm"])
    assert np.array_equal(fa["models"][0]["offsets"].astype(np.uint64), ob["offsets"])
    assert np.array_equal(fa["models"][0]["tri_idx"].astype(np.uint64), ob["tri_idx"])
    assert fa["mat_type"][0] == 1 and fa["materials"][0].tolist() == [0.5, 0.25, 1.0]


def test_render_without_gpu_fails_loudly(rtm):
    """No CPU fallback: on a box without a GPU the render entry points must raise, never return pixels."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    s = rtm.Scene(text="size 8 8\nsphere (0,0,3) 1 white")
    with pytest.raises(rtm.RtmError):
        s.render(spp=1)


def test_front_scene_limits_guard(rtm):
    """the ray-front path forms `list position * 10` in 32 bits, so scenes are routed to it only up to
    0xFFFFFFFF // 10 list entries over all models (and below 2^27 grid cells); host arithmetic, no device."""
    lim = 0xFFFFFFFF // 10
    assert lim * 10 <= 0xFFFFFFFF < (lim + 1) * 10
    fits = rtm.front_scene_fits
    assert fits(0, 0) and fits(44343, 32768) and fits(lim, 1 << 20)
    assert not fits(lim + 1, 32768)           # the product would wrap
    assert not fits(1 << 32, 32768) and not fits(0xFFFFFFF0, 32768)  # (the old bound let these through)
    assert fits(1000, (1 << 27) - 1) and not fits(1000, 1 << 27)
