"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs and against the committed golden fixtures.

Bars (BASELINE.json north_star): closest-hit primitive ids bit-exact at 1 spp; colour within 1e-4 absolute per
channel at 1 spp; PSNR >= 45 dB at N spp.  The kernel accumulates samples in the reference's order, so these
tests additionally require the f64 framebuffer to be BIT-IDENTICAL and say so when it is not.
"""
import os

import numpy as np
import pytest

from conftest import ASSETS, ROOT, SCENES

pytestmark = pytest.mark.gpu
COLOUR_TOL = 1e-4   # north_star: absolute, per channel, at 1 spp
PSNR_MIN_DB = 45.0  # north_star: at N spp


def psnr(a, b):
    mse = float(np.mean((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


def assert_frames_match(g, c, what):
    assert np.array_equal(g["obj_id"], c["obj_id"]), f"{what}: object ids differ"
    assert np.array_equal(g["tri_id"], c["tri_id"]), f"{what}: triangle ids differ"
    err = float(np.abs(g["rgb"] - c["rgb"]).max())
    assert err <= COLOUR_TOL, f"{what}: max colour error {err}"
    nbits = int((g["rgb"].view(np.uint64) != c["rgb"].view(np.uint64)).sum())
    assert nbits == 0, f"{what}: {nbits} f64 words differ (max abs err {err})"


def test_f64_primitives_are_ieee(rtm_gpu):
    """device /, sqrt, fmin, fmax, unfused a*b+a and the saturating cell cast vs the host (x86-64
    SSE2 is IEEE-754 correctly rounded), on 10^7 random + edge inputs.  Bar: bit-exact."""
    rng = np.random.default_rng(1)
    n = 10_000_000
    a = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
    b = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1e308, 31.0,
                     31.999999999999996, 32.0, 1e-5, 0.001, 3.0, 1 / 3.0])
    ea, eb = np.meshgrid(edge, edge)
    a[: ea.size] = ea.ravel()
    b[: eb.size] = eb.ravel()
    out = rtm_gpu.f64_ops_selftest(a, b)
    with np.errstate(all="ignore"):
        exp = np.stack([a / b, np.sqrt(np.abs(a)), np.fmin(a, b), np.fmax(a, b), a * b + a,
                        np.where(a > 0, np.minimum(np.nan_to_num(np.floor(np.where(a > 0, a, 0)), posinf=31), 31), 0)])
    for k, name in enumerate(("div", "sqrt", "fmin", "fmax", "mul_add_unfused", "cell_cast")):
        g, e = out[k], exp[k]
        both_nan = np.isnan(g) & np.isnan(e)
        same = (g.view(np.uint64) == e.view(np.uint64)) | both_nan
        if name in ("fmin", "fmax"):  # +0/-0 choice is unspecified for min/max (Rust's f64::min too)
            same |= (g == 0) & (e == 0)
        assert same.all(), f"{name}: {int((~same).sum())} of {n} differ, e.g. a={a[~same][0]!r} b={b[~same][0]!r}"


KERNELS = {"front": {}, "wave": dict(kernel_wave=True)}


@pytest.mark.parametrize("kernel", tuple(KERNELS))
@pytest.mark.parametrize("name", SCENES)
def test_full_size_1spp_matches_oracle(name, kernel, rtm_gpu, rtm_scene, oracle_scene):
    """BASELINE configs 1-4 at the reference's own 1200x1200, 1 spp, same seed; both device formulations (the
    ray-front pipeline that is the product path and the wave-scheduled single kernel) must give the oracle's bits and
    its exact work counters."""
    g = rtm_scene(name).render(spp=1, seed=11, stats=True, **KERNELS[kernel])
    c = oracle_scene(name).render(spp=1, seed=11)
    assert g["rgb"].shape == (1200, 1200, 3)
    assert_frames_match(g, c, name)
    assert g["counters"] == c["counters"]
    assert g["counters"]["rays_camera"] == 1200 * 1200


@pytest.mark.parametrize("name", SCENES)
def test_golden_fixtures(name, rtm_gpu, rtm_scene):
    """Committed oracle outputs (tests/golden/make_golden.py): 96x96, 1 spp ids + colour, 4 spp colour."""
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{name}_96.npz"))
    size, seed = int(z["size"]), int(z["seed"])
    g1 = rtm_scene(name).render(spp=1, seed=seed, width=size, height=size, stats=True)
    assert_frames_match(g1, dict(rgb=z["rgb_1spp"], obj_id=z["obj_id"], tri_id=z["tri_id"]), name + " golden")
    assert [g1["counters"][k] for k in rtm_gpu.STAT_NAMES] == z["counters_1spp"].tolist()
    g4 = rtm_scene(name).render(spp=4, seed=seed, width=size, height=size)
    assert psnr(g4["rgb"], z["rgb_4spp"]) >= PSNR_MIN_DB
    assert np.array_equal(g4["rgb"], z["rgb_4spp"])


@pytest.mark.parametrize("name", ("bunny", "multisphere"))
def test_n_spp_psnr(name, rtm_gpu, rtm_scene, oracle_scene):
    """N spp: PSNR >= 45 dB vs the oracle image (north_star); in-order accumulation makes it exact."""
    g = rtm_scene(name).render(spp=20, seed=5, width=200, height=200)
    c = oracle_scene(name).render(spp=20, seed=5, width=200, height=200)
    assert psnr(g["rgb"], c["rgb"]) >= PSNR_MIN_DB
    assert_frames_match(g, c, name + " 20spp")


@pytest.mark.parametrize("kernel", tuple(KERNELS))
def test_kernels_agree_at_n_spp(kernel, rtm_gpu, rtm_scene, oracle_scene):
    """8 spp, 10 bounces, without the stats build (the ray-front path then skips shadow rays that cannot contribute):
    every formulation gives the oracle's bits."""
    c = oracle_scene("bunny").render(spp=8, seed=21, width=160, height=120)
    g = rtm_scene("bunny").render(spp=8, seed=21, width=160, height=120, **KERNELS[kernel])
    assert_frames_match(g, c, "bunny 8spp " + kernel)


def test_front_bands_cull_and_refill_do_not_change_bits(rtm_gpu, rtm_scene, oracle_scene):
    """The ray-front scheduling knobs (rtm_set_tuning): many small bands of rows, ragged last band, culling off,
    eager and lazy refill, stream count, claim size and stealing -- same image, ids and counters."""
    s = rtm_scene("suzanne")
    ref = oracle_scene("suzanne").render(spp=3, seed=4, width=203, height=77)
    try:
        for key, value in (("front_max_paths", 64), ("front_max_paths", 26 * 64 * 3 * 4), ("front_cull", 0), ("front_refill", 1),
                           ("front_refill", 64), ("front_streams", 1), ("front_streams", 4), ("front_streams", 0), ("front_claim", 16), ("front_steal", 512),
                           ("front_steal", 0), ("front_coop", 0), ("front_rehit", 0),
                           ("front_inlane", 0), ("front_inlane", 1), ("front_inlane", 64), ("front_inlane_lights", 0)):
            rtm_gpu.set_tuning(key, value)
            g = s.render(spp=3, seed=4, width=203, height=77, stats=True)
            assert_frames_match(g, ref, f"suzanne {key}={value}")
            assert g["counters"] == ref["counters"]
            g = s.render(spp=3, seed=4, width=203, height=77)
            assert np.array_equal(g["rgb"], ref["rgb"])
    finally:
        rtm_gpu.set_tuning("front_max_paths", 0)
        rtm_gpu.set_tuning("front_cull", 1)
        rtm_gpu.set_tuning("front_refill", 24)
        rtm_gpu.set_tuning("front_streams", 2)
        rtm_gpu.set_tuning("front_claim", 128)
        rtm_gpu.set_tuning("front_steal", 8)
        rtm_gpu.set_tuning("front_coop", 1)
        rtm_gpu.set_tuning("front_rehit", 1)
        rtm_gpu.set_tuning("front_inlane", -1)
        rtm_gpu.set_tuning("front_inlane_lights", 1)
    with pytest.raises(rtm_gpu.RtmError):
        rtm_gpu.set_tuning("no_such_knob", 1)


def test_wide_frames_many_samples_and_packed_path_ids(rtm_gpu, rtm_scene):
    """A path's (pixel of its tile, tile column, tile row of the chunk, sample) travel packed in one word (rt_front.h
    PU_HOME) and the rows of a chunk are bounded so that it fits: 4096 columns x 200 samples leave two bits for the tile
    row (chunks of four bands), 2050 x 33 is ragged in both directions.  The ray-front path must give the bits of the
    single-kernel formulation, which has no such word."""
    s = rtm_scene("simple_sphere")
    for (w, h, spp) in ((4096, 48, 200), (2050, 35, 33), (9, 9, 1)):
        a = s.render(spp=spp, seed=5, width=w, height=h)
        b = s.render(spp=spp, seed=5, width=w, height=h, kernel_wave=True)
        assert np.array_equal(a["rgb"], b["rgb"]), (w, h, spp)
        assert np.array_equal(a["obj_id"], b["obj_id"])


def test_timed_events(rtm_gpu, rtm_scene):
    """rtm_timed_events_collect: one event pair per trace launch of ordinary frames, and none when switched off."""
    s = rtm_scene("suzanne")
    ref = s.render(spp=2, seed=1, width=96, height=64)
    rtm_gpu.set_tuning("front_timed_events", 1)
    try:
        g = s.render(spp=2, seed=1, width=96, height=64)
    finally:
        rtm_gpu.set_tuning("front_timed_events", 0)
    ms, n = rtm_gpu.timed_events_collect()
    assert np.array_equal(g["rgb"], ref["rgb"])
    assert n == 22 and ms > 0.0  # two chunks in flight, 11 trace launches each (10 bounces, one batch of two lights per vertex)
    s.render(spp=2, seed=1, width=96, height=64)
    assert rtm_gpu.timed_events_collect() == (0.0, 0)


def test_profile_entry_point(rtm_gpu, rtm_scene):
    """rtm_profile_render_device (bench.py's roofline source): kernel times add up, and the work it reports is the
    reference's work minus what the product path provably skips (culled shadow rays, entries already tested)."""
    import torch
    s = rtm_scene("bunny")
    w = h = 200
    p = rtm_gpu.make_params(w, h, spp=4, seed=3)
    rgb = torch.empty(h, w, 3, dtype=torch.float64, device="cuda")
    pr = s.profile_render_device(p, rgb.data_ptr())
    ref = s.render(spp=4, seed=3, width=w, height=h, stats=True)
    assert np.array_equal(rgb.cpu().numpy(), ref["rgb"])
    c = ref["counters"]
    ref_rays = c["rays_camera"] + c["rays_bounce"] + c["rays_shadow"]
    assert c["rays_camera"] + c["rays_bounce"] < pr["rays_total"] < ref_rays  # issued: the reference's minus the culled shadow queries
    rtm_gpu.set_tuning("front_cull", 0)
    try:
        assert s.profile_render_device(p, rgb.data_ptr())["rays_total"] == ref_rays
    finally:
        rtm_gpu.set_tuning("front_cull", 1)
    assert 0 < pr["traced_rays"] <= c["bbox_tests"]
    assert 0 < pr["tri_tests"] <= c["tri_tests"] and 0 < pr["cells_visited"] <= c["cells_visited"]
    assert pr["trace_launches"] >= 1 and pr["shade_launches"] >= 2
    assert pr["trace_ms"] > 0 and abs(pr["total_ms"] - (pr["shade_ms"] + pr["trace_ms"] + pr["resolve_ms"])) < 1e-9
    pw = s.profile_render_device(rtm_gpu.make_params(w, h, spp=4, seed=3, kernel_wave=True), rgb.data_ptr())
    assert pw["shade_launches"] == 0 and pw["tri_tests"] == c["tri_tests"]


def test_zero_jitter_known_answer(rtm_gpu, rtm_scene):
    """Hand-derived KAT through the C ABI (see tests/test_oracle_kat.py): centre pixel hits the sphere, corner the plane."""
    g = rtm_scene("simple_sphere").render(spp=1, zero_jitter=True)
    assert g["obj_id"][600, 600] == 0 and g["obj_id"][0, 0] == 1
    assert (g["tri_id"] == rtm_gpu.NO_ID).all()
    t = rtm_scene("simple_sphere").trace(np.array([[0, 0, 0, 0, 0, 1.0]]))
    assert t["t"][0] == 2.0 and t["normal"][0].tolist() == [0.0, 0.0, -1.0]


def test_row_interleave_is_invariant(rtm_gpu, rtm_scene):
    """Multi-GPU split: rendering rows y = g (mod G) separately gives the same bits as one frame,
    for divisible and ragged heights; rtm_deinterleave_device reassembles them."""
    import torch
    s = rtm_scene("bunny")
    for (w, h, G) in ((160, 128, 4), (100, 77, 8), (64, 30, 2)):
        full = s.render(spp=2, seed=9, width=w, height=h)
        blocks = [s.render(spp=2, seed=9, width=w, height=h, row_begin=g, row_stride=G) for g in range(G)]
        for g in range(G):
            assert blocks[g]["rgb"].shape[0] == len(range(g, h, G))
            assert np.array_equal(blocks[g]["rgb"], full["rgb"][g::G])
            assert np.array_equal(blocks[g]["obj_id"], full["obj_id"][g::G])
            assert np.array_equal(blocks[g]["tri_id"], full["tri_id"][g::G])
        packed = torch.from_numpy(np.concatenate([b["rgb"] for b in blocks], axis=0)).cuda()
        out = torch.empty(h, w, 3, dtype=torch.float64, device="cuda")
        rtm_gpu.deinterleave_device(packed.data_ptr(), w, h, G, out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), full["rgb"])


def test_device_entry_points_and_tonemap(rtm_gpu, rtm_scene, orc):
    import torch
    s = rtm_scene("suzanne")
    w = h = 128
    p = rtm_gpu.make_params(w, h, spp=3, seed=2)
    rgb = torch.empty(h, w, 3, dtype=torch.float64, device="cuda")
    obj = torch.empty(h, w, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        st = s.render_device(p, rgb.data_ptr(), d_obj=obj.data_ptr(), stats=True, stream=stream.cuda_stream)
    stream.synchronize()
    host = s.render(spp=3, seed=2, width=w, height=h, stats=True)
    assert np.array_equal(rgb.cpu().numpy(), host["rgb"]) and st == host["counters"]
    rgb8 = torch.empty(h, w, 3, dtype=torch.uint8, device="cuda")
    rtm_gpu.tonemap_device(rgb.data_ptr(), w * h, False, rgb8.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(rgb8.cpu().numpy(), orc.to_rgb8(host["rgb"]))  # main.rs:35-37 exact
    rtm_gpu.tonemap_device(rgb.data_ptr(), w * h, True, rgb8.data_ptr())
    torch.cuda.synchronize()
    d = np.abs(rgb8.cpu().numpy().astype(int) - orc.to_rgb8(host["rgb"], gamma=True).astype(int))
    assert d.max() <= 1  # powf on device vs libm: not bit-pinned, at most one code value (main.rs:42-48)


def test_two_caller_streams_do_not_corrupt_each_other(rtm_gpu, rtm_scene):
    """rtm_render_device is asynchronous and the library keeps ONE work space and one set of internal streams: frames
    queued back to back on different caller streams (no host synchronisation in between) are ordered by the library's
    frame-done event; different sizes make the work-space layouts of the two frames differ, and the second frame is
    large enough to regrow the work space while the first may still be queued."""
    import torch
    s, t = rtm_scene("bunny"), rtm_scene("suzanne")
    pa, pb = rtm_gpu.make_params(96, 64, spp=4, seed=2), rtm_gpu.make_params(320, 200, spp=6, seed=3)
    ref_a = s.render(spp=4, seed=2, width=96, height=64)["rgb"]
    ref_b = t.render(spp=6, seed=3, width=320, height=200)["rgb"]
    for rel in (s, t):  # (scenes are freed before the library lets go of the device, as the header asks)
        rel.release_device()
    rtm_gpu.shutdown()  # drop the work space so that the frames below have to grow it
    rtm_gpu.init(0)
    try:
        for streams in (1, 2):
            rtm_gpu.set_tuning("front_streams", streams)
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            a = [torch.empty(64, 96, 3, dtype=torch.float64, device="cuda") for _ in range(3)]
            b = [torch.empty(200, 320, 3, dtype=torch.float64, device="cuda") for _ in range(3)]
            torch.cuda.synchronize()
            for i in range(3):
                s.render_device(pa, a[i].data_ptr(), stream=sa.cuda_stream)
                t.render_device(pb, b[i].data_ptr(), stream=sb.cuda_stream)
            torch.cuda.synchronize()
            for i in range(3):
                assert np.array_equal(a[i].cpu().numpy(), ref_a), f"streams={streams} frame a{i}"
                assert np.array_equal(b[i].cpu().numpy(), ref_b), f"streams={streams} frame b{i}"
    finally:
        rtm_gpu.set_tuning("front_streams", 2)


def test_edge_cases(rtm_gpu, orc):
    """Empty scene, no lights, ragged tiles, one bounce, single pixel."""
    for txt, kw in (("size 20 13", {}),                                        # nothing to hit: sky (main.rs:127)
                    ("size 33 17\nsphere (0,0,3) 1 red", {}),                   # no lights, ragged 16x16 tiles
                    ("size 1 1\nplane (0,0,1) 4 metal:white\nlight (0,0,0) white", {}),
                    ("size 40 40\nsphere (0,0,3) 1 metal:white\nplane (0,1,0) 1 white\nlight (2,-3,0) orange",
                     dict(max_bounces=1)),
                    ("size 40 40\nsphere (0,0,3) 1 metal:white\nplane (0,1,0) 1 white\nlight (2,-3,0) orange",
                     dict(max_bounces=64, spp=3)),
                    ("size 48 40\nsphere (0,0,4) 1 white\nsphere (1.5,0.5,3) 0.5 metal:cyan\nplane (0,1,0) 1 white\n"
                     "light (2,-3,0) orange\nlight (-2,-3,1) white\nlight (0,-4,4) red\nlight (3,-1,6) green\nlight (0,2,2) blue",
                     dict(spp=2, max_bounces=5))):                              # five lights: three batches of shadow rays
        c = orc.OracleScene(text=txt).render(seed=4, **kw)
        for kernel, sel in KERNELS.items():
            g = rtm_gpu.Scene(text=txt).render(seed=4, stats=True, **kw, **sel)
            assert_frames_match(g, c, txt.splitlines()[0] + " " + kernel)
            assert g["counters"] == c["counters"]
            g = rtm_gpu.Scene(text=txt).render(seed=4, **kw, **sel)  # (no counters: the ray-front path culls)
            assert np.array_equal(g["rgb"], c["rgb"])
    sky = rtm_gpu.Scene(text="size 20 13").render()
    assert (sky["rgb"] == 1.0).all() and (sky["obj_id"] == rtm_gpu.NO_ID).all()


def test_argument_errors(rtm_gpu):
    s = rtm_gpu.Scene(text="size 8 8\nsphere (0,0,3) 1 white")
    for kw in (dict(spp=0), dict(max_bounces=0), dict(max_bounces=65), dict(row_stride=0), dict(row_begin=8)):
        with pytest.raises(rtm_gpu.RtmError) as e:
            s.render(**{**dict(spp=1), **kw})
        assert e.value.code == -1


def _random_scene_text(rng, n_sph, n_pl):
    lines = ["size 64 64", "camera from (0.3, -0.2, -1) to (0, 0, 5)"]
    for _ in range(n_sph):
        c = rng.uniform(-3, 3, 3) + np.array([0, 0, 6])
        lines.append("sphere (%.3f, %.3f, %.3f) %.3f %s" % (*c, rng.uniform(0.2, 1.5), rng.choice(["white", "metal:red", "cyan"])))
    for _ in range(n_pl):
        n = rng.normal(size=3)
        lines.append("plane (%.3f, %.3f, %.3f) %.3f %s" % (*n, rng.uniform(2, 9), rng.choice(["white", "metal:white"])))
    for _ in range(int(rng.integers(0, 4))):
        lines.append("light (%.2f, %.2f, %.2f) %s" % (*rng.uniform(-4, 4, 3), rng.choice(["red", "green", "blue", "white"])))
    return "\n".join(lines)


def _soup(rng, n, centre, spread, s):
    c = rng.uniform(-spread, spread, (n, 1, 3)) + np.asarray(centre)
    pos = (c + rng.uniform(-s, s, (n, 3, 3))).astype(np.float32)
    e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    g = np.cross(e1, e2)
    g /= np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
    return pos.reshape(n, 9), np.repeat(g, 3, axis=0).reshape(n, 9).astype(np.float32)


def test_differential_fuzz_random_scenes(rtm_gpu, orc):
    """random rays vs random small scenes (spheres, planes, triangle soups, several models so that
    object ties and model order matter), oracle vs kernel, ids and t and normal bit-exact; includes axis-parallel
    rays (zero direction components: quirk Q6), origins inside the grid box (Q5) and non-unit directions (Q4)."""
    rng = np.random.default_rng(77)
    for trial in range(6):
        # (trials 4 and 5: more than four spheres -- the shade kernel's packed passes and f32 sphere screen -- together with models)
        txt = _random_scene_text(rng, int(rng.integers(0, 6)) if trial < 4 else int(rng.integers(6, 48)), int(rng.integers(0, 3)))
        a, b = rtm_gpu.Scene(text=txt), orc.OracleScene(text=txt)
        for m in range(int(rng.integers(1, 3))):
            pos, nrm = _soup(rng, int(rng.integers(50, 3000)), (rng.uniform(-1, 1), rng.uniform(-1, 1), 5.0 + m), 1.5,
                             rng.uniform(0.05, 0.6))
            a.add_triangles(pos, nrm, metal=bool(m))
            b.add_triangles(pos, nrm, metal=bool(m))
        n = 40000
        o = rng.uniform(-4, 4, (n, 3)) + np.array([0, 0, 3.0])
        d = rng.normal(size=(n, 3)) * rng.uniform(0.1, 3.0, (n, 1))
        d[: n // 10, rng.integers(0, 3)] = 0.0                    # one zero component
        d[n // 10: n // 8] = np.eye(3)[rng.integers(0, 3, n // 8 - n // 10)] * rng.choice([-1.0, 1.0])  # axis rays
        o[n // 2: n // 2 + 5000] = rng.uniform(-1, 1, (5000, 3)) + np.array([0, 0, 5.0])  # inside the soup's box
        rays = np.concatenate([o, d], axis=1)
        g, c = a.trace(rays), b.trace(rays)
        assert np.array_equal(g["obj_id"], c["obj_id"]), f"trial {trial}"
        assert np.array_equal(g["tri_id"], c["tri_id"]), f"trial {trial}"
        assert np.array_equal(g["t"].view(np.uint64), c["t"].view(np.uint64)), f"trial {trial}"
        nn = (g["normal"].view(np.uint64) != c["normal"].view(np.uint64)) & ~(np.isnan(g["normal"]) & np.isnan(c["normal"]))
        assert not nn.any(), f"trial {trial}: {int(nn.sum())} normal words differ"
        assert (g["obj_id"] != rtm_gpu.NO_ID).mean() > 0.05
        # and a full render of the same random scene
        gr = a.render(spp=2, seed=trial, stats=True)
        cr = b.render(spp=2, seed=trial)
        assert_frames_match(gr, cr, f"fuzz scene {trial}")
        assert gr["counters"] == cr["counters"]
        assert np.array_equal(a.render(spp=2, seed=trial)["rgb"], cr["rgb"])                    # culling on
        assert np.array_equal(a.render(spp=2, seed=trial, kernel_wave=True)["rgb"], cr["rgb"])


def test_grazing_rays_on_long_cells(rtm_gpu, orc):
    """the box screen of long cells (more than 32 entries: each entry's 8-byte box is tested before its record is
    read) must not drop a hit the reference accepts when the ray nearly lies in a triangle's plane (small |det|, u and v
    amplified by 1/det).  Flat patches of many small coplanar triangles -- hundreds per grid cell -- seen edge-on from a
    camera a hair above / below / exactly in their planes, axis-aligned and tilted, with and without jitter: oracle bits."""
    rng = np.random.default_rng(5)

    def patch(n, y, tilt):
        c = np.stack([rng.uniform(-0.6, 0.6, n), np.full(n, y), rng.uniform(4.8, 6.0, n)], axis=1)[:, None, :]
        e = rng.uniform(-0.03, 0.03, (n, 3, 3))
        e[:, :, 1] = 0.0                                  # coplanar: every triangle lies in the plane y = const ...
        pos = c + e
        pos[:, :, 1] += tilt * pos[:, :, 0]               # ... or in a plane tilted about z
        pos = pos.astype(np.float32)
        g = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
        g /= np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
        return pos.reshape(n, 9), np.repeat(g, 3, axis=0).reshape(n, 9).astype(np.float32)

    for cam_y, tilt in ((0.3, 0.0), (0.3000001, 0.0), (0.2999, 0.0), (0.31, 0.0), (0.3, 1e-4), (0.25, 0.03)):
        txt = f"size 96 64\ncamera from (0, {cam_y}, 0) to (0, 0.3, 5.5)\nplane (0, 1.0, 0) 2 white\nlight (0, -3, 5) white\nlight (1, 0.3, 0) red"
        a, b = rtm_gpu.Scene(text=txt), orc.OracleScene(text=txt)
        for y in (0.3, 0.45):
            pos, nrm = patch(60000, y, tilt)
            a.add_triangles(pos, nrm)
            b.add_triangles(pos, nrm)
        lens = np.concatenate([np.diff(np.concatenate([[0], m["offsets"]])) for m in a.flat()["models"]])
        assert (lens > 32).sum() > 500, "the patches must make long cells (the box screen runs only there)"
        for zero_jitter in (True, False):
            g = a.render(spp=2, seed=3, stats=True, zero_jitter=zero_jitter)
            c = b.render(spp=2, seed=3, zero_jitter=zero_jitter)
            assert_frames_match(g, c, f"grazing cam_y={cam_y} tilt={tilt} zero_jitter={zero_jitter}")
            assert g["counters"] == c["counters"]
            assert np.array_equal(a.render(spp=2, seed=3, zero_jitter=zero_jitter)["rgb"], c["rgb"])  # culling and the screen on
        assert (g["obj_id"] >= 1).any()


def test_sphere_screen_keeps_every_hit(rtm_gpu, orc):
    """Scenes with more than four spheres run a packed-f32 screen in front of the f64 sphere tests (rt_front.h analytic(), MANY):
    it may only drop spheres the reference's f64 test rejects.  Random scenes of 5..300 spheres whose radii and distances span
    five orders of magnitude (one, two and three blocks of 128; tails that are not multiples of 32), scenes scaled down and up
    by 1e5, rays tangent to a sphere to the last bit (discriminant exactly 0: core.rs:188 accepts it), a sphere and a camera
    beyond the range the screen's error bound covers (every sphere is kept there): oracle bits, ids and work counters, with the
    screen on and off."""
    rng = np.random.default_rng(11)

    def f(x):  # (the reference's number syntax has no exponent: parser.rs)
        return np.format_float_positional(float(x), precision=17, unique=True, trim="-")

    def scene(n, scale, spread, cam=None):
        lines = ["size 48 40", cam or "camera from (%s, %s, %s) to (0, 0, %s)" % (f(0.3 * scale), f(-0.2 * scale), f(-1 * scale), f(5 * scale))]
        for _ in range(n):
            c = (rng.uniform(-spread, spread, 3) + np.array([0, 0, 6 + spread])) * scale
            r = scale * 10 ** rng.uniform(-3, 0.3) * (spread if rng.random() < 0.3 else 1.0)
            lines.append("sphere (%s, %s, %s) %s %s" % (f(c[0]), f(c[1]), f(c[2]), f(r), rng.choice(["white", "metal:red", "cyan"])))
        lines.append("plane (0, 1, 0) %s white" % f(2 * spread * scale))
        for _ in range(3):
            lines.append("light (%s, %s, %s) white" % tuple(f(v) for v in rng.uniform(-4, 4, 3) * spread * scale))
        return "\n".join(lines)

    cases = [scene(n, sc, sp) for n, sc, sp in ((5, 1, 3), (31, 1, 3), (33, 1, 3), (64, 1, 30), (127, 1e-5, 3), (129, 1e5, 3), (300, 1, 300), (140, 1, 3))]
    # the centre ray of a zero-jitter frame runs along +z exactly: spheres it touches in one point
    tangent = ["size 32 32", "camera from (0, 0, 0) to (0, 0, 1)", "sphere (1, 0, 5) 1 white", "sphere (0, -0.5, 9) 0.5 cyan",
               "sphere (0.25, 0, 14) 0.25 metal:red", "sphere (0, 3, 20) 3 white"]
    tangent += ["sphere (%.6f, %.6f, %.6f) 0.3 white" % tuple(rng.uniform(-20, 20, 3) + np.array([0, 0, 40])) for _ in range(60)]
    tangent += ["light (0, -30, 0) white", "light (5, -3, 2) red"]
    cases.append("\n".join(tangent))
    cases.append(scene(40, 1, 3) + "\nsphere (1000000000, 0, 5) 1 white")                            # a sphere out of range: no screen at all
    cases.append(scene(40, 1, 3, cam="camera from (300000000, 0, 0) to (0, 0, 5)"))                  # rays out of range: every sphere kept
    try:
        for k, txt in enumerate(cases):
            a, b = rtm_gpu.Scene(text=txt), orc.OracleScene(text=txt)
            for zero_jitter in (True, False):
                c = b.render(spp=2, seed=k, zero_jitter=zero_jitter)
                for on in (1, 0):
                    rtm_gpu.set_tuning("front_sph_screen", on)
                    g = a.render(spp=2, seed=k, stats=True, zero_jitter=zero_jitter)
                    assert_frames_match(g, c, f"sphere screen case {k} screen={on} zero_jitter={zero_jitter}")
                    assert g["counters"] == c["counters"]
                    assert np.array_equal(a.render(spp=2, seed=k, zero_jitter=zero_jitter)["rgb"], c["rgb"])
            if k < 8:
                assert (c["obj_id"] != rtm_gpu.NO_ID).mean() > 0.2
    finally:
        rtm_gpu.set_tuning("front_sph_screen", 1)


def test_teapot_and_soup_scene(rtm_gpu, orc):
    """BASELINE config 5 in miniature: a seeded triangle soup plus teapot.obj (no `vn`: geometric normals are
    synthesised on both sides -- NOT reference behaviour, the reference loader refuses this file)."""
    txt = ('size 160 160\nplane (0, 1.0, 0) 2 metal:white\n'
           'model "./models/teapot.obj" white > scale 0.4 > translate (0, 0.5, 6)\nlight (1, -2, 2) white\nlight (-2, -1, 3) orange')
    a = rtm_gpu.Scene(text=txt, base_dir=ASSETS, synth_normals=True)
    b = orc.OracleScene(text=txt, base_dir=ASSETS, allow_missing_normals=True)
    pos, nrm = _soup(np.random.default_rng(3), 20000, (0, -1, 7), 2.0, 0.05)
    a.add_triangles(pos, nrm)
    b.add_triangles(pos, nrm)
    g, c = a.render(spp=2, seed=1, stats=True), b.render(spp=2, seed=1)
    assert_frames_match(g, c, "teapot+soup")
    assert g["counters"] == c["counters"]
    assert (g["obj_id"] == 1).any() and (g["obj_id"] == 2).any()


@pytest.mark.parametrize("inlane", (0, 1, 40))
def test_inlane_advance_does_not_change_bits(inlane, rtm_gpu, rtm_scene, oracle_scene):
    """Paths whose queries are all settled by the analytic primitives advance inside the shade kernel's lane instead of
    waiting a round (rtm_set_tuning("front_inlane", min lanes)): scenes without models then finish in round 0, with three
    and four lights (several light batches per vertex) -- oracle bits and counters whatever the threshold."""
    try:
        rtm_gpu.set_tuning("front_inlane", inlane)
        for name in ("simple_sphere", "multisphere", "bunny"):
            ref = oracle_scene(name).render(spp=3, seed=13, width=150, height=90)
            g = rtm_scene(name).render(spp=3, seed=13, width=150, height=90, stats=True)
            assert_frames_match(g, ref, f"{name} inlane={inlane}")
            assert g["counters"] == ref["counters"]
            assert np.array_equal(rtm_scene(name).render(spp=3, seed=13, width=150, height=90)["rgb"], ref["rgb"])
    finally:
        rtm_gpu.set_tuning("front_inlane", -1)


def test_pool_trace_kernel_gives_the_same_bits(rtm_gpu, rtm_scene, oracle_scene, orc):
    """The trace kernel with its rays parked in per-wave LDS pools (rt_pool.h, rtm_set_tuning("front_pool", 1); measured
    slower than the one-ray-per-lane kernel and therefore not the default): oracle bits, ids and counters on a single
    model, on a model with long cells (suzanne: lists up to 37) and on two models (teapot + soup: the enter phase)."""
    try:
        rtm_gpu.set_tuning("front_pool", 1)
        for name, spp in (("bunny", 4), ("suzanne", 3)):
            ref = oracle_scene(name).render(spp=spp, seed=9, width=240, height=160)
            g = rtm_scene(name).render(spp=spp, seed=9, width=240, height=160, stats=True)
            assert_frames_match(g, ref, name + " pool")
            assert g["counters"] == ref["counters"]
            assert np.array_equal(rtm_scene(name).render(spp=spp, seed=9, width=240, height=160)["rgb"], ref["rgb"])  # culling on
        txt = ('size 160 160\nplane (0, 1.0, 0) 2 metal:white\n'
               'model "./models/teapot.obj" white > scale 0.4 > translate (0, 0.5, 6)\nlight (1, -2, 2) white\nlight (-2, -1, 3) orange')
        a = rtm_gpu.Scene(text=txt, base_dir=ASSETS, synth_normals=True)
        b = orc.OracleScene(text=txt, base_dir=ASSETS, allow_missing_normals=True)
        pos, nrm = _soup(np.random.default_rng(3), 20000, (0, -1, 7), 2.0, 0.05)
        a.add_triangles(pos, nrm)
        b.add_triangles(pos, nrm)
        g, c = a.render(spp=2, seed=1, stats=True), b.render(spp=2, seed=1)
        assert_frames_match(g, c, "teapot+soup pool")
        assert g["counters"] == c["counters"]
        assert np.array_equal(a.render(spp=2, seed=1)["rgb"], c["rgb"])
    finally:
        rtm_gpu.set_tuning("front_pool", 0)


def test_full_size_20spp_properties(rtm_gpu, rtm_scene):
    """BASELINE's headline size (bunny 1200x1200, 20 spp) is too slow for the CPU oracle inside a test, so it is
    checked through size-independent properties: determinism (same seed -> same bits), seed sensitivity, the
    camera-ray count, 1-spp ids embedded in the 20-spp frame, and row-split invariance on a row subset."""
    s = rtm_scene("bunny")
    a = s.render(spp=20, seed=0, stats=True)
    b = s.render(spp=20, seed=0)
    assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["tri_id"], b["tri_id"])
    c = a["counters"]
    assert c["rays_camera"] == 1200 * 1200 * 20
    assert c["rays_shadow"] % 2 == 0  # two lights: every hit issues exactly two shadow queries (main.rs:78-86)
    one = s.render(spp=1, seed=0)
    assert np.array_equal(one["obj_id"], a["obj_id"]) and np.array_equal(one["tri_id"], a["tri_id"])
    assert np.isfinite(a["rgb"]).all() and a["rgb"].min() >= 0.0
    part = s.render(spp=20, seed=0, row_begin=5, row_stride=64)
    assert np.array_equal(part["rgb"], a["rgb"][5::64])
    assert not np.array_equal(s.render(spp=20, seed=1, row_begin=5, row_stride=64)["rgb"], part["rgb"])


def test_dda_known_answers_on_the_device(rtm_gpu):
    """The hand-derived DDA cases of tests/test_oracle_kat.py through the C ABI: ray A reaches the target after fifteen
    cells (t = 11.5, triangle 2); the axis-parallel ray B walks along -z (quirk Q6) and misses the triangle in its way."""
    from test_oracle_kat import DDA_NRMS, DDA_TRIS, DDA_TRIS_B, RAY_A, RAY_B
    for tris, ray, expect in ((DDA_TRIS, RAY_A, (0, 2, 11.5)), (DDA_TRIS_B, RAY_B, (rtm_gpu.NO_ID, rtm_gpu.NO_ID, np.inf))):
        s = rtm_gpu.Scene(text="size 8 8")
        s.add_triangles(tris, DDA_NRMS)
        r = s.trace(np.array([ray]))
        assert (r["obj_id"][0], r["tri_id"][0], r["t"][0]) == expect
