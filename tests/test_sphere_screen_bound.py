"""The packed-f32 sphere screen of the shade kernel (rt_front.h analytic(), MANY) restated in numpy and
checked against the reference's f64 sphere test (core.rs:183-195) on rays built to graze: the screen may drop a sphere only
if the f64 test finds no hit with t > eps.  CPU only -- the device side of the same statement is
tests/test_parity_gpu.py::test_sphere_screen_keeps_every_hit."""
import numpy as np

EPS = 1e-5  # core.rs:10
f32 = np.float32


def fma32(a, b, c):
    # f32 operands: the product is exact in f64, the sum rounds once in f64 and once more to f32 (a double rounding that can
    # differ from a true fma by one f32 ulp in rare ties: far inside the margin's slack)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def screen_keeps(o, d, c, r, lc, r2max):
    """rt_front.h: the survivors' condition, per (ray, sphere) pair; lc / r2max are the scene's bounds as render.hip forms them."""
    of, df, cf = o.astype(f32), d.astype(f32), c.astype(f32)
    r2f = (r * r).astype(f32)
    a = (d * d).sum(-1)  # (the kernel rounds its f64 d.d to f32)
    af = a.astype(f32)
    lo, dm = np.abs(of).max(-1), np.abs(df).max(-1)
    sc = (lo + lc).astype(f32)
    sc2 = (sc * sc + r2max).astype(f32)
    md = (f32(2.0 ** -14) * (dm * dm) * sc2).astype(f32)
    mb = (f32(2.0 ** -17) * sc * dm).astype(f32)
    tame = (lo < 1e8) & (dm < 1e8) & (dm > 1e-8) & (sc2 > 1e-12) & (sc2 < 1e17)
    oc = (of - cf).astype(f32)
    bh = fma32(oc[..., 2], df[..., 2], fma32(oc[..., 1], df[..., 1], (oc[..., 0] * df[..., 0]).astype(f32)))
    cc = fma32(oc[..., 2], oc[..., 2], fma32(oc[..., 1], oc[..., 1], fma32(oc[..., 0], oc[..., 0], -r2f)))
    tt = fma32(af, cc, -md)
    dd = fma32(bh, bh, -tt)
    v = np.minimum(dd, (mb - bh).astype(f32))
    return ~tame | ~np.signbit(v)


def reference_hits(o, d, c, r):
    """core.rs:183-195 with core.rs:75: the operations in the reference's order, in f64."""
    oc = o - c
    a = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    b = 2.0 * (d[..., 0] * oc[..., 0] + d[..., 1] * oc[..., 1] + d[..., 2] * oc[..., 2])
    cc = (oc[..., 0] * oc[..., 0] + oc[..., 1] * oc[..., 1] + oc[..., 2] * oc[..., 2]) - r * r
    disc = b * b - 4.0 * a * cc
    with np.errstate(invalid="ignore", divide="ignore"):
        t = (-b - np.sqrt(disc)) / (2.0 * a)
    return ~(disc < 0.0) & (t > EPS)


def grazing_rays(rng, n, c, r, scene_scale):
    """Rays that pass a sphere at distance r (1 + delta), delta from 0 to +-1e-2 with every magnitude in between."""
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = np.cross(nrm, rng.normal(size=(n, 3)))
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    delta = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-17, -2, n)
    delta[: n // 10] = 0.0
    p = c + nrm * (r * (1.0 + delta))[:, None]
    back = 10.0 ** rng.uniform(-3, 1.5, n) * np.maximum(r, scene_scale * 1e-3)
    dlen = 10.0 ** rng.uniform(-3, 3, n)
    dlen[n // 2:] = 1.0  # (the reference normalises most of its rays)
    return p - tan * back[:, None], tan * dlen[:, None]


def test_screen_never_drops_a_reference_hit():
    rng = np.random.default_rng(3)
    worst_kept_fraction = []
    for scene_scale in (1e-4, 1.0, 13.0, 1e3, 1e6):
        for r_rel in (1e-4, 1e-2, 0.05, 1.0, 30.0):
            n = 200_000
            c = rng.uniform(-1, 1, (n, 3)) * scene_scale
            r = np.full(n, r_rel * scene_scale) * 10.0 ** rng.uniform(-0.5, 0.5, n)
            lc = np.nextafter(f32(np.abs(c).max()), f32(np.inf))
            r2max = np.nextafter(f32((r * r).max()), f32(np.inf))
            o, d = grazing_rays(rng, n, c, r, scene_scale)
            hit = reference_hits(o, d, c, r)
            keep = screen_keeps(o, d, c, r, lc, r2max)
            assert hit.sum() > n // 50, "the construction must produce grazing hits"
            assert not (hit & ~keep).any(), f"scale {scene_scale} r {r_rel}: the screen dropped {int((hit & ~keep).sum())} hits"
            # and it is a screen: rays that pass a sphere at twice its radius or more, or that start beyond it, are dropped
            o2 = o + (o - c) / np.linalg.norm(o - c, axis=1, keepdims=True) * (2.0 * r)[:, None] * (1.0 + rng.uniform(0, 3, n))[:, None]
            far = ~screen_keeps(o2, d, c, r, lc, r2max)
            worst_kept_fraction.append(1.0 - far.mean())
            assert not reference_hits(o2, d, c, r)[far].any()
    # (how sharp the screen is depends on r / scene size: the margin is relative to the scene's extent)
    assert np.median(worst_kept_fraction) < 0.2


def test_screen_on_the_lattice_scene_is_sharp():
    """multisphere.scene's geometry (5 x 5 x 5 spheres of radius 0.5, one apart): random rays keep only a few of the 125."""
    rng = np.random.default_rng(4)
    g = np.arange(5.0)
    c = np.stack(np.meshgrid(g - 2.5, g - 2.5, g + 6.0, indexing="ij"), -1).reshape(-1, 3)
    r = np.full(len(c), 0.5)
    lc, r2max = np.nextafter(f32(np.abs(c).max()), f32(np.inf)), np.nextafter(f32(0.25), f32(np.inf))
    n = 4000
    o = rng.uniform(-3, 3, (n, 1, 3)) + np.array([0, 0, 8.0])
    d = rng.normal(size=(n, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    keep = screen_keeps(np.broadcast_to(o, (n, 125, 3)), np.broadcast_to(d, (n, 125, 3)), np.broadcast_to(c, (n, 125, 3)), r, lc, r2max)
    hit = reference_hits(o, d, c, r)
    assert not (hit & ~keep).any()
    assert keep.sum(1).mean() < 4.0 and keep.sum(1).mean() < 1.3 * hit.sum(1).mean() + 0.5
