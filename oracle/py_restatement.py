"""Second, independent restatement of the reference's intersection path in plain Python floats
(IEEE f64, no contraction).  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Written from the reference's Rust lines (cited below, relative to /root/reference/src/raytracing/), NOT from
oracle/rt_oracle.hpp: its only purpose is to lower the transcription risk of the C++ oracle (SURVEY 8c,
"honest limitation").  tests/test_oracle_cross.py requires both to agree bit for bit on object id, triangle id,
t and normal for thousands of rays per scene.  It takes the flattened geometry (what Model::get_triangle /
get_triangle_normals return, model.rs:108-139, and the grid arrays, model.rs:205-210) as input, so it restates
rows a5-a14 of SURVEY 8(a): hit, Solid::hit, ray_intersect, Triangle::intersect, normal_on_point, Box3::intersect,
the ModelGrid accessors and the Vec3 operators.
"""
import math

EPS = 1e-5                      # core.rs:10
F64_EPS = 2.220446049250313e-16  # f64::EPSILON, model.rs:51
INF = float("inf")


def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])          # vec3.rs:51-57
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])          # vec3.rs:29-35
def mul(a, s): return (a[0] * s, a[1] * s, a[2] * s)                   # vec3.rs:64-70
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]          # vec3.rs:178-180
def cross(a, b):                                                        # vec3.rs:183-189
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def fdiv(a, b):
    """IEEE f64 division including x/0 and 0/0 (Python raises instead)."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float("nan")
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -INF if neg else INF


def normalize(a):                                                       # vec3.rs:213-215
    l = math.sqrt(dot(a, a))
    return (fdiv(a[0], l), fdiv(a[1], l), fdiv(a[2], l))


def at(o, d, t): return add(o, mul(d, t))                              # ray.rs:10-12


def rmin(a, b):  # Rust f64::min: the non-NaN operand when one is NaN
    if a != a: return b
    if b != b: return a
    return a if a < b else b


def rmax(a, b):
    if a != a: return b
    if b != b: return a
    return a if a > b else b


def as_u32(v):  # Rust `as u32`: saturating, NaN -> 0
    if not (v > 0.0): return 0
    if v >= 4294967295.0: return 4294967295
    return int(v)


def box_intersect(center, half, o, d):                                  # box3D.rs:64-93
    bmin, bmax = sub(center, half), add(center, half)
    df = (fdiv(1.0, d[0]), fdiv(1.0, d[1]), fdiv(1.0, d[2]))
    lo, hi = sub(bmin, o), sub(bmax, o)
    t1, t2 = lo[0] * df[0], hi[0] * df[0]
    t3, t4 = lo[1] * df[1], hi[1] * df[1]
    t5, t6 = lo[2] * df[2], hi[2] * df[2]
    tmin = rmax(rmax(rmin(t1, t2), rmin(t3, t4)), rmin(t5, t6))
    tmax = rmin(rmin(rmax(t1, t2), rmax(t3, t4)), rmax(t5, t6))
    if tmax < 0.0: return None
    if tmin > tmax: return None
    return tmin


def tri_intersect(v0, v1, v2, o, d):                                    # model.rs:43-74
    e1, e2 = sub(v1, v0), sub(v2, v0)
    p = cross(d, e2)
    det = dot(e1, p)
    if abs(det) < F64_EPS: return None
    inv = fdiv(1.0, det)
    tv = sub(o, v0)
    u = dot(tv, p) * inv
    if u < 0.0 or u > 1.0: return None
    q = cross(tv, e1)
    v = dot(d, q) * inv
    if v < 0.0 or u + v > 1.0: return None
    t = dot(e2, q) * inv
    if t < 0.0: return None
    return t


def normal_on_point(v0, v1, v2, n0, n1, n2, p):                         # model.rs:18-39
    s0, s1, s2 = sub(v1, v0), sub(v2, v0), sub(p, v0)
    d00, d01, d11, d20, d21 = dot(s0, s0), dot(s0, s1), dot(s1, s1), dot(s2, s0), dot(s2, s1)
    den = d00 * d11 - d01 * d01
    v = fdiv(d11 * d20 - d01 * d21, den)
    w = fdiv(d00 * d21 - d01 * d20, den)
    u = 1.0 - v - w
    return normalize(add(add(mul(n0, u), mul(n1, v)), mul(n2, w)))


class Model:
    """pos/nrm: n_tris x 9 nested lists; offsets: inclusive prefix sums (model.rs:185-193); tri_idx: list."""

    def __init__(self, center, half, n, pos, nrm, offsets, tri_idx):
        self.center, self.half, self.n = tuple(center), tuple(half), n
        self.pos, self.nrm, self.offsets, self.tri_idx = pos, nrm, offsets, tri_idx

    def tri(self, i):
        p = self.pos[i]
        return (p[0], p[1], p[2]), (p[3], p[4], p[5]), (p[6], p[7], p[8])

    def ray_intersect(self, o, d, visited=None):                        # core.rs:94-178 (visited: list that receives the cells walked)
        tb = box_intersect(self.center, self.half, o, d)
        if tb is None: return None
        P0 = at(o, d, tb)
        n = self.n
        bmin, bmax = sub(self.center, self.half), add(self.center, self.half)
        cs = (fdiv(bmax[0] - bmin[0], n), fdiv(bmax[1] - bmin[1], n), fdiv(bmax[2] - bmin[2], n))   # model.rs:243-245
        rel = sub(P0, bmin)
        idx = [min(as_u32(fdiv(rel[c], cs[c])), n - 1) for c in range(3)]                          # model.rs:267-280
        step = [1 if d[c] > 0.0 else -1 for c in range(3)]                                          # core.rs:102-104
        hc = (self.half[0] / n, self.half[1] / n, self.half[2] / n)                                 # model.rs:232
        tmax = [0.0] * 3
        for c in range(3):                                                                          # core.rs:107-119
            nxt = idx[c] + max(step[c], 0)
            cell_min = (bmin[c] + (hc[c] * 2.0) * (float(nxt) + 0.5)) - hc[c]                      # model.rs:231-240, box3D.rs:56
            tmax[c] = fdiv(cell_min - P0[c], d[c])
        tdelta = [fdiv(float(step[c]) * cs[c], d[c]) for c in range(3)]                             # core.rs:122-125
        best = None
        while 0 <= idx[0] < n and 0 <= idx[1] < n and 0 <= idx[2] < n:                              # core.rs:129-135
            cell = idx[2] * n * n + idx[1] * n + idx[0]                                             # model.rs:219
            if visited is not None: visited.append(tuple(idx))
            start = self.offsets[cell - 1] if cell > 0 else 0
            for k in range(start, self.offsets[cell]):
                ti = self.tri_idx[k]
                t = tri_intersect(*self.tri(ti), o, d)
                if t is not None and (best is None or t < best[1]):                                 # core.rs:140-142
                    best = (ti, t)
            if best is not None: break                                                              # core.rs:147-149
            if tmax[0] < tmax[1]:                                                                   # core.rs:151-167
                c = 0 if tmax[0] < tmax[2] else 2
            else:
                c = 1 if tmax[1] < tmax[2] else 2
            idx[c] += step[c]
            tmax[c] += tdelta[c]
        if best is None: return None
        ti, t = best
        q = self.nrm[ti]
        nn = normal_on_point(*self.tri(ti), (q[0], q[1], q[2]), (q[3], q[4], q[5]), (q[6], q[7], q[8]), at(o, d, t))
        return t, nn, ti


def hit(objects, o, d):
    """objects: list in scene-file order of ('sphere', c, r) | ('plane', n, dist) | ('model', Model).   core.rs:68-92"""
    closest, res = INF, None
    for oi, ob in enumerate(objects):
        if ob[0] == "sphere":                                           # core.rs:183-197
            oc = sub(o, ob[1])
            a = dot(d, d)
            b = 2.0 * dot(d, oc)
            c = dot(oc, oc) - ob[2] * ob[2]
            disc = b * b - 4.0 * a * c
            if disc < 0.0: continue
            t = fdiv(-b - math.sqrt(disc), 2.0 * a)
            nn, tri = normalize(sub(at(o, d, t), ob[1])), None
        elif ob[0] == "plane":                                          # core.rs:198-210
            dv = dot(ob[1], d)
            center = mul(ob[1], ob[2])
            if abs(dv) < EPS: continue
            t = fdiv(dot(sub(center, o), ob[1]), dv)
            if t < EPS: continue
            nn, tri = ob[1], None
        else:
            r = ob[1].ray_intersect(o, d)
            if r is None: continue
            t, nn, tri = r
        if t <= EPS: continue                                           # core.rs:75-77
        if t < closest:                                                 # core.rs:79-83
            closest, res = t, (oi, tri, nn)
    if res is None: return None
    return res[0], res[1], closest, res[2]
