"""Second, independent restatement of the REST of the reference's program path in plain Python floats: scene parser,
OBJ loading, Mat4, Camera, Model::new, create_grid, the pixel loop, cast and get_bounce_direction.
TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ imports it).

Like oracle/py_restatement.py (whose `hit` it calls for the intersection path) it is written from the reference's
Rust lines, cited below relative to /root/reference/src/, NOT from oracle/rt_oracle.hpp: tests/test_oracle_cross.py
requires the C++ oracle to reproduce, bit for bit, the flattened scenes this module builds from the scene and OBJ
files (positions, normals, boxes, offset_array, triangle_indices, camera) and the colour of >= 10^4 samples per
named scene.  Two things here are NOT the reference's: the random numbers (the reference's thread_rng is OS-seeded;
both oracles use Philox4x32-10 with the stream layout of DESIGN.md section 2, restated here from the Random123
specification) and obj-rs (a crates.io dependency, absent from /root/reference: restated from its published behaviour,
see load_obj).
"""
import math
import os
from fractions import Fraction

import numpy as np

import py_restatement as P
from py_restatement import add, cross, dot, fdiv, mul, normalize, sub

INF = float("inf")


# ---------------------------------------------------------------------------------------------- raytracing/math/mat4.rs
def mat_identity():                                                    # mat4.rs:9-15
    return [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]


def mat_scale(f):                                                      # mat4.rs:17-24
    return [f, 0.0, 0.0, 0.0, 0.0, f, 0.0, 0.0, 0.0, 0.0, f, 0.0, 0.0, 0.0, 0.0, 1.0]


def mat_translate(o):                                                  # mat4.rs:26-33
    return [1.0, 0.0, 0.0, o[0], 0.0, 1.0, 0.0, o[1], 0.0, 0.0, 1.0, o[2], 0.0, 0.0, 0.0, 1.0]


def mat_rotate(axis, angle):                                           # mat4.rs:35-60 (not a true rotation: terms kept as written)
    ux, uy, uz = normalize(axis)
    c, s = math.cos(angle), math.sin(angle)
    return [c + (ux * ux) * (1.0 - c), ux * uy * (1.0 - c) - uz * s, ux * uz * (1.0 - c) - uy * s, 0.0,
            ux * uy * (1.0 - c) - ux * s, c + (uy * uy) * (1.0 - c), uz * uy * (1.0 - c) - ux * s, 0.0,
            ux * uz * (1.0 - c) - uy * s, uz * uy * (1.0 - c) + ux * s, c + uz * uz * (1.0 - c), 0.0,
            0.0, 0.0, 0.0, 1.0]


def _det3(m):                                                          # mat4.rs:63-67
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6])


def _sub_matrix(m, row_rm, col_rm):                                    # mat4.rs:70-86
    out, idx = [0.0] * 16, 0
    for row in range(4):
        if row == row_rm:
            continue
        for col in range(4):
            if col == col_rm:
                continue
            out[idx] = m[row * 4 + col]
            idx += 1
    return out


def _powi_m1(i):                                                       # (-1.0f64).powi(i)
    return 1.0 if i % 2 == 0 else -1.0


def mat_determinant(m):                                                # mat4.rs:89-96
    det = 0.0
    for i in range(4):
        det += _powi_m1(i) * m[i] * _det3(_sub_matrix(m, 0, i))
    return det


def mat_inverse(m):                                                    # mat4.rs:98-121
    det = mat_determinant(m)
    if det == 0.0:
        return None
    cof = [0.0] * 16
    for row in range(4):
        for col in range(4):
            cof[col * 4 + row] = _powi_m1(row + col) * _det3(_sub_matrix(m, row, col))
    return [fdiv(c, det) for c in cof]


def mat_transpose(m):                                                  # mat4.rs:123-144
    return [m[0], m[4], m[8], m[12], m[1], m[5], m[9], m[13], m[2], m[6], m[10], m[14], m[3], m[7], m[11], m[15]]


def mat_then(s, o):                                                    # mat4.rs:146-222: other * self, sums left to right
    out = [0.0] * 16
    for r in range(4):
        for c in range(4):
            out[r * 4 + c] = o[r * 4] * s[c] + o[r * 4 + 1] * s[4 + c] + o[r * 4 + 2] * s[8 + c] + o[r * 4 + 3] * s[12 + c]
    return out


def mat_apply(m, v):                                                   # mat4.rs:224-230
    x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
    y = m[4] * v[0] + m[5] * v[1] + m[6] * v[2] + m[7]
    z = m[8] * v[0] + m[9] * v[1] + m[10] * v[2] + m[11]
    w = m[12] * v[0] + m[13] * v[1] + m[14] * v[2] + m[15]
    return (fdiv(x, w), fdiv(y, w), fdiv(z, w))


# ------------------------------------------------------------------------------------------------ raytracing/camera.rs
class Camera:
    def __init__(self, position):                                      # camera.rs:13-21
        self.up, self.forward, self.right, self.position, self.zoom = (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0), position, 1.0

    @staticmethod
    def look_at(position, point):                                      # camera.rs:30-43
        cam = Camera(position)
        world_up = (0.0, 1.0, 0.0)
        cam.forward = normalize(sub(point, position))
        cam.right = normalize(cross(world_up, cam.forward))
        cam.up = normalize(cross(cam.forward, cam.right))
        return cam

    def shoot_to(self, u, v):                                          # camera.rs:46-52
        d = add(add(mul(self.forward, self.zoom), mul(self.up, v)), mul(self.right, u))
        return self.position, normalize(d)


# ---------------------------------------------------------------------------------- obj-rs 0.7.1 load_obj (Cargo.lock:633)
def _parse_f32(tok):
    """Decimal -> f32, correctly rounded (what Rust's str::parse::<f32> does), widened to f64 (vec3.rs:18-22)."""
    exact = Fraction(tok)
    c = np.float32(float(tok))
    cands = {float(c), float(np.nextafter(c, np.float32(np.inf))), float(np.nextafter(c, np.float32(-np.inf)))}
    best = min(cands, key=lambda x: (abs(Fraction(x) - exact), int(np.float32(x).view(np.uint32)) & 1))
    return best


def load_obj(path):
    """obj-rs `load_obj::<Vertex, _, u16>` as the reference uses it (core.rs:57-58): `v` / `vn` parsed as f32; every face
    must be a triangle whose corners carry a normal index (`p//n` or `p/t/n`); vertices are the distinct (p, n) pairs in
    order of first appearance; indices are u16.  Returns (vertices [(pos, nrm)], indices)."""
    pos, nrm, verts, index_of, indices = [], [], [], {}, []
    for line in open(path):
        w = line.split()
        if not w or w[0].startswith("#"):
            continue
        if w[0] == "v":
            pos.append(tuple(_parse_f32(x) for x in w[1:4]))
        elif w[0] == "vn":
            nrm.append(tuple(_parse_f32(x) for x in w[1:4]))
        elif w[0] == "f":
            if len(w) != 4:
                raise ValueError("Model should be triangulated first to be loaded properly")
            for corner in w[1:]:
                parts = corner.split("/")
                if len(parts) != 3 or not parts[2]:
                    raise ValueError("face corner without a normal index")
                pi, ni = int(parts[0]), int(parts[2])
                key = (pi - 1 if pi > 0 else len(pos) + pi, ni - 1 if ni > 0 else len(nrm) + ni)
                if key not in index_of:
                    index_of[key] = len(verts)
                    verts.append((pos[key[0]], nrm[key[1]]))
                if index_of[key] > 0xFFFF:
                    raise ValueError("vertex index does not fit u16")
                indices.append(index_of[key])
    return verts, indices


# ------------------------------------------------------------------------------------------------ raytracing/model.rs
def box_include(center, half, point):                                  # box3D.rs:31-45 (quirk Q9: the updated half extent is used)
    center, half = list(center), list(half)
    dist = sub(point, center)
    for c in range(3):
        if abs(dist[c]) > half[c]:
            half[c] = (abs(dist[c]) + half[c]) / 2.0
            center[c] += (dist[c] - half[c] * math.copysign(1.0, dist[c])) / 2.0
    return tuple(center), tuple(half)


def _as_usize(v):                                                      # `as usize`: saturating, NaN -> 0
    if not (v > 0.0):
        return 0
    if v >= 18446744073709551615.0:
        return 18446744073709551615
    return int(v)


class Model:
    def __init__(self, verts, indices, trasform):                      # model.rs:91-103
        self.verts, self.indices, self.trasform = verts, indices, trasform
        mx, mn = [-INF] * 3, [INF] * 3                                 # calculate_bounding_box, model.rs:296-323
        for p, _ in verts:
            v = mat_apply(trasform, p)
            for c in range(3):
                if mx[c] < v[c]:
                    mx[c] = v[c]
                if mn[c] > v[c]:
                    mn[c] = v[c]
        self.center = tuple((mn[c] + mx[c]) * 0.5 for c in range(3))  # Box3::from_min_max, box3D.rs:24-29
        self.half = tuple((mx[c] - mn[c]) * 0.5 for c in range(3))
        inv = mat_inverse(trasform)
        if inv is None:
            raise ValueError("Error: the provided matrix to transform the model is not invertible")
        self.normal_trasform = mat_transpose(inv)

    def n_triangles(self):
        return len(self.indices) // 3

    def get_triangle(self, i):                                         # model.rs:108-122
        return tuple(mat_apply(self.trasform, self.verts[self.indices[i * 3 + k]][0]) for k in range(3))

    def get_triangle_normals(self, i):                                 # model.rs:125-139
        return tuple(normalize(mat_apply(self.normal_trasform, self.verts[self.indices[i * 3 + k]][1])) for k in range(3))

    def create_grid(self, n):                                          # model.rs:147-201
        cells = [[] for _ in range(n * n * n)]
        bmin, bmax = sub(self.center, self.half), add(self.center, self.half)       # box3D.rs:55-61
        cs = (fdiv(bmax[0] - bmin[0], n), fdiv(bmax[1] - bmin[1], n), fdiv(bmax[2] - bmin[2], n))
        for i in range(self.n_triangles()):
            v0, v1, v2 = self.get_triangle(i)
            center, half = v0, (0.0, 0.0, 0.0)                         # Box3::from_single_point, box3D.rs:17-22
            center, half = box_include(center, half, v1)
            center, half = box_include(center, half, v2)
            lo_p, hi_p = sub(sub(center, half), bmin), sub(add(center, half), bmin)
            lo = [min(_as_usize(fdiv(lo_p[c], cs[c])), n - 1) for c in range(3)]
            hi = [min(_as_usize(fdiv(hi_p[c], cs[c])), n - 1) for c in range(3)]
            for ix in range(lo[0], hi[0] + 1):
                for iy in range(lo[1], hi[1] + 1):
                    for iz in range(lo[2], hi[2] + 1):
                        lst = cells[iz * n * n + iy * n + ix]
                        if lst and lst[-1] == i:
                            continue
                        lst.append(i)
        offsets, tri_idx, off = [0] * (n * n * n), [], 0
        for c, lst in enumerate(cells):
            off += len(lst)
            offsets[c] = off
            tri_idx.extend(lst)
        return offsets, tri_idx


# ----------------------------------------------------------------------------------------------- raytracing/parser.rs
NAMED_COLOURS = {"red": (1.0, 0.0, 0.0), "blue": (0.0, 0.0, 1.0), "green": (0.0, 1.0, 0.0), "white": (1.0, 1.0, 1.0),
                 "black": (0.0, 0.0, 0.0), "cyan": (0.0, 1.0, 1.0), "violet": (1.0, 0.0, 1.0), "fuchsia": (0.96, 0.0, 96.0),
                 "yellow": (1.0, 1.0, 0.0), "orange": (0.98, 0.45, 0.02)}  # parser.rs:291-316


def tokenize(text):                                                    # parser.rs:126-214
    toks, i, n = [], 0, len(text)
    while True:
        while i < n and (text[i].isspace() or text[i] == "#"):
            if text[i] == "#":
                while i < n and text[i] != "\n":
                    i += 1
            else:
                i += 1
        if i >= n:
            return toks
        c, j = text[i], i
        if c in ",():>":
            j = i + 1
        elif c == '"':
            j = text.index('"', i + 1) + 1
        elif c in ".+-0123456789":
            if c in "+-":
                j += 1
            while j < n and text[j].isdigit():
                j += 1
            if j < n and text[j] == ".":
                j += 1
                while j < n and text[j].isdigit():
                    j += 1
        else:
            while j < n and text[j].isalpha():
                j += 1
            if j == i:
                raise ValueError(f"unexpected character {c!r}")
        toks.append(text[i:j])
        i = j


class ParsedScene:
    """ImageData (parser.rs:79-84): width, height, camera, objects in file order, lights."""

    def __init__(self, text, base_dir, grid_cells=32):
        t, self.pos = tokenize(text), 0
        self.t = t
        self._expect("size")                                           # parser.rs:270-276, 472-473
        self.width, self.height = int(self._float()), int(self._float())
        if self._maybe("camera"):                                      # parser.rs:365-380
            position = (0.0, 0.0, 0.0)
            if self._maybe("from"):
                position = self._vec3()
            point = self._vec3() if self._maybe("to") else add(position, (0.0, 0.0, 1.0))
            self.camera = Camera.look_at(position, point)
        else:
            self.camera = Camera((0.0, 0.0, 0.0))
        self.objects, self.materials, self.lights = [], [], []
        while self.pos < len(t):                                       # parser.rs:445-469
            kw = t[self.pos]
            if kw == "light":                                          # parser.rs:427-436
                self.pos += 1
                p = self._vec3()
                self.lights.append((p, self._colour(), 2.0))
            elif kw == "sphere":                                       # parser.rs:334-343
                self.pos += 1
                c, r = self._vec3(), self._float()
                self.materials.append(self._material())
                self.objects.append(("sphere", c, r))
            elif kw == "plane":                                        # parser.rs:345-354
                self.pos += 1
                nrm, d = self._vec3(), self._float()
                self.materials.append(self._material())
                self.objects.append(("plane", nrm, d))
            elif kw == "model":                                        # parser.rs:410-425, core.rs:56-65
                self.pos += 1
                path = self._next()[1:-1]
                material = self._material()
                trasform = mat_identity()                              # parser.rs:382-408
                while self._maybe(">"):
                    op = self._next()
                    if op == "scale":
                        nxt = mat_scale(self._float())
                    elif op == "translate":
                        nxt = mat_translate(self._vec3())
                    elif op == "rotate":
                        axis = self._vec3()
                        nxt = mat_rotate(axis, self._float())
                    else:
                        raise ValueError("unexpected token while parsing trasform")
                    trasform = mat_then(trasform, nxt)
                model = Model(*load_obj(os.path.join(base_dir, path)), trasform)
                offsets, tri_idx = model.create_grid(grid_cells)
                self.materials.append(material)
                self.objects.append(("model", model, offsets, tri_idx))
            else:
                raise ValueError(f"unexpected token '{kw}'")

    def _next(self):
        self.pos += 1
        return self.t[self.pos - 1]

    def _maybe(self, tok):
        if self.pos < len(self.t) and self.t[self.pos] == tok:
            self.pos += 1
            return True
        return False

    def _expect(self, tok):
        if not self._maybe(tok):
            raise ValueError(f"expected {tok}")

    def _float(self):
        return float(self._next())                                     # str::parse::<f64>, parser.rs:231

    def _vec3(self):                                                   # parser.rs:279-288
        self._expect("(")
        x = self._float()
        self._expect(",")
        y = self._float()
        self._expect(",")
        z = self._float()
        self._expect(")")
        return (x, y, z)

    def _colour(self):
        if self.pos < len(self.t) and self.t[self.pos] in NAMED_COLOURS:
            return NAMED_COLOURS[self._next()]
        return self._vec3()

    def _material(self):                                               # parser.rs:318-332: both prefixes give Metal
        metal = False
        if self._maybe("metal") or self._maybe("plastic"):
            metal = True
            self._expect(":")
        return (self._colour(), metal)

    def hit_objects(self):
        """The scene as py_restatement.hit takes it (flattened models: get_triangle evaluated once per triangle)."""
        out = []
        for ob in self.objects:
            if ob[0] != "model":
                out.append(ob)
                continue
            m = ob[1]
            pos = [[c for v in m.get_triangle(i) for c in v] for i in range(m.n_triangles())]
            nrm = [[c for v in m.get_triangle_normals(i) for c in v] for i in range(m.n_triangles())]
            out.append(("model", P.Model(m.center, m.half, 32, pos, nrm, ob[2], ob[3])))
        return out


# ------------------------------------------------------------------------ random numbers (NOT the reference's: DESIGN.md 2)
def philox4x32_10(key, ctr):
    """Random123's Philox4x32-10: key = (k0, k1), ctr = (c0, c1, c2, c3), all u32."""
    k0, k1 = key
    c0, c1, c2, c3 = ctr
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u01(lo, hi):                                                      # 52 random bits -> [1, 2) - 1.0 (rand 0.8.5 UniformFloat)
    bits = (((hi << 32) | lo) >> 12) | 0x3FF0000000000000
    return float(np.array([bits], dtype=np.uint64).view(np.float64)[0]) - 1.0


def gen_range(u, lo, hi):                                              # u * (hi - lo) + lo
    return u * (hi - lo) + lo


# ------------------------------------------------------------------------------------------------------------ main.rs
def cast(objects, materials, lights, o, d, rnd):                       # main.rs:57-132; rnd(bounce) = Vec3::random()
    color, attenuation = (0.0, 0.0, 0.0), 1.0
    for bounce in range(10):                                           # K, main.rs:63
        h = P.hit(objects, o, d)
        if h is None:
            color = add(color, mul((1.0, 1.0, 1.0), attenuation))     # main.rs:127
            break
        oi, _, t, n_raw = h
        hit_point = P.at(o, d, t)                                      # core.rs:89
        normal = mul(n_raw, -1.0) if dot(d, n_raw) > 0.001 else n_raw  # main.rs:68-73
        diffuse = (0.0, 0.0, 0.0)
        for lp, lc, lr in lights:                                      # main.rs:78-105
            v = normalize(sub(lp, hit_point))
            e = sub(hit_point, lp)
            dist2 = dot(e, e)                                          # squared_distance, vec3.rs:192-194
            occ = P.hit(objects, hit_point, v)
            if occ is not None:
                eo = sub(P.at(hit_point, v, occ[2]), hit_point)
                if dot(eo, eo) <= dist2:
                    continue
            eff = dot(v, normal)
            if eff > 0.001:
                dd = P.rmax(fdiv(dist2, lr * lr), 1.0)
                decay = fdiv(1.0, dd)
                diffuse = add(diffuse, mul(mul(lc, decay), eff))
        mcol, metal = materials[oi]
        diffuse_k, specular_k = (0.2, 0.8) if metal else (0.9, 0.1)    # main.rs:108-115
        if metal:                                                      # main.rs:50-55, vec3.rs:172-175
            bounce_dir = sub(d, mul(mul(normal, 2.0), dot(d, normal)))
        else:
            bounce_dir = normalize(add(normal, rnd(bounce)))
        k = mul(diffuse, diffuse_k)
        ca = mul(mcol, attenuation)
        color = add(color, (ca[0] * k[0], ca[1] * k[1], ca[2] * k[2]))  # main.rs:118
        o, d = hit_point, bounce_dir
        attenuation *= specular_k
    return color


def render_pixel(scene, objects, x, y, spp, seed, width=None, height=None):
    """One element of `pixels` (main.rs:159-170) with the shared counter-based random numbers."""
    w, h = width or scene.width, height or scene.height
    key = (seed & 0xFFFFFFFF, seed >> 32)
    pixel = x + w * y                                                  # main.rs:160-161
    px = (0.0, 0.0, 0.0)
    for s in range(spp):
        r = philox4x32_10(key, (pixel, s, 0, 0))
        x_off = gen_range(_u01(r[0], r[1]), -0.5, 0.5)                 # main.rs:162-163
        y_off = gen_range(_u01(r[2], r[3]), -0.5, 0.5)
        u = fdiv((float(x) + x_off) - float(w) * 0.5, float(w))        # main.rs:165-166
        v = fdiv((float(y) + y_off) - float(h) * 0.5, float(h))
        o, d = scene.camera.shoot_to(u, v)

        def rnd(bounce, s=s):                                          # vec3.rs:159-164: three gen_range(0.0..1.0)
            a = philox4x32_10(key, (pixel, s, 1 + 2 * bounce, 0))
            b = philox4x32_10(key, (pixel, s, 2 + 2 * bounce, 0))
            return (gen_range(_u01(a[0], a[1]), 0.0, 1.0), gen_range(_u01(a[2], a[3]), 0.0, 1.0), gen_range(_u01(b[0], b[1]), 0.0, 1.0))
        c = cast(objects, scene.materials, scene.lights, o, d, rnd)
        px = add(px, (fdiv(c[0], float(spp)), fdiv(c[1], float(spp)), fdiv(c[2], float(spp))))  # main.rs:169
    return px
