"""ctypes binding of the CPU ORACLE (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  See oracle/rt_oracle.hpp for citations and the
parity status ("parity unpinned" by the reference itself).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
NO_ID = 0xFFFFFFFF
COUNTER_NAMES = ("rays_camera", "rays_bounce", "rays_shadow", "sphere_tests", "plane_tests", "bbox_tests",
                 "cells_visited", "tri_tests", "model_hits")


class OrcParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_bounces", C.c_uint32),
                ("seed", C.c_uint64), ("row_begin", C.c_uint32), ("row_stride", C.c_uint32), ("flags", C.c_uint32),
                ("faithful", C.c_uint32), ("n_threads", C.c_uint32), ("reserved", C.c_uint32)]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("rt_oracle_capi.cpp", "rt_oracle.hpp")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 5
        L.orc_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_object.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_uint32)]
        L.orc_light.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_model_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32), C.c_void_p]
        L.orc_model_export.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_add_triangles.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_char_p, C.c_int]
        L.orc_trace_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int]
        L.orc_shoot.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_sample_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_to_rgb8.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """A parsed scene (parser.rs:438-477) held by the oracle."""

    def __init__(self, path: str | None = None, text: str | None = None, base_dir: str = "",
                 allow_missing_normals: bool = False):
        err = C.create_string_buffer(512)
        self._h = lib().orc_load(path.encode() if path else None, text.encode() if text is not None else None,
                                 base_dir.encode(), int(allow_missing_normals), err, 512)
        if not self._h:
            raise ValueError(err.value.decode())
        v = [C.c_uint32() for _ in range(5)]
        lib().orc_info(self._h, *[C.byref(x) for x in v])
        self.width, self.height, self.n_objects, self.n_lights, self.n_models = (x.value for x in v)

    def close(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def _refresh(self):
        v = [C.c_uint32() for _ in range(5)]
        lib().orc_info(self._h, *[C.byref(x) for x in v])
        self.width, self.height, self.n_objects, self.n_lights, self.n_models = (x.value for x in v)

    def camera(self) -> np.ndarray:
        out = np.zeros(13)
        lib().orc_camera(self._h, _p(out))
        return out

    def object(self, oi: int):
        kind, mt = C.c_uint32(), C.c_uint32()
        abcd, col = np.zeros(4), np.zeros(3)
        lib().orc_object(self._h, oi, C.byref(kind), _p(abcd), _p(col), C.byref(mt))
        return kind.value, abcd, col, mt.value

    def light(self, li: int) -> np.ndarray:
        out = np.zeros(7)
        lib().orc_light(self._h, li, _p(out))
        return out

    def model(self, oi: int) -> dict:
        nt, ni, cps = C.c_uint64(), C.c_uint64(), C.c_uint32()
        bbox = np.zeros(6)
        if lib().orc_model_info(self._h, oi, C.byref(nt), C.byref(ni), C.byref(cps), _p(bbox)) != 0:
            raise ValueError("object %d is not a model" % oi)
        pos = np.zeros((nt.value, 9))
        nrm = np.zeros((nt.value, 9))
        off = np.zeros(cps.value ** 3, dtype=np.uint64)
        idx = np.zeros(ni.value, dtype=np.uint64)
        lib().orc_model_export(self._h, oi, _p(pos), _p(nrm), _p(off), _p(idx))
        return dict(n_tris=nt.value, cells_per_side=cps.value, bbox_center=bbox[:3], bbox_half=bbox[3:], pos=pos,
                    nrm=nrm, offsets=off, tri_idx=idx)

    def add_triangles(self, pos9: np.ndarray, nrm9: np.ndarray, color=(1.0, 1.0, 1.0), metal: bool = False,
                      transform: np.ndarray | None = None):
        pos9 = np.ascontiguousarray(pos9, dtype=np.float32).reshape(-1, 9)
        nrm9 = np.ascontiguousarray(nrm9, dtype=np.float32).reshape(-1, 9)
        col = np.asarray(color, dtype=np.float64)
        tr = None if transform is None else np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        err = C.create_string_buffer(256)
        if lib().orc_add_triangles(self._h, pos9.shape[0], _p(pos9), _p(nrm9), _p(col), int(metal), _p(tr), err,
                                   256) != 0:
            raise ValueError(err.value.decode())
        self._refresh()

    def render(self, spp: int = 1, seed: int = 0, width: int = 0, height: int = 0, max_bounces: int = 10,
               row_begin: int = 0, row_stride: int = 1, zero_jitter: bool = False, faithful: bool = False,
               n_threads: int = 0, counters: bool = True) -> dict:
        W = width or self.width
        H = height or self.height
        n_rows = (H - row_begin + row_stride - 1) // row_stride
        rgb = np.zeros((n_rows, W, 3))
        obj = np.zeros((n_rows, W), dtype=np.uint32)
        tri = np.zeros((n_rows, W), dtype=np.uint32)
        cnt = np.zeros(9, dtype=np.uint64) if counters else None
        p = OrcParams(W, H, spp, max_bounces, seed, row_begin, row_stride, 1 if zero_jitter else 0,
                      1 if faithful else 0, n_threads, 0)
        rc = lib().orc_render(self._h, C.byref(p), _p(rgb), _p(obj), _p(tri), _p(cnt))
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        out = dict(rgb=rgb, obj_id=obj, tri_id=tri)
        if counters:
            out["counters"] = dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))
        return out

    def trace(self, rays6: np.ndarray, faithful: bool = False) -> dict:
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        n = rays6.shape[0]
        obj = np.zeros(n, dtype=np.uint32)
        tri = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n)
        nrm = np.zeros((n, 3))
        lib().orc_trace(self._h, n, _p(rays6), _p(obj), _p(tri), _p(t), _p(nrm), int(faithful))
        return dict(obj_id=obj, tri_id=tri, t=t, normal=nrm)

    def trace_counters(self, ray6) -> dict:
        """Work counters of ONE hit() query (cells visited, triangle tests, ...)."""
        ray6 = np.ascontiguousarray(ray6, dtype=np.float64).reshape(6)
        cnt = np.zeros(9, dtype=np.uint64)
        lib().orc_trace_counters(self._h, _p(ray6), _p(cnt))
        return dict(zip(("rays_camera", "rays_bounce", "rays_shadow", "sphere_tests", "plane_tests", "bbox_tests",
                         "cells_visited", "tri_tests", "model_hits"), (int(v) for v in cnt)))

    def shoot(self, uv: np.ndarray) -> np.ndarray:
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 6))
        lib().orc_shoot(self._h, uv.shape[0], _p(uv), _p(out))
        return out


def philox(seed: int, c0: int, c1: int, c2: int, c3: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(seed, c0, c1, c2, c3, _p(out))
    return out


def sample_rng(seed: int, pixel: int, sample: int, bounce: int):
    j, r = np.zeros(2), np.zeros(3)
    lib().orc_sample_rng(seed, pixel, sample, bounce, _p(j), _p(r))
    return j, r


def to_rgb8(rgb: np.ndarray, gamma: bool = False) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.zeros(rgb.shape, dtype=np.uint8)
    lib().orc_to_rgb8(rgb.size // 3, _p(rgb), int(gamma), _p(out))
    return out
