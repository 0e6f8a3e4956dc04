"""raytracer-rs_amd -- Python host binding of the MI355X rendering back end.

The product is the C-ABI shared library ``lib/libraytracer_rs_amd.so`` (``include/rt_mi355x.h``);
this module only binds it with ctypes for tests, ``bench.py`` and multi-GPU runs under
``torch.distributed``.  It mirrors the reference's render boundary: a parsed ``ImageData``
(``parser.rs:79-84``) goes in, the f64 ``pixels`` buffer (``main.rs:153``) comes out.

There is NO CPU fallback: if the HIP library is missing, importing this module raises.

The directory name contains a hyphen (the layout the project asks for), so import it through
``load_package()`` in ``rtm_loader.py`` at the repo root, which registers it as ``raytracer_rs_amd``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # one HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's), so
    import torch  # noqa: F401  -- it must be mapped first or torch later finds "No HIP GPUs"
except ImportError:  # torch is plumbing (device buffers, torch.distributed), not a dependency of the library
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTM_LIB_PATH") or os.path.join(_HERE, "lib", "libraytracer_rs_amd.so")  # override: A/B builds
INCLUDE_PATH = os.path.join(os.path.dirname(_HERE), "include", "rt_mi355x.h")
NO_ID = 0xFFFFFFFF
FLAG_ZERO_JITTER = 1
FLAG_KERNEL_WAVE = 4
STAT_NAMES = ("rays_camera", "rays_bounce", "rays_shadow", "sphere_tests", "plane_tests", "bbox_tests",
              "cells_visited", "tri_tests", "model_hits")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C raytracer-rs_amd/csrc` (or __graft_entry__.build()). "
        "There is no CPU fallback for the render path.")


class RtmProfile(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("shade_ms", C.c_double), ("trace_ms", C.c_double), ("resolve_ms", C.c_double),
                ("shade_launches", C.c_uint32), ("trace_launches", C.c_uint32), ("traced_rays", C.c_uint64),
                ("cells_visited", C.c_uint64), ("tri_tests", C.c_uint64), ("rays_total", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64), ("bbox_tests", C.c_uint64)]


class RtmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rtm error {code}: {message}")
        self.code = code
        self.message = message


class RtmParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_bounces", C.c_uint32),
                ("seed", C.c_uint64), ("row_begin", C.c_uint32), ("row_stride", C.c_uint32), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class RtmStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_NAMES]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n in STAT_NAMES}


class RtmCamera(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("forward", C.c_double * 3), ("up", C.c_double * 3),
                ("right", C.c_double * 3), ("zoom", C.c_double)]


_PD = C.POINTER(C.c_double)
_PU = C.POINTER(C.c_uint32)


class RtmModel(C.Structure):
    _fields_ = [("bbox_center", C.c_double * 3), ("bbox_half", C.c_double * 3), ("cells_per_side", C.c_uint32),
                ("n_tris", C.c_uint32), ("offsets", _PU), ("tri_idx", _PU), ("n_tri_idx", C.c_uint64),
                ("pos", _PD * 9), ("nrm", _PD * 9), ("mat_idx", C.c_uint32), ("obj_idx", C.c_uint32)]


class RtmScene(C.Structure):
    _fields_ = [("camera", RtmCamera), ("n_objects", C.c_uint32),
                ("n_spheres", C.c_uint32), ("sph_cx", _PD), ("sph_cy", _PD), ("sph_cz", _PD), ("sph_r", _PD),
                ("sph_mat", _PU), ("sph_obj", _PU),
                ("n_planes", C.c_uint32), ("pl_nx", _PD), ("pl_ny", _PD), ("pl_nz", _PD), ("pl_d", _PD),
                ("pl_mat", _PU), ("pl_obj", _PU),
                ("n_models", C.c_uint32), ("models", C.POINTER(RtmModel)),
                ("n_lights", C.c_uint32), ("li_px", _PD), ("li_py", _PD), ("li_pz", _PD), ("li_r", _PD),
                ("li_g", _PD), ("li_b", _PD), ("li_radius", _PD),
                ("n_materials", C.c_uint32), ("mat_r", _PD), ("mat_g", _PD), ("mat_b", _PD), ("mat_type", _PU)]


_lib = C.CDLL(LIB_PATH)
_vp = C.c_void_p
_lib.rtm_local_rows.restype = C.c_uint32
_lib.rtm_local_rows.argtypes = [C.POINTER(RtmParams)]
_lib.rtm_init.argtypes = [C.c_int]
_lib.rtm_last_error.restype = C.c_char_p
_lib.rtm_version.restype = C.c_char_p
_lib.rtm_render.argtypes = [C.POINTER(RtmScene), C.POINTER(RtmParams), _vp, _vp, _vp, C.POINTER(RtmStats)]
_lib.rtm_scene_upload.argtypes = [C.POINTER(RtmScene), C.POINTER(_vp)]
_lib.rtm_scene_free.argtypes = [_vp]
_lib.rtm_render_device.argtypes = [_vp, C.POINTER(RtmParams), _vp, _vp, _vp, C.POINTER(RtmStats), _vp]
_lib.rtm_render_device_noalloc.argtypes = [_vp, C.POINTER(RtmParams), _vp, _vp, _vp, _vp]
_lib.rtm_noalloc_scratch_bytes.restype = C.c_uint64
_lib.rtm_time_render_device.argtypes = [_vp, C.POINTER(RtmParams), _vp, C.c_int, _vp, C.POINTER(C.c_double)]
_lib.rtm_profile_render_device.argtypes = [_vp, C.POINTER(RtmParams), _vp, _vp, C.POINTER(RtmProfile)]
_lib.rtm_trace_rays.argtypes = [_vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp]
_lib.rtm_tonemap_device.argtypes = [_vp, C.c_uint64, C.c_int, _vp, _vp]
_lib.rtm_deinterleave_device.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp]
_lib.rtm_timed_events_collect.argtypes = [C.POINTER(C.c_double), _PU]
_lib.rtm_render_rgb8.argtypes = [_vp, C.POINTER(RtmParams), C.c_int, _vp, C.POINTER(C.c_double)]
_lib.rtm_write_bmp.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _vp]
_lib.rtm_cli_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
_lib.rtm_f64_ops_selftest.argtypes = [C.c_uint64, _vp, _vp, _vp]
_lib.rtm_grid_build.argtypes = [C.POINTER(_PD), C.c_uint32, _PD, _PD, C.c_uint32, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]
_lib.rtm_grid_build_device.argtypes = [C.POINTER(_vp), C.c_uint32, _PD, _PD, C.c_uint32, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp]
_lib.rtm_frontend_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(_vp)]
_lib.rtm_frontend_add_triangles.argtypes = [_vp, C.c_uint64, _vp, _vp, _vp, C.c_uint32, _vp]
_lib.rtm_frontend_scene.restype = C.POINTER(RtmScene)
_lib.rtm_frontend_scene.argtypes = [_vp]
_lib.rtm_frontend_size.argtypes = [_vp, _PU, _PU]
_lib.rtm_frontend_free.argtypes = [_vp]

EXPORTED_SYMBOLS = (
    "rtm_local_rows", "rtm_init", "rtm_shutdown", "rtm_last_error", "rtm_version", "rtm_set_tuning", "rtm_render", "rtm_scene_upload",
    "rtm_scene_free", "rtm_render_device", "rtm_time_render_device", "rtm_profile_render_device", "rtm_trace_rays", "rtm_tonemap_device",
    "rtm_deinterleave_device", "rtm_f64_ops_selftest", "rtm_frontend_load", "rtm_frontend_add_triangles", "rtm_frontend_scene",
    "rtm_frontend_size", "rtm_frontend_free", "rtm_render_rgb8", "rtm_write_bmp", "rtm_cli_main", "rtm_timed_events_collect",
    "rtm_render_device_noalloc", "rtm_noalloc_scratch_bytes", "rtm_grid_build", "rtm_grid_build_device", "rtm_front_scene_fits")


def raw_lib() -> C.CDLL:
    return _lib


def _check(rc: int):
    if rc != 0:
        raise RtmError(rc, (_lib.rtm_last_error() or b"").decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def version() -> str:
    return _lib.rtm_version().decode()


_inited = False


def init(device: int = -1):
    """Bind this process to one GPU (one process per GPU).  Raises RtmError when no GPU is visible."""
    global _inited
    _check(_lib.rtm_init(device))
    _inited = True


def shutdown():
    """rtm_shutdown: frees the work space, internal streams and events (scenes must be released first)."""
    global _inited
    _check(_lib.rtm_shutdown())
    _inited = False


def noalloc_scratch_bytes() -> int:
    return int(_lib.rtm_noalloc_scratch_bytes())


def set_tuning(key: str, value: int):
    """rtm_set_tuning: scheduling knobs of the device path (results never change)."""
    _lib.rtm_set_tuning.argtypes = [C.c_char_p, C.c_longlong]
    _check(_lib.rtm_set_tuning(key.encode(), int(value)))


def make_params(width: int, height: int, spp: int = 1, seed: int = 0, max_bounces: int = 10, row_begin: int = 0,
                row_stride: int = 1, zero_jitter: bool = False, kernel_wave: bool = False) -> RtmParams:
    flags = (FLAG_ZERO_JITTER if zero_jitter else 0) | (FLAG_KERNEL_WAVE if kernel_wave else 0)
    return RtmParams(width, height, spp, max_bounces, seed, row_begin, row_stride, flags, 0)


_lib.rtm_front_scene_fits.restype = C.c_int
_lib.rtm_front_scene_fits.argtypes = [C.c_uint64, C.c_uint64]


def front_scene_fits(list_entries: int, cells: int) -> bool:
    """Whether a scene of that size runs the ray-front pipeline (host arithmetic only)."""
    return bool(_lib.rtm_front_scene_fits(list_entries, cells))


def local_rows(p: RtmParams) -> int:
    return int(_lib.rtm_local_rows(C.byref(p)))


class Scene:
    """The flattened scene produced by the host front end (stand-in for the Rust side + FFI crate)."""

    def __init__(self, path: str | None = None, text: str | None = None, base_dir: str = "",
                 synth_normals: bool = False):
        h = _vp()
        _check(_lib.rtm_frontend_load(path.encode() if path else None, text.encode() if text is not None else None,
                                      base_dir.encode(), 1 if synth_normals else 0, C.byref(h)))
        self._fe = h
        self._dev = None
        w, hgt = C.c_uint32(), C.c_uint32()
        _check(_lib.rtm_frontend_size(self._fe, C.byref(w), C.byref(hgt)))
        self.width, self.height = w.value, hgt.value

    # -- host view ------------------------------------------------------------------------------
    @property
    def c_scene(self) -> RtmScene:
        return _lib.rtm_frontend_scene(self._fe).contents

    def add_triangles(self, pos9, nrm9, color=(1.0, 1.0, 1.0), metal: bool = False, transform=None):
        pos9 = np.ascontiguousarray(pos9, dtype=np.float32).reshape(-1, 9)
        nrm9 = np.ascontiguousarray(nrm9, dtype=np.float32).reshape(-1, 9)
        col = np.asarray(color, dtype=np.float64)
        tr = None if transform is None else np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        self.release_device()
        _check(_lib.rtm_frontend_add_triangles(self._fe, pos9.shape[0], _ptr(pos9), _ptr(nrm9), _ptr(col), int(metal),
                                               _ptr(tr)))

    def flat(self) -> dict:
        """Copies of the flattened buffers as numpy arrays (for structural tests)."""
        s = self.c_scene

        def arr(p, n, dt=np.float64):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, dtype=dt)

        out = dict(
            camera=dict(position=np.array(s.camera.position[:]), forward=np.array(s.camera.forward[:]),
                        up=np.array(s.camera.up[:]), right=np.array(s.camera.right[:]), zoom=s.camera.zoom),
            n_objects=s.n_objects,
            spheres=np.stack([arr(s.sph_cx, s.n_spheres), arr(s.sph_cy, s.n_spheres), arr(s.sph_cz, s.n_spheres),
                              arr(s.sph_r, s.n_spheres)], axis=1),
            sph_mat=arr(s.sph_mat, s.n_spheres, np.uint32), sph_obj=arr(s.sph_obj, s.n_spheres, np.uint32),
            planes=np.stack([arr(s.pl_nx, s.n_planes), arr(s.pl_ny, s.n_planes), arr(s.pl_nz, s.n_planes),
                             arr(s.pl_d, s.n_planes)], axis=1),
            pl_mat=arr(s.pl_mat, s.n_planes, np.uint32), pl_obj=arr(s.pl_obj, s.n_planes, np.uint32),
            lights=np.stack([arr(getattr(s, f), s.n_lights) for f in
                             ("li_px", "li_py", "li_pz", "li_r", "li_g", "li_b", "li_radius")], axis=1),
            materials=np.stack([arr(s.mat_r, s.n_materials), arr(s.mat_g, s.n_materials), arr(s.mat_b, s.n_materials)],
                               axis=1),
            mat_type=arr(s.mat_type, s.n_materials, np.uint32), models=[])
        for i in range(s.n_models):
            m = s.models[i]
            n3 = m.cells_per_side ** 3
            out["models"].append(dict(
                bbox_center=np.array(m.bbox_center[:]), bbox_half=np.array(m.bbox_half[:]),
                cells_per_side=m.cells_per_side, n_tris=m.n_tris, offsets=arr(m.offsets, n3, np.uint32),
                tri_idx=arr(m.tri_idx, m.n_tri_idx, np.uint32),
                pos=np.stack([arr(m.pos[a], m.n_tris) for a in range(9)], axis=1),
                nrm=np.stack([arr(m.nrm[a], m.n_tris) for a in range(9)], axis=1), mat_idx=m.mat_idx, obj_idx=m.obj_idx))
        return out

    # -- device ---------------------------------------------------------------------------------
    def device_handle(self):
        if self._dev is None:
            if not _inited:
                init()
            h = _vp()
            _check(_lib.rtm_scene_upload(_lib.rtm_frontend_scene(self._fe), C.byref(h)))
            self._dev = h
        return self._dev

    def release_device(self):
        if self._dev is not None:
            _lib.rtm_scene_free(self._dev)
            self._dev = None

    def close(self):
        self.release_device()
        if getattr(self, "_fe", None):
            _lib.rtm_frontend_free(self._fe)
            self._fe = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, spp: int = 1, seed: int = 0, width: int = 0, height: int = 0, max_bounces: int = 10,
               row_begin: int = 0, row_stride: int = 1, zero_jitter: bool = False, ids: bool = True,
               stats: bool = False, kernel_wave: bool = False) -> dict:
        """rtm_render: host buffers in, host buffers out (the drop-in for main.rs:153-171)."""
        if not _inited:
            init()
        p = make_params(width or self.width, height or self.height, spp, seed, max_bounces, row_begin, row_stride,
                        zero_jitter, kernel_wave)
        rows = local_rows(p)
        rgb = np.zeros((rows, p.width, 3))
        obj = np.zeros((rows, p.width), dtype=np.uint32) if ids else None
        tri = np.zeros((rows, p.width), dtype=np.uint32) if ids else None
        st = RtmStats() if stats else None
        _check(_lib.rtm_render(_lib.rtm_frontend_scene(self._fe), C.byref(p), _ptr(rgb), _ptr(obj), _ptr(tri),
                               C.byref(st) if stats else None))
        out = dict(rgb=rgb, obj_id=obj, tri_id=tri)
        if stats:
            out["counters"] = st.as_dict()
        return out

    def render_device(self, params: RtmParams, d_rgb: int, d_obj: int = 0, d_tri: int = 0, stats: bool = False,
                      stream: int = 0):
        """rtm_render_device on raw device pointers (e.g. torch tensors' data_ptr())."""
        st = RtmStats() if stats else None
        _check(_lib.rtm_render_device(self.device_handle(), C.byref(params), d_rgb, d_obj or None, d_tri or None,
                                      C.byref(st) if stats else None, stream or None))
        return st.as_dict() if stats else None

    def render_device_noalloc(self, params: RtmParams, d_rgb: int, d_obj: int = 0, d_tri: int = 0, stream: int = 0):
        """rtm_render_device_noalloc: the same frame out of the library's static scratch array; allocates nothing."""
        _check(_lib.rtm_render_device_noalloc(self.device_handle(), C.byref(params), d_rgb, d_obj or None, d_tri or None,
                                              stream or None))

    def time_render_device(self, params: RtmParams, d_rgb: int, iters: int, stream: int = 0) -> float:
        ms = C.c_double()
        _check(_lib.rtm_time_render_device(self.device_handle(), C.byref(params), d_rgb, iters, stream or None,
                                           C.byref(ms)))
        return ms.value

    def profile_render_device(self, params: RtmParams, d_rgb: int, stream: int = 0) -> dict:
        """rtm_profile_render_device: per-kernel device time and the counters of the work actually done."""
        pr = RtmProfile()
        _check(_lib.rtm_profile_render_device(self.device_handle(), C.byref(params), d_rgb, stream or None, C.byref(pr)))
        return {k: getattr(pr, k) for k, _ in RtmProfile._fields_}

    def render_rgb8(self, spp: int = 20, seed: int = 0, gamma: bool = False, width: int | None = None,
                    height: int | None = None, **kw) -> tuple[np.ndarray, float]:
        """rtm_render_rgb8 (main.rs:153-185): the frame as [rows, width, 3] u8 and the wall seconds it took."""
        p = make_params(width or self.width, height or self.height, spp=spp, seed=seed, **kw)
        out = np.zeros((local_rows(p), p.width, 3), dtype=np.uint8)
        sec = C.c_double()
        _check(_lib.rtm_render_rgb8(self.device_handle(), C.byref(p), int(gamma), _ptr(out), C.byref(sec)))
        return out, sec.value

    def trace(self, rays6) -> dict:
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        n = rays6.shape[0]
        obj = np.zeros(n, dtype=np.uint32)
        tri = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n)
        nrm = np.zeros((n, 3))
        _check(_lib.rtm_trace_rays(self.device_handle(), n, _ptr(rays6), _ptr(obj), _ptr(tri), _ptr(t), _ptr(nrm)))
        return dict(obj_id=obj, tri_id=tri, t=t, normal=nrm)


def grid_build(pos9, bbox_center, bbox_half, cells_per_side: int = 32):
    """rtm_grid_build: create_grid (model.rs:147-201) on the GPU.  pos9 = [n_tris, 9] f64 (v0 xyz, v1 xyz, v2 xyz).
    Returns (offsets [cells^3] u32 inclusive prefix sums, tri_idx u32)."""
    if not _inited:
        init()
    pos9 = np.asarray(pos9, dtype=np.float64).reshape(-1, 9)
    n = pos9.shape[0]
    cols = [np.ascontiguousarray(pos9[:, a]) for a in range(9)]
    arr = (_PD * 9)(*[c.ctypes.data_as(_PD) for c in cols])
    ctr = (C.c_double * 3)(*[float(v) for v in bbox_center])
    half = (C.c_double * 3)(*[float(v) for v in bbox_half])
    offsets = np.zeros(cells_per_side ** 3, dtype=np.uint32)
    need = C.c_uint64()
    rc = _lib.rtm_grid_build(arr, n, ctr, half, cells_per_side, _ptr(offsets), None, 0, C.byref(need))
    if rc != 0 and not (rc == -1 and need.value > 0):
        _check(rc)
    tri_idx = np.zeros(need.value, dtype=np.uint32)
    if need.value:
        _check(_lib.rtm_grid_build(arr, n, ctr, half, cells_per_side, _ptr(offsets), _ptr(tri_idx), need.value, C.byref(need)))
    return offsets, tri_idx


def tonemap_device(d_rgb: int, n_pixels: int, gamma: bool, d_rgb8: int, stream: int = 0):
    _check(_lib.rtm_tonemap_device(d_rgb, n_pixels, int(gamma), d_rgb8, stream or None))


def timed_events_collect() -> tuple[float, int]:
    """rtm_timed_events_collect: (summed trace-kernel ms, launches) since set_tuning("front_timed_events", 1)."""
    ms = C.c_double()
    n = C.c_uint32()
    _check(_lib.rtm_timed_events_collect(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def write_bmp(path: str, rgb8) -> None:
    """rtm_write_bmp: rgb8 = [height, width, 3] u8, top row first (what main.rs:190 saves)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    _check(_lib.rtm_write_bmp(os.fsencode(path), rgb8.shape[1], rgb8.shape[0], _ptr(rgb8)))


def cli_main(argv) -> int:
    """rtm_cli_main: the reference's main() (main.rs:134-192); argv[0] is the program name.  Returns the exit status."""
    args = [os.fsencode(a) for a in argv]
    arr = (C.c_char_p * len(args))(*args)
    return int(_lib.rtm_cli_main(len(args), arr))


def deinterleave_device(d_blocks: int, width: int, height: int, n_ranks: int, d_rgb: int, stream: int = 0):
    _check(_lib.rtm_deinterleave_device(d_blocks, width, height, n_ranks, d_rgb, stream or None))


def f64_ops_selftest(a, b) -> np.ndarray:
    """Device evaluation of a/b, sqrt|a|, fmin, fmax, a*b+a (unfused), cell cast -- rows of the result."""
    if not _inited:
        init()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros((6, a.size))
    _check(_lib.rtm_f64_ops_selftest(a.size, _ptr(a), _ptr(b), _ptr(out)))
    return out
