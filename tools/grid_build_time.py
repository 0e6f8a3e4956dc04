"""create_grid (model.rs:147-201) for the 1M-triangle soup: rtm_grid_build_device with the positions resident in HBM
(HIP events around the call), rtm_grid_build from host buffers, and the C++ front end's host build for comparison."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from rtm_loader import load_package
R = load_package(); R.init(0)
lib = R.raw_lib()
t = time.time(); s = bench.soup_scene(R, 1_000_000, 20250725); t_front = time.time() - t
for m, what in zip(s.flat()["models"], ("teapot", "soup")):
    pos9 = np.asarray(m["pos"], dtype=np.float64).reshape(-1, 9)
    n, cells = pos9.shape[0], int(m["cells_per_side"])
    t = time.time(); off, idx = R.grid_build(pos9, m["bbox_center"], m["bbox_half"], cells); t_host_api = time.time() - t
    assert np.array_equal(off, m["offsets"]) and np.array_equal(idx, m["tri_idx"])
    cols = [torch.from_numpy(np.ascontiguousarray(pos9[:, a])).cuda() for a in range(9)]
    arr = (C.c_void_p * 9)(*[c.data_ptr() for c in cols])
    ctr = (C.c_double * 3)(*[float(v) for v in m["bbox_center"]]); half = (C.c_double * 3)(*[float(v) for v in m["bbox_half"]])
    d_off = torch.zeros(cells ** 3, dtype=torch.int32, device="cuda")
    d_idx = torch.zeros(idx.size, dtype=torch.int32, device="cuda")
    need = C.c_uint64()
    st = torch.cuda.current_stream()
    best = 1e9
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        rc = lib.rtm_grid_build_device(arr, n, ctr, half, cells, C.c_void_p(d_off.data_ptr()), C.c_void_p(d_idx.data_ptr()), idx.size, C.byref(need), C.c_void_p(st.cuda_stream))
        e1.record(st); torch.cuda.synchronize()
        assert rc == 0 and need.value == idx.size, (rc, need.value)
        best = min(best, e0.elapsed_time(e1))
    assert np.array_equal(d_off.cpu().numpy().view(np.uint32), off) and np.array_equal(d_idx.cpu().numpy().view(np.uint32), idx)
    print(f"{what}: {n} triangles, {idx.size} list entries: rtm_grid_build_device {best:.3f} ms (resident), rtm_grid_build from host buffers {t_host_api * 1e3:.1f} ms", flush=True)
print(f"C++ front end for the whole scene (parse, OBJ, both grids on the host, numpy generation of the soup): {t_front:.2f} s", flush=True)
