#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X back end (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bunny|suzanne|multisphere|simple_sphere|soup]

A "step" is one pass of the hot path over one frame: render (one rtm_render_device launch per rank on its
interleaved rows) and, for N > 1, the RCCL gather to rank 0 plus the de-interleave.  The scene and the
framebuffer are resident in HBM before the timed region (the PCIe-inclusive rate is in DESIGN.md).
At N = 1 the workload is the configuration the metric is quoted on: bunny 1200x1200 at 20 spp.
value = hit() queries (camera + bounce + shadow; counted exactly by the library's stats instantiation, which the
parity tests pin to the oracle's counters) x K / wall seconds of the K timed steps, max over ranks.
One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ASSETS = os.path.join(ROOT, "assets")
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # public vector FP64 figure (FMA = 2 flop); BASELINE.md section 4


def algorithmic_bytes(c: dict, w: int, h: int) -> float:
    """SURVEY 8(d) B_ray summed with the exact counters: 48 B ray + 40 B hit record per query, 32 B per sphere or
    plane test, 48 B per model bbox test, 8 B per cell visited, 76 B per triangle test, 72 B of normals per model
    hit, plus the 24 B/pixel framebuffer write."""
    rays = c["rays_camera"] + c["rays_bounce"] + c["rays_shadow"]
    return (88.0 * rays + 32.0 * (c["sphere_tests"] + c["plane_tests"]) + 48.0 * c["bbox_tests"] + 8.0 * c["cells_visited"] +
            76.0 * c["tri_tests"] + 72.0 * c["model_hits"] + 24.0 * w * h)


def algorithmic_flops(c: dict) -> float:
    """SURVEY 8(d) hand counts (div and sqrt = 1 flop each): sphere 25 to the discriminant (+6 +normal 21 when it
    wins: folded into 31 per test as an upper-side estimate is avoided -- only the discriminant part is charged),
    plane 17, bbox 31, DDA set-up 54 per bbox test that enters the grid (charged per model hit or miss alike via
    bbox tests), 9 per cell step, 52 per full triangle test (30 to the u-reject: 41 on average), 66 per model hit
    for the barycentric normal."""
    return (25.0 * c["sphere_tests"] + 17.0 * c["plane_tests"] + 31.0 * c["bbox_tests"] + 54.0 * c["bbox_tests"] * 0.5 +
            9.0 * c["cells_visited"] + 41.0 * c["tri_tests"] + 66.0 * c["model_hits"])


SOUP_TEXT = ("size 3840 3840\n\ncamera from (4.0, -8, 0) to (0, 0, 7)\n\nplane (0, 0, 1.0) 10 white\n"
             "plane (0, 1.0, 0) 2 metal:white\n"
             'model "./models/teapot.obj" white > scale 0.4 > translate (2.5, 1.2, 6)\n'
             "light (1, 0, 2) green\nlight (-1, 0, 2) blue")


def soup_arrays(n_tris: int, seed: int):
    """BASELINE config 5's synthetic soup: N triangles, centres U([-2,2]^3)+(0,0,7), edge vectors U([-s,s]^3)
    with s = 0.02, geometric normals, numpy Philox(seed).  Returns (pos9, nrm9) as float32 [N, 9]."""
    import numpy as np
    rng = np.random.Generator(np.random.Philox(seed))
    c = rng.uniform(-2, 2, (n_tris, 1, 3)) + np.array([0.0, 0.0, 7.0])
    pos = (c + rng.uniform(-0.02, 0.02, (n_tris, 3, 3))).astype(np.float32)
    g = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
    g /= np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
    return pos.reshape(n_tris, 9), np.repeat(g, 3, axis=0).reshape(n_tris, 9).astype(np.float32)


def soup_scene(rtm, n_tris: int, seed: int):
    """Config 5's scene on the library's front end: SOUP_TEXT (two planes and two lights as in the bunny scene, the
    teapot) plus the soup as a second model."""
    # teapot.obj has no `vn` lines: the reference loader would refuse it; geometric normals are synthesised (stated)
    s = rtm.Scene(text=SOUP_TEXT, base_dir=ASSETS, synth_normals=True)
    s.add_triangles(*soup_arrays(n_tris, seed))
    return s


def cpu_baseline(workload: str, width: int, height: int, soup_tris: int, target_s: float = 15.0):
    """The oracle (kind "port": C++ restatement of the rayon path, faithful-cost mode) on ALL of this box's host cores
    (rayon's default, main.rs:158), on a bounded sample of the same workload: the full frame at a reduced spp (named
    scenes) or a reduced frame at 1 spp (config 5, whose 3840^2 frame costs minutes per sample on a CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # CPU baseline leg only
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)  # (the GPU box pins its share in OMP_NUM_THREADS)
    cores = max(1, min(cores, os.cpu_count() or 1))
    if workload == "soup":
        sc = oracle.OracleScene(text=SOUP_TEXT, base_dir=ASSETS, allow_missing_normals=True)
        sc.add_triangles(*soup_arrays(soup_tris, 20250725))
        width = height = 512
    else:
        sc = oracle.OracleScene(path=os.path.join(ASSETS, "scene", workload + ".scene"), base_dir=ASSETS)
    t0 = time.perf_counter()
    sc.render(spp=1, seed=0, width=width, height=height, faithful=True, n_threads=cores, counters=False)
    t1 = time.perf_counter() - t0
    if workload == "soup":  # grow the frame, keep 1 spp
        side = int(min(3840, 512 * max(1.0, (target_s / max(t1, 1e-3)) ** 0.5))) // 8 * 8
        width = height = side
        spp = 1
    else:
        spp = int(max(1, min(20, target_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    sc.render(spp=spp, seed=0, width=width, height=height, faithful=True, n_threads=cores, counters=False)
    dt = time.perf_counter() - t0
    cnt = sc.render(spp=spp, seed=0, width=width, height=height, faithful=False, n_threads=cores)["counters"]
    rays = cnt["rays_camera"] + cnt["rays_bounce"] + cnt["rays_shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"{workload} {width}x{height} at {spp} spp, all pixels, faithful-cost mode "
                      f"(per-access Mat4::apply, model.rs:108-122), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bunny")
    ap.add_argument("--spp", type=int, default=20)
    ap.add_argument("--size", type=int, default=0, help="override width=height (0 = scene file's size)")
    ap.add_argument("--soup-tris", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the extra pass with one chunk in flight (rocprofv3 runs: every launch of the process is then in the "
                         "regime of the timed steps, and the profiler's average kernel duration is comparable with roofline.kernel_ms)")
    ap.add_argument("--streams", type=int, default=2, help="chunks of a frame in flight at once (rtm_set_tuning front_streams; default 2, 0 = by frame size)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gather + de-interleave of the N > 1 step even with one rank "
                         "(rehearsal of the multi-GPU control flow on a one-GPU box; launch under torch.distributed.run)")
    ap.add_argument("--gather", choices=("f64", "u8"), default="f64",
                    help="N > 1: what rank 0 collects -- the f64 framebuffer (default, the parity path) or the 8-bit image "
                         "after the device tonemap (main.rs:173-185; an eighth of the bytes)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rtm_loader import load_package
    rtm = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    rtm.init(local_rank)
    rtm.set_tuning("front_streams", args.streams)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.workload == "soup":
        scene = soup_scene(rtm, args.soup_tris, 20250725)
    else:
        scene = rtm.Scene(path=os.path.join(ASSETS, "scene", args.workload + ".scene"), base_dir=ASSETS)
    W = args.size or scene.width
    H = args.size or scene.height
    p_local = rtm.make_params(W, H, spp=args.spp, seed=0, row_begin=rank, row_stride=world)
    rows = rtm.local_rows(p_local)
    stream = torch.cuda.current_stream()
    local = torch.empty(rows, W, 3, dtype=torch.float64, device="cuda")
    scene.device_handle()  # scene resident in HBM before the timed region
    if world > 1 and H % world != 0:
        sys.exit("bench.py: height must be divisible by the number of GPUs")
    from raytracer_rs_amd.distributed import gather_frame
    scratch = {}

    local8 = torch.empty(rows, W, 3, dtype=torch.uint8, device="cuda") if args.gather == "u8" else None

    def step():
        scene.render_device(p_local, local.data_ptr(), stream=stream.cuda_stream)
        if use_dist:
            # one gather of the compact row blocks to rank 0 over xGMI (RCCL), then de-interleave
            if local8 is not None:
                rtm.tonemap_device(local.data_ptr(), rows * W, False, local8.data_ptr(), stream=stream.cuda_stream)
                return gather_frame(local8, H, W, rank, world, scratch=scratch, force=use_dist)
            return gather_frame(local, H, W, rank, world, deinterleave_device=rtm.deinterleave_device, scratch=scratch, force=use_dist)
        return local

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def progress(what):  # (stderr: a long run must be seen to be alive; stdout carries the one JSON line)
        if rank == 0:
            print(f"[bench] {what}", file=sys.stderr, flush=True)

    progress(f"{args.workload} {W}x{H} at {args.spp} spp: counting frame")
    # exact ray / work counts of the whole frame (outside the timed region)
    cnt = scene.render_device(p_local, local.data_ptr(), stats=True, stream=stream.cuda_stream)
    if use_dist:
        t = torch.tensor([cnt[k] for k in rtm.STAT_NAMES], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        cnt = dict(zip(rtm.STAT_NAMES, (int(v) for v in t.tolist())))
    rays = cnt["rays_camera"] + cnt["rays_bounce"] + cnt["rays_shadow"]

    progress(f"{args.warmup} warm-up + {args.steps} timed steps")
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    progress(f"{dt / args.steps * 1e3:.2f} ms per step; timing the dominant kernel")
    n_ranks_seen = 1
    if use_dist:
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one)  # the ranks RCCL actually connects: must equal n_gpus
        n_ranks_seen = int(one.item())
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # dominant kernel: the same K steps once more with a HIP event pair around every trace-kernel launch, on the stream
    # the kernel is launched on (the frames above carry no events, so `value` is not perturbed) ...
    rtm.set_tuning("front_timed_events", 1)
    for _ in range(args.steps):
        step()
    barrier()
    ev_ms, ev_n = rtm.timed_events_collect()
    # ... once more with ONE chunk of the frame in flight at a time: no other kernel runs beside the one being timed, so
    # these durations add up to less than the frame (the default, two chunks on two internal streams, overlaps the late
    # rounds of one chunk with the heavy rounds of the other: faster frames, but each launch is stretched by its neighbour)
    iso_steps = max(1, min(args.steps, 5))
    iso_ms, iso_n, iso_frame_ms = 0.0, 0, None
    if not args.no_isolated:
        progress("... and with one chunk of the frame in flight at a time")
        rtm.set_tuning("front_streams", 1)
        step()
        barrier()
        rtm.timed_events_collect()
        t0 = time.perf_counter()
        for _ in range(iso_steps):
            step()
        barrier()
        iso_frame_ms = (time.perf_counter() - t0) / iso_steps * 1e3
        iso_ms, iso_n = rtm.timed_events_collect()
    rtm.set_tuning("front_timed_events", 0)
    # ... and the per-kernel split of one frame (still one chunk at a time) plus the counters of the work really done
    progress("per-kernel split and work counters")
    prof = scene.profile_render_device(p_local, local.data_ptr(), stream=stream.cuda_stream)
    rtm.set_tuning("front_streams", args.streams)
    frame_ms = scene.time_render_device(p_local, local.data_ptr(), max(1, min(args.steps, 10)), stream=stream.cuda_stream)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = rays * args.steps / dt / 1e6
        # SURVEY 8(d) per-unit figures x the units the dominant kernel processed.  Scenes with models: the trace kernel (48 B ray
        # + 40 B result per traced ray, 8 B per cell visited, 76 B per triangle test).  Scenes whose time is in the shade kernel
        # (no models, or few rays that reach them): its queries -- 88 B per ray issued, 32 B per sphere or plane test, 48 B per
        # model-box test -- plus the 24 B/pixel framebuffer.
        # (per launch in the regime of the timed frames: two chunks in flight make twice as many launches of half the size)
        single = prof["shade_launches"] == 0
        shade_dominant = (not single) and prof["shade_ms"] > prof["trace_ms"]
        if shade_dominant:
            # timed by rtm_profile_render_device: a HIP event pair around every launch, on the stream it is launched on, one
            # chunk of the frame in flight at a time (the shade launches of ordinary frames carry no events)
            n_launch = n_launch_iso = max(1, prof["shade_launches"])
            frame_bytes = (88.0 * prof["rays_total"] + 32.0 * (prof["sphere_tests"] + prof["plane_tests"]) + 48.0 * prof["bbox_tests"] + 24.0 * W * rows)
            kern_ms = kern_ms_iso = prof["shade_ms"] / n_launch
            flops = (25.0 * prof["sphere_tests"] + 17.0 * prof["plane_tests"] + 31.0 * prof["bbox_tests"]) / n_launch
            ev_n_out, iso_n_out = prof["shade_launches"], prof["shade_launches"]
            iso_frame = prof["total_ms"]
        else:
            n_launch = max(1, ev_n // args.steps if ev_n else prof["trace_launches"])
            n_launch_iso = max(1, iso_n // iso_steps if iso_n else prof["trace_launches"])
            frame_bytes = 88.0 * prof["traced_rays"] + 8.0 * prof["cells_visited"] + 76.0 * prof["tri_tests"]
            kern_ms = ev_ms / ev_n if ev_n else prof["trace_ms"] / n_launch  # (single-kernel formulations record no events)
            kern_ms_iso = iso_ms / iso_n if iso_n else kern_ms
            flops = (9.0 * prof["cells_visited"] + 41.0 * prof["tri_tests"]) / n_launch
            ev_n_out, iso_n_out = ev_n, iso_n
            iso_frame = iso_frame_ms
        abytes = frame_bytes / n_launch
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f"{args.workload}_{W}x{H}_{args.spp}spp_n{world}", {}).get("hbm_bytes_per_launch")
        headline = args.workload == "bunny" and W == 1200 and args.spp == 20
        if single:
            kname = "rtw::render_wave_kernel<false>"
        elif shade_dominant:
            kname = "rtf::front_shade_kernel<false, *, true> (all rounds)"
        else:
            kname = "rtf::front_trace_kernel<false, true, true>" if args.workload == "soup" else "rtf::front_trace_kernel<false, true, false>"
        out = {
            "metric": "Mrays/s (primary+shadow) bunny 1200^2 20spp" if headline
            else f"Mrays/s (primary+shadow) {args.workload} {W}x{H} {args.spp}spp",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "reference scene files (assets/), counter-based RNG seed 0" if args.workload != "soup"
            else f"synthetic {args.soup_tris}-triangle soup (numpy Philox seed 20250725) + teapot.obj with synthesised normals",
            "config": {"workload": f"{args.workload} {W}x{H} at {args.spp} spp, 10 bounces, grid 32^3",
                       "rays_per_frame": rays, "rays_camera": cnt["rays_camera"], "rays_bounce": cnt["rays_bounce"],
                       "rays_shadow": cnt["rays_shadow"], "samples_per_s": W * H * args.spp * args.steps / dt,
                       "rays_issued": prof["rays_total"], "rays_traced": prof["traced_rays"],
                       "mrays_issued_per_s": prof["rays_total"] * world / (ms_per_step * 1e-3) / 1e6 if world == 1 else None,
                       "rays_note": "value counts the rays the REFERENCE casts for this frame (rays_per_frame: exact counters, pinned to "
                                    "the oracle's). The device path answers shadow rays whose light is behind the surface (main.rs:96 "
                                    "fails) without tracing them: rays_issued is what it does issue, and of those only the rays that "
                                    "enter a model's box reach the trace kernel: rays_traced (both on this rank)",
                       "n_ranks": n_ranks_seen, "process_group": ("nccl (RCCL), %d rank(s)" % dist.get_world_size()) if use_dist else None,
                       "front_streams": args.streams or "by frame size (3 chunks in flight from 12 Mi sample paths per frame, else 2)", "parallelism": f"row-interleave x{world}" + (f" + RCCL gather ({args.gather}) to rank 0" if use_dist else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": kname,
                         "bytes_are": "algorithmic (SURVEY 8d), not measured: " + ("88 B per ray issued + 32 B per sphere or plane test + 48 B per model-box test + 24 B per pixel"
                                      if shade_dominant else "88 B per traced ray + 8 B per cell + 76 B per triangle test") +
                                      "; `traffic` is the measured HBM figure per launch and frac_measured_traffic what it comes to",
                         "frac_measured_traffic": (traffic / (kern_ms_iso * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel_ms": kern_ms, "kernel_launches_timed": ev_n_out, "launches_per_frame": n_launch,
                         "kernel_ms_isolated": kern_ms_iso if iso_n_out else None, "isolated_launches_timed": iso_n_out, "isolated_frame_ms": iso_frame,
                         "frac_isolated": frame_bytes / n_launch_iso / (kern_ms_iso * 1e-3) / 1e9 / HBM_PEAK_GBS if iso_n_out else None,
                         "launches_per_frame_isolated": n_launch_iso if iso_n_out else None,
                         "algorithmic_bytes_per_launch": abytes,
                         "limiter": "valu-issue",
                         "binding_limit": ("vector issue of the analytic primitives (scenes with more than four spheres: a packed-f32 screen over every sphere, then the f64 test "
                                           "of core.rs:183-195 on the few spheres a ray can meet; plane tests) and the streamed path state; the scene is L2-resident; "
                                           "valu counts the REFERENCE's f64 operations, which the screen does not all execute" if shade_dominant else
                                           "VALU issue (f64 triangle tests, f32 box screen) at ~35 of 64 lanes and vector-memory latency of per-lane gathers: "
                                           "the scene and the queues of a round are L2/MALL-resident (measured traffic is a fraction of the algorithmic bytes), see valu and DESIGN.md")},
            "valu": {"achieved_tflops": flops / (kern_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TF,
                     "frac": flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, "algorithmic_flops_per_launch": flops},
            "pipeline": {"frame_ms": frame_ms, "one_chunk_at_a_time": (not args.no_isolated) or args.streams == 1, "shade_ms": prof["shade_ms"], "trace_ms": prof["trace_ms"],
                         "resolve_ms": prof["resolve_ms"], "shade_launches": prof["shade_launches"], "trace_launches": prof["trace_launches"],
                         "traced_rays": prof["traced_rays"], "cells_visited": prof["cells_visited"], "tri_tests": prof["tri_tests"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            progress("CPU baseline (oracle, all host cores)")
            out["cpu_baseline"] = cpu_baseline(args.workload, W, H, args.soup_tris)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
