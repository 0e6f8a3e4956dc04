"""The RCCL leg of the multi-GPU assembly on the one GPU a test box has: a world-size-1 `nccl` process group (RCCL
initialises, its gather runs on the device, on torch's current stream) carries the row blocks that four "ranks" rendered
one after the other on this GPU, and rtm_deinterleave_device -- queued on the same stream, with no host
synchronisation in between -- assembles the frame.  The N > 1 host logic is covered with gloo on CPU
(test_multigpu_gloo.py); more than one rank cannot share a GPU under RCCL, so this is what can run here."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_gather_then_deinterleave_on_one_stream(rtm_gpu, rtm_scene):
    import torch
    import torch.distributed as dist
    s = rtm_scene("bunny")
    w, h, G = 160, 128, 4
    port = 29600 + os.getpid() % 2000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        full = torch.empty(h, w, 3, dtype=torch.float64, device="cuda")
        s.render_device(rtm_gpu.make_params(w, h, spp=2, seed=9), full.data_ptr())
        stream = torch.cuda.Stream()
        gathered = torch.empty(G, h // G, w, 3, dtype=torch.float64, device="cuda")
        frame = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for g in range(G):  # rank g's share, rendered on the stream the gather will run on
                p = rtm_gpu.make_params(w, h, spp=2, seed=9, row_begin=g, row_stride=G)
                blk = torch.empty(rtm_gpu.local_rows(p), w, 3, dtype=torch.float64, device="cuda")
                s.render_device(p, blk.data_ptr(), stream=stream.cuda_stream)
                dist.gather(blk, [gathered[g]], dst=0)  # RCCL, world of one: the device-side copy of a gather
            rtm_gpu.deinterleave_device(gathered.data_ptr(), w, h, G, frame.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(frame.cpu().numpy(), full.cpu().numpy())
        # the 8-bit path bench.py --gather u8 uses: tonemap on the device, gather of the bytes
        from raytracer_rs_amd.distributed import gather_frame
        rgb8 = torch.empty(h, w, 3, dtype=torch.uint8, device="cuda")
        rtm_gpu.tonemap_device(full.data_ptr(), h * w, False, rgb8.data_ptr())
        out8 = [torch.empty_like(rgb8)]
        dist.gather(rgb8, out8, dst=0)
        assert gather_frame(rgb8, h, w, 0, 1) is rgb8
        torch.cuda.synchronize()
        assert torch.equal(out8[0], rgb8)
    finally:
        dist.destroy_process_group()


def test_bench_step_under_torchrun_with_a_process_group(rtm_gpu):
    """bench.py's own N > 1 step -- render, RCCL gather of the row blocks, de-interleave, barrier and max-over-ranks timing --
    launched the way the driver launches it (torch.distributed.run starts before anything touches the GPU), with the one rank
    a one-GPU box allows (--force-dist): the first execution of that control flow must not be the driver's 8-GPU run."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    port = 29700 + os.getpid() % 2000
    for gather in ("f64", "u8"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist",
               "--spp", "2", "--size", "256", "--no-cpu-baseline", "--no-isolated", "--gather", gather]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 1 and line["config"]["n_ranks"] == 1 and "RCCL" in line["config"]["process_group"]
        assert line["value"] > 0 and "RCCL gather (%s)" % gather in line["config"]["parallelism"]
