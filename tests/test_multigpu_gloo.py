"""World-size-2 (and 3, ragged) runs of the multi-GPU frame assembly on CPU with gloo: the same gather_frame()
bench.py uses with RCCL.  The oracle stands in for the per-rank renderer (this is a test; no GPU here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ASSETS, ROOT, scene_path


def _worker(rank, world, port, height, width, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from rtm_loader import load_package
    rtm = load_package()
    from raytracer_rs_amd.distributed import gather_frame, rows_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = oracle.OracleScene(path=scene_path("bunny"), base_dir=ASSETS)
    p = rtm.make_params(width, height, spp=2, seed=3, row_begin=rank, row_stride=world)
    assert rtm.local_rows(p) == rows_of_rank(height, rank, world)
    blk = sc.render(spp=2, seed=3, width=width, height=height, row_begin=rank, row_stride=world, n_threads=2)["rgb"]
    scratch = {}
    for _ in range(2):  # second call reuses the buffers
        frame = gather_frame(torch.from_numpy(blk), height, width, rank, world, scratch=scratch)
    # the 8-bit variant: every rank converts its rows (main.rs:33-37), one gather of an eighth of the bytes
    frame8 = gather_frame(torch.from_numpy(oracle.to_rgb8(blk)), height, width, rank, world, scratch=scratch)
    if rank == 0:
        full = sc.render(spp=2, seed=3, width=width, height=height, n_threads=2)["rgb"]
        ok8 = frame8.dtype == torch.uint8 and np.array_equal(frame8.numpy(), oracle.to_rgb8(full))
        np.save(out_path, np.array([np.array_equal(frame.numpy(), full) and ok8]))
    else:
        assert frame is None and frame8 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 48), (3, 50), (8, 64)])
def test_row_interleave_gather_gloo(world, height, tmp_path):
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, height, 64, out), nprocs=world, join=True)
    assert np.load(out)[0]
