"""Multi-GPU assembly of a frame rendered with the static row-interleave split.

Rank g of G renders rows y = g (mod G) (rtm_params.row_begin = g, row_stride = G); rank 0 receives every rank's
compact row block with ONE gather (RCCL over xGMI when the tensors are on GPUs, gloo on CPU in the tests) and
de-interleaves them into the row-major framebuffer the reference's main() reads (main.rs:176-178).
There is no other exchange on this path: pixels are independent (main.rs:158-171).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def rows_of_rank(height: int, rank: int, world: int) -> int:
    return 0 if rank >= height else (height - rank + world - 1) // world


def gather_frame(local_rows: torch.Tensor, height: int, width: int, rank: int, world: int, deinterleave_device=None,
                 scratch: dict | None = None, force: bool = False):
    """local_rows: [rows_of_rank, width, 3] on this rank's device -- f64 (the parity path) or u8 after
    rtm_tonemap_device (an eighth of the bytes when only the bitmap is wanted, main.rs:173-185).
    Returns the [height, width, 3] frame on rank 0 and None elsewhere.  `scratch` (optional dict) keeps the
    gather/frame buffers between frames.  `force`: run the gather and the de-interleave even in a process group of one
    rank (bench.py --force-dist: the N > 1 control flow on the one GPU a test box has)."""
    if world == 1 and not force:
        return local_rows
    max_rows = rows_of_rank(height, 0, world)
    pad = local_rows
    if local_rows.shape[0] != max_rows:  # ragged heights: pad to the largest block so the gather is regular
        pad = local_rows.new_zeros((max_rows, width, 3))
        pad[: local_rows.shape[0]] = local_rows
    scratch = scratch if scratch is not None else {}
    if rank == 0:
        key = (max_rows, width, world, str(local_rows.device), local_rows.dtype)
        if scratch.get("key") != key:
            scratch["key"] = key
            scratch["gathered"] = local_rows.new_empty((world, max_rows, width, 3))
            scratch["frame"] = local_rows.new_empty((height, width, 3))
        gathered, frame = scratch["gathered"], scratch["frame"]
        dist.gather(pad, list(gathered.unbind(0)), dst=0)
        if deinterleave_device is not None and height % world == 0 and gathered.is_cuda and gathered.dtype == torch.float64:
            # on the stream the gather was queued on: rtm's launches do not wait for the null stream
            deinterleave_device(gathered.data_ptr(), width, height, world, frame.data_ptr(),
                                stream=torch.cuda.current_stream(gathered.device).cuda_stream)
        else:
            for g in range(world):
                n = rows_of_rank(height, g, world)
                frame[g::world] = gathered[g, :n]
        return frame
    dist.gather(pad, None, dst=0)
    return None
