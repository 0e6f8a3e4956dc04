"""rtm_render_device_noalloc (the entry point that allocates nothing, include/rt_mi355x.h) against rtm_render_device,
through the C ABI: bit-equal framebuffer and hit ids on two calls in a row, the second with a smaller input, so that
whatever the first call left in the static scratch array must not leak into the second.  Collected last (file name):
the round's end runs `pytest -x`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(rtm, scene, w, h, spp, seed):
    import torch
    p = rtm.make_params(w, h, spp=spp, seed=seed)
    out = {}
    for name in ("noalloc", "ref"):
        rgb = torch.full((h, w, 3), -7.0, dtype=torch.float64, device="cuda")
        obj = torch.full((h, w), 12345, dtype=torch.int32, device="cuda")
        tri = torch.full((h, w), 12345, dtype=torch.int32, device="cuda")
        if name == "noalloc":
            scene.render_device_noalloc(p, rgb.data_ptr(), d_obj=obj.data_ptr(), d_tri=tri.data_ptr())
        else:
            scene.render_device(p, rgb.data_ptr(), d_obj=obj.data_ptr(), d_tri=tri.data_ptr())
        torch.cuda.synchronize()
        out[name] = (rgb.cpu().numpy(), obj.cpu().numpy(), tri.cpu().numpy())
    return out


def test_noalloc_matches_render_device_twice(rtm_gpu, rtm_scene):
    import torch
    s = rtm_scene("bunny")
    assert rtm_gpu.noalloc_scratch_bytes() == 256 << 20
    # first call: 1200 wide at 20 spp -- one 8-row band (192,000 sample paths) fills most of the array, so the 64 rows
    # take eight groups of bands; second call: a much smaller frame (fewer sub-queues in use, stale state all around it)
    for (w, h, spp, seed) in ((1200, 64, 20, 3), (160, 96, 3, 8)):
        o = _both(rtm_gpu, s, w, h, spp, seed)
        for k, what in enumerate(("rgb", "obj_id", "tri_id")):
            a, b = o["noalloc"][k], o["ref"][k]
            if what == "rgb":
                assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), f"{w}x{h}@{spp}: f64 framebuffer differs"
            else:
                assert np.array_equal(a, b), f"{w}x{h}@{spp}: {what} differs"
        assert (o["noalloc"][0] >= 0).all()
    # no device memory is taken or returned by the call (hipMemGetInfo before/after, caches of torch untouched)
    p = rtm_gpu.make_params(160, 96, spp=3, seed=8)
    rgb = torch.empty(96, 160, 3, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    s.render_device_noalloc(p, rgb.data_ptr())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0


def test_noalloc_rejects_what_does_not_fit(rtm_gpu, rtm_scene):
    import torch
    s = rtm_scene("suzanne")
    rgb = torch.empty(8, 3840, 3, dtype=torch.float64, device="cuda")
    with pytest.raises(rtm_gpu.RtmError) as e:   # one band of 3840 x 20 spp needs more than the 256 MiB array
        s.render_device_noalloc(rtm_gpu.make_params(3840, 8, spp=20), rgb.data_ptr())
    assert e.value.code == -1 and "static scratch" in e.value.message
    with pytest.raises(rtm_gpu.RtmError) as e:   # single-kernel formulations have no work space: not this entry point
        s.render_device_noalloc(rtm_gpu.make_params(64, 8, spp=1, kernel_wave=True), rgb.data_ptr())
    assert e.value.code == -1
    # and what fits at that width still renders, equal to rtm_render_device
    o = _both(rtm_gpu, s, 3840, 8, 8, 1)
    assert np.array_equal(o["noalloc"][0].view(np.uint64), o["ref"][0].view(np.uint64))
