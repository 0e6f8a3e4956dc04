"""The whole-program row: the bitmap writer and the command line of main.rs:17-31,134-192.

CPU tests cover the writer's bytes and the argument handling (everything before the first device call); the GPU test
renders through rtm_cli_main and compares the file with the oracle's frame converted by main.rs:33-37.
The `image` crate's encoder is not under /root/reference: the expected bytes are built here from the BMP format
(BITMAPFILEHEADER + BITMAPINFOHEADER, 24 bpp, bottom-up BGR rows padded to 4 bytes) -- parity unpinned by a fixture."""
import os
import struct

import numpy as np
import pytest

from conftest import ASSETS, scene_path


def expected_bmp(rgb8: np.ndarray) -> bytes:
    h, w, _ = rgb8.shape
    pad = (4 - (w * 3) % 4) % 4
    size = (w * 3 + pad) * h
    head = b"BM" + struct.pack("<IHHI", 54 + size, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, size, 0, 0, 0, 0)
    rows = [rgb8[y, :, ::-1].tobytes() + b"\0" * pad for y in range(h - 1, -1, -1)]
    return head + b"".join(rows)


@pytest.mark.parametrize("shape", [(3, 5), (4, 4), (1, 1), (7, 2), (2, 3)])
def test_write_bmp_bytes(rtm, tmp_path, shape):
    h, w = shape
    rng = np.random.default_rng(h * 100 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    path = tmp_path / "x.bmp"
    rtm.write_bmp(str(path), img)
    data = path.read_bytes()
    assert data == expected_bmp(img)
    assert len(data) == 54 + ((w * 3 + 3) // 4 * 4) * h


def test_write_bmp_errors(rtm, tmp_path):
    with pytest.raises(rtm.RtmError):
        rtm.write_bmp(str(tmp_path / "no_such_dir" / "x.bmp"), np.zeros((2, 2, 3), dtype=np.uint8))


def test_cli_arguments_without_device(rtm, tmp_path, capfd):
    """clap's behaviour (main.rs:17-31): help and version exit 0, usage errors exit 2; a scene that cannot be read or
    parsed makes main() return Err (exit 1) before anything touches the device."""
    assert rtm.cli_main(["raytracer_mi355x", "--help"]) == 0
    assert "--sample-rate" in capfd.readouterr().out
    assert rtm.cli_main(["raytracer_mi355x", "-V"]) == 0
    assert rtm.cli_main(["raytracer_mi355x"]) == 2
    assert rtm.cli_main(["raytracer_mi355x", "a.scene", "b.scene"]) == 2
    assert rtm.cli_main(["raytracer_mi355x", "a.scene", "-s", "many"]) == 2
    assert rtm.cli_main(["raytracer_mi355x", "a.scene", "--no-such-flag"]) == 2
    assert rtm.cli_main(["raytracer_mi355x", "a.scene", "-o"]) == 2
    assert rtm.cli_main(["raytracer_mi355x", str(tmp_path / "missing.scene")]) == 1
    bad = tmp_path / "bad.scene"
    bad.write_text("size 10 10\nsphere (0,0,3 1 red\n")
    capfd.readouterr()
    assert rtm.cli_main(["raytracer_mi355x", str(bad)]) == 1
    assert "Error" in capfd.readouterr().err


@pytest.mark.gpu
def test_cli_renders_the_oracles_frame(rtm_gpu, orc, tmp_path, capfd):
    txt = ("size 67 41\ncamera from (0, -1, -4) to (0, 0, 3)\nsphere (0,0,3) 1 metal:white\nplane (0,1,0) 1 white\n"
           'model "./models/monkey.obj" orange > scale 0.5 > translate (1.5, 0, 3)\nlight (2,-3,0) white\nlight (-2,-3,1) blue')
    scene = tmp_path / "t.scene"
    scene.write_text(txt)
    ref = orc.OracleScene(text=txt, base_dir=ASSETS).render(spp=3, seed=9)
    for gamma in (False, True):
        out = tmp_path / f"o{int(gamma)}.bmp"
        args = ["raytracer_mi355x", str(scene), "-o", str(out), "-s", "3", "--seed", "9", "--base-dir", ASSETS]
        capfd.readouterr()
        assert rtm_gpu.cli_main(args + (["--gamma-correction"] if gamma else [])) == 0
        assert capfd.readouterr().out.startswith(f"Rendered {out} in ")
        want = expected_bmp(orc.to_rgb8(ref["rgb"], gamma=gamma))
        got = out.read_bytes()
        assert len(got) == len(want) and got[:54] == want[:54]
        # exact with and without gamma: main.rs:35-37 on the device, main.rs:42-48 with the host libm's pow (as the oracle, as Rust)
        assert got == want


@pytest.mark.gpu
def test_render_rgb8_equals_tonemapped_frame(rtm_gpu, rtm_scene, orc):
    s = rtm_scene("suzanne")
    host = s.render(spp=2, seed=5, width=96, height=80)
    img, seconds = s.render_rgb8(spp=2, seed=5, width=96, height=80)
    assert img.shape == (80, 96, 3) and seconds > 0
    assert np.array_equal(img, orc.to_rgb8(host["rgb"]))
    part, _ = s.render_rgb8(spp=2, seed=5, width=96, height=80, row_begin=1, row_stride=4)  # a rank's rows
    assert np.array_equal(part, img[1::4])
    # --gamma-correction (main.rs:42-48,179-183): the power is taken on the host, so the bytes are the oracle's, not +-1
    big = s.render(spp=2, seed=5, width=400, height=300)
    img_g, _ = s.render_rgb8(spp=2, seed=5, gamma=True, width=400, height=300)
    assert np.array_equal(img_g, orc.to_rgb8(big["rgb"], gamma=True))
