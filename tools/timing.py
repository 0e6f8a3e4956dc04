"""Frame timings on the one GPU of a box (rtm_time_render_device: HIP events around back-to-back frames).

    python tools/timing.py scenes [--spp 20]                  the four named scenes, frame time and checksum
    python tools/timing.py shares [--workload bunny|soup] [--spp N] [--streams 1,2,3,4]
                                                              rank 0's share (rows y = 0 mod G) for G = 1, 2, 4, 8: the
                                                              strong-scaling proxy (no 8-GPU node is ours to launch on)
    python tools/timing.py spheres                            the f32 sphere screen on / off for lattices of 5..343 spheres
    python tools/timing.py env                                bunny frame and its G = 8 share under the current RTM_* environment
    python tools/timing.py sweep KEY=V[,V...] ...             one child process per setting (the knobs are read at rtm_init)

Replaces round 1-2's time_scenes / time_env / time_split* / time_w8 / time_*variants scripts."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ASSETS = os.path.join(ROOT, "assets")


def load(workload):
    import torch  # noqa: F401  (device memory for the framebuffers)
    from rtm_loader import load_package
    rtm = load_package()
    rtm.init(0)
    if workload == "soup":
        import bench
        return rtm, bench.soup_scene(rtm, 1_000_000, 20250725)
    return rtm, rtm.Scene(path=os.path.join(ASSETS, "scene", workload + ".scene"), base_dir=ASSETS)


def frame_ms(rtm, scene, spp, world=1, iters=5):
    import torch
    p = rtm.make_params(scene.width, scene.height, spp=spp, seed=0, row_begin=0, row_stride=world)
    buf = torch.empty(rtm.local_rows(p) * scene.width * 3, dtype=torch.float64, device="cuda")
    scene.time_render_device(p, buf.data_ptr(), 1)
    return scene.time_render_device(p, buf.data_ptr(), iters), float(buf.sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=("scenes", "shares", "env", "sweep", "spheres"))
    ap.add_argument("settings", nargs="*")
    ap.add_argument("--workload", default="bunny")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--streams", default="")
    a = ap.parse_args()
    if a.mode == "sweep":  # KEY=V1,V2 ... -> one `env` run per value (environment variable RTM_<KEY>)
        for st in a.settings:
            key, vals = st.split("=")
            for v in vals.split(","):
                env = dict(os.environ, **{"RTM_" + key.upper(): v})
                subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "env", "--workload", a.workload, "--spp", str(a.spp)], env=env, check=True, timeout=300)
        return
    if a.mode == "spheres":  # the f32 sphere screen on / off against the number of spheres (a lattice like multisphere.scene's)
        import torch  # noqa: F401
        from rtm_loader import load_package
        rtm = load_package()
        rtm.init(0)
        for n in (5, 8, 12, 16, 27, 64, 125, 343):
            side = max(2, round(n ** (1 / 3)))
            pts = [(x - side / 2 + 0.5, y - side / 2 + 0.5, 6 + z) for x in range(side) for y in range(side) for z in range(side)][:n]
            txt = "size 1200 1200\ncamera from (-0.5, -0.5, 0)\n" + "\n".join("sphere (%g, %g, %g) .45 white" % p for p in pts)
            txt += "\nplane (0, 1, 0) 4 white\nlight (0, -8, 0) white\nlight (4, -6, 2) red\nlight (-4, -6, 2) blue"
            s = rtm.Scene(text=txt)
            out = []
            for on in (1, 0):
                rtm.set_tuning("front_sph_screen", on)
                out.append(frame_ms(rtm, s, a.spp or 20)[0])
            print(f"{n:4d} spheres: screen on {out[0]:7.2f} ms   off {out[1]:7.2f} ms", flush=True)
        return
    if a.mode == "scenes":
        for name in ("simple_sphere", "multisphere", "suzanne", "bunny"):
            rtm, s = load(name)
            ms, chk = frame_ms(rtm, s, a.spp or 20)
            print(f"{name:14s} {ms:8.2f} ms   checksum {chk:.9f}", flush=True)
        return
    rtm, s = load(a.workload)
    spp = a.spp or (16 if a.workload == "soup" else 20)
    iters = 2 if a.workload == "soup" else 5
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RTM_") and k != "RTM_LIB_PATH")
    if a.mode == "env":
        w1, w8 = frame_ms(rtm, s, spp, 1, iters)[0], frame_ms(rtm, s, spp, 8, iters)[0]
        print(f"{tag} : w1 {w1:.2f} ms  w8 {w8:.2f} ms  w1/w8 {w1 / w8:.2f}", flush=True)
        return
    for streams in ([int(x) for x in a.streams.split(",")] if a.streams else [None]):
        if streams is not None:
            rtm.set_tuning("front_streams", streams)
        out = [frame_ms(rtm, s, spp, w, iters)[0] for w in (1, 2, 4, 8)]
        print(f"{a.workload} {s.width}x{s.height} at {spp} spp, front_streams {streams if streams is not None else 'default'}: " +
              "  ".join(f"w{w} {m:.2f} ms" for w, m in zip((1, 2, 4, 8), out)) + f"   w1/w8 = {out[0] / out[3]:.2f}", flush=True)


if __name__ == "__main__":
    main()
