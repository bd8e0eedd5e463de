"""Developer probe: frame time over the first N frames of a view (the list-share feedback re-ranks the schedule during the first 16), for
the depth-20 fractal at 4K (BASELINE config 5, primary rays) and monu9 at 1080p (config 2).  Run twice: as is, and with SVO_NO_LIST_BALANCE=1.
usage: python tools/balance_probe.py [--frames 60]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
a = ap.parse_args()
pkg = entry.load_package()
import torch  # noqa: E402

gpu = pkg.Gpu(0)
gpu.set_option(pkg.gpu.OPT_VARIANT, 1)


def series(render, hits, n):
    ms = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); render.render(hits=hits); e1.record(); e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return np.array(ms)


def report(name, ms):
    print(f"{name}: frames 1-4 {ms[:4].round(3).tolist()}, 5-16 median {np.median(ms[4:16]):.4f}, 17-24 median {np.median(ms[16:24]):.4f}, "
          f"last 20 median {np.median(ms[-20:]):.4f} min {ms[-20:].min():.4f}", flush=True)


words = pkg.scenes.fractal(seed=0, max_depth=20, cam=(-0.9999, -0.9999, -0.9999), lod_c=3000.0, min_depth=4, max_words=120_000_000)
gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 20)
render = pkg.Render(gpu, (3840, 2160), words, capacity=words.size)
render.set_flags(pause_adaptive=True, shadows=False)
render.update(pkg.Settings(), pkg.Character((-0.9990, -0.9985, -0.9980), (-1.0, -1.2, -0.9)))
report("fractal depth 20, 4K primary", series(render, render.alloc_hits(3840 * 2160), a.frames))
gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)
z = np.load(os.path.join(ROOT, "tests", "golden", "monu9_vox.npz"))
words = pkg.CpuOctree.from_voxels(int(z["size"][0]), z["xyzi"], z["palette"]).to_octree_words()
render = pkg.Render(gpu, (1920, 1080), words, capacity=words.size)
render.set_flags(pause_adaptive=True, shadows=False)
render.update(pkg.Settings(), pkg.Character((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)))
report("monu9, 1080p", series(render, render.alloc_hits(1920 * 1080), a.frames))
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
render = pkg.Render(gpu, (1920, 1080), words, capacity=words.size)
render.set_flags(pause_adaptive=True, shadows=False)
render.update(pkg.Settings(), pkg.Character(cam, look))
report("terrain depth 16, 1080p (the benchmark frame)", series(render, render.alloc_hits(1920 * 1080), a.frames))
gpu.close()
