"""Developer probe: the adaptive (hit counters live) trace on the benchmark tree, 1080p: kernel time per frame when every
frame starts from cleared counters (what the streaming loop does: the scan clears them) and when they carry over.
usage: python tools/count_probe.py [--reps 20] [--res 1920x1080]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--res", default="1920x1080")
ap.add_argument("--census", action="store_true")
ap.add_argument("--only", type=int, default=-1, help="0 static, 1 cleared, 2 carried (for a counter pass of one mode)")
a = ap.parse_args()
pkg = entry.load_package()
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
gpu.set_option(pkg.gpu.OPT_SCAN_CLEARS_COUNTERS, 1)
gpu.set_option(pkg.gpu.OPT_TIMING, 1)
W, H = (int(v) for v in a.res.split("x"))
render = pkg.Render(gpu, (W, H), words, capacity=words.size)
compute = pkg.Compute.new(gpu, render)
hits = render.alloc_hits(W * H)


def run(clear, flags):
    render.set_flags(**flags)
    render.update(pkg.Settings(), pkg.Character(cam, look))
    ms = []
    for i in range(a.reps + 3):
        if clear:
            compute.update(int(words.size))
            compute.read_lists()
        render.render(hits=hits)
        gpu.sync()
        if i >= 3:
            ms.append(gpu.last_render_ms())
    return float(np.median(ms)), float(min(ms))


def census():
    """What one frame from cleared counters leaves behind: how many words were touched, and how many increments that took."""
    render.set_flags(pause_adaptive=False, shadows=False)
    render.update(pkg.Settings(), pkg.Character(cam, look))
    compute.update(int(words.size)); compute.read_lists()
    render.render(hits=hits); gpu.sync()
    w = render.read_nodes()
    c = (w & 15).astype(np.int64)
    leaf = (w >> 4) >= (1 << 27)
    steps = (hits[:, 2] & 0xFF).sum().item()
    print(f"census: {int((c > 0).sum()):,} of {w.size:,} words touched ({int((leaf & (c > 0)).sum()):,} leaves), sum of counters "
          f"{int(c.sum()):,} (leaves {int(c[leaf].sum()):,}), saturated {int((c == 15).sum()):,} (leaves {int((leaf & (c == 15)).sum()):,}); "
          f"ray steps {steps:,}", flush=True)
    print("  counter histogram, leaves:  ", np.bincount(c[leaf], minlength=16).tolist())
    print("  counter histogram, interior:", np.bincount(c[~leaf], minlength=16).tolist(), flush=True)
    compute.update(int(words.size)); compute.read_lists()


if "--census" in sys.argv:
    census()
    sys.exit(0)
for name, clear, flags in (("static (counters paused)", False, dict(pause_adaptive=True, shadows=False)),
                           ("counters live, cleared before every frame", True, dict(pause_adaptive=False, shadows=False)),
                           ("counters live, carried over", False, dict(pause_adaptive=False, shadows=False)))[slice(None) if a.only < 0 else slice(a.only, a.only + 1)]:
    med, best = run(clear, flags)
    print(f"{name}: median {med:.3f} ms, min {best:.3f} ms per {W}x{H} frame", flush=True)
