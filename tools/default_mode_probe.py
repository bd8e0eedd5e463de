"""Developer probe: the reference's DEFAULT frame (Uniforms::new, render.rs:306-321: pause_adaptive false, shadows true) on
the benchmark tree at 1080p, as the streaming loop runs it: counter scan (which also clears the counters it has read,
SVO_OPT_SCAN_CLEARS_COUNTERS) + list read-back, then trace with live hit counters + shadow rays + shading to RGBA8.
Meant to run under `rocprofv3 --kernel-trace --stats` / tools/pmc_profile.sh (PMC_PROG) as well as alone.
usage: python tools/default_mode_probe.py [--frames 30] [--fused 0|1|2] [--carry]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--fused", type=int, default=2)
ap.add_argument("--carry", action="store_true", help="counters carry over from frame to frame (no scan between frames)")
ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE for gpu.set_option (e.g. REFILL_MIN=8)")
a = ap.parse_args()
pkg = entry.load_package()
import torch  # noqa: E402
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
gpu.set_option(pkg.gpu.OPT_SCAN_CLEARS_COUNTERS, 1)
gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, a.fused)
gpu.set_option(pkg.gpu.OPT_TIMING, 4)
for o in a.opt:
    k, v = o.split("=")
    gpu.set_option(getattr(pkg.gpu, "OPT_" + k), int(v))
W, H = 1920, 1080
render = pkg.Render(gpu, (W, H), words, capacity=words.size)
compute = pkg.Compute.new(gpu, render)
render.set_flags(pause_adaptive=False, shadows=True)
render.update(pkg.Settings(), pkg.Character(cam, look))
hits, rgba = render.alloc_hits(W * H), render.alloc_rgba(W * H)
frame_ms, scan_ms, trace_ms = [], [], []
for i in range(a.frames + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not a.carry:
        compute.update(int(words.size))
        compute.read_lists()
    t1 = time.perf_counter()
    render.render(hits=hits, rgba=rgba)
    gpu.sync()
    t2 = time.perf_counter()
    if i >= 3:
        scan_ms.append((t1 - t0) * 1e3)
        frame_ms.append((t2 - t1) * 1e3)
        trace_ms.append(gpu.last_render_ms())
print(f"default mode ({'counters carry over' if a.carry else 'counters cleared by the scan every frame'}, fused shadows {a.fused}): "
      f"trace + shadow rays + shade {np.median(frame_ms):.3f} ms wall per 1080p frame (primary trace kernel {np.median(trace_ms):.3f} ms), "
      f"scan + list read-back {np.median(scan_ms):.3f} ms wall", flush=True)
