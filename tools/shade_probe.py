"""Developer probe: the full fs_main frame (trace + shadow rays + shading to RGBA8) on the benchmark tree."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
import torch
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
W, H = 1920, 1080
render = pkg.Render(gpu, (W, H), words, capacity=words.size)
hits = render.alloc_hits(W * H)
rgba = render.alloc_rgba(W * H)
def timed(fn, reps=30):
    for _ in range(4): fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for name, flags in (("records only, static", dict(pause_adaptive=True, shadows=False)),
                    ("RGBA, no shadows, static", dict(pause_adaptive=True, shadows=False)),
                    ("RGBA + shadow rays, static", dict(pause_adaptive=True, shadows=True)),
                    ("RGBA + shadow rays, counters live (the reference's default settings)", dict(pause_adaptive=False, shadows=True))):
    render.set_flags(**flags)
    render.update(pkg.Settings(), pkg.Character(cam, look))
    if name.startswith("records"):
        t = timed(lambda: render.render(hits=hits))
    else:
        t = timed(lambda: render.render(hits=hits, rgba=rgba))
    print(f"{name}: {t:.3f} ms per 1080p frame", flush=True)
