"""Developer probe: counter scan (svo_scan_dispatch) throughput on the benchmark tree, after an adaptive frame."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
import torch
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
render = pkg.Render(gpu, (1920, 1080), words, capacity=words.size)
compute = pkg.Compute(gpu, render)
for mode, flags in (("counters all zero (static tree)", dict(pause_adaptive=True, shadows=False)),
                    ("after one adaptive frame", dict(pause_adaptive=False, shadows=False))):
    render.write_nodes(words)
    render.set_flags(**flags)
    render.update(pkg.Settings(), pkg.Character(cam, look))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); render.render(); e1.record(); e1.synchronize()
    t_trace = e0.elapsed_time(e1)
    ms = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); compute.update(int(words.size)); e1.record(); e1.synchronize()
        ms.append(e0.elapsed_time(e1))
        sub, unsub = compute.read_lists()
    m = float(np.median(ms[1:]))
    print(f"{mode}: trace {t_trace:.3f} ms; scan {m:.4f} ms = {words.nbytes / m / 1e6:.0f} GB/s over {words.nbytes / 1e6:.0f} MB; "
          f"lists {sub.size} / {unsub.size}", flush=True)
