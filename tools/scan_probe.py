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

# the reference's per-frame upload against its incremental form
gpu.set_option(pkg.gpu.OPT_SCAN_CLEARS_COUNTERS, 1)
render.set_flags(pause_adaptive=False, shadows=False)
render.update(pkg.Settings(), pkg.Character(cam, look))
render.render()
rng = np.random.default_rng(0)
idx = np.unique(rng.integers(0, words.size, 540_000)).astype(np.uint32)  # ~60 k subdivisions x 9 words
val = words[idx]
gpu.sync()
for name, fn in (("full re-upload (svo_nodes_write, 428 MB)", lambda: render.write_nodes(words)),
                 ("scan with counter reset + svo_nodes_scatter of 540 k words", lambda: (compute.update(int(words.size)), render.scatter_nodes(idx, val)))):
    ts = []
    for _ in range(4):
        render.render(); gpu.sync()
        t0 = time.perf_counter(); fn(); gpu.sync(); ts.append(time.perf_counter() - t0)
    print(f"{name}: {np.median(ts) * 1e3:.2f} ms", flush=True)
assert (render.read_nodes(words.size) & 15 == 0).all()

# steady-state adaptive frames (counters reset by the scan every frame, schedule from the previous frame)
ts = []
for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); render.render(); e1.record()
    compute.update(int(words.size)); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
    compute.read_lists()
print(f"adaptive trace, steady state: {np.median(ts[2:]):.3f} ms per 1080p frame", flush=True)
