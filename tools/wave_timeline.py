"""Diagnostic: per-wave timeline of the STACK kernel (start / work-queue-dry / end) on the bench workload."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
import torch
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
render = pkg.Render(gpu, (W, H), words, capacity=words.size)
render.set_flags(pause_adaptive=True, shadows=False)
render.update(pkg.Settings(), pkg.Character(cam, look))
dbg = torch.zeros((16384, 8), dtype=torch.int32, device="cuda")
hits = render.alloc_hits(W * H)
for _ in range(3):
    render.render(hits=hits)
gpu.sync()
gpu.set_option(pkg.gpu.OPT_DEBUG_BUFFER, dbg.data_ptr())
gpu.set_option(pkg.gpu.OPT_TIMING, 1)
render.render(hits=hits)
ms = gpu.last_render_ms()
gpu.sync()
d = dbg.cpu().numpy().view(np.uint32)
d = d[d[:, 2] != 0]
t0 = d[:, 0].min()
start, dry, end = (d[:, 0] - t0) * 0.01, (d[:, 1] - t0) * 0.01, (d[:, 2] - t0) * 0.01  # us
print(f"kernel {ms * 1e3:.1f} us, waves {len(d)}")
pct = [0, 1, 5, 25, 50, 75, 95, 99, 100]
print("pct       ", pct)
print("start us  ", np.percentile(start, pct).round(1))
print("dry   us  ", np.percentile(dry, pct).round(1))
print("end   us  ", np.percentile(end, pct).round(1))
print("drain us  ", np.percentile(end - dry, pct).round(1))
print("rounds    ", np.percentile(d[:, 3], pct).round(0))
print("us/round  ", np.percentile((end - start) / np.maximum(d[:, 3], 1), pct).round(2))
print("active lanes per round (mean over waves):", (d[:, 4].sum() / d[:, 3].sum()).round(2), " refills/round", (d[:, 6].sum() / d[:, 3].sum()).round(3), " gens", int(d[:, 7].sum()), " total rounds", int(d[:, 3].sum()))
print("steps of the last rays of a wave, by end-time decile:", [int(np.median(d[np.argsort(end)][i * len(d) // 10:(i + 1) * len(d) // 10, 5])) for i in range(10)])
idx = np.argsort(end)[-8:]
print("latest waves: id, dry, end, rounds, steps of its last ray")
ids = np.flatnonzero(dbg.cpu().numpy().view(np.uint32)[:, 2] != 0)
for i in idx:
    print(int(ids[i]), round(float(dry[i]), 1), round(float(end[i]), 1), int(d[i, 3]), int(d[i, 5]))
