"""Developer probe: kernel time of each of the first 120 frames of the benchmark workload after start-up (the chip's clocks and
caches settle over the first ~40): why bench.py traces untimed pre-roll frames.  usage: python tools/ramp_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
render = pkg.Render(gpu, (1920,1080), words, capacity=words.size)
render.set_flags(pause_adaptive=True, shadows=False)
render.update(pkg.Settings(), pkg.Character(cam, look))
gpu.set_option(pkg.gpu.OPT_TIMING, 1)
hits = render.alloc_hits(1920*1080)
ms=[]
for i in range(120):
    render.render(hits=hits); ms.append(round(gpu.last_render_ms(),4))
print(ms[:40]); print(ms[40:80]); print(ms[80:])
