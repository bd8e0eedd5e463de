"""Developer probe: the first frame of a layout (no strip schedule yet), as bench.py's config.cold_frame_ms measures it, and the frame after it.
usage: [SVO_NO_DEFAULT_LISTS=1] python tools/cold_probe.py [--w 1920 --h 1080]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    a = ap.parse_args()
    pkg = entry.load_package()
    cam, look = pkg.scenes.terrain_camera(0, 16)
    words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
    gpu = pkg.Gpu(0)
    render = pkg.Render(gpu, (a.w, a.h), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(), pkg.Character(cam, look))
    gpu.set_option(pkg.gpu.OPT_TIMING, 1)
    hits = render.alloc_hits(a.w * a.h)
    for _ in range(40):
        render.render(hits=hits)
    cold, second = [], []
    for _ in range(9):
        gpu.set_option(pkg.gpu.OPT_SCHEDULE, 2)  # (setting the period forgets the schedule)
        render.render(hits=hits)
        cold.append(gpu.last_render_ms())
        render.render(hits=hits)
        second.append(gpu.last_render_ms())
    sig = int(np.bitwise_xor.reduce(hits.cpu().numpy().view(np.uint32).reshape(-1)))
    print(f"{a.w}x{a.h}: first frame of a layout {np.median(cold):.4f} ms (kernel), the frame after it {np.median(second):.4f} ms, default lists "
          f"{'off' if os.environ.get('SVO_NO_DEFAULT_LISTS') else 'on'}, sig {sig}")


if __name__ == "__main__":
    main()
