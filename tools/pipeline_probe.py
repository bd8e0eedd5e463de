"""Developer probe: one rank's share of a tile-sharded frame (rank 0 of `world`) traced with F frames in flight
(F HIP streams / contexts over one node buffer), no gather: how far pipelining hides the per-frame latency.
usage: python tools/pipeline_probe.py [--world 8] [--w 1920 --h 1080] [--frames 300]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="8,4,2,1")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--inflight", default="1,2,3,4")
    a = ap.parse_args()
    pkg = entry.load_package()
    import torch
    cam, look = pkg.scenes.terrain_camera(0, 16)
    words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
    W, H, tw, th = a.w, a.h, 64, 8
    gpu0 = pkg.Gpu(0)
    r0 = pkg.Render(gpu0, (W, H), words, capacity=words.size)
    r0.set_flags(pause_adaptive=True, shadows=False)
    r0.update(pkg.Settings(), pkg.Character(cam, look))
    lanes = [(gpu0, r0, torch.cuda.current_stream())]
    for _ in range(max(int(x) for x in a.inflight.split(",")) - 1):
        s = torch.cuda.Stream()
        g = pkg.Gpu(0, stream=s.cuda_stream)
        r = pkg.Render.share_nodes(g, r0)
        lanes.append((g, r, s))
    for world in [int(x) for x in a.world.split(",")]:
        n_mine = pkg.sharding.local_tile_count(W, H, tw, th, 0, world)
        bufs = [r0.alloc_hits(n_mine * tw * th) for _ in lanes]
        for F in [int(x) for x in a.inflight.split(",")]:
            def run(n):
                for i in range(n):
                    g, r, s = lanes[i % F]
                    r.render_tiles(tw, th, 0, world, hits=bufs[i % F])
            run(20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(a.frames)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.frames
            print(f"world {world} (rank 0: {n_mine * tw * th} rays)  frames in flight {F}: {dt * 1e3:.4f} ms/frame  "
                  f"{n_mine * tw * th / dt / 1e9:.2f} Grays/s per rank", flush=True)


if __name__ == "__main__":
    main()
