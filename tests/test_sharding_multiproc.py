"""The N > 1 path on CPU: two processes over gloo shard the frame's tiles round-robin, trace their own
tiles (here with the oracle standing in for the GPU kernel), do the ONE gather of the design, and rank 0
un-permutes.  Checks the sharding arithmetic and the collective the GPU ranks use with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, tw, th, out_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    O = entry.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        words = pkg.scenes.random_tree(seed=5, max_depth=6, p_split=0.5, p_solid=0.35, max_words=1 << 17)
        u = O.make_uniforms(width=W, height=H, flags=O.F_PAUSE_ADAPTIVE)
        sh = pkg.sharding
        n_pad = sh.padded_tile_count(W, H, tw, th, world)
        n_mine = sh.local_tile_count(W, H, tw, th, rank, world)
        local = torch.zeros((n_pad, th * tw, 4), dtype=torch.int32)
        tiles_x = W // tw
        for k in range(n_mine):  # the tiles svo_render_tiles(first_tile=rank, tile_stride=world) would trace
            t = rank + k * world
            ty, tx = divmod(t, tiles_x)
            rec = O.trace_frame(words, u, tile=(tx * tw, ty * th, tw, th)).reshape(-1)
            local[k] = torch.from_numpy(rec.view(np.uint32).reshape(-1, 4).view(np.int32).copy())
        g = sh.gather_frame(local, rank, world)
        ok = True
        if rank == 0:
            frame = sh.assemble_frame(g, W, H, tw, th).contiguous().numpy().view(np.uint32)
            want = O.trace_frame(words, u).reshape(-1).view(np.uint32).reshape(H, W, 4)
            ok = bool(np.array_equal(frame, want))
        else:
            assert g is None
        # the benchmark's double-buffered loop (bench.py, N > 1): three frames through FramePipeline
        calls = []

        def trace(buf):
            calls.append(1)
            buf.copy_(local + len(calls))  # frame k carries records + k, so stale buffers would show

        pipe = sh.FramePipeline(trace, W, H, tw, th, rank, world, "cpu")
        for k in range(3):
            done = pipe.step()
        last = pipe.drain()
        if rank == 0:
            got = last.numpy().view(np.uint32)
            ok = ok and bool(np.array_equal(got, want + np.uint32(3)))
        # three lanes (bench.py's default for N > 1: consecutive frames on different streams / contexts; no
        # streams on the CPU): 7 frames rotate over 3 buffers, each lane sees every third frame, in order
        seen = [[], [], []]
        count = [0]

        def lane_trace(lane):
            def trace(buf):
                count[0] += 1
                seen[lane].append(count[0])
                buf.copy_(local)
                buf[..., 0] += 100 * count[0]  # frame k is marked in the voxel-index word (words 2 and 3 stay consistent:
                return None                     # the 12-byte wire form rebuilds word 3 from word 2)
            return trace

        pipe = sh.FramePipeline([lane_trace(k) for k in range(3)], W, H, tw, th, rank, world, "cpu", pack=True)
        done = [pipe.step() for k in range(7)]
        last = pipe.drain()
        ok = ok and seen == [[1, 4, 7], [2, 5], [3, 6]]
        if rank == 0:
            def marked(n):
                m = want.copy()
                m[..., 0] += np.uint32(100 * n)
                return m
            # step k (0-based) first frees its buffer: the frame that used it, k - 3, is complete by then
            for k in range(3, 7):
                ok = ok and bool(np.array_equal(done[k].numpy().view(np.uint32), marked(k - 2)))
            ok = ok and done[2] is None
            ok = ok and bool(np.array_equal(last.numpy().view(np.uint32), marked(7)))
        # colour frames (bench.py --wire rgba8): one RGBA8 word per pixel travels instead of records
        colour = (local[..., 0:1] * 7 + 3).contiguous()  # any per-pixel word

        def colour_trace(buf):
            buf.copy_(colour)

        pipe = sh.FramePipeline(colour_trace, W, H, tw, th, rank, world, "cpu", words=1)
        for k in range(3):
            pipe.step()
        last = pipe.drain()
        if rank == 0:
            ok = ok and last.shape == (H, W, 1) and bool(np.array_equal(last.numpy().view(np.uint32)[..., 0], want[..., 0] * np.uint32(7) + np.uint32(3)))
            np.save(out_path, np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,tw,th", [(2, 96, 40, 32, 8), (3, 64, 48, 16, 8)])
def test_tile_sharded_frame_over_gloo(tmp_path, world, W, H, tw, th):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, tw, th, out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def test_sharding_arithmetic(pkg):
    sh = pkg.sharding
    assert [sh.local_tile_count(1920, 1080, 64, 8, r, 8) for r in range(8)] == [507, 507, 506, 506, 506, 506, 506, 506]
    assert sh.padded_tile_count(1920, 1080, 64, 8, 8) == 507
    assert sum(sh.local_tile_count(3840, 2160, 64, 8, r, 4) for r in range(4)) == 60 * 270
    g = torch.arange(2 * 3 * 4 * 4, dtype=torch.int32).reshape(2, 3, 4, 4)  # world 2, 3 slots, 2x2 tiles
    f = sh.assemble_frame(g, 4, 6, 2, 2)  # 2 x 3 tiles: tile t = slot * 2 + rank
    assert f.shape == (6, 4, 4)
    assert torch.equal(f[0:2, 2:4].reshape(4, 4), g[1, 0])  # tile 1 -> rank 1 slot 0
    assert torch.equal(f[2:4, 0:2].reshape(4, 4), g[0, 1])  # tile 2 -> rank 0 slot 1
