"""Helper process of test_frame_pipeline_nccl_single_rank: the N > 1 code path of bench.py (three lanes on three HIP
streams, asynchronous RCCL gather, svo_assemble_tiles on rank 0) with a process group of ONE rank on one GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    port = sys.argv[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = entry.load_package()
    words = pkg.scenes.random_tree(seed=3, max_depth=8, p_split=0.6, p_solid=0.3, max_words=1 << 20)
    W, H, tw, th = 256, 128, 64, 8
    gpu = pkg.Gpu(0)
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(), pkg.Character((0.3, 0.4, -1.6), (-0.2, -0.3, 1.5)))
    want = pkg.render.hits_to_numpy(render.render()).view(np.uint32).reshape(H, W, 4)
    gpu.sync()
    lanes = [(gpu, render, torch.cuda.current_stream())]
    for _ in range(2):
        s = torch.cuda.Stream()
        g = pkg.Gpu(0, stream=s.cuda_stream)
        lanes.append((g, pkg.Render.share_nodes(g, render), s))
    traces = [(lambda buf, r=r: r.render_tiles(tw, th, 0, 1, hits=buf)) for _, r, _ in lanes]
    assemble = [(lambda g_, out, r=r: r.assemble_tiles(g_, tw, th, out=out)) for _, r, _ in lanes]
    pack = [(lambda rec, wire, r=r: r.pack_records(rec, wire)) for _, r, _ in lanes]
    ok = True
    for packers in (pack, None):  # 12-byte wire records (bench.py's default) and full records
        pipe = pkg.sharding.FramePipeline(traces, W, H, tw, th, 0, 1, "cuda:0", streams=[s for _, _, s in lanes],
                                          assemble=assemble, pack=packers)
        for k in range(8):
            frame = pipe.step()
            if frame is not None:
                torch.cuda.synchronize()
                ok = ok and bool(np.array_equal(frame.cpu().numpy().view(np.uint32), want))
        last = pipe.drain()
        torch.cuda.synchronize()
        ok = ok and bool(np.array_equal(last.cpu().numpy().view(np.uint32), want))
    dist.barrier()
    dist.destroy_process_group()
    print("PIPELINE_OK" if ok else "PIPELINE_MISMATCH")


if __name__ == "__main__":
    main()
