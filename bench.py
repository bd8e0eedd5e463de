#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the SVO ray-traversal hot path.

One "step" = one frame: every primary ray of the workload traced once by the HIP kernel
(svo_render / svo_render_tiles through the C ABI), node array resident in HBM, hit records left in
HBM.  With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the frame's tiles are
dealt round-robin to ranks and each step ends with ONE RCCL gather of hit records to rank 0 plus
the un-permute on rank 0 (strong scaling: the frame is fixed).

Workload (BASELINE.json metric "Mrays/sec at 1920x1080, depth-16 SVO"): deterministic LOD terrain,
max depth 16, ~107 M words (428 MB, larger than the 256 MiB Infinity Cache, under the 2^27-word
layout cap), camera standing on the terrain, 1920x1080 primary rays, static tree
(pause_adaptive), no shadows.  Data is synthetic (seeded generator, no reference counterpart).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (width, height, scene kwargs)
    "terrain16_1080p": dict(width=1920, height=1080, seed=0, max_depth=16, lod_c=1500.0, max_words=125_000_000),
    "terrain16_4k": dict(width=3840, height=2160, seed=0, max_depth=16, lod_c=1500.0, max_words=125_000_000),
}
# Mean algorithmic bytes per ray, B_ray = 4 * W_ray + 16, W_ray counted exactly by the
# oracle over the FULL frame of the workload (tools/ray_stats.py; stated in DESIGN.md).  bench.py
# re-measures it on the cpu_baseline sample and reports both.
ALGO_BYTES_PER_RAY = {"terrain16_1080p": 243.0228, "terrain16_4k": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="terrain16_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--tile-w", type=int, default=64)
    ap.add_argument("--tile-h", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: validation mode for boxes with fewer GPUs than ranks -- ranks share the visible GPUs and "
                         "the gather is staged through host memory (not a performance configuration)")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6],
                    help="consecutive frames rotate over this many HIP streams / device contexts, so the serial tail of one "
                         "frame's rays overlaps the next frames.  0 = default: 1 at N=1 (frames serial, which is what "
                         "roofline.* describes), 3 at N>1 (a rank's share of a sharded frame is too small to fill a GPU).  "
                         "More than 3 needs more hardware queues than ROCm's default of 4: GPU_MAX_HW_QUEUES=8 is set then")
    ap.add_argument("--wire", default="packed12", choices=["packed12", "full16"],
                    help="N>1: what the frame-end gather carries per ray: the 12-byte wire record (the fourth word of svo_hit "
                         "repeats bits of the third; rank 0 rebuilds it while un-permuting) or the full 16-byte record")
    ap.add_argument("--force-pipeline", action="store_true",
                    help="validation: run the N>1 code path (lanes, wire records, RCCL gather, assemble) with a one-rank "
                         "process group on a single GPU")
    ap.add_argument("--cpu-frac", type=int, default=1, help="cpu_baseline traces the top 1/n of the frame's rows")
    a = ap.parse_args()
    if a.frames_in_flight > 3:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: one hardware queue per lane

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if a.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    pipelined = world > 1 or a.force_pipeline
    if pipelined:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only the single-process validation mode gets here without a launcher
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # the library normally travels prebuilt; build it (rank 0) if it does not
    lib_path = os.path.join(ROOT, "octree-tracer_amd", "libsvo_hip.so")
    if not os.path.exists(lib_path) and rank == 0:
        entry.build()
    if pipelined:
        dist.barrier()
    pkg = entry.load_package()
    wl = WORKLOADS[a.workload]
    W, H = wl["width"], wl["height"]
    cam, look = pkg.scenes.terrain_camera(wl["seed"], wl["max_depth"])
    t0 = time.time()
    words = pkg.scenes.terrain(seed=wl["seed"], max_depth=wl["max_depth"], cam=cam, lod_c=wl["lod_c"],
                               max_words=wl["max_words"])
    gen_s = time.time() - t0

    gpu = pkg.Gpu(local_rank)
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(fov=90.0), pkg.Character(cam, look))
    gpu.set_option(pkg.gpu.OPT_TIMING, max(a.steps, 1))

    tw, th = a.tile_w, a.tile_h
    n_rays = W * H
    if a.frames_in_flight == 0:
        a.frames_in_flight = 1 if (not pipelined or a.backend == "gloo") else 3
    # lane 0 = the context above on torch's current stream; further lanes: own HIP stream + context, same node buffer
    lanes = [(gpu, render, torch.cuda.current_stream())]
    for _ in range(a.frames_in_flight - 1):
        s_k = torch.cuda.Stream()
        gpu_k = pkg.Gpu(local_rank, stream=s_k.cuda_stream)
        render_k = pkg.Render.share_nodes(gpu_k, render)
        gpu_k.set_option(pkg.gpu.OPT_TIMING, max(a.steps, 1))
        lanes.append((gpu_k, render_k, s_k))
    if not pipelined and len(lanes) > 1:
        bufs = [r.alloc_hits(n_rays) for _, r, _ in lanes]
        counter = [0]

        def step():
            k = counter[0] % len(lanes)
            counter[0] += 1
            lanes[k][1].render(hits=bufs[k])
            return bufs[k]
    elif not pipelined:
        hits = render.alloc_hits(n_rays)

        def step():
            render.render(hits=hits)
            return hits
    else:
        assert W % tw == 0 and H % th == 0
        # frame i's RCCL gather overlaps frame i+1's trace (double-buffered); rank 0 un-permutes each frame
        if a.backend == "nccl":
            traces = [(lambda buf, r=r: r.render_tiles(tw, th, rank, world, hits=buf)) for _, r, _ in lanes]
            assemble = [(lambda g, out, r=r: r.assemble_tiles(g, tw, th, out=out)) for _, r, _ in lanes]
            pack = [(lambda rec, wire, r=r: r.pack_records(rec, wire)) for _, r, _ in lanes] if a.wire == "packed12" else None
            pipe = pkg.sharding.FramePipeline(traces, W, H, tw, th, rank, world, f"cuda:{local_rank}",
                                              streams=[s for _, _, s in lanes], assemble=assemble, pack=pack)
        else:
            n_pad_v = pkg.sharding.padded_tile_count(W, H, tw, th, world)

            def via_host(r):
                dev_buf = torch.zeros((n_pad_v, th * tw, 4), dtype=torch.int32, device=f"cuda:{local_rank}")

                def trace(buf):
                    r.render_tiles(tw, th, rank, world, hits=dev_buf)
                    buf.copy_(dev_buf)  # blocking D2H on the lane's stream: validation only
                return trace

            pipe = pkg.sharding.FramePipeline([via_host(r) for _, r, _ in lanes], W, H, tw, th, rank, world, "cpu",
                                              streams=[s for _, _, s in lanes], pack=True if a.wire == "packed12" else None)
        step = pipe.step

    def barrier():
        torch.cuda.synchronize()
        if pipelined:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    if pipelined:
        pipe.drain()
    barrier()
    for g, _, _ in lanes:
        g.timing_collect()  # drop the warm-up launches' records
    t_start = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    if pipelined:
        out = pipe.drain()  # every one of the K frames is gathered and assembled inside the timed region
    barrier()
    elapsed = time.perf_counter() - t_start
    # per-launch kernel durations of exactly the K timed launches: HIP event pairs recorded by the C ABI
    # around each launch on the launch stream
    kms = np.concatenate([g.timing_collect() for g, _, _ in lanes])
    assert len(kms) == a.steps
    if pipelined:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gpu.sync()

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = n_rays * a.steps / elapsed / 1e6
        kernel_avg_ms = float(np.mean(kms))
        # rays this rank's launch traced (rank 0 owns the most tiles)
        rays_per_launch = n_rays if not pipelined else pkg.sharding.local_tile_count(W, H, tw, th, 0, world) * tw * th
        frame = out.reshape(-1, 4).cpu().numpy().view(np.uint32)
        result = {
            "metric": "Mrays/sec at 1920x1080, depth-16 SVO; achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32+u32", "data": "synthetic",
            "config": {"workload": a.workload, "width": W, "height": H, "octree_max_depth": wl["max_depth"],
                       "node_words": int(words.size), "node_bytes": int(words.size) * 4, "rays_per_step": n_rays,
                       "kernel_variant": "stack", "frames_in_flight": a.frames_in_flight, "backend": a.backend if pipelined else None, "sharding": "none" if not pipelined else f"tiles {tw}x{th} round-robin, 1 RCCL gather per frame ({12 if a.wire == 'packed12' else 16} B/ray) overlapped with the following frames' traces",
                       "scene_gen_s": round(gen_s, 1)},
        }
        cpu = None
        bytes_per_ray = ALGO_BYTES_PER_RAY.get(a.workload)
        if not a.no_cpu_baseline:
            O = entry.load_oracle()
            u = O.Uniforms()
            for f in ("camera", "camera_inverse", "dimensions", "sun_dir"):
                getattr(u, f)[:] = list(getattr(render.uniforms, f))
            u.flags, u.misc_value = render.uniforms.flags, render.uniforms.misc_value
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = max(1, min(cores, 64))
            # bounded sample: the top H / cpu_frac rows... a full frame is only seconds of CPU work, so by
            # default (cpu_frac = 1) the sample is the whole frame and the byte count below is exact
            rows = H // a.cpu_frac
            t0 = time.perf_counter()
            rec, st = O.trace_frame(words, u, tile=(0, 0, W, rows), stats=True, threads=cores)
            cpu_s = time.perf_counter() - t0
            rec = rec.reshape(-1)
            st = st.reshape(-1, 2).astype(np.float64)
            sample_bpr = float(4.0 * st[:, 1].mean() + 16.0)
            if bytes_per_ray is None or a.cpu_frac == 1:
                bytes_per_ray = sample_bpr
            # parity of the frame the GPU just produced, on the sampled rows (checker, not the product)
            got = frame.reshape(H, W, 4)[:rows].reshape(-1, 4)
            parity = bool(np.array_equal(got, rec.view(np.uint32).reshape(-1, 4)))
            cpu = {"value": round(len(rec) / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"rows 0..{rows - 1} of the same frame ({len(rec)} rays, {cpu_s:.1f} s), "
                             f"oracle/svo_oracle.c (restart-from-root algorithm of shader.wgsl), {cores} pthreads",
                   "sample_algo_bytes_per_ray": round(sample_bpr, 3), "w_restart_words_per_ray": round(float(st[:, 0].mean()), 2),
                   "gpu_frame_matches_oracle_on_sample": parity}
        if bytes_per_ray is not None:
            achieved = bytes_per_ray * rays_per_launch / (kernel_avg_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("workload") == a.workload and world == 1:
                    traffic = tj.get("hbm_bytes_per_launch")
            result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                                  "kernel": "trace_stack_kernel", "kernel_avg_ms": round(kernel_avg_ms, 4),
                                  "algo_bytes_per_ray": round(bytes_per_ray, 3), "rays_per_launch": rays_per_launch}
        if cpu is not None:
            result["cpu_baseline"] = cpu
        print(json.dumps(result), flush=True)
    if pipelined:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
