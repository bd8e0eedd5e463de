#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the SVO ray-traversal hot path.

One "step" = one frame: every primary ray of the workload traced once by the HIP kernel
(svo_render / svo_render_tiles through the C ABI), node array resident in HBM, hit records left in
HBM.  With --gpus N > 1 (one rank per GPU; started by torch.distributed.run, or by this script itself
as a child process when it is called as plain `python bench.py --gpus N`) the frame's tiles are dealt
round-robin to ranks and each step ends with ONE RCCL gather of hit records to rank 0 (svo_gather_frame,
behind the C ABI) plus the un-permute on rank 0 (strong scaling: the frame is fixed).

Workload (BASELINE.json metric "Mrays/sec at 1920x1080, depth-16 SVO"): deterministic LOD terrain,
max depth 16, ~107 M words (428 MB, larger than the 256 MiB Infinity Cache, under the 2^27-word
layout cap), camera standing on the terrain, 1920x1080 primary rays, static tree
(pause_adaptive), no shadows.  Data is synthetic (seeded generator, no reference counterpart).
The headline is that 1080p frame for EVERY N (an N-GPU headline on another frame is not comparable with the
1-GPU line).  Beside it the line carries, measured in the same run outside the timed region: the first frame of a layout (no
strip schedule yet), a moving camera, and the 4K frame of the same scene -- the frame BASELINE.json's 1 -> 8 GPU target refers
to -- under config.also; for N > 1 both frames come with the colour wire, the one-GPU rate and the link bound.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# vector-instruction issue peak (round 4, by wall clock: tools/issue_rate.hip, profiles/r04_issue_rate*.log; the microarch guide says
# the same): a SIMD issues a wave64 instruction of the fast group -- v_add / v_mul / v_fma_f32, v_add_u32, v_and / v_or / v_xor,
# right shifts, v_mov -- every 2 cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 2.  Everything else (compares, selects, min / max, conversions,
# v_bfe, shift-or and other three-operand integer forms, left shifts) takes 4 cycles, but an f32 instruction of the fast group
# issues beside one of those for nothing.  (Round 2's 1-instruction-per-cycle peak came from s_memtime deltas of a loop whose
# scalar overhead was left out)
VALU_PEAK_GINSTR_S = 256 * 4 * 2.4 / 2

WORKLOADS = {
    # name: (width, height, scene kwargs)
    "terrain16_1080p": dict(width=1920, height=1080, seed=0, max_depth=16, lod_c=1500.0, max_words=125_000_000),
    "terrain16_4k": dict(width=3840, height=2160, seed=0, max_depth=16, lod_c=1500.0, max_words=125_000_000),
}
# Mean algorithmic bytes per ray, B_ray = 4 * W_ray + 16, W_ray counted exactly by the
# oracle over the FULL frame of the workload (stated in DESIGN.md).  bench.py re-measures it on the
# cpu_baseline sample and reports both.
ALGO_BYTES_PER_RAY = {"terrain16_1080p": 243.0228, "terrain16_4k": 243.0108}  # tools/ray_stats.py


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited"""
    try:
        t = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if t[0] == "max" else float(t[0]) / float(t[1])
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        return None if q < 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    except (OSError, ValueError):
        return None


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (torch.distributed.run) before
    anything here has touched the GPU, relay what it prints and exit with its code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="terrain16_1080p", choices=sorted(WORKLOADS),
                    help="default: terrain16_1080p, the frame BASELINE.json's metric is quoted on, for every N; the 4K frame of the same "
                         "scene -- the one the 1 -> 8 GPU target refers to -- is measured beside it (config.also)")
    ap.add_argument("--tile-w", type=int, default=64)
    ap.add_argument("--tile-h", type=int, default=8)
    ap.add_argument("--preroll", type=int, default=120,
                    help="untimed frames traced once before the W warm-up steps of the headline measurement: the first frames after "
                         "start-up run 4 - 5 %% slower than the steady state the chip reaches after 40 - 80 frames (12 - 25 ms) of this kernel "
                         "(profiles/r03_ramp_probe.log); reported as config.preroll_frames.  0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold-frame / moving-camera / other-workload measurements")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: validation mode for boxes with fewer GPUs than ranks -- ranks share the visible GPUs and "
                         "the gather is staged through host memory (not a performance configuration)")
    ap.add_argument("--gather", default="abi", choices=["abi", "torch"],
                    help="N>1, backend nccl: the frame-end gather goes through the C ABI (svo_gather_frame: the library's own RCCL "
                         "communicator, one per lane) or through torch.distributed.gather")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6],
                    help="consecutive frames rotate over this many HIP streams / device contexts, so the serial tail of one "
                         "frame's rays overlaps the next frames.  0 = default: 1 at N=1 (frames serial, which is what "
                         "roofline.* describes), 3 at N>1 (a rank's share of a sharded frame is too small to fill a GPU).  "
                         "More than 3 needs more hardware queues than ROCm's default of 4: GPU_MAX_HW_QUEUES=8 is set then")
    ap.add_argument("--wire", default="packed12", choices=["packed12", "full16", "rgba8"],
                    help="N>1: what the frame-end gather carries per ray: the 12-byte wire record (the fourth word of svo_hit "
                         "repeats bits of the third; rank 0 rebuilds it while un-permuting), the full 16-byte record, or -- rgba8 -- "
                         "the shaded RGBA8 colour: the ranks shade their own tiles (fs_main, shadows as the uniforms say) and the "
                         "frame that travels is the image the reference displays, 4 bytes per ray (the records stay on the ranks)")
    ap.add_argument("--force-pipeline", action="store_true",
                    help="validation: run the N>1 code path (lanes, wire records, RCCL gather, assemble) with a one-rank "
                         "process group on a single GPU")
    ap.add_argument("--opt", action="append", default=[], help="developer: NAME=VALUE for svo_set_option on every lane (e.g. PAIR_TABLE=0)")
    ap.add_argument("--cpu-frac", type=int, default=1, help="cpu_baseline traces the top 1/n of the frame's rows")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    if a.frames_in_flight > 3:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: one hardware queue per lane

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if a.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} with backend nccl needs {world} GPUs, {torch.cuda.device_count()} visible "
                         "(--backend gloo shares the visible ones: validation only)")
    if a.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    pipelined = world > 1 or a.force_pipeline
    if pipelined:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only the single-process validation mode gets here without a launcher
            os.environ["MASTER_PORT"] = str(free_port())
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
    cam, look = pkg.scenes.terrain_camera(wl["seed"], wl["max_depth"])
    t0 = time.time()
    words = pkg.scenes.terrain(seed=wl["seed"], max_depth=wl["max_depth"], cam=cam, lod_c=wl["lod_c"],
                               max_words=wl["max_words"])
    gen_s = time.time() - t0

    gpu = pkg.Gpu(local_rank)
    render = pkg.Render(gpu, (wl["width"], wl["height"]), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    tw, th = a.tile_w, a.tile_h
    a.frames_in_flight_auto = a.frames_in_flight == 0
    if a.frames_in_flight == 0:
        a.frames_in_flight = 1 if (not pipelined or a.backend == "gloo") else 3
    # lane 0 = the context above on torch's current stream; further lanes: own HIP stream + context, same node buffer
    lanes = [(gpu, render, torch.cuda.current_stream())]
    for _ in range(a.frames_in_flight - 1):
        s_k = torch.cuda.Stream()
        gpu_k = pkg.Gpu(local_rank, stream=s_k.cuda_stream)
        lanes.append((gpu_k, pkg.Render.share_nodes(gpu_k, render), s_k))
    for o in a.opt:
        k, v = o.split("=")
        for g, _, _ in lanes:
            g.set_option(getattr(pkg.gpu, "OPT_" + k), int(v))
    gather_mode = None
    if pipelined and a.backend == "nccl":
        gather_mode = a.gather
        if gather_mode == "abi":
            # one RCCL communicator per lane behind the C ABI; rank 0's ids travel over the launcher's process group.
            # A rank on which the library cannot set one up (librccl missing, ...) says so before the collective call, and
            # all ranks then fall back to torch.distributed's gather together.
            ok = torch.tensor([1], dtype=torch.int32, device=f"cuda:{local_rank}")
            try:
                pkg.Gpu.comm_unique_id()  # (loads librccl)
            except pkg.SvoError:
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                gather_mode = "torch"
            else:
                for g, _, _ in lanes:
                    ids = [pkg.Gpu.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    g.comm_init_rank(ids[0], world, rank)

    def barrier():
        torch.cuda.synchronize()
        if pipelined:
            dist.barrier()
            torch.cuda.synchronize()

    lanes_all = lanes
    in_flight_used = {}

    def frame_loop(W, H, wire, serial_one_gpu=False):
        """(step, drain) for W x H frames of the scene on the lanes above; wire: what the frame-end gather of a sharded frame
        carries; serial_one_gpu: this rank traces the whole frame alone on lane 0 (the one-GPU reference of an N > 1 run)"""
        lanes = lanes_all
        for _, r, _ in lanes:
            r.resize((W, H))
            r.update(pkg.Settings(fov=90.0), pkg.Character(cam, look))
        n_rays = W * H
        if not pipelined and len(lanes) > 1 and not serial_one_gpu:
            bufs = [r.alloc_hits(n_rays) for _, r, _ in lanes]
            counter = [0]

            def step():
                k = counter[0] % len(lanes)
                counter[0] += 1
                lanes[k][1].render(hits=bufs[k])
                return bufs[k]
            return step, None
        if not pipelined or serial_one_gpu:
            hits = render.alloc_hits(n_rays)
            if serial_one_gpu and wire == "rgba8":  # the one-GPU comparator of the colour wire: trace + fs_main's shading, serial
                rgba = torch.empty((n_rays, 1), dtype=torch.int32, device=f"cuda:{local_rank}")

                def step():
                    render.render(hits=hits, rgba=rgba)
                    return rgba
                return step, None

            def step():
                render.render(hits=hits)
                return hits
            return step, None
        assert W % tw == 0 and H % th == 0
        # frame i's RCCL gather overlaps frame i+1's trace (double-buffered); rank 0 un-permutes each frame
        # (a rank's share of at least ~0.7 M rays fills the GPU with TWO frames in flight; a third only costs -- tools/pipeline_probe.py,
        # profiles/r05_pipeline_probe.log: world 8 at 4K 6.8 / 10.0 / 9.5 Grays/s per rank with 1 / 2 / 3, at 1080p 2.3 / 4.1 / 5.7)
        lanes = lanes_all[:2] if (len(lanes_all) == 3 and a.frames_in_flight_auto and (W * H) // world >= 700_000) else lanes_all
        in_flight_used[(W, H)] = len(lanes)
        if a.backend == "nccl" and wire == "rgba8":
            n_pad_c = pkg.sharding.padded_tile_count(W, H, tw, th, world)
            recs = [r.alloc_hits(n_pad_c * tw * th) for _, r, _ in lanes]  # this rank's records stay here
            traces = [(lambda buf, r=r, h=h: r.render_tiles(tw, th, rank, world, hits=h, rgba=buf)) for (_, r, _), h in zip(lanes, recs)]
            assemble = [(lambda g, out, r=r: r.assemble_tiles_rgba(g, tw, th, out=out)) for _, r, _ in lanes]
            gather = [((lambda send, recv, g=g: g.gather_frame(send, recv, 0)), g.gather_wait) for g, _, _ in lanes] \
                if gather_mode == "abi" else None
            pipe = pkg.sharding.FramePipeline(traces, W, H, tw, th, rank, world, f"cuda:{local_rank}",
                                              streams=[s for _, _, s in lanes], assemble=assemble, gather=gather, words=1)
        elif a.backend == "nccl":
            traces = [(lambda buf, r=r: r.render_tiles(tw, th, rank, world, hits=buf)) for _, r, _ in lanes]
            assemble = [(lambda g, out, r=r: r.assemble_tiles(g, tw, th, out=out)) for _, r, _ in lanes]
            pack = [(lambda rec, w12, r=r: r.pack_records(rec, w12)) for _, r, _ in lanes] if wire == "packed12" else None
            gather = [((lambda send, recv, g=g: g.gather_frame(send, recv, 0)), g.gather_wait) for g, _, _ in lanes] \
                if gather_mode == "abi" else None
            pipe = pkg.sharding.FramePipeline(traces, W, H, tw, th, rank, world, f"cuda:{local_rank}",
                                              streams=[s for _, _, s in lanes], assemble=assemble, pack=pack, gather=gather)
        else:
            # validation mode (gloo): ranks share the visible GPUs, the gather is staged through host memory
            n_pad_v = pkg.sharding.padded_tile_count(W, H, tw, th, world)
            colour = wire == "rgba8"

            def via_host(r):
                dev_rec = torch.zeros((n_pad_v, th * tw, 4), dtype=torch.int32, device=f"cuda:{local_rank}")
                dev_col = torch.zeros((n_pad_v, th * tw, 1), dtype=torch.int32, device=f"cuda:{local_rank}") if colour else None

                def trace(buf):
                    if colour:
                        r.render_tiles(tw, th, rank, world, hits=dev_rec, rgba=dev_col)
                        buf.copy_(dev_col)
                    else:
                        r.render_tiles(tw, th, rank, world, hits=dev_rec)
                        buf.copy_(dev_rec)  # blocking D2H on the lane's stream: validation only
                return trace

            pipe = pkg.sharding.FramePipeline([via_host(r) for _, r, _ in lanes], W, H, tw, th, rank, world, "cpu",
                                              streams=[s for _, _, s in lanes], pack=True if wire == "packed12" else None,
                                              words=1 if colour else 4)
        return pipe.step, pipe.drain

    preroll = [0 if (pipelined and a.backend == "gloo") else max(0, a.preroll)]  # (gloo: validation mode, not a performance configuration)

    def measure(W, H, steps, warmup, wire, serial_one_gpu=False):
        """K timed frames: (elapsed s [max over ranks], kernel ms of the K launches, last frame)"""
        step, drain = frame_loop(W, H, wire, serial_one_gpu)
        bar = torch.cuda.synchronize if serial_one_gpu else barrier  # (the one-GPU reference runs on rank 0 alone: no collective)
        for g, _, _ in lanes:
            g.set_option(pkg.gpu.OPT_TIMING, max(steps, 1))
        if preroll[0] > 0:  # once per process: the chip's clocks settle over the first ~40 frames (same count on every rank)
            for _ in range(preroll[0]):
                step()
            preroll[0] = 0
        for _ in range(warmup):
            step()
        if drain:
            drain()
        bar()
        for g, _, _ in lanes:
            g.timing_collect()  # drop the warm-up launches' records
        t_start = time.perf_counter()
        for _ in range(steps):
            out = step()
        if drain:
            out = drain()  # every one of the K frames is gathered and assembled inside the timed region
        bar()
        elapsed = time.perf_counter() - t_start
        # per-launch kernel durations of exactly the K timed launches: HIP event pairs recorded by the C ABI
        # around each launch on the launch stream
        kms = np.concatenate([g.timing_collect() for g, _, _ in lanes])
        assert len(kms) == steps, (len(kms), steps)
        if pipelined and not serial_one_gpu:
            t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, kms, out

    W, H = wl["width"], wl["height"]
    n_rays = W * H
    elapsed, kms, out = measure(W, H, a.steps, a.warmup, a.wire)
    gpu.sync()
    colour_wire = pipelined and a.wire == "rgba8"
    frame = out.reshape(-1, 1 if colour_wire else 4).cpu().numpy().view(np.uint32) if rank == 0 else None

    # ---- beside the headline, same run, outside its timed region ----
    extras = {}
    frame_rgba = None
    if not a.no_extras:
        k_extra = max(10, min(a.steps, 50))
        if not pipelined:
            # (1) the first frame of a layout: no strip schedule yet (screen order); the schedule is dropped before every sample
            hits = render.alloc_hits(n_rays)
            cold = []
            for _ in range(7):
                gpu.set_option(pkg.gpu.OPT_SCHEDULE, 2)  # (setting the period forgets the schedule)
                render.render(hits=hits)
                cold.append(gpu.last_render_ms())
            extras["cold_frame_ms"] = round(float(np.median(cold)), 4)
            # (2) a moving camera: 1 degree of yaw per frame, schedule rebuilt every 2nd frame (the default period)
            gpu.timing_collect()
            lx, ly, lz = look
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k_extra):
                ang = math.radians(1.0) * (i + 1)
                render.update(pkg.Settings(fov=90.0), pkg.Character(cam, (lx * math.cos(ang) + lz * math.sin(ang), ly,
                                                                          -lx * math.sin(ang) + lz * math.cos(ang))))
                render.render(hits=hits)
            torch.cuda.synchronize()
            extras["motion_ms"] = round((time.perf_counter() - t0) / k_extra * 1e3, 4)
            extras["motion_kernel_ms"] = round(float(np.mean(gpu.timing_collect())), 4)
            extras["motion"] = "1 degree of yaw per frame, strip schedule rebuilt every 2nd frame; wall time per frame incl. the schedule kernels"
            render.update(pkg.Settings(fov=90.0), pkg.Character(cam, look))
        # (3) the other frame size of the same scene and pose (the 4K frame is the one the 1 -> 8 GPU target refers to)
        other = "terrain16_4k" if a.workload == "terrain16_1080p" else "terrain16_1080p"
        ow = WORKLOADS[other]
        e2, k2, _ = measure(ow["width"], ow["height"], k_extra, 3, a.wire)
        extras["also"] = {"workload": other, "value": round(ow["width"] * ow["height"] * k_extra / e2 / 1e6, 2), "unit": "Mrays/s",
                          "steps": k_extra, "ms_per_step": round(e2 / k_extra * 1e3, 4), "kernel_avg_ms": round(float(np.mean(k2)), 4)}
        if ALGO_BYTES_PER_RAY.get(other) and not pipelined:
            extras["also"]["roofline_frac"] = round(ALGO_BYTES_PER_RAY[other] * ow["width"] * ow["height"] / (float(np.mean(k2)) * 1e-3) / 1e9
                                                    / HBM_PEAK_GBS, 5)

        def sharded_extras(Wx, Hx, name, keep_frame):
            """N > 1, one frame size: (4) the same sharded frame with the COLOUR wire -- the ranks shade their own tiles and the RGBA8
            image, what the reference displays (shader.wgsl:261-304, render.rs:246-248), is what travels: 4 bytes per ray instead
            of 12; (5) one GPU, the same frame, unsharded and serial (rank 0 alone, the others wait): what N GPUs are compared with;
            (6) what the links allow: every ray's wire bytes end on rank 0, (N - 1) / N of them over rank 0's N - 1 inbound xGMI links,
            one peer per link (7 links per GPU, ~64 GB/s per link and direction sustained: DESIGN.md 7)"""
            nonlocal frame_rgba
            nx = Wx * Hx
            res = {}
            if not colour_wire:
                e3, k3, out3 = measure(Wx, Hx, k_extra, 3, "rgba8")
                gpu.sync()
                if keep_frame:
                    frame_rgba = out3.reshape(-1, 1).cpu().numpy().view(np.uint32) if rank == 0 else None
                res["also_rgba8"] = {"workload": name, "wire": "rgba8: 4 B/ray (shaded RGBA8 colour; records stay on the ranks)",
                                     "value": round(nx * k_extra / e3 / 1e6, 2), "unit": "Mrays/s", "steps": k_extra,
                                     "ms_per_step": round(e3 / k_extra * 1e3, 4), "kernel_avg_ms": round(float(np.mean(k3)), 4)}
            if rank == 0:
                try:  # (rank 0 only, no collective inside: a failure here must not keep the other ranks waiting at the barrier below)
                    e1, k1, _ = measure(Wx, Hx, k_extra, 3, a.wire, serial_one_gpu=True)
                    res["one_gpu_same_workload"] = {"value": round(nx * k_extra / e1 / 1e6, 2), "unit": "Mrays/s", "steps": k_extra,
                                                    "ms_per_step": round(e1 / k_extra * 1e3, 4), "kernel_avg_ms": round(float(np.mean(k1)), 4),
                                                    "what": "rank 0 traces the whole frame alone, frames serial, same run"}
                except Exception as ex:  # noqa: BLE001
                    res["one_gpu_same_workload"] = {"error": repr(ex)}
                try:  # (the like-for-like denominator of also_rgba8: one GPU produces the same product, the shaded image -- VERDICT r4 next 4b)
                    e1c, k1c, _ = measure(Wx, Hx, k_extra, 3, "rgba8", serial_one_gpu=True)
                    res["one_gpu_same_workload_rgba8"] = {"value": round(nx * k_extra / e1c / 1e6, 2), "unit": "Mrays/s", "steps": k_extra,
                                                          "ms_per_step": round(e1c / k_extra * 1e3, 4), "kernel_avg_ms": round(float(np.mean(k1c)), 4),
                                                          "what": "rank 0 traces AND shades the whole frame alone (svo_render with rgba_out), frames serial, same run"}
                except Exception as ex:  # noqa: BLE001
                    res["one_gpu_same_workload_rgba8"] = {"error": repr(ex)}
            barrier()
            link_gbs = 64.0
            bound = {}
            for wname, bpr in (("packed12", 12), ("full16", 16), ("rgba8", 4)):
                per_link = nx * bpr / world  # bytes one peer sends per frame
                bound[wname] = {"bytes_into_rank0_per_frame": int(nx * bpr * (world - 1) // world), "bytes_per_link_per_frame": int(per_link),
                                "min_ms_per_frame": round(per_link / (link_gbs * 1e9) * 1e3, 4),
                                "max_mrays_s": round(nx / (per_link / (link_gbs * 1e9)) / 1e6, 1)}
            res["scaling_read_on"] = ("BASELINE.json's >= 6x at 1 -> 8 GPUs (4K frame: config.also) is read on the colour wire: also_rgba8.value / "
                                      "one_gpu_same_workload_rgba8.value -- the frame that travels is the image the reference displays; the record "
                                      "wire (the headline: value / one_gpu_same_workload.value) is link-bound below that, see link_bound")
            res["link_bound"] = {"assumed_gb_s_per_link_and_direction": link_gbs, "links_into_rank0": world - 1, **bound,
                                 "note": "an upper bound at perfect link efficiency; the gathers of consecutive frames overlap the traces"}
            return res

        if pipelined:
            extras.update(sharded_extras(W, H, a.workload, True))                        # the headline frame
            extras["also"].update(sharded_extras(ow["width"], ow["height"], other, False))  # the other frame (4K: the scaling target's)
        for _, r, _ in lanes:
            r.resize((W, H))
            r.update(pkg.Settings(fov=90.0), pkg.Character(cam, look))

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = n_rays * a.steps / elapsed / 1e6
        kernel_avg_ms = float(np.mean(kms))
        # rays this rank's launch traced (rank 0 owns the most tiles)
        rays_per_launch = n_rays if not pipelined else pkg.sharding.local_tile_count(W, H, tw, th, 0, world) * tw * th
        if not pipelined:
            sharding = "none"
        else:
            via = {"abi": "svo_gather_frame (RCCL behind the C ABI)", "torch": "torch.distributed.gather (RCCL)", None: "gloo via host memory"}[gather_mode]
            sharding = (f"tiles {tw}x{th} round-robin, 1 gather per frame through {via}, "
                        f"{ {'packed12': '12 B/ray (wire records)', 'full16': '16 B/ray (records)', 'rgba8': '4 B/ray (shaded RGBA8 colour; records stay on the ranks)'}[a.wire] }, "
                        "overlapped with the following frames' traces")
        result = {
            "metric": "Mrays/sec at 1920x1080, depth-16 SVO; achieved HBM GB/s vs peak",  # (BASELINE.json's name; config.workload says which frame)
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32+u32", "data": "synthetic",
            "config": {"workload": a.workload, "width": W, "height": H, "octree_max_depth": wl["max_depth"],
                       "node_words": int(words.size), "node_bytes": int(words.size) * 4, "rays_per_step": n_rays,
                       "kernel_variant": "stack", "frames_in_flight": in_flight_used.get((W, H), a.frames_in_flight),
                       "frames_in_flight_by_frame_size": {f"{k[0]}x{k[1]}": v for k, v in in_flight_used.items()} or None,
                       "preroll_frames": 0 if (pipelined and a.backend == "gloo") else max(0, a.preroll),
                       "preroll": "untimed frames before the warm-up steps, so that the K timed steps run at the chip's steady-state clocks",
                       "step_semantics": ("frames are serial: ms_per_step is one frame's latency and roofline.kernel_avg_ms one un-overlapped launch"
                                          if a.frames_in_flight == 1 else
                                          f"{a.frames_in_flight} frames in flight on separate HIP streams: ms_per_step is the interval between completed "
                                          "frames (throughput), not a frame's latency, and kernel durations are of overlapping launches"),
                       "backend": a.backend if pipelined else None, "gather": gather_mode, "sharding": sharding,
                       "scene_gen_s": round(gen_s, 1), **extras},
        }
        cpu = None
        bytes_per_ray = ALGO_BYTES_PER_RAY.get(a.workload)
        if not a.no_cpu_baseline:
            O = entry.load_oracle()
            u = O.Uniforms()
            for f in ("camera", "camera_inverse", "dimensions", "sun_dir"):
                getattr(u, f)[:] = list(getattr(render.uniforms, f))
            u.flags, u.misc_value = render.uniforms.flags, render.uniforms.misc_value
            host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            # every host core this process may USE: the affinity mask says 256 on the GPU box, its cgroup quota 16 CPUs -- and past the
            # quota more threads only get throttled (tools/cpu_scaling_probe.py, profiles/r04_cpu_scaling.log: 3.8 / 7.6 / 8.4 / 7.3 / 6.8
            # Mrays/s with 8 / 16 / 32 / 64 / 256 threads; round 3's 256-thread figure was the slow end of that curve)
            quota = cgroup_cpu_quota()
            cores = max(1, min(host_cores, 256, int(math.ceil(quota)) if quota else 256))
            # bounded sample: the top H / cpu_frac rows... a full frame is only seconds of CPU work, so by
            # default (cpu_frac = 1) the sample is the whole frame and the byte count below is exact
            rows = H // a.cpu_frac
            rec, st = O.trace_frame(words, u, tile=(0, 0, W, rows), stats=True, threads=cores)  # (also warms the host caches)
            reps, cpu_s = 0, 0.0
            while cpu_s < 3.0 and reps < 40:  # a frame is a fraction of a second on a many-core host: repeat it for a stable figure
                t0 = time.perf_counter()
                O.trace_frame(words, u, tile=(0, 0, W, rows), threads=cores)
                cpu_s += time.perf_counter() - t0
                reps += 1
            rec = rec.reshape(-1)
            st = st.reshape(-1, 2).astype(np.float64)
            sample_bpr = float(4.0 * st[:, 1].mean() + 16.0)
            if bytes_per_ray is None or a.cpu_frac == 1:
                bytes_per_ray = sample_bpr
            # parity of the frame the GPU just produced, on the sampled rows (checker, not the product)
            if colour_wire:
                # the assembled frame is the RGBA8 image: against the oracle's fs_main (pow() differs between libm and the GPU
                # by at most one code value per channel, tests/test_parity_gpu.py::test_shaded_frame)
                want = O.shade_frame(words, u, tile=(0, 0, W, rows), threads=cores)
                want8 = np.floor(np.clip(want, 0, 1) * 255.0 + 0.5).astype(np.int32).reshape(-1, 4)
                got8 = frame.reshape(H, W)[:rows].reshape(-1).view(np.uint8).reshape(-1, 4).astype(np.int32)
                parity = bool(np.abs(got8 - want8).max() <= 1)
            else:
                got = frame.reshape(H, W, 4)[:rows].reshape(-1, 4)
                parity = bool(np.array_equal(got, rec.view(np.uint32).reshape(-1, 4)))
            if extras.get("also_rgba8") is not None and frame_rgba is not None:
                want = O.shade_frame(words, u, tile=(0, 0, W, rows), threads=cores)
                want8 = np.floor(np.clip(want, 0, 1) * 255.0 + 0.5).astype(np.int32).reshape(-1, 4)
                got8 = frame_rgba.reshape(H, W)[:rows].reshape(-1).view(np.uint8).reshape(-1, 4).astype(np.int32)
                result["config"]["also_rgba8"]["image_matches_oracle_fs_main_within_1_code_value"] = bool(np.abs(got8 - want8).max() <= 1)
            cpu = {"value": round(len(rec) * reps / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores,
                   "host_cores_available": host_cores, "host_cores_total": os.cpu_count(), "cgroup_cpu_quota": quota, "kind": "port",
                   "sample": f"rows 0..{rows - 1} of the same frame ({len(rec)} rays) traced {reps} times ({cpu_s:.1f} s), "
                             f"oracle/svo_oracle.c (restart-from-root algorithm of shader.wgsl), {cores} pthreads (persistent pool, "
                             "256-pixel spans from an atomic cursor)",
                   "sample_algo_bytes_per_ray": round(sample_bpr, 3), "w_restart_words_per_ray": round(float(st[:, 0].mean()), 2),
                   "gpu_frame_matches_oracle_on_sample": parity}
        if bytes_per_ray is not None:
            achieved = bytes_per_ray * rays_per_launch / (kernel_avg_ms * 1e-3) / 1e9
            traffic = valu = salu = None
            tj = {}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("workload") == a.workload and world == 1:
                    traffic = tj.get("hbm_bytes_per_launch")
                    valu = tj.get("valu_wave_instructions_per_launch")
                    salu = tj.get("salu_wave_instructions_per_launch")
            # `bound` names the contractual ceiling (BASELINE.json: fraction of the HBM-read roofline on ALGORITHMIC bytes); what
            # the frame time actually follows is a balance of instruction issue and dependent-load latency: memory-side traffic is
            # a fraction of the algorithmic bytes (L2 absorbs the shared ancestors), see roofline_valu and DESIGN.md 4.7 / 6
            result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                                  "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (not this run)",
                                  "achieved_algorithmic_gbs": round(achieved, 2),
                                  "measured_hbm_gbs": round(traffic / (kernel_avg_ms * 1e-3) / 1e9, 2) if traffic else None,
                                  "measured_limiter": "not HBM bandwidth: the latency of the walk's dependent loads (half of a wave's time: most wave-loads have a line "
                                                      "that misses the 32 KB L1, and every line an XCD's L2 sees for the first time costs its wave a trip to the "
                                                      "memory side) against seven waves per SIMD, and a wave's serial instruction issue "
                                                      "(DESIGN.md 4.8, 6; profiles/r05_*)",
                                  # (the headline is the best of the three frames a renderer sees -- static camera, schedule from this very view,
                                  # shares settled: the same fraction for the first frame of a layout and for a camera in motion, same run)
                                  "frac_cold": (round(bytes_per_ray * rays_per_launch / (extras["cold_frame_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                                if extras.get("cold_frame_ms") else None),
                                  "frac_motion": (round(bytes_per_ray * rays_per_launch / (extras["motion_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                                  if extras.get("motion_kernel_ms") else None),
                                  "kernel": "trace_stack_kernel", "kernel_avg_ms": round(kernel_avg_ms, 4),
                                  "algo_bytes_per_ray": round(bytes_per_ray, 3), "rays_per_launch": rays_per_launch}
            if valu:
                ach = valu / (kernel_avg_ms * 1e-3) / 1e9
                salu_ps = (salu / (kernel_avg_ms * 1e-3) / 1e9) if salu else None
                # the clock the chip HOLDS inside this kernel (timeline build: s_memtime against s_memrealtime in every wave), not the 2.4 GHz
                # it can reach: VERDICT r4 weak 3
                clock = tj.get("shader_clock_ghz_in_kernel") or 2.4
                peak = 256 * 4 * clock / 2
                result["roofline_valu"] = {"bound": "valu-issue", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
                                           "frac": round(ach / peak, 4), "instructions_per_launch": valu,
                                           "shader_clock_ghz_in_kernel": clock, "peak_at_2p4_ghz": VALU_PEAK_GINSTR_S,
                                           "issue_ledger": tj.get("issue_ledger"),
                                           "salu_instructions_per_launch": salu, "salu_g_instr_s": round(salu_ps, 1) if salu_ps else None,
                                           "slow_group_share_static": tj.get("valu_slow_group_share_static"),
                                           "source": "profiles/traffic.json: SQ_INSTS_VALU / SQ_INSTS_SALU of this workload's launch (rocprofv3 --pmc passes, not this "
                                                     "run); peak = 1024 SIMDs x one wave64 instruction of the 2-cycle group per 2 cycles x the clock held in this kernel "
                                                     "(tools/issue_rate.hip by wall clock, profiles/r04_issue_rate.log; MI355X_MICROARCH.md: v_fma_f32 2 cycles). "
                                                     "frac is a LOWER bound of how busy the issue port is: instructions of the 4-cycle group count as one here "
                                                     "(slow_group_share_static: their share in the hot loop's ISA, profiles/r04_hot_loop_isa_table.md), while f32 "
                                                     "instructions issued beside them cost nothing"}
        if cpu is not None:
            result["cpu_baseline"] = cpu
        print(json.dumps(result), flush=True)
    if pipelined:
        dist.barrier()
        for g, _, _ in lanes:
            g.close()  # (communicators go before the process group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
