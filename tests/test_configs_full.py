"""BASELINE.json configs 2..5 at their STATED sizes on one GPU: whole frames against the oracle, plus the tile-sharded
form of the multi-GPU configs (4-way and 8-way splits traced rank by rank on the one GPU and reassembled).
Scenes and poses: tests/config_scenes.py."""
import os

import numpy as np
import pytest

import config_scenes as cs
from conftest import assert_hits_equal

pytestmark = pytest.mark.gpu

THREADS = os.cpu_count() or 8


def _oracle_uniforms(O, render):
    u = O.Uniforms()
    for f in ("camera", "camera_inverse", "dimensions", "sun_dir"):
        getattr(u, f)[:] = list(getattr(render.uniforms, f))
    u.flags, u.misc_value = render.uniforms.flags, render.uniforms.misc_value
    return u


def _reassemble(pkg, per_rank, W, H, tw, th, world, n_sets=1):
    """per_rank[r]: records of rank r's tiles (svo_render_tiles layout; n_sets record sets ray-major) -> [n_sets, H, W]."""
    tiles_x = W // tw
    frame = np.zeros((n_sets, H, W), dtype=pkg.HIT_DTYPE)
    for r, rec in enumerate(per_rank):
        rec = rec.reshape(n_sets, -1, th, tw)
        for k in range(rec.shape[1]):
            ty, tx = divmod(r + k * world, tiles_x)
            frame[:, ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = rec[:, k]
    return frame


def _full_frame_twice(pkg, gpu, render, n):
    """two frames into poisoned buffers (a dropped ray would keep the poison; the second frame runs on the schedule
    built from the first)"""
    out = []
    for _ in range(2):
        buf = render.alloc_hits(n)
        buf.fill_(-1)
        out.append(pkg.render.hits_to_numpy(render.render(hits=buf)))
        gpu.sync()
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)), "records depend on the strip schedule"
    return out[1]


def test_config2_monu9_1080p_all_poses(pkg, gpu, O):
    """configs[1]: files/monu9.vox, 1920x1080 primary rays, default pose + 3 orbit poses."""
    words, poses, (W, H) = cs.config2(pkg)
    assert words.size == 65184
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    for i, pose in enumerate(poses):
        render.update(pkg.Settings(), pkg.Character(*pose))
        got = _full_frame_twice(pkg, gpu, render, W * H)
        assert_hits_equal(got, O.trace_frame(words, _oracle_uniforms(O, render), threads=THREADS), f"config 2 pose {i}, 1080p")


def test_config3_standin_1080p_both_poses(pkg, gpu, O):
    """configs[2] stand-in (statuette.rsvo is absent): the 27.5 M-word block-instanced .rsvo shell at 1920x1080, from
    outside the shell and from inside it."""
    words, poses, (W, H) = cs.config3(pkg)
    assert words.size > 20_000_000 and pkg.scenes.max_depth(words) == 10
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    for i, pose in enumerate(poses):
        render.update(pkg.Settings(), pkg.Character(*pose))
        got = _full_frame_twice(pkg, gpu, render, W * H)
        want = O.trace_frame(words, _oracle_uniforms(O, render), threads=THREADS)
        assert_hits_equal(got, want, f"config 3 stand-in pose {i}, 1080p")
    assert ((want["info"] >> 16) & 1).mean() > 0.9  # the inside pose


def test_config4_terrain16_4k_and_4way_tiles(pkg, gpu, O):
    """configs[3]: procedural terrain depth 16 at 3840x2160: the whole frame on one GPU against the oracle, then the
    frame as 4 ranks (and 8) would trace it -- svo_render_tiles(first_tile = rank, stride = world) -- reassembled."""
    words, poses, (W, H) = cs.config4(pkg)
    assert words.size > 64 * 1024 * 1024
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(fov=90.0), pkg.Character(*poses[0]))
    got = _full_frame_twice(pkg, gpu, render, W * H)
    want = O.trace_frame(words, _oracle_uniforms(O, render), threads=THREADS)
    assert_hits_equal(got, want, "config 4, full 4K frame")
    tw, th = 64, 8
    for world in (4, 8):
        per_rank = []
        for r in range(world):
            per_rank.append(pkg.render.hits_to_numpy(render.render_tiles(tw, th, r, world)))
            gpu.sync()
        frame = _reassemble(pkg, per_rank, W, H, tw, th, world)[0]
        assert np.array_equal(frame.reshape(-1).view(np.uint32), want.reshape(-1).view(np.uint32)), f"{world}-way tile split"


def test_config5_fractal20_4k_four_secondary_rays(pkg, gpu, O):
    """configs[4]: depth-20 fractal (87 M words) at 3840x2160 with 4 secondary rays per hit pixel: primary and all four
    secondary record sets of the whole frame against the oracle, then the 8-way tile split reassembled."""
    words, poses, (W, H) = cs.config5(pkg)
    assert pkg.scenes.max_depth(words) == 20 and words.size > 80_000_000
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 20)
    try:
        render = pkg.Render(gpu, (W, H), words, capacity=words.size)
        render.set_flags(pause_adaptive=True, shadows=True)
        render.update(pkg.Settings(), pkg.Character(*poses[0]))
        n = W * H
        prim, sec = render.alloc_hits(n), render.alloc_hits(4 * n)
        for _ in range(2):
            prim.fill_(-1)
            sec.fill_(-1)
            render.render_secondary(4, hits=prim, secondary=sec)
            gpu.sync()
        oprim, osec = O.secondary_frame(words, _oracle_uniforms(O, render), 4, threads=THREADS)
        gprim, gsec = pkg.render.hits_to_numpy(prim), pkg.render.hits_to_numpy(sec)
        assert_hits_equal(gprim, oprim, "config 5 primary, 4K")
        assert_hits_equal(gsec, osec, "config 5, 4 secondary rays per hit pixel, 4K")
        assert 0.1 < ((oprim["info"] >> 16) & 1).mean() < 0.6
        del prim, sec
        tw, th, world = 64, 8, 8
        pr, sr = [], []
        for r in range(world):
            p, s = render.render_tiles_secondary(tw, th, r, world, 4)
            gpu.sync()
            pr.append(pkg.render.hits_to_numpy(p))
            sr.append(pkg.render.hits_to_numpy(s))
            del p, s
        fp = _reassemble(pkg, pr, W, H, tw, th, world)[0]
        fs = _reassemble(pkg, sr, W, H, tw, th, world, n_sets=4)
        assert np.array_equal(fp.reshape(-1).view(np.uint32), gprim.view(np.uint32)), "8-way tile split, primary"
        assert np.array_equal(fs.reshape(-1).view(np.uint32), gsec.view(np.uint32)), "8-way tile split, secondary"
    finally:
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)


def test_back_to_back_launches_with_growing_layouts(pkg, gpu, O):
    """Regression for the use-after-free fixed in round 1 (schedule / deferred-list / shading buffers were freed and
    reallocated while a launch still used them): frames of growing size, tile layouts and secondary-ray counts issued
    back to back WITHOUT a sync in between, so that every reallocation path runs with work in flight; then every
    frame is checked against the oracle."""
    words, poses, _ = cs.config2(pkg)
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    g2 = pkg.Gpu(0)  # a fresh context: its scratch buffers start empty, so each step below has to grow them
    try:
        jobs = []
        sizes = [(64, 64), (320, 192), (640, 384), (1280, 768), (1920, 1152)]
        render = pkg.Render(g2, sizes[0], words, capacity=words.size)
        render.set_flags(pause_adaptive=True, shadows=True)
        for (W, H) in sizes:
            render.resize((W, H))
            render.update(pkg.Settings(), pkg.Character(*poses[0]))
            u = _oracle_uniforms(O, render)
            jobs.append(("frame", u, None, render.render()))
            jobs.append(("tiles", u, (64, 64, 0, 3), render.render_tiles(64, 64, 0, 3)))
            jobs.append(("sec2", u, 2, render.render_secondary(2)))
            jobs.append(("sec4", u, 4, render.render_secondary(4)))
            rgba = render.alloc_rgba(W * H)
            jobs.append(("shaded", u, rgba, render.render(rgba=rgba)))
        g2.sync()
        for kind, u, arg, out in jobs:
            W, H = int(u.dimensions[0]), int(u.dimensions[1])
            if kind in ("frame", "shaded"):
                assert_hits_equal(pkg.render.hits_to_numpy(out), O.trace_frame(words, u, threads=THREADS), f"{kind} {W}x{H}")
                if kind == "shaded":
                    want = O.shade_frame(words, u, threads=THREADS)
                    want8 = np.floor(np.clip(want, 0, 1) * 255.0 + 0.5).astype(np.int32).reshape(-1, 4)
                    got = arg.cpu().numpy().view(np.uint8).reshape(-1, 4).astype(np.int32)
                    assert np.abs(got - want8).max() <= 1, f"shaded {W}x{H}"  # pow() differs by at most one code value
            elif kind == "tiles":
                tw, th, first, stride = arg
                want = O.trace_frame(words, u, threads=THREADS).reshape(H, W)
                got = pkg.render.hits_to_numpy(out).reshape(-1, th, tw)
                tiles_x = W // tw
                for k in range(got.shape[0]):
                    ty, tx = divmod(first + k * stride, tiles_x)
                    assert np.array_equal(got[k].view(np.uint32), want[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].copy().view(np.uint32)), \
                        f"tiles {W}x{H}, tile {k}"
            else:
                oprim, osec = O.secondary_frame(words, u, arg, threads=THREADS)
                assert_hits_equal(pkg.render.hits_to_numpy(out[0]), oprim, f"{kind} primary {W}x{H}")
                assert_hits_equal(pkg.render.hits_to_numpy(out[1]), osec, f"{kind} secondary {W}x{H}")
    finally:
        g2.close()
