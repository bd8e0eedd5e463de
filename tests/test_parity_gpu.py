"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_hits_equal, set_uniforms_from_oracle

pytestmark = pytest.mark.gpu

VARIANTS = [0, 1]  # RESTART, STACK (the two round-3 experiments left the library: tools/experiments/)


def _render(pkg, gpu, words, u, variant, capacity=None, tile=None):
    render = pkg.Render(gpu, (int(u.dimensions[0]), int(u.dimensions[1])), words,
                        capacity=capacity or max(words.size, 64))
    set_uniforms_from_oracle(render, u)
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    n = int(u.dimensions[0]) * int(u.dimensions[1]) if tile is None else tile[2] * tile[3]
    # output buffers are poisoned first: a ray that is dropped (never traced, record never written) must show
    buf = render.alloc_hits(n)
    buf.fill_(-1)
    first = pkg.render.hits_to_numpy(render.render(hits=buf, tile=tile))
    gpu.sync()
    # the second frame of the same layout is scheduled from the first one's step counts (longest rays
    # first); the records must not depend on the schedule
    buf2 = render.alloc_hits(n)
    buf2.fill_(-1)
    hits = render.render(hits=buf2, tile=tile)
    gpu.sync()
    second = pkg.render.hits_to_numpy(hits)
    assert np.array_equal(first.view(np.uint32), second.view(np.uint32)), "records depend on the strip schedule"
    return second


@pytest.mark.parametrize("variant", VARIANTS)
def test_config1_small_vox_golden(pkg, gpu, O, small_words, variant):
    """Config 1 (small.vox, 256x256, default camera) against the committed golden records."""
    z = np.load(f"{GOLDEN}/config1_small_256.npz")
    u = O.make_uniforms(width=256, height=256, flags=O.F_PAUSE_ADAPTIVE)
    u.camera[:] = z["camera"].tolist()
    u.camera_inverse[:] = z["camera_inverse"].tolist()
    got = _render(pkg, gpu, small_words, u, variant)
    want = z["hits"].view(pkg.HIT_DTYPE).reshape(-1)
    assert_hits_equal(got, want, "config1 golden")
    assert_hits_equal(got, O.trace_frame(small_words, u, threads=4), "config1 oracle")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("pose", [((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)), ((1.3, 0.9, 1.2), (-1.0, -0.6, -1.0)),
                                  ((0.02, 0.31, 0.05), (0.3, -0.2, 1.0)), ((-1.6, 0.1, 0.2), (1.0, 0.0, 0.0))])
def test_monu9_frame(pkg, gpu, O, monu9_words, variant, pose):
    """Config 2 geometry at 480x270 (the oracle finishes in seconds); poses outside, inside, axis-aligned."""
    u = O.make_uniforms(pos=pose[0], look=pose[1], width=480, height=270, flags=O.F_PAUSE_ADAPTIVE)
    got = _render(pkg, gpu, monu9_words, u, variant)
    assert_hits_equal(got, O.trace_frame(monu9_words, u, threads=8), f"monu9 pose {pose}")


@pytest.mark.parametrize("variant", VARIANTS)
def test_monu9_golden_samples_1080p(pkg, gpu, O, monu9_words, variant):
    """Config 2 at its full 1920x1080 size: 4096 committed oracle samples."""
    z = np.load(f"{GOLDEN}/config2_monu9_samples.npz")
    u = O.make_uniforms(width=1920, height=1080, flags=O.F_PAUSE_ADAPTIVE)
    u.camera[:] = z["camera"].tolist()
    u.camera_inverse[:] = z["camera_inverse"].tolist()
    got = _render(pkg, gpu, monu9_words, u, variant).reshape(1080, 1920)
    sel = got[z["py"], z["px"]]
    assert_hits_equal(sel, z["hits"].view(pkg.HIT_DTYPE).reshape(-1), "monu9 1080p samples")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("misc_bool", [False, True])
def test_random_trees(pkg, gpu, O, variant, misc_bool):
    """Seeded random sparse trees (depth 9), both tie-break modes (shader.wgsl:138-150)."""
    flags = O.F_PAUSE_ADAPTIVE | (O.F_MISC_BOOL if misc_bool else 0)
    for seed in (1, 2):
        words = pkg.scenes.random_tree(seed=seed, max_depth=9, p_split=0.45, p_solid=0.25, max_words=1 << 20)
        u = O.make_uniforms(pos=(0.3, 0.4, -1.7), look=(-0.1, -0.2, 1.0), width=320, height=200, flags=flags)
        got = _render(pkg, gpu, words, u, variant)
        assert_hits_equal(got, O.trace_frame(words, u, threads=8), f"random tree seed {seed}")


@pytest.mark.parametrize("variant", VARIANTS)
def test_deep_terrain_depth16(pkg, gpu, O, variant):
    """Depth-16 LOD terrain (the benchmark scene family at reduced size), camera inside the cube."""
    cam, look = pkg.scenes.terrain_camera(0, 16)
    words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=150.0, max_words=4_000_000)
    assert pkg.scenes.max_depth(words) == 16
    u = O.make_uniforms(pos=cam, look=look, width=384, height=216, flags=O.F_PAUSE_ADAPTIVE)
    got = _render(pkg, gpu, words, u, variant)
    assert_hits_equal(got, O.trace_frame(words, u, threads=8), "terrain depth 16")


@pytest.mark.parametrize("variant", VARIANTS)
def test_explicit_rays_edge_cases(pkg, gpu, O, monu9_words, variant):
    """Axis-aligned rays (zero direction components, shader.wgsl:193-194 vs :67-72), rays that miss,
    rays starting inside, on faces, with NaN/inf components, plus random rays."""
    import torch
    rng = np.random.default_rng(5)
    rays = [
        [0.5, 0.5, -3, 0, 0, 1], [0.5, 0.5, -3, 0, 0, -1], [-0.5, -0.5, -3, 0, 0, 1], [0.1, 0.2, 0.3, 1, 0, 0],
        [0.1, 0.2, 0.3, 0, -1, 0], [-1.0, 0.0, 0.0, 1, 0, 0], [1.0, 0.0, 0.0, -1, 0, 0], [0.0, 0.0, 0.0, 0, 0, 0],
        [0, 5, 0, 0, -1, 0], [0, 5, 0, 0, 1, 0], [3, 3, 3, -1, -1, -1], [-1, -1, -1, 1, 1, 1],
        [np.nan, 0, 0, 1, 0, 0], [0, 0, -2, np.nan, 0, 1], [0, 0, -2, 0, 0, np.inf], [np.inf, 0, 0, -1, 0, 0],
        [0.25, 0.25, -1.0000001, 0, 0, 1], [0.999999, 0.999999, 0.999999, 1, 1, 1],
    ]
    rnd = np.concatenate([rng.uniform(-2, 2, (4000, 3)), rng.normal(size=(4000, 3))], axis=1)
    rnd[:, 3:] /= np.linalg.norm(rnd[:, 3:], axis=1, keepdims=True)
    inside = np.concatenate([rng.uniform(-1, 1, (4000, 3)), rng.normal(size=(4000, 3))], axis=1)
    axis = inside.copy()
    axis[:, 3:] = np.eye(3)[rng.integers(0, 3, 4000)] * rng.choice([-1.0, 1.0], (4000, 1))
    rays = np.concatenate([np.array(rays, dtype=np.float32), rnd.astype(np.float32), inside.astype(np.float32),
                           axis.astype(np.float32)])
    render = pkg.Render(gpu, (8, 8), monu9_words, capacity=monu9_words.size)
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for flags in (O.F_PAUSE_ADAPTIVE, O.F_PAUSE_ADAPTIVE | O.F_MISC_BOOL):
        render.uniforms.flags = flags
        render.upload_uniforms()
        got = pkg.render.hits_to_numpy(render.trace_rays(torch.from_numpy(rays).cuda()))
        gpu.sync()
        assert_hits_equal(got, O.trace_rays(monu9_words, rays, flags=flags, threads=8), f"explicit rays flags={flags}")


def test_known_answers_single_level(pkg, gpu, O):
    """SURVEY 8c KATs 4-7 on the device: one-level tree with only child 7 solid."""
    import torch
    words = np.array([0x80000000] * 7 + [0x8FF00000], dtype=np.uint32)
    rays = np.array([[0.5, 0.5, -3, 0, 0, 1], [0.5, 0.5, -3, 0, 0, -1], [-0.5, -0.5, -3, 0, 0, 1]], dtype=np.float32)
    render = pkg.Render(gpu, (8, 8), words, capacity=64)
    render.uniforms.flags = O.F_PAUSE_ADAPTIVE
    render.upload_uniforms()
    for variant in VARIANTS:
        gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
        h = pkg.render.hits_to_numpy(render.trace_rays(torch.from_numpy(rays).cuda()))
        gpu.sync()
        assert h["value"].tolist() == [7, 0, 0x20202000]
        assert h["t"].tolist() == [3.0, 0.0, 4.0]
        assert (h["info"] & 0xFF).tolist() == [1, 0, 1]          # steps
        assert ((h["info"] >> 8) & 0xFF).tolist() == [1, 0, 1]   # depth
        assert ((h["info"] >> 16) & 1).tolist() == [1, 0, 0]     # hit
        assert h["normal_bits"].tolist() == [2 << 4, 0, 0]       # (0, 0, -1)


@pytest.mark.parametrize("variant", VARIANTS)
def test_hand_derived_known_answers(pkg, gpu, O, variant):
    """tests/kat_cases.py on the device, through svo_trace_rays: the records derived on paper from the WGSL (two-level descent,
    multi-axis ties, a start on a centre plane under both tie-break modes, the 100 / 101 step boundary) -- the same literals the
    oracle is pinned with."""
    import torch
    import kat_cases as K
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for name, words, flags, rays, expected in K.cases():
        render = pkg.Render(gpu, (8, 8), words, capacity=max(words.size, 64))
        render.uniforms.flags = flags
        render.upload_uniforms()
        h = pkg.render.hits_to_numpy(render.trace_rays(torch.from_numpy(rays).cuda()))
        gpu.sync()
        K.check(h, expected, f"{name} (variant {variant})")


@pytest.mark.parametrize("variant", VARIANTS)
def test_hand_derived_hit_counters(pkg, gpu, O, variant):
    """tests/kat_cases.py counter_case on the device: the node words a 1 x 1 frame of one hand-derived ray leaves behind, frame after
    frame up to saturation, against the counters worked out on paper (shader.wgsl:157-161)."""
    import kat_cases as K
    words, cinv, record, after = K.counter_case()
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    u = O.make_uniforms(width=1, height=1, flags=0)  # counters live, no shadow ray
    u.camera_inverse[:] = cinv.tolist()
    render = pkg.Render(gpu, (1, 1), words, capacity=64)
    set_uniforms_from_oracle(render, u)
    for n in range(1, 10):
        hits = pkg.render.hits_to_numpy(render.render())
        gpu.sync()
        K.check(hits, [record], f"the 1 x 1 frame's ray, frame {n}")
        assert np.array_equal(render.read_nodes(words.size), after(n)), f"counters after {n} frames (variant {variant})"


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("fused_shadows", [0, 1])
def test_hand_derived_shading(pkg, gpu, O, variant, fused_shadows):
    """tests/kat_cases.py shading_cases on the device: the RGBA8 pixel of fs_main for a lit hit (with and without its shadow ray)
    and for a miss, against the colours worked out on paper (one code value of slack for pow)."""
    import kat_cases as K
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, fused_shadows)
    try:
        for name, words, flags, cinv, sun, rgba in K.shading_cases():
            u = O.make_uniforms(width=1, height=1, flags=flags, sun_dir=sun)
            u.camera_inverse[:] = cinv.tolist()
            render = pkg.Render(gpu, (1, 1), words, capacity=64)
            set_uniforms_from_oracle(render, u)
            hits, img = render.render_host(rgba=True)
            got = img.reshape(4).astype(np.int32)
            assert np.abs(got - np.array(rgba)).max() <= 1 and got[3] == rgba[3], f"{name} (variant {variant}): {got.tolist()} against {rgba}"
    finally:
        gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, 2)


@pytest.mark.parametrize("variant", VARIANTS)
def test_step_limit_and_malformed(pkg, gpu, O, variant):
    """>100 steps sentinel (shader.wgsl:242-244) and a malformed array (zero words = a pointer cycle):
    the kernel must terminate and report the sentinel like the oracle."""
    z = np.load(f"{GOLDEN}/config1_small_256.npz")
    want = z["hits"].view(pkg.HIT_DTYPE).reshape(-1)
    assert (want["value"] == 0xFF000000).any(), "fixture should contain step-limit rays"
    words = np.zeros(64, dtype=np.uint32)  # every word is 'interior -> group 0'
    u = O.make_uniforms(width=64, height=64, flags=O.F_PAUSE_ADAPTIVE)
    if variant >= 1:
        # the STACK kernels resolve 16 (24) levels; deeper trees are refused loudly (the oracle's guard is at depth 31)
        render = pkg.Render(gpu, (64, 64), words, capacity=64)
        set_uniforms_from_oracle(render, u)
        gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
        hits = render.render()
        with pytest.raises(pkg.SvoError):
            gpu.sync()
        h = pkg.render.hits_to_numpy(hits)
        inside = h["value"] != 0
        assert (h["value"][inside] == 0xFF000000).all()
    else:
        got = _render(pkg, gpu, words, u, 0)
        assert_hits_equal(got, O.trace_frame(words, u, threads=2), "malformed array")


@pytest.mark.parametrize("variant", VARIANTS)
def test_unaligned_child_groups(pkg, gpu, O, variant):
    """A node array whose child groups are not 8-aligned (nothing in the layout forbids it, LAYOUT.md; Octree::subdivide
    never makes one): child indices are ADDED to the group's index, never or-ed into it -- same records as the oracle."""
    V = 134217728
    words = np.zeros(64, dtype=np.uint32)
    words[0:8] = (V + 0) << 4                    # root group: empty leaves ...
    words[3] = 12 << 4                           # ... except child 3 -> a group at word 12 (unaligned)
    words[6] = (V + 0x00FF00) << 4               # and a solid child
    words[12:20] = [(V + (0x10 * (i + 1) if i % 3 == 0 else 0)) << 4 for i in range(8)]
    words[17] = 28 << 4                          # one level further down, unaligned again
    words[28:36] = [(V + (0x2000 + i if i % 2 else 0)) << 4 for i in range(8)]
    seen = set()
    for pos, look in (((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)), ((0.3, -0.4, 0.2), (0.5, 0.7, -0.3)), ((-1.8, 1.1, 0.4), (1.0, -0.6, -0.2))):
        u = O.make_uniforms(pos=pos, look=look, width=96, height=64, flags=O.F_PAUSE_ADAPTIVE)
        got = _render(pkg, gpu, words, u, variant)
        want = O.trace_frame(words, u, threads=2)
        seen |= set(np.unique(want["value"]).tolist())
        assert_hits_equal(got, want, f"unaligned child groups, pose {pos}")
    assert {6, 12, 29} <= seen, "the poses should hit leaves of all three groups"


CULL_POSES = [  # (pos, look): all outside the cube
    ((0.2, 0.1, -1.2), (0.0, 0.0, 1.0)),            # close: the cube fills most of the frame
    ((0.3, 0.2, -3.0), (0.0, 0.0, 1.0)),            # a cube with sky all around
    ((2.5, 1.5, -6.0), (-0.4, -0.25, 1.0)),         # far
    ((10.0, 8.0, -30.0), (-0.31, -0.25, 1.0)),      # very far: the cube covers a few blocks
    ((0.0, 1.0005, -2.0), (0.0, 0.0, 1.0)),         # grazing the face y = 1: rays nearly parallel to it, just above
    ((0.0, 1.3, -2.0), (0.0, -0.02, 1.0)),          # looking along the top face from slightly above
    ((2.0, 2.0, -2.0), (-1.0, -1.0, 1.0)),          # a cube corner pointing at the camera (three faces, six silhouette edges)
    ((2.0, 2.0, -2.0), (-1.0, -0.62, 1.0)),         # the same corner near the frame's edge: partly visible cube
    ((1.5, 0.0, 0.0), (1.0, 0.0, 0.0)),             # looking away from the cube, which lies close behind: every ray misses it (no single
                                                    # side plane of a block's cone separates it: blocks are not culled, records still zero)
    ((6.0, 0.5, 0.0), (1.0, 0.1, 0.0)),             # the same from further away
]


@pytest.mark.parametrize("variant", [1])
def test_culling_pass_forced(pkg, gpu, O, monu9_words, variant):
    """SVO_OPT_CULL = 1: the pre-trace pass that writes the all-zero records of 64-pixel blocks whose rays all miss the cube
    (ray_box_dist returns 0, shader.wgsl:66-80,197-205) and keeps them out of the trace -- the one place where a wrong decision
    would fabricate output.  Whole frames against the oracle: camera outside at several distances, grazing a face, a corner on
    the silhouette, odd frame sizes with partial blocks, sub-rectangles, tile sharding, shaded frames (the pass also zeroes
    t_current / shadow records); and the decisions themselves: every culled block's oracle records are 'never entered'."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    gpu.set_option(pkg.gpu.OPT_CULL, 1)
    try:
        culled_total = 0
        for W, H in ((256, 144), (250, 131), (67, 45)):
            for pos, look in CULL_POSES:
                u = O.make_uniforms(pos=pos, look=look, width=W, height=H, flags=O.F_PAUSE_ADAPTIVE)
                want = O.trace_frame(monu9_words, u, threads=8)
                got = _render(pkg, gpu, monu9_words, u, variant).reshape(H, W)
                assert_hits_equal(got, want, f"cull forced, {W}x{H}, pose {pos}")
                # the decisions of the second of _render's two frames
                bpr, bpc = (W + 7) // 8, (H + 7) // 8
                cls = gpu.strip_classes(bpr * bpc).reshape(bpc, bpr)
                never = (want["value"] == 0) & ((want["info"] & 0xFF) == 0) & (((want["info"] >> 16) & 1) == 0)
                for by, bx in np.argwhere(cls == 0xFF):
                    assert never[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].all(), f"block ({bx},{by}) was culled but a ray of it enters the cube"
                culled_total += int((cls == 0xFF).sum())
        assert culled_total > 1000, "the pass should have culled blocks on these poses"
        # sub-rectangles and tile sharding (svo_render_tiles: the pass works per tile rectangle)
        W, H = 256, 144
        u = O.make_uniforms(pos=(0.3, 0.2, -3.0), look=(0.0, 0.0, 1.0), width=W, height=H, flags=O.F_PAUSE_ADAPTIVE)
        full = O.trace_frame(monu9_words, u, threads=8)
        render = pkg.Render(gpu, (W, H), monu9_words, capacity=monu9_words.size)
        set_uniforms_from_oracle(render, u)
        for tile in [(3, 5, 61, 37), (96, 40, 90, 77), (255, 143, 1, 1), (0, 7, 256, 1)]:
            buf = render.alloc_hits(tile[2] * tile[3])
            buf.fill_(-1)
            got = pkg.render.hits_to_numpy(render.render(hits=buf, tile=tile))
            gpu.sync()
            x0, y0, w, h = tile
            assert_hits_equal(got, full[y0:y0 + h, x0:x0 + w], f"cull forced, rect {tile}")
        tw, th, ranks = 64, 8, 3
        frame = np.zeros((H, W), dtype=pkg.HIT_DTYPE)
        for r in range(ranks):
            got = pkg.render.hits_to_numpy(render.render_tiles(tw, th, r, ranks)).reshape(-1, th, tw)
            gpu.sync()
            for k in range(got.shape[0]):
                ty, tx = divmod(r + k * ranks, W // tw)
                frame[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = got[k]
        assert_hits_equal(frame, full, "cull forced, tile-sharded frame")
        # shaded frames with shadow rays, fused into the primary launch and as a second launch
        for fused in (1, 0):
            gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, fused)
            for pos, look in (CULL_POSES[1], CULL_POSES[6]):
                us = O.make_uniforms(pos=pos, look=look, width=240, height=136, flags=O.F_PAUSE_ADAPTIVE | O.F_SHADOWS)
                rs = pkg.Render(gpu, (240, 136), monu9_words, capacity=monu9_words.size)
                set_uniforms_from_oracle(rs, us)
                for _ in range(2):
                    hits, img = rs.render_host(rgba=True)
                assert_hits_equal(hits, O.trace_frame(monu9_words, us, threads=8), f"cull forced, shaded, fused={fused}")
                want8 = np.floor(np.clip(O.shade_frame(monu9_words, us, threads=8), 0, 1) * 255.0 + 0.5).astype(np.int32)
                assert np.abs(img.astype(np.int32) - want8).max() <= 1
    finally:
        gpu.set_option(pkg.gpu.OPT_CULL, 2)
        gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, 2)


@pytest.mark.parametrize("variant", VARIANTS)
def test_tiles_and_rectangles(pkg, gpu, O, monu9_words, variant):
    """svo_render on sub-rectangles (ragged sizes) and svo_render_tiles sharding reassemble the frame."""
    W, H = 256, 144
    u = O.make_uniforms(width=W, height=H, flags=O.F_PAUSE_ADAPTIVE)
    full = O.trace_frame(monu9_words, u, threads=8)
    render = pkg.Render(gpu, (W, H), monu9_words, capacity=monu9_words.size)
    set_uniforms_from_oracle(render, u)
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for tile in [(0, 0, W, H), (3, 5, 61, 37), (200, 100, 56, 44), (255, 143, 1, 1), (0, 7, 256, 1)]:
        got = pkg.render.hits_to_numpy(render.render(tile=tile))
        gpu.sync()
        x0, y0, w, h = tile
        assert_hits_equal(got, full[y0:y0 + h, x0:x0 + w], f"rect {tile}")
    tw, th, ranks = 64, 8, 3
    tiles_x = W // tw
    frame = np.zeros((H, W), dtype=pkg.HIT_DTYPE)
    for r in range(ranks):
        got = pkg.render.hits_to_numpy(render.render_tiles(tw, th, r, ranks)).reshape(-1, th, tw)
        gpu.sync()
        for k in range(got.shape[0]):
            t = r + k * ranks
            ty, tx = divmod(t, tiles_x)
            frame[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = got[k]
    assert_hits_equal(frame, full, "tile-sharded frame")
    host = render.render_host(tile=(3, 5, 61, 37))
    assert_hits_equal(host, full[5:42, 3:64], "render_host")


def test_scan_kernel(pkg, gpu, O):
    """Counter scan (compute.wgsl:26-47) against the oracle, as sets (the append order is unordered)."""
    rng = np.random.default_rng(11)
    n = 300_000
    ptr = np.where(rng.random(n) < 0.4, rng.integers(0, 1 << 20, n), pkg.VOXEL_OFFSET + rng.integers(0, 1 << 24, n))
    ptr[rng.random(n) < 0.05] = pkg.VOXEL_OFFSET  # empty leaves
    words = ((ptr.astype(np.uint64) << 4) | rng.integers(0, 16, n).astype(np.uint64)).astype(np.uint32)
    words[rng.random(n) < 0.02] = 0
    render = pkg.Render(gpu, (8, 8), words, capacity=n + 1000)
    compute = pkg.Compute(gpu, render)
    for node_length in (n, n // 3, 0):
        compute.update(node_length)
        sub, unsub = compute.read_lists()
        osub, ounsub = O.scan(words, node_length=node_length)
        assert sorted(sub.tolist()) == osub[1:1 + osub[0]].tolist()
        assert sorted(unsub.tolist()) == ounsub[1:1 + ounsub[0]].tolist()
    compute.update(n)  # counters were reset by the read (adaptive.rs:23,87)
    sub2, _ = compute.read_lists()
    assert sub2.size == O.scan(words)[0][0]


def test_scan_list_overflow_and_ragged_lengths(pkg, gpu, O):
    """More candidates than a list holds: the read-back shows MAX - 1 entries (adaptive.rs:22,86), every one a real
    candidate, none twice; the other list stays exact.  Plus lengths that are not multiples of the kernel's chunking."""
    rng = np.random.default_rng(12)
    n = 4_200_017
    ptr = np.where(rng.random(n) < 0.6, rng.integers(1, 1 << 20, n), pkg.VOXEL_OFFSET + 1 + rng.integers(0, 1 << 24, n))
    cnt = np.where(ptr < pkg.VOXEL_OFFSET, (rng.random(n) > 0.2).astype(np.int64), rng.integers(0, 16, n))  # a fifth of the interior nodes cold
    words = ((ptr.astype(np.uint64) << 4) | cnt.astype(np.uint64)).astype(np.uint32)
    render = pkg.Render(gpu, (8, 8), words, capacity=n)
    compute = pkg.Compute(gpu, render)
    osub, ounsub = O.scan(words, capacity=n + 1)  # exact candidate sets
    want_sub, want_unsub = set(osub[1:1 + osub[0]].tolist()), set(ounsub[1:1 + ounsub[0]].tolist())
    assert len(want_unsub) < pkg.compute.MAX_UNSUBDIVISIONS_PER_FRAME < len(want_sub)
    compute.update(n)
    sub, unsub = compute.read_lists()
    assert sub.size == pkg.compute.MAX_SUBDIVISIONS_PER_FRAME - 1
    assert np.unique(sub).size == sub.size and set(sub.tolist()) <= want_sub
    assert set(unsub.tolist()) == want_unsub and unsub.size == len(want_unsub)
    for node_length in (1, 3, 255, 8191, 8193, 100_003):
        compute.update(node_length)
        sub, unsub = compute.read_lists()
        assert set(sub.tolist()) == {i for i in want_sub if i < node_length}
        assert sorted(unsub.tolist()) == sorted(i for i in want_unsub if i < node_length)


def test_assemble_tiles_kernel(pkg, gpu):
    """svo_assemble_tiles (rank 0's un-permute after the gather) equals the torch expression the CPU tests use."""
    import torch
    for world, (W, H, tw, th) in ((1, (128, 64, 64, 8)), (3, (192, 40, 64, 8)), (8, (1920, 1080, 64, 8)), (5, (96, 48, 32, 16))):
        n_pad = pkg.sharding.padded_tile_count(W, H, tw, th, world)
        g = torch.randint(-2 ** 31, 2 ** 31 - 1, (world, n_pad, th * tw, 4), dtype=torch.int32, device="cuda")
        render = pkg.Render(gpu, (W, H), np.array([0x80000000] * 8, dtype=np.uint32), capacity=64)
        got = render.assemble_tiles(g, tw, th)
        gpu.sync()
        want = pkg.sharding.assemble_frame(g, W, H, tw, th).contiguous()
        assert torch.equal(got, want), (world, W, H)
        # 12-byte wire records: pack kernel == torch expression, packed assemble rebuilds word 3 from word 2
        wire = render.pack_records(g, torch.empty((world, n_pad, th * tw, 3), dtype=torch.int32, device="cuda"))
        gpu.sync()
        assert torch.equal(wire, pkg.sharding.pack_records(g))
        got = render.assemble_tiles(wire, tw, th)
        gpu.sync()
        want = pkg.sharding.assemble_frame(wire, W, H, tw, th).contiguous()
        assert torch.equal(got, want) and torch.equal(got[..., 3], (got[..., 2] >> 17) & 63)
    with pytest.raises(pkg.SvoError):
        render.assemble_tiles(g[:1], tw, th)  # fewer tiles than the frame has


def test_assemble_tiles_rgba_kernel(pkg, gpu):
    """svo_assemble_tiles_rgba against the torch expression of the sharding module (colour frames, one word per pixel)."""
    import torch
    W, H, tw, th, world = 192, 48, 64, 8, 5
    n_pad = pkg.sharding.padded_tile_count(W, H, tw, th, world)
    g = torch.randint(-2**31, 2**31 - 1, (world, n_pad, th * tw), dtype=torch.int32, device="cuda")
    render = pkg.Render(gpu, (W, H), np.zeros(8, dtype=np.uint32), capacity=64)
    got = render.assemble_tiles_rgba(g, tw, th)
    gpu.sync()
    want = pkg.sharding.assemble_frame(g.unsqueeze(-1), W, H, tw, th)[..., 0]
    assert torch.equal(got, want)


def test_frame_pipeline_nccl_single_rank(pkg, gpu):
    """bench.py's N > 1 loop (lanes on separate HIP streams, async RCCL gather, svo_assemble_tiles) with a one-rank
    process group: every completed frame equals the directly rendered one.  (More ranks need more GPUs; the rank
    arithmetic is covered over gloo in test_sharding_multiproc.py.)"""
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pipeline_world1.py")
    r = subprocess.run([sys.executable, script, str(port)], capture_output=True, text=True, timeout=300)
    assert "PIPELINE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_error_paths(pkg, gpu):
    """Call-order and argument errors come back as statuses, not aborts."""
    g = pkg.Gpu(0)
    with pytest.raises(pkg.SvoError):
        g.check(pkg._lib.lib().svo_scan_dispatch(g._h, 8))  # no node buffer yet
    render = pkg.Render(g, (16, 16), np.array([0x80000000] * 8, dtype=np.uint32), capacity=64)
    with pytest.raises(pkg.SvoError):
        render.render()  # uniforms not set
    render.update(pkg.Settings(), pkg.Character())
    with pytest.raises(pkg.SvoError):
        render.render(tile=(8, 8, 16, 16))  # outside the frame
    with pytest.raises(pkg.SvoError):
        render.write_nodes(np.zeros(100, dtype=np.uint32))  # past capacity
    with pytest.raises(pkg.SvoError):
        g.set_option(pkg.gpu.OPT_VARIANT, 7)
    g.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_hit_counters_adaptive_mode(pkg, gpu, O, small_words, monu9_words, variant):
    """pause_adaptive off: every word a primary ray's descent visits gets +1, saturating at 15
    (shader.wgsl:157-161); the device words after a frame equal the oracle's order-independent count.  Both kernels:
    RESTART walks those words, STACK reconstructs their addresses from its tables and stack."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for words, size in ((small_words, (96, 64)), (monu9_words, (160, 90)), (monu9_words, (640, 360))):
        u = O.make_uniforms(width=size[0], height=size[1], flags=0)  # adaptive on, shadows off
        render = pkg.Render(gpu, size, words, capacity=words.size + 64)
        set_uniforms_from_oracle(render, u)
        hits = pkg.render.hits_to_numpy(render.render())
        gpu.sync()
        assert_hits_equal(hits, O.trace_frame(words, u, threads=4), "adaptive-mode hits")
        want = O.count_frame(words, u)
        got = render.read_nodes(words.size)
        assert np.array_equal(got >> 4, words >> 4), "pointers must not change"
        assert np.array_equal(got, want)
        # second frame keeps counting from the first frame's state
        render.render()
        gpu.sync()
        assert np.array_equal(render.read_nodes(words.size), O.count_frame(want, u))
        # the scan then sees the hot leaves (counter >= 4) and the never-visited interior nodes (counter == 0)
        compute = pkg.Compute(gpu, render)
        compute.update(int(words.size))
        sub, unsub = compute.read_lists()
        now = render.read_nodes(words.size)
        osub, ounsub = O.scan(now)
        assert sorted(sub.tolist()) == osub[1:1 + osub[0]].tolist()
        assert sorted(unsub.tolist()) == ounsub[1:1 + ounsub[0]].tolist()
        assert osub[0] > 0 or words.size > 1000  # (monu9 leaves are too small to collect 4 visits at this size)


def test_hit_counters_large_tree_both_stacks(pkg, gpu, O):
    """Counting on the ancestor stack at scale: a depth-14 terrain (camera inside, long rays, deep restarts) and a
    depth-18 fractal through the deep-stack instantiation; counters after two frames equal the oracle's, records too."""
    cam, look = pkg.scenes.terrain_camera(0, 14)
    cases = [(pkg.scenes.terrain(seed=0, max_depth=14, cam=cam, lod_c=400.0, max_words=6_000_000), cam, look, 14),
             (pkg.scenes.fractal(seed=1, max_depth=18, cam=(-0.999, -0.999, -0.999), lod_c=300.0, min_depth=4, max_words=4_000_000),
              (-0.9990, -0.9985, -0.9980), (-1.0, -1.2, -0.9), 18)]
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    try:
        for words, pos, lookv, depth in cases:
            gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, depth)
            u = O.make_uniforms(pos=pos, look=lookv, width=320, height=180, flags=0)  # adaptive on, shadows off
            render = pkg.Render(gpu, (320, 180), words, capacity=words.size)
            set_uniforms_from_oracle(render, u)
            want = words
            for frame in range(2):
                hits = pkg.render.hits_to_numpy(render.render())
                gpu.sync()
                assert_hits_equal(hits, O.trace_frame(words, u, threads=8), f"depth {depth} frame {frame}")
                want = O.count_frame(want, u)
                got = render.read_nodes(words.size)
                assert np.array_equal(got, want), f"depth {depth} frame {frame}: {np.count_nonzero(got != want)} counters differ"
            assert (want & 15).max() == 15 and np.count_nonzero(want & 15) > 1000
    finally:
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)


def test_debug_show_hits_mode(pkg, gpu, O, small_words):
    """pause_adaptive && show_hits: the hit test reads the counter bits (shader.wgsl:220-224)."""
    u0 = O.make_uniforms(width=64, height=64, flags=0)
    counted = O.count_frame(small_words, u0)
    u = O.make_uniforms(width=64, height=64, flags=O.F_PAUSE_ADAPTIVE | O.F_SHOW_HITS)
    render = pkg.Render(gpu, (64, 64), counted, capacity=counted.size)
    set_uniforms_from_oracle(render, u)
    for variant in VARIANTS:
        gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
        got = pkg.render.hits_to_numpy(render.render())
        gpu.sync()
        assert_hits_equal(got, O.trace_frame(counted, u, threads=4), "show_hits debug mode")


@pytest.mark.parametrize("variant", VARIANTS)
def test_shaded_frame(pkg, gpu, O, monu9_words, small_words, variant):
    """fs_main end to end (shader.wgsl:261-304): ambient + Lambert, shadow ray, gamma, debug views.
    pow() differs between libm and the GPU, so the RGBA8 image may differ by one code value; the hit
    records produced on the way stay bit-exact."""
    F = O
    cases = [
        (monu9_words, F.F_PAUSE_ADAPTIVE | F.F_SHADOWS, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))),
        (monu9_words, F.F_PAUSE_ADAPTIVE | F.F_SHADOWS, ((0.02, 0.31, 0.05), (0.3, -0.2, 1.0))),
        (monu9_words, F.F_PAUSE_ADAPTIVE, ((1.3, 0.9, 1.2), (-1.0, -0.6, -1.0))),
        (monu9_words, F.F_PAUSE_ADAPTIVE | F.F_SHADOWS | F.F_MISC_BOOL, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))),
        (small_words, F.F_PAUSE_ADAPTIVE | F.F_SHOW_STEPS, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))),
        (small_words, F.F_PAUSE_ADAPTIVE | F.F_SHADOWS, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))),
    ]
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for words, flags, pose in cases:
        u = O.make_uniforms(pos=pose[0], look=pose[1], width=240, height=136, flags=flags)
        render = pkg.Render(gpu, (240, 136), words, capacity=words.size)
        set_uniforms_from_oracle(render, u)
        for frame in range(2):  # the second frame runs with both schedules (primary + shadow rays) in place
            hits, img = render.render_host(rgba=True)
        assert_hits_equal(hits, O.trace_frame(words, u, threads=8), f"hits under shading flags={flags}")
        want = O.shade_frame(words, u, threads=8)
        want8 = np.floor(np.clip(want, 0, 1) * 255.0 + 0.5).astype(np.int32)
        diff = np.abs(img.astype(np.int32) - want8)
        assert diff[..., 3].max() == 0 and (img[..., 3] == 128).all()
        assert diff.max() <= 1, f"flags={flags}: colour off by {diff.max()}"
        assert (diff == 0).mean() > 0.98
        # the shadowed / lit split must agree exactly wherever the lit colour is not black
        assert img[..., :3].any()


@pytest.mark.parametrize("variant", VARIANTS)
def test_shaded_frame_counts_shadow_rays(pkg, gpu, O, small_words, variant):
    """Adaptive mode + shadows: the shadow ray passes primary = true (shader.wgsl:276), so it bumps counters too."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    u = O.make_uniforms(width=80, height=48, flags=O.F_SHADOWS)
    render = pkg.Render(gpu, (80, 48), small_words, capacity=small_words.size)
    set_uniforms_from_oracle(render, u)
    render.render_host(rgba=True)
    assert np.array_equal(render.read_nodes(small_words.size), O.count_frame(small_words, u))


def test_fused_shadow_rays_match_two_pass(pkg, gpu, O, monu9_words, small_words):
    """SVO_OPT_FUSED_SHADOWS: the shadow ray of a hit traced by the lane that found it, inside the primary launch, against
    the two-launch form (generator + explicit rays): identical images, identical records, and -- with live counters --
    identical node arrays; also inside the cube, with the >= tie-break, a sun along an axis (zero components get
    octree_ray's bias) and on a deep tree."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, pkg.gpu.VARIANT_STACK)
    terrain = pkg.scenes.terrain(seed=3, max_depth=13, cam=(0.2, 0.6, -0.7), lod_c=300.0, max_words=6_000_000)
    cases = [
        (monu9_words, O.F_PAUSE_ADAPTIVE | O.F_SHADOWS, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)), None),
        (monu9_words, O.F_PAUSE_ADAPTIVE | O.F_SHADOWS | O.F_MISC_BOOL, ((0.02, 0.31, 0.05), (0.3, -0.2, 1.0)), None),
        (monu9_words, O.F_PAUSE_ADAPTIVE | O.F_SHADOWS, ((1.3, 0.9, 1.2), (-1.0, -0.6, -1.0)), (0.0, -1.0, 0.0)),
        (monu9_words, O.F_SHADOWS, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)), None),              # counters live
        (small_words, O.F_SHADOWS, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)), (1.0, -0.3, 0.0)),
        (terrain, O.F_PAUSE_ADAPTIVE | O.F_SHADOWS, ((0.2, 0.6, -0.7), (0.1, -0.5, 1.0)), None),
        (terrain, O.F_SHADOWS, ((0.2, 0.6, -0.7), (0.1, -0.5, 1.0)), (-0.4, -1.0, 0.3)),
    ]
    # last case: every primary ray has a direction component of 1e-30 -- outside the range of the fast arithmetic, so the
    # whole frame (and its shadow rays) goes through the deferred list of the post pass
    cases.append((monu9_words, O.F_PAUSE_ADAPTIVE | O.F_SHADOWS, ((0.0, 0.0, -1.5), (0.0, 0.0, 1.5)), (-0.3, -1.0, 0.2)))
    for case, (words, flags, pose, sun) in enumerate(cases):
        u = O.make_uniforms(pos=pose[0], look=pose[1], width=320, height=200, flags=flags)
        if sun is not None:
            u.sun_dir[:3] = list(sun)
        if case == len(cases) - 1:
            u.camera_inverse[0] *= 1e-30
        results = []
        for fused in (1, 0):
            gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, fused)
            render = pkg.Render(gpu, (320, 200), words, capacity=words.size)
            set_uniforms_from_oracle(render, u)
            for frame in range(2):  # second frame: schedules in place, counters carried over
                hits, img = render.render_host(rgba=True)
            results.append((hits, img, render.read_nodes(words.size)))
        gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, 2)  # back to the default (automatic)
        (h1, i1, n1), (h0, i0, n0) = results
        what = f"flags={flags} sun={sun} words={words.size}"
        assert_hits_equal(h1, h0, "fused vs two-pass records, " + what)
        assert np.array_equal(i1, i0), "fused vs two-pass image, " + what
        assert np.array_equal(n1, n0), "fused vs two-pass counters, " + what
        assert (i1[..., :3].reshape(-1, 3).max(axis=1) > 0).mean() > 0.1, what
        if case == len(cases) - 1:
            assert ((h1["info"] >> 16) & 1).sum() > 100, "the deferred frame should hit the model"


def test_adaptive_streaming_loop(pkg, gpu, O, monu9_words):
    """The reference's frame loop with live counters (app.rs:94-118 + adaptive.rs): starting from the 8-word
    root tree, hot leaves are subdivided from the CPU world and the device tree converges towards the view;
    every intermediate device tree traces bit-exactly like the oracle on the same words."""
    from conftest import load_vox_fixture
    size, xyzi, pal, n, _ = load_vox_fixture("monu9")
    world = pkg.adaptive.World(pkg.CpuOctree.from_voxels(size, xyzi, pal))
    octree = world.root_octree()
    assert len(octree) == 8
    render = pkg.Render.new(gpu, (160, 96), octree, capacity=200_000)
    render.set_flags(pause_adaptive=False, shadows=False)
    compute = pkg.Compute.new(gpu, render)
    loop = pkg.adaptive.AdaptiveLoop(gpu, render, compute, octree, world)
    settings, character = pkg.Settings(), pkg.Character()
    sizes, total_sub = [], 0
    for frame in range(12):
        before = octree.raw_data()
        hits, n_sub, n_unsub = loop.frame(settings, character, deterministic=True)
        gpu.sync()
        total_sub += n_sub
        sizes.append(len(octree))
        # parity of this frame's records against the oracle on the words the frame was traced on
        u = O.make_uniforms(width=160, height=96, flags=0)
        for f in ("camera", "camera_inverse"):
            getattr(u, f)[:] = list(getattr(render.uniforms, f))
        assert_hits_equal(pkg.render.hits_to_numpy(hits), O.trace_frame(before, u, threads=4), f"adaptive frame {frame}")
    assert total_sub > 50 and sizes[-1] > sizes[0] and sizes[-1] % 8 == 0
    words = octree.raw_data()
    ptr = words >> 4
    interior = ptr < pkg.VOXEL_OFFSET
    assert (ptr[interior] % 8 == 0).all() and (ptr[interior] + 8 <= words.size).all()
    assert (words & 15 == 0).all()  # host words carry counter 0 (octree.rs:28-30,164-166)
    # a leaf of the streamed tree shows the world's colour (mip or voxel) at that position and depth
    rng = np.random.default_rng(1)
    ptrs, rgb = world.chunk(0).raw()
    for p in rng.uniform(-0.9, 0.9, (50, 3)).astype(np.float32):
        idx, depth, _ = octree.find_voxel(p.tolist())
        _, cidx, _, _ = world.find_voxel(p.tolist(), depth)
        c = rgb[cidx]
        assert ptr[idx] - pkg.VOXEL_OFFSET == (int(c[0]) << 16 | int(c[1]) << 8 | int(c[2]))
    # pausing freezes the tree (app.rs:97)
    render.set_flags(pause_adaptive=True)
    _, n_sub, n_unsub = loop.frame(settings, character)
    assert (n_sub, n_unsub) == (0, 0) and len(octree) == sizes[-1]


def test_adaptive_loop_incremental_upload(pkg, gpu, O):
    """The streaming loop with the device-side reset (scan clears the counters it read, only changed words are sent,
    svo_nodes_scatter) against the reference's form (whole array re-uploaded every frame, app.rs:113-118): same lists,
    same host tree, and the SAME device array after every frame -- counters included."""
    from conftest import load_vox_fixture
    size, xyzi, pal, n, _ = load_vox_fixture("monu9")
    loops = []
    for incremental in (False, True):
        g = pkg.Gpu(0)
        world = pkg.adaptive.World(pkg.CpuOctree.from_voxels(size, xyzi, pal))
        octree = world.root_octree()
        render = pkg.Render.new(g, (160, 96), octree, capacity=200_000)
        render.set_flags(pause_adaptive=False, shadows=True)
        compute = pkg.Compute.new(g, render)
        loops.append((g, render, octree, pkg.adaptive.AdaptiveLoop(g, render, compute, octree, world, incremental=incremental)))
    settings = pkg.Settings()
    for frame in range(10):
        character = pkg.Character((0.1 + 0.02 * frame, 0.2, -1.5), (0.0, 0.0, 1.5))  # a moving camera: unsubdivisions too
        results = []
        for g, render, octree, loop in loops:
            hits, n_sub, n_unsub = loop.frame(settings, character, deterministic=True)
            g.sync()
            results.append((pkg.render.hits_to_numpy(hits).view(np.uint32).copy(), n_sub, n_unsub, octree.raw_data(),
                            render.read_nodes(len(octree))))
        a, b = results
        assert np.array_equal(a[0], b[0]), f"frame {frame}: records differ"
        assert a[1:3] == b[1:3] and np.array_equal(a[3], b[3])
        assert np.array_equal(a[4], b[4]), f"frame {frame}: device arrays differ"
        assert (b[4] & 15 == 0).all() and np.array_equal(b[4], b[3])  # counters reset, device == host tree
    assert len(loops[1][2]) > 1000
    for g, *_ in loops:
        g.close()


def test_config3_rsvo_shell(pkg, gpu, O):
    """Config 3 stand-in (files/statuette.rsvo is absent from the checkout): a sphere shell voxelised at depth 8,
    serialised as an .rsvo child-mask stream, loaded by load_octree at two depths (cpu_octree.rs:128-175) and traced."""
    tree = pkg.CpuOctree.new(0)
    n = 1 << 8
    ax = (np.arange(n) + 0.5) / n * 2 - 1
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z)
    shell = np.argwhere(np.abs(r - 0.8) < 1.2 / n)
    for i, j, k in shell[::3]:
        tree.put_in_voxel((float(ax[i]), float(ax[j]), float(ax[k])), pkg.Voxel(200, 120, 40), 8)
    blob = tree.to_rsvo()
    for depth in (8, 6):
        words = pkg.CpuOctree.load_octree(blob, depth).to_octree_words()
        assert np.array_equal(words, O.Tree.from_rsvo(blob, depth).to_octree())
        assert pkg.scenes.max_depth(words) == depth
        u = O.make_uniforms(width=320, height=180, flags=O.F_PAUSE_ADAPTIVE)
        for variant in VARIANTS:
            got = _render(pkg, gpu, words, u, variant)
            assert_hits_equal(got, O.trace_frame(words, u, threads=8), f"rsvo shell depth {depth}")


@pytest.fixture
def fused_shadows(request, pkg, gpu):
    gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, request.param)
    yield request.param
    gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, 2)  # the default (automatic)


@pytest.mark.parametrize("variant,fused_shadows", [(0, 0), (1, 0), (1, 1)], indirect=["fused_shadows"])
def test_secondary_rays_per_hit_pixel(pkg, gpu, O, monu9_words, small_words, variant, fused_shadows):
    """svo_render_secondary: ray 0 is fs_main's shadow ray (shader.wgsl:275-280), rays 1..3 share its origin;
    records are bit-exact against the oracle, the tile-sharded call equals the full frame, and ray 0 decides
    exactly the pixels the shaded image shows in shadow.  With fused shadow rays (STACK kernel) ray 0 is traced by
    the lane that found the hit, inside the primary launch: its records must be the same."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for words, pose in ((monu9_words, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))), (monu9_words, ((0.02, 0.31, 0.05), (0.3, -0.2, 1.0))),
                        (small_words, ((0.9, 0.8, -1.1), (-0.9, -0.8, 1.1)))):
        W, H = 256, 128
        u = O.make_uniforms(pos=pose[0], look=pose[1], width=W, height=H, flags=O.F_PAUSE_ADAPTIVE | O.F_SHADOWS)
        render = pkg.Render(gpu, (W, H), words, capacity=words.size)
        set_uniforms_from_oracle(render, u)
        oprim, osec = O.secondary_frame(words, u, 4, threads=8)
        for frame in range(2):
            prim, sec = render.render_secondary(4)
            gpu.sync()
        assert_hits_equal(pkg.render.hits_to_numpy(prim), oprim, "primary records of the secondary call")
        sec = pkg.render.hits_to_numpy(sec).reshape(4, H, W)
        for k in range(4):
            assert_hits_equal(sec[k], osec[k], f"secondary ray {k}")
        hit = ((oprim["info"] >> 16) & 1).astype(bool)
        assert hit.any() and not hit.all()
        assert (sec["value"][:, ~hit] == 0).all() and (((sec["info"] >> 16) & 1)[:, ~hit] == 0).all()
        # fewer rays: a prefix of the same rays
        _, sec2 = render.render_secondary(2)
        gpu.sync()
        assert_hits_equal(pkg.render.hits_to_numpy(sec2).reshape(2, H, W), osec[:2], "n_secondary = 2")
        # tile sharding: two "ranks" with interleaved 64x8 tiles reassemble to the full frame
        tw, th = 64, 8
        full = np.empty((4, H, W), dtype=pkg.HIT_DTYPE)
        for rank in range(2):
            _, part = render.render_tiles_secondary(tw, th, rank, 2, 4)
            gpu.sync()
            n_mine = pkg.sharding.local_tile_count(W, H, tw, th, rank, 2)
            part = pkg.render.hits_to_numpy(part).reshape(4, n_mine, th, tw)
            for j in range(n_mine):
                t = rank + 2 * j
                ty, tx = divmod(t, W // tw)
                full[:, ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = part[:, j]
        assert_hits_equal(full, osec, "secondary rays, tile-sharded")
        # a rectangle inside the frame
        _, rect = render.render_secondary(3, tile=(40, 16, 100, 50))
        gpu.sync()
        assert_hits_equal(pkg.render.hits_to_numpy(rect).reshape(3, 50, 100), osec[:3, 16:66, 40:140], "secondary rays of a rectangle")
    with pytest.raises(pkg.SvoError):
        render.render_secondary(5)


def test_config3_block_instanced_shell(pkg, gpu, O):
    """Config 3 stand-in as SURVEY 8d describes it: an .rsvo shell whose leaves reference the eight 16^3 block
    models (cpu_octree.rs:37, world.rs:19-58), expanded by the streaming loop's rule (svo_world_expand) to
    shell depth + 4 levels, then traced."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "blocks_vox.npz"))
    world = pkg.World.new("")
    for i, name in enumerate(pkg.world.BLOCK_NAMES):
        world.insert(i + 1, pkg.CpuOctree.from_voxels(16, z[name + "_xyzi"], z[name + "_palette"]))
        world.generate_mip_tree(i + 1)
    depth = 5
    tree = pkg.CpuOctree.new(0)
    n = 1 << depth
    ax = (np.arange(n) + 0.5) / n * 2 - 1
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z)
    for i, j, k in np.argwhere(np.abs(r - 0.75) < 1.0 / n):
        tree.put_in_voxel((float(ax[i]), float(ax[j]), float(ax[k])), pkg.Voxel(1, 1, 1), depth)
    world.insert(0, pkg.CpuOctree.load_octree(tree.to_rsvo(), depth))
    world.generate_mip_tree(0)
    octree = world.root_octree()
    world.expand(octree, max_depth=depth + 4)
    words = octree.raw_data()
    assert pkg.scenes.max_depth(words) == depth + 4 and words.size > 3_000_000
    u = O.make_uniforms(pos=(0.3, 0.4, -1.6), look=(-0.2, -0.3, 1.5), width=480, height=270, flags=O.F_PAUSE_ADAPTIVE)
    want = O.trace_frame(words, u, threads=8)
    assert ((want["info"] >> 16) & 1).mean() > 0.10
    for variant in VARIANTS:
        assert_hits_equal(_render(pkg, gpu, words, u, variant), want, "block-instanced shell")
    # the view-limited expansion is a prefix-closed subtree of the same world: still a well-formed array
    lod = world.root_octree()
    world.expand(lod, max_depth=depth + 4, cam=(0.3, 0.4, -1.6), lod_c=40.0)
    lw = lod.raw_data()
    assert 8 < lw.size < words.size
    assert_hits_equal(_render(pkg, gpu, lw, u, 1), O.trace_frame(lw, u, threads=8), "block-instanced shell, LOD expansion")


def test_config5_fractal_depth20(pkg, gpu, O):
    """Config 5 family: depth-20 fractal (voxel 1.9e-6 < the reference's 2e-6 nudge, SURVEY section 0.5: the
    images are not meaningful but parity with the oracle stays exact), primary frame plus secondary rays
    (explicit rays from inside the cube).  Needs the deep ancestor stack (SVO_OPT_TREE_DEPTH)."""
    import torch
    # the corner (-1,-1,-1) belongs to the Sierpinski set: refine to depth 20 around it and look at it closely
    words = pkg.scenes.fractal(seed=0, max_depth=20, cam=(-0.9999, -0.9999, -0.9999), lod_c=300.0, min_depth=4,
                               max_words=6_000_000)
    assert pkg.scenes.max_depth(words) == 20
    cam = (-0.9990, -0.9985, -0.9980)
    u = O.make_uniforms(pos=cam, look=(-1.0, -1.2, -0.9), width=256, height=144, flags=O.F_PAUSE_ADAPTIVE)
    want = O.trace_frame(words, u, threads=8)
    render = pkg.Render(gpu, (256, 144), words, capacity=words.size)
    set_uniforms_from_oracle(render, u)
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    try:
        # default kernel resolves 16 levels: it must refuse loudly, not return wrong voxels
        render.render()
        with pytest.raises(pkg.SvoError):
            gpu.sync()
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 20)
        for _ in range(2):
            got = pkg.render.hits_to_numpy(render.render())
            gpu.sync()
            assert_hits_equal(got, want, "fractal depth 20, deep stack")
        rng = np.random.default_rng(20)
        rays = np.concatenate([rng.uniform(-1.0, -0.998, (20000, 3)), rng.normal(size=(20000, 3))], axis=1).astype(np.float32)
        rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
        got = pkg.render.hits_to_numpy(render.trace_rays(torch.from_numpy(rays).cuda()))
        gpu.sync()
        assert_hits_equal(got, O.trace_rays(words, rays, threads=8), "fractal secondary rays")
        # config 5 proper: 4 secondary rays per hit pixel (shadow ray + 3 hashed directions), both layouts
        oprim, osec = O.secondary_frame(words, u, 4, threads=8)
        prim, sec = render.render_secondary(4)
        gpu.sync()
        assert_hits_equal(pkg.render.hits_to_numpy(prim), oprim, "fractal primary (secondary call)")
        assert_hits_equal(pkg.render.hits_to_numpy(sec), osec, "fractal 4 secondary rays / pixel")
        gpu.set_option(pkg.gpu.OPT_VARIANT, 0)
        assert_hits_equal(_render(pkg, gpu, words, u, 0), want, "fractal depth 20, general kernel")
    finally:
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)
        gpu.set_option(pkg.gpu.OPT_VARIANT, 1)


def test_deep_stack_limit_depth22_and_beyond(pkg, gpu, O):
    """Round 4: the STACK variant's path codes have 23 bits, i.e. resolve 22 levels.  A depth-22 tree runs on the deep-stack
    instantiation, a depth-23 tree is handed to the general kernel by SVO_OPT_TREE_DEPTH = 23 (and refused loudly by the deep-stack
    kernel if the option claims 22) -- records equal the oracle's either way."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    try:
        for depth in (22, 23):
            words = pkg.scenes.fractal(seed=0, max_depth=depth, cam=(-0.99999, -0.99999, -0.99999), lod_c=300.0, min_depth=4, max_words=6_000_000)
            assert pkg.scenes.max_depth(words) == depth
            u = O.make_uniforms(pos=(-0.99990, -0.99985, -0.99980), look=(-1.0, -1.2, -0.9), width=192, height=108, flags=O.F_PAUSE_ADAPTIVE)
            want = O.trace_frame(words, u, threads=8)
            render = pkg.Render(gpu, (192, 108), words, capacity=words.size)
            set_uniforms_from_oracle(render, u)
            gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, depth)
            got = pkg.render.hits_to_numpy(render.render())
            gpu.sync()
            assert_hits_equal(got, want, f"fractal depth {depth}, SVO_OPT_TREE_DEPTH {depth}")
            if depth == 23:
                gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 22)  # a lie: the deep-stack kernel must notice
                render.render()
                with pytest.raises(pkg.SvoError):
                    gpu.sync()
    finally:
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)


def test_tree_deeper_than_declared_is_refused(pkg, gpu, O):
    """Round 4: the STACK walk counts its iterations instead of checking every lane's level, and the ancestor stacks lie at the end of
    the workgroup's LDS so that a push below the last level falls off the allocation.  A tree deeper than SVO_OPT_TREE_DEPTH says --
    18 and 23 levels under a declared 16, static and with live hit counters -- must still be refused by svo_sync, must not fault,
    and the frames after it (same context, an honest tree) must be right."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)
    honest = pkg.scenes.fractal(seed=2, max_depth=12, cam=(-0.99, -0.99, -0.99), lod_c=200.0, min_depth=3, max_words=2_000_000)
    u = O.make_uniforms(pos=(-0.99990, -0.99985, -0.99980), look=(-1.0, -1.2, -0.9), width=192, height=108, flags=O.F_PAUSE_ADAPTIVE)
    for depth in (18, 23):
        words = pkg.scenes.fractal(seed=0, max_depth=depth, cam=(-0.99999, -0.99999, -0.99999), lod_c=300.0, min_depth=4, max_words=6_000_000)
        assert pkg.scenes.max_depth(words) == depth
        for flags in (O.F_PAUSE_ADAPTIVE, 0):
            render = pkg.Render(gpu, (192, 108), words, capacity=words.size)
            uu = O.make_uniforms(pos=(-0.99990, -0.99985, -0.99980), look=(-1.0, -1.2, -0.9), width=192, height=108, flags=flags)
            set_uniforms_from_oracle(render, uu)
            render.render()
            with pytest.raises(pkg.SvoError):
                gpu.sync()
            del render
        render = pkg.Render(gpu, (192, 108), honest, capacity=honest.size)
        set_uniforms_from_oracle(render, u)
        got = pkg.render.hits_to_numpy(render.render())
        gpu.sync()
        assert_hits_equal(got, O.trace_frame(honest, u, threads=8), f"honest tree after the refused depth-{depth} one")


def test_bench_workload_full_size(pkg, gpu, O):
    """The benchmark configuration at its full size (depth-16 terrain, ~107 M words, 1920x1080): the whole frame
    against the oracle, frame-to-frame idempotence under the adaptive schedule, and tile sharding == full frame."""
    cam, look = pkg.scenes.terrain_camera(0, 16)
    words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
    assert words.size > 64 * 1024 * 1024  # larger than the 256 MiB Infinity Cache
    W, H = 1920, 1080
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(fov=90.0), pkg.Character(cam, look))
    gpu.set_option(pkg.gpu.OPT_VARIANT, 1)
    frames = []
    for _ in range(3):
        buf = render.alloc_hits(W * H)
        buf.fill_(-1)  # poisoned: a dropped ray would keep it
        frames.append(pkg.render.hits_to_numpy(render.render(hits=buf)))
        gpu.sync()
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    assert np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))
    u = O.Uniforms()
    for f in ("camera", "camera_inverse", "dimensions", "sun_dir"):
        getattr(u, f)[:] = list(getattr(render.uniforms, f))
    u.flags = render.uniforms.flags
    want = O.trace_frame(words, u, threads=os.cpu_count() or 8)
    assert_hits_equal(frames[0], want, "bench workload, full 1080p frame")
    steps = want["info"].reshape(-1) & 0xFF
    assert 20 < steps.mean() < 30 and steps.max() == 101
    # 8-way tile sharding reassembles to the same frame
    tw, th, world = 64, 8, 8
    tiles_x = W // tw
    frame = np.zeros((H, W), dtype=pkg.HIT_DTYPE)
    for r in range(world):
        got = pkg.render.hits_to_numpy(render.render_tiles(tw, th, r, world)).reshape(-1, th, tw)
        gpu.sync()
        t = r + np.arange(got.shape[0]) * world
        for k, tt in enumerate(t):
            ty, tx = divmod(int(tt), tiles_x)
            frame[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = got[k]
    assert np.array_equal(frame.reshape(-1).view(np.uint32), want.reshape(-1).view(np.uint32))


@pytest.mark.parametrize("fused_shadows", [0, 1], indirect=True)
def test_shaded_tiles_match_full_frame(pkg, gpu, O, monu9_words, fused_shadows):
    """svo_render_tiles with rgba_out: the sharded, shaded image equals the unsharded one pixel for pixel (shadow rays as a
    second launch and fused into the primary one)."""
    gpu.set_option(pkg.gpu.OPT_VARIANT, pkg.gpu.VARIANT_STACK)
    W, H, tw, th, ranks = 192, 96, 64, 8, 3
    u = O.make_uniforms(width=W, height=H, flags=O.F_PAUSE_ADAPTIVE | O.F_SHADOWS)
    render = pkg.Render(gpu, (W, H), monu9_words, capacity=monu9_words.size)
    set_uniforms_from_oracle(render, u)
    _, full = render.render_host(rgba=True)
    full = full.view(np.uint32).reshape(H, W)
    tiles_x = W // tw
    img = np.zeros((H, W), dtype=np.uint32)
    for r in range(ranks):
        n = pkg.sharding.local_tile_count(W, H, tw, th, r, ranks)
        rgba = render.alloc_rgba(n * tw * th)
        render.render_tiles(tw, th, r, ranks, rgba=rgba)
        gpu.sync()
        got = rgba.cpu().numpy().view(np.uint32).reshape(n, th, tw)
        for k in range(n):
            ty, tx = divmod(r + k * ranks, tiles_x)
            img[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = got[k]
    assert np.array_equal(img, full)


@pytest.mark.parametrize("variant", VARIANTS)
def test_lattice_rays_ties_and_boundaries(pkg, gpu, O, monu9_words, variant):
    """Rays built to land exactly on cell boundaries and to tie in t_max (several axes in the step mask,
    shader.wgsl:231): origins on dyadic lattice points, directions along axes, face and space diagonals with
    exactly equal components, both tie-break modes (shader.wgsl:138-150)."""
    import itertools
    import torch
    trees = [monu9_words, pkg.scenes.random_tree(seed=11, max_depth=7, p_split=0.6, p_solid=0.3, max_words=1 << 19)]
    dirs = []
    for d in itertools.product((-1.0, 0.0, 1.0), repeat=3):
        if any(d):
            v = np.array(d, dtype=np.float32)
            dirs.append(v / np.float32(np.sqrt(np.float32((v * v).sum()))))  # equal components stay bit-equal
    dirs = np.array(dirs, dtype=np.float32)
    rng = np.random.default_rng(31)
    for m in (2, 4, 7):
        k = rng.integers(-(1 << m), (1 << m) + 1, (3000, 3)).astype(np.float32) / np.float32(1 << m)
        # a third of the origins sit on cell centres (odd multiples of 2^-(m+1)), a few outside the cube
        k[::3] += np.float32(1.0 / (1 << (m + 1)))
        k[::50] *= np.float32(1.5)
        d = dirs[rng.integers(0, len(dirs), k.shape[0])]
        rays = np.concatenate([k, d], axis=1).astype(np.float32)
        for words in trees:
            render = pkg.Render(gpu, (8, 8), words, capacity=words.size)
            gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
            for flags in (O.F_PAUSE_ADAPTIVE, O.F_PAUSE_ADAPTIVE | O.F_MISC_BOOL):
                render.uniforms.flags = flags
                render.upload_uniforms()
                got = pkg.render.hits_to_numpy(render.trace_rays(torch.from_numpy(rays).cuda()))
                gpu.sync()
                want = O.trace_rays(words, rays, flags=flags, threads=8)
                assert_hits_equal(got, want, f"lattice rays m={m} flags={flags}")
                multi = np.isin(want["normal_bits"], [0b000101, 0b000110, 0b001001, 0b001010, 0b010001, 0b010010,
                                                      0b100001, 0b100010, 0b010100, 0b011000, 0b100100, 0b101000])
                assert multi.any(), "the set should contain steps that tie on two axes"


@pytest.mark.parametrize("variant", VARIANTS)
def test_tiny_origin_components(pkg, gpu, O, monu9_words, variant):
    """Origins with components far below the voxel size -- exactly 0, +-1e-12, +-1e-30, subnormal -- and a camera
    sitting exactly on the x = 0 / y = 0 centre planes (its matrix inverse yields |pos| ~ 1e-17): these take the
    fast path of the STACK kernel (no lower bound on |pos|) and must stay bit-exact."""
    import torch
    words = pkg.scenes.random_tree(seed=5, max_depth=9, p_split=0.55, p_solid=0.25, max_words=1 << 20)
    rng = np.random.default_rng(77)
    tiny = np.array([0.0, 1e-12, -1e-12, 1e-30, -1e-30, 1e-40, -1e-40, 1e-7, -1e-7, 2.0 ** -24, -(2.0 ** -24)], dtype=np.float32)
    n = 30000
    org = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
    for axis in range(3):
        sel = rng.random(n) < 0.5
        org[sel, axis] = tiny[rng.integers(0, tiny.size, int(sel.sum()))]
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::7, 0] = 0.0  # exercises the 1e-6 bias of zero direction components next to tiny origins (shader.wgsl:193-194)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    gpu.set_option(pkg.gpu.OPT_VARIANT, variant)
    for tree in (words, monu9_words):
        render = pkg.Render(gpu, (8, 8), tree, capacity=tree.size)
        for flags in (O.F_PAUSE_ADAPTIVE, O.F_PAUSE_ADAPTIVE | O.F_MISC_BOOL):
            render.uniforms.flags = flags
            render.upload_uniforms()
            got = pkg.render.hits_to_numpy(render.trace_rays(torch.from_numpy(rays).cuda()))
            gpu.sync()
            assert_hits_equal(got, O.trace_rays(tree, rays, flags=flags, threads=8), f"tiny origins flags={flags}")
    for pos, look in (((0.0, 0.0, -0.2), (0.3, 0.1, 1.0)), ((0.0, 0.25, 0.0), (1.0, -0.2, 0.4))):
        u = O.make_uniforms(pos=pos, look=look, width=320, height=180, flags=O.F_PAUSE_ADAPTIVE)
        assert_hits_equal(_render(pkg, gpu, words, u, variant), O.trace_frame(words, u, threads=8), f"camera on a centre plane {pos}")


def test_launch_options_do_not_change_results(pkg, gpu, O, monu9_words):
    """Every tuning knob of svo_set_option (refill threshold, strip size, static / dynamic claiming, grid size, schedule
    period, block shape) only changes HOW the rays are traced: the records stay bit-identical to the oracle's."""
    terrain = pkg.scenes.terrain(seed=2, max_depth=12, cam=(0.1, 0.3, -0.2), lod_c=300.0, max_words=3_000_000)
    G = pkg.gpu
    defaults = {G.OPT_REFILL_MIN: 32, G.OPT_STRIP_ITEMS: 64, G.OPT_DYNAMIC_STRIPS: 1, G.OPT_GRID_BLOCKS: 0, G.OPT_SCHEDULE: 2,
                G.OPT_BLOCK_SHAPE: 3}
    combos = [{G.OPT_REFILL_MIN: 1}, {G.OPT_REFILL_MIN: 64}, {G.OPT_STRIP_ITEMS: 256, G.OPT_SCHEDULE: 0},
              {G.OPT_DYNAMIC_STRIPS: 0, G.OPT_SCHEDULE: 0}, {G.OPT_GRID_BLOCKS: 7}, {G.OPT_GRID_BLOCKS: 4000},
              {G.OPT_SCHEDULE: 1}, {G.OPT_SCHEDULE: 0}, {G.OPT_BLOCK_SHAPE: 4}, {G.OPT_BLOCK_SHAPE: 6, G.OPT_REFILL_MIN: 5},
              {G.OPT_GRID_BLOCKS: 1, G.OPT_REFILL_MIN: 33}]
    gpu.set_option(G.OPT_VARIANT, 1)
    try:
        for words, pose in ((monu9_words, ((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))), (terrain, ((0.1, 0.3, -0.2), (0.4, -0.3, 1.0)))):
            u = O.make_uniforms(pos=pose[0], look=pose[1], width=333, height=187, flags=O.F_PAUSE_ADAPTIVE)
            want = O.trace_frame(words, u, threads=8)
            for combo in combos:
                for k, v in {**defaults, **combo}.items():
                    gpu.set_option(k, v)
                assert_hits_equal(_render(pkg, gpu, words, u, 1), want, f"options {combo}")
    finally:
        for k, v in defaults.items():
            gpu.set_option(k, v)


def test_bench_multi_gpu_path_with_one_rank(pkg, gpu):
    """bench.py's N > 1 code path end to end (three lanes, 12-byte wire records, asynchronous RCCL gather, assemble on
    rank 0, timing collection, oracle check of the assembled frame) with a one-rank process group on this GPU."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-pipeline", "--steps", "24", "--warmup", "3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["cpu_baseline"]["gpu_frame_matches_oracle_on_sample"] is True
    # (three lanes exist; a rank's share of 0.7 M rays or more -- here the whole frame -- runs two deep since round 5: bench.py)
    assert line["config"]["frames_in_flight"] == 2 and line["config"]["frames_in_flight_by_frame_size"]["1920x1080"] == 2
    assert "12 B/ray" in line["config"]["sharding"]
    assert line["steps"] == 24 and line["value"] > 100
    # the same path carrying the shaded colour frame instead of records (4 bytes per ray on the links)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-pipeline", "--wire", "rgba8", "--steps", "12", "--warmup", "2",
                        "--no-extras"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["cpu_baseline"]["gpu_frame_matches_oracle_on_sample"] is True and "4 B/ray" in line["config"]["sharding"]
