"""Oracle pinning: the reference has no tests or golden vectors, so the oracle is
pinned by the hand-derivable known answers of SURVEY.md 8c and by the committed fixtures."""
import numpy as np

from conftest import GOLDEN, load_vox_fixture


def test_word_encodings(O):
    """KAT 1: octree.rs:28-30,164-166."""
    t = O.Tree.new(0)
    assert t.to_octree().tolist() == [0x80000000] * 8          # empty = VOXEL_OFFSET << 4
    t.put_in_voxel((0.5, 0.5, 0.5), (255, 0, 0), 1)
    assert t.to_octree()[7] == 0x8FF00000                      # Voxel(255,0,0).to_value()
    t2 = O.Tree.new(0)
    t2.put_in_voxel((-0.9, -0.9, -0.9), (1, 2, 3), 2)          # forces one subdivision: create_node(8) = 0x80
    w = t2.to_octree()
    assert w[0] == 0x00000080 and w.size == 16 and w[8] == ((O.VOXEL_OFFSET + 0x010203) << 4)


def test_child_mask_bit_test(O):
    """cpu_octree.rs:32-45: bit i of the mask -> child i is a block reference CHUNK_OFFSET + i % 8 + 1."""
    t = O.Tree.new(0b01011011)
    ptrs, rgb = t.raw()
    for i in range(8):
        if (0b01011011 >> i) & 1:
            assert ptrs[i] == O.CHUNK_OFFSET + i + 1 and rgb[i].tolist() == [255, 0, 0]
        else:
            assert ptrs[i] == O.CHUNK_OFFSET and rgb[i].tolist() == [0, 0, 0]


def test_node_counts_from_assets(O):
    """KAT 3: 8 * (1 + #interior): small.vox 256 words, monu9.vox 65 184."""
    for name, n_vox, n_words in (("small", 45, 256), ("monu9", 32832, 65184)):
        size, xyzi, pal, n, crc = load_vox_fixture(name)
        assert xyzi.shape[0] == n_vox and n == n_words
        words = O.Tree.from_voxels(size, xyzi, pal).to_octree()
        assert words.size == n_words
        assert int(np.bitwise_xor.reduce(words * np.arange(1, words.size + 1, dtype=np.uint32))) == crc
        solid = ((words >> 4) > O.VOXEL_OFFSET).sum()
        assert solid == n_vox  # every voxel of the model is one solid leaf


def test_node_counts_of_the_other_assets(O):
    """KAT 3, the rest: the node counts SURVEY.md 8c computed independently from the .vox files under the load_vox rules
    (cpu_octree.rs:177-210) -- monu10 194 856, defualt 108 088, phantom_mansion 1 132 944 words -- and a full 16^3 block:
    8 * (1 + 8 + 64 + 512) = 4 680 words."""
    for name, n_vox, n_words in (("monu10", 150764, 194856), ("defualt", 56784, 108088), ("phantom_mansion", 631021, 1132944)):
        size, xyzi, pal, n, crc = load_vox_fixture(name)
        assert xyzi.shape[0] == n_vox and n == n_words
        words = O.Tree.from_voxels(size, xyzi, pal).to_octree()
        assert words.size == n_words
        assert int(np.bitwise_xor.reduce(words * np.arange(1, words.size + 1, dtype=np.uint32))) == crc
        # every voxel of the model is one leaf; a voxel whose palette colour is black encodes as VOXEL_OFFSET + 0, the empty
        # leaf (octree.rs:28-35: the reference cannot tell the two apart either) -- phantom_mansion has 101 of them
        black = int(((pal[(xyzi[:, 3].astype(np.int64) - 1) & 255] & 0xFFFFFF) == 0).sum())
        assert ((words >> 4) > O.VOXEL_OFFSET).sum() == n_vox - black and black == (101 if name == "phantom_mansion" else 0)
    g = np.arange(16, dtype=np.uint8)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    full = np.stack([x.ravel(), y.ravel(), z.ravel(), np.ones(4096, dtype=np.uint8)], axis=1)
    pal = np.full(256, 0xFF3366CC, dtype=np.uint32)
    words = O.Tree.from_voxels(16, full, pal).to_octree()
    assert words.size == 4680 and ((words >> 4) > O.VOXEL_OFFSET).sum() == 4096 and ((words >> 4) < O.VOXEL_OFFSET).sum() == 584


def test_hand_derived_rays(O):
    """tests/kat_cases.py: two-level descent, three- and two-axis ties, a start on a centre plane under both tie-break modes,
    the 100 / 101 step boundary for hits and for rays that leave the cube -- every record derived on paper from the WGSL."""
    import kat_cases as K
    for name, words, flags, rays, expected in K.cases():
        K.check(O.trace_rays(words, rays, flags=flags), expected, name)


def test_hand_derived_hit_counters(O):
    """tests/kat_cases.py counter_case: a 1 x 1 frame whose camera matrix makes its only ray a hand-derived one; the counters the
    frame leaves in the node words are worked out on paper (every word a walk reads, leaf included, +1 per walk, up to 15)."""
    import kat_cases as K
    words, cinv, record, after = K.counter_case()
    u = O.make_uniforms(width=1, height=1, flags=0)  # counters live, no shadow ray
    u.camera_inverse[:] = cinv.tolist()
    K.check(O.trace_frame(words, u).reshape(-1), [record], "the 1 x 1 frame's ray")
    w = words
    for n in range(1, 10):  # word 6 saturates in frame 8
        w = O.count_frame(w, u)
        assert np.array_equal(w, after(n)), f"counters after {n} frames"


def test_hand_derived_shading(O):
    """tests/kat_cases.py shading_cases: fs_main's colour of a hit pixel and of a miss pixel, worked out on paper."""
    import kat_cases as K
    for name, words, flags, cinv, sun, rgba in K.shading_cases():
        u = O.make_uniforms(width=1, height=1, flags=flags, sun_dir=sun)
        u.camera_inverse[:] = cinv.tolist()
        got = np.floor(np.clip(O.shade_frame(words, u).reshape(4), 0, 1) * 255.0 + 0.5).astype(np.int32)
        assert np.abs(got - np.array(rgba)).max() <= 1 and got[3] == rgba[3], f"{name}: {got.tolist()} against {rgba}"


def hand_made_rsvo():
    """A .rsvo stream assembled by hand from the reader's rules (cpu_octree.rs:128-175): 16 header bytes (ignored), byte 16 =
    top_level, from byte 20 top_level + 1 little-endian u32 node counts per level, then one child-mask byte per node in
    breadth-first order, the root's first.  Here: top_level 2; counts 1, 2, 3; root mask 0b101 (children 0 and 2), their masks
    0b10000000 (child 7) and 0b00000011 (children 0, 1), then three level-2 masks that a depth-2 load must not look at."""
    import struct
    return (b"hand-made header" + bytes([2, 0, 0, 0]) + struct.pack("<3I", 1, 2, 3) + bytes([0b101, 0b10000000, 0b00000011, 0xFF, 0x01, 0x80]))


def check_hand_made_rsvo(load, CHUNK):
    """Expected trees (CpuOctree::new / add_voxels, cpu_octree.rs:23-45: a set bit becomes a block reference CHUNK_OFFSET + (index % 8)
    + 1 coloured (255, 0, 0), a clear bit CHUNK_OFFSET coloured black; load_octree turns a reference into a pointer to 8 new nodes
    while masks of the first `octree_depth` levels are left).
    depth 2: node 0 -> group 8, node 2 -> group 16; group 8 holds one reference (node 15 = CHUNK_OFFSET + 8), group 16 two (nodes 16,
    17 = CHUNK_OFFSET + 1, + 2); they stay references (their masks belong to level 2).  depth 1: the root's own 8 nodes only."""
    red = [255, 0, 0]
    ptrs, rgb, words = load(2)
    want = [CHUNK] * 24
    want[0], want[2], want[15], want[16], want[17] = 8, 16, CHUNK + 8, CHUNK + 1, CHUNK + 2
    assert ptrs.tolist() == want
    assert [i for i in range(24) if rgb[i].tolist() == red] == [0, 2, 15, 16, 17] and not rgb[[1, 3, 8, 23]].any()
    empty, solid = 0x80000000, 0x8FF00000
    assert words.tolist() == [8 << 4, empty, 16 << 4] + [empty] * 12 + [solid, solid, solid] + [empty] * 6   # to_octree, :233-252
    ptrs, rgb, words = load(1)
    assert ptrs.tolist() == [CHUNK + 1, CHUNK, CHUNK + 3] + [CHUNK] * 5
    assert words.tolist() == [solid, empty, solid] + [empty] * 5
    try:
        load(3)
        raise AssertionError("a depth above the stream's top level must be refused")
    except ValueError as e:
        assert "greater than top level" in str(e)


def test_hand_made_rsvo_stream(O):
    """The .rsvo reader against a stream NOT produced by this repository's writer (no real .rsvo model is in the reference checkout)."""
    def load(depth):
        t = O.Tree.from_rsvo(hand_made_rsvo(), depth)
        ptrs, rgb = t.raw()
        return ptrs, rgb, t.to_octree()
    check_hand_made_rsvo(load, O.CHUNK_OFFSET)


def hand_made_vox():
    """A MagicaVoxel file assembled by hand (format: 'VOX ' 150, MAIN { SIZE, XYZI, RGBA }): a 2 x 2 x 2 model with two voxels,
    (x, y, z, colour index) = (0, 0, 0, 1) and (1, 0, 1, 2); RGBA entry k colours index k + 1: entry 0 = (10, 20, 30), entry 1 =
    (200, 100, 50)."""
    import struct

    def chunk(cid, content, children=b""):
        return cid + struct.pack("<II", len(content), len(children)) + content + children
    size = chunk(b"SIZE", struct.pack("<III", 2, 2, 2))
    xyzi = chunk(b"XYZI", struct.pack("<I", 2) + bytes([0, 0, 0, 1]) + bytes([1, 0, 1, 2]))
    pal = bytearray(1024)
    pal[0:4] = bytes([10, 20, 30, 255])
    pal[4:8] = bytes([200, 100, 50, 255])
    rgba = chunk(b"RGBA", bytes(pal))
    return b"VOX " + struct.pack("<I", 150) + chunk(b"MAIN", b"", size + xyzi + rgba)


def check_hand_made_vox(words, V):
    """load_vox (cpu_octree.rs:177-210): depth = log2(2) = 1; pos = ((size - x - 1, z, y) / size) * 2 - 1:
    voxel 1 -> (0, -1, -1): CpuOctree::find_voxel compares with >= (cpu_octree.rs:57-61): child 4 * 1 + 0 + 0 = 4, colour of index 1 =
    RGBA entry 0; voxel 2 -> (-1, 0, -1): child 0 + 2 * 1 + 0 = 2, colour of index 2 = RGBA entry 1.  Voxel::to_value = r << 16 | g << 8 | b
    on top of VOXEL_OFFSET (octree.rs:28-35)."""
    want = [V << 4] * 8
    want[4] = (V + (10 << 16 | 20 << 8 | 30)) << 4
    want[2] = (V + (200 << 16 | 100 << 8 | 50)) << 4
    assert words.tolist() == want


def test_hand_made_vox_file(O):
    """The .vox parser, the colour-index convention inferred for dot_vox 4.1 (voxel.i = file index - 1) and load_vox's axis swap,
    on a file assembled by hand."""
    check_hand_made_vox(O.Tree.from_vox(hand_made_vox()).to_octree(), O.VOXEL_OFFSET)


def test_single_level_rays(O):
    """KATs 4-6: tree with only child 7 solid."""
    words = np.array([0x80000000] * 7 + [0x8FF00000], dtype=np.uint32)
    rays = np.array([[0.5, 0.5, -3, 0, 0, 1], [0.5, 0.5, -3, 0, 0, -1], [-0.5, -0.5, -3, 0, 0, 1]], dtype=np.float32)
    h = O.trace_rays(words, rays)
    i = O.unpack_info(h["info"])
    # KAT 4: dist = 2, first leaf child 6 (empty), one step t = 1, normal (0,0,-1), hit index 7, t_hit = 3
    assert h["value"][0] == 7 and h["t"][0] == 3.0 and i["steps"][0] == 1 and i["depth"][0] == 1 and i["hit"][0] == 1
    assert h["normal_bits"][0] == (2 << 4)
    # KAT 5: pointing away: ray_box_dist = 0 -> miss with value 0
    assert h["value"][1] == 0 and i["hit"][1] == 0 and h["t"][1] == 0.0
    # KAT 6: crosses the cube without hitting: value 0x20202000
    assert h["value"][2] == 0x20202000 and i["hit"][2] == 0 and i["steps"][2] == 1


def test_step_limit_sentinel(O):
    """KAT 7: > 100 steps -> hit = true, value 0xFF000000, depth 100 (shader.wgsl:242-244)."""
    z = np.load(f"{GOLDEN}/config1_small_256.npz")
    hits = z["hits"].view(O.HIT_DTYPE).reshape(-1)
    i = O.unpack_info(hits["info"])
    capped = hits["value"] == 0xFF000000
    assert capped.any()
    assert (i["steps"][capped] == 101).all() and (i["depth"][capped] == 100).all() and (i["hit"][capped] == 1).all()


def test_find_voxel_tie_breaks(O):
    """shader.wgsl:138-150: `>` (default) vs `>=` (misc_bool) on a boundary position."""
    words = np.array([(O.VOXEL_OFFSET + i + 1) << 4 for i in range(8)], dtype=np.uint32)
    v, pos, d = O.find_voxel(words, (0.0, 0.0, 0.0), misc_bool=False)
    assert (v, d, pos) == (0, 1, (-0.5, -0.5, -0.5))
    v, pos, d = O.find_voxel(words, (0.0, 0.0, 0.0), misc_bool=True)
    assert (v, d, pos) == (7, 1, (0.5, 0.5, 0.5))


def test_scan_rules(O):
    """KAT 8: compute.wgsl:34-46."""
    VO = O.VOXEL_OFFSET
    words = np.array([
        ((VO + 5) << 4) | 4,    # leaf, counter 4 -> subdivide
        ((VO + 5) << 4) | 3,    # leaf, counter 3 -> nothing
        (VO << 4) | 15,         # empty leaf (not > VOXEL_OFFSET) -> nothing
        (16 << 4) | 0,          # interior, counter 0 -> unsubdivide
        (16 << 4) | 1,          # interior, counter 1 -> nothing
        0,                      # zero word -> skipped
        ((VO + 1) << 4) | 15,   # leaf, counter 15 -> subdivide
        (24 << 4) | 0,          # interior, counter 0, but beyond node_length below
    ], dtype=np.uint32)
    sub, unsub = O.scan(words)
    assert sub[0] == 2 and sub[1:3].tolist() == [0, 6]
    assert unsub[0] == 2 and unsub[1:3].tolist() == [3, 7]
    sub, unsub = O.scan(words, node_length=7)
    assert unsub[0] == 1 and unsub[1] == 3


def test_golden_regression(O, small_words):
    """The committed config-1 records are what the oracle produces today (guards the oracle itself)."""
    z = np.load(f"{GOLDEN}/config1_small_256.npz")
    u = O.make_uniforms(width=256, height=256, flags=O.F_PAUSE_ADAPTIVE)
    assert np.array_equal(np.array(u.camera_inverse, dtype=np.float32), z["camera_inverse"])
    hits, stats = O.trace_frame(small_words, u, stats=True, threads=4)
    assert np.array_equal(hits.reshape(-1).view(np.uint32).reshape(-1, 4), z["hits"])
    assert np.array_equal(stats.reshape(-1, 2), z["stats"])
    # perfect-reuse words never exceed the restart-from-root words
    assert (stats[..., 1] <= stats[..., 0]).all()


def test_counter_side_effect_is_order_independent(O, small_words):
    """shader.wgsl:157-161 as a saturating count of visits: tiles in any order give the same words."""
    u = O.make_uniforms(width=64, height=64, flags=0)  # adaptive on, no shadows
    whole = O.count_frame(small_words, u)
    a = O.count_frame(O.count_frame(small_words, u, tile=(0, 0, 64, 32)), u, tile=(0, 32, 64, 32))
    b = O.count_frame(O.count_frame(small_words, u, tile=(0, 32, 64, 32)), u, tile=(0, 0, 64, 32))
    assert np.array_equal(whole, a) and np.array_equal(whole, b)
    assert np.array_equal(whole >> 4, small_words >> 4)  # pointers untouched
    assert (whole & 15).max() == 15
