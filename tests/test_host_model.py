"""Host data model of the product (libsvo_hip.so svo_cpu_octree_* / svo_octree_* / camera / generators)
against the oracle's independent restatement, plus the C-ABI export check.  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_vox_fixture


def test_every_declared_symbol_is_exported(pkg):
    """Every function declared in include/*.h resolves in libsvo_hip.so (no compute calls here)."""
    lib = C.CDLL(pkg._lib.LIB_PATH)
    declared = []
    for h in ("svo_hip.h", "svo_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared += re.findall(r"\b(svo_[a-z0-9_]+)\s*\(", text)
    declared = sorted(set(declared))
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(pkg._lib.DEVICE_SYMBOLS + pkg._lib.HOST_SYMBOLS) == set(declared)


def test_no_device_is_reported_not_crashed(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = pkg._lib.lib().svo_ctx_create(0, C.byref(h))
    assert rc == -4 and not h.value  # SVO_ERR_NO_DEVICE
    with pytest.raises(pkg.SvoError):
        pkg.Gpu(0)


@pytest.mark.parametrize("name", ["small", "monu9", "monu10", "defualt", "phantom_mansion"])
def test_vox_to_node_words_matches_oracle(pkg, O, name):
    size, xyzi, pal, n, _ = load_vox_fixture(name)
    ours = pkg.CpuOctree.from_voxels(size, xyzi, pal)
    theirs = O.Tree.from_voxels(size, xyzi, pal)
    assert np.array_equal(ours.to_octree_words(), theirs.to_octree())
    p1, c1 = ours.raw()
    p2, c2 = theirs.raw()
    assert np.array_equal(p1, p2) and np.array_equal(c1, c2)
    # .vox writer -> reader round trip through both parsers (ragged chunk handling)
    blob = pkg.cpu_octree.vox_write(size, xyzi, pal)
    s1, x1, q1 = pkg.cpu_octree.vox_parse(blob)
    s2, x2, q2 = O.vox_parse(blob)
    assert s1 == s2 == (size, size, size) and np.array_equal(x1, xyzi) and np.array_equal(x2, xyzi)
    assert np.array_equal(q1, pal) and np.array_equal(q2, pal)
    assert np.array_equal(pkg.CpuOctree.load_vox(blob).to_octree_words(), theirs.to_octree())


def test_vox_errors(pkg, O, tmp_path):
    """cpu_octree.rs:113-125,180-189 error strings."""
    pal = np.arange(256, dtype=np.uint32)
    xyzi = np.array([[0, 0, 0, 1]], dtype=np.uint8)
    with pytest.raises(ValueError, match="power of 2"):
        pkg.CpuOctree.from_voxels(12, xyzi, pal)
    with pytest.raises(ValueError, match="power of 2"):
        O.Tree.from_voxels(12, xyzi, pal)
    with pytest.raises(ValueError):
        pkg.CpuOctree.load_vox(b"not a vox file at all....")
    p = tmp_path / "x.obj"
    p.write_bytes(b"hello")
    with pytest.raises(ValueError, match="Unknown file type"):
        pkg.CpuOctree.load_file(p)
    blob = pkg.cpu_octree.vox_write(8, xyzi, pal)
    q = tmp_path / "m.vox"
    q.write_bytes(blob)
    assert len(pkg.CpuOctree.load_file(q)) == 8 * 3  # one voxel at depth 3: the root group + 2 subdivisions


def test_hand_made_rsvo_stream(pkg):
    """The product's .rsvo reader on the stream assembled by hand in tests/test_oracle_kat.py (not a product of this repository's
    writer): same expected trees as the oracle's reader."""
    from test_oracle_kat import check_hand_made_rsvo, hand_made_rsvo

    def load(depth):
        t = pkg.CpuOctree.load_octree(hand_made_rsvo(), depth)
        ptrs, rgb = t.raw()
        return ptrs, rgb, t.to_octree_words()
    check_hand_made_rsvo(load, pkg.CHUNK_OFFSET)


def test_hand_made_vox_file(pkg):
    """The product's .vox reader on the file assembled by hand in tests/test_oracle_kat.py."""
    from test_oracle_kat import check_hand_made_vox, hand_made_vox
    check_hand_made_vox(pkg.CpuOctree.load_vox(hand_made_vox()).to_octree_words(), pkg.VOXEL_OFFSET)


def test_rsvo_round_trip(pkg, O):
    """.rsvo BFS child-mask stream (cpu_octree.rs:128-175): writer -> both loaders, and depth truncation."""
    size, xyzi, pal, n, _ = load_vox_fixture("small")
    tree = pkg.CpuOctree.from_voxels(size, xyzi, pal)
    blob = tree.to_rsvo()
    for depth in (3, 2, 1):
        a = pkg.CpuOctree.load_octree(blob, depth)
        b = O.Tree.from_rsvo(blob, depth)
        assert np.array_equal(a.to_octree_words(), b.to_octree())
        pa, _ = a.raw()
        # the loader's leaves are block references CHUNK_OFFSET + i % 8 + 1, drawn red
        leaves = pa > pkg.CHUNK_OFFSET
        assert ((pa[leaves] - pkg.CHUNK_OFFSET - 1) == (np.flatnonzero(leaves) % 8)).all()
    # the loader rebuilds the array breadth-first, so indices differ from the insertion-ordered tree;
    # the geometry must be the same: every cell centre of the 8^3 grid locates a leaf of equal depth/solidity
    full = pkg.CpuOctree.load_octree(blob, 3).to_octree_words()
    want = tree.to_octree_words()
    assert full.size == want.size
    for x in range(8):
        for y in range(8):
            for z in range(8):
                p = ((x + 0.5) / 4 - 1, (y + 0.5) / 4 - 1, (z + 0.5) / 4 - 1)
                v1, _, d1 = O.find_voxel(full, p)
                v2, _, d2 = O.find_voxel(want, p)
                assert d1 == d2
                assert ((full[v1] >> 4) > pkg.VOXEL_OFFSET) == ((want[v2] >> 4) > pkg.VOXEL_OFFSET)
    with pytest.raises(ValueError, match="greater than top level"):
        pkg.CpuOctree.load_octree(blob, 9)
    with pytest.raises(ValueError):
        O.Tree.from_rsvo(blob, 9)


def test_mip_tree_matches_oracle(pkg, O):
    """World::generate_mip_tree (world.rs:234-336): bottom-up averages, clamp >= 1."""
    size, xyzi, pal, n, _ = load_vox_fixture("monu9")
    ours = pkg.CpuOctree.from_voxels(size, xyzi, pal)
    theirs = O.Tree.from_voxels(size, xyzi, pal)
    top_a = ours.generate_mip_tree()
    top_b = theirs.generate_mips()
    assert (top_a.r, top_a.g, top_a.b) == tuple(top_b)
    pa, ca = ours.raw()
    pb, cb = theirs.raw()
    assert np.array_equal(ca, cb)
    interior = pa < pkg.CHUNK_OFFSET
    assert (ca[interior] >= 1).all()
    # hand check of one interior node: mean of its non-black children, truncated
    i = int(np.flatnonzero(interior)[-1])
    kids = ca[pa[i]:pa[i] + 8].astype(np.float32)
    nz = kids[(kids != 0).any(axis=1)]
    assert ca[i].tolist() == np.maximum(nz.mean(axis=0).astype(np.uint8), 1).tolist()


def test_mip_colours_by_hand(pkg, O):
    """World::generate_mip_tree (world.rs:234-336) on a tree small enough to do on paper.  Root child 0 is subdivided (put_in_voxel
    at depth 2, cpu_octree.rs:100-111) and holds (10, 20, 30) and (20, 40, 61) among six black children; root child 5 is the leaf
    (100, 0, 7).  A node's colour = the f32 mean of its non-black children, truncated to u8, each channel at least 1:
    node 0 = ((10 + 20) / 2, (20 + 40) / 2, (30 + 61) / 2 = 45.5 -> 45).  Root child 3 is subdivided into eight black leaves: 0 / 0 = NaN ->
    `as u8` 0 -> max(1): (1, 1, 1) -- which then counts as a non-black child of the root: top_mip = mean of (15, 30, 45), (1, 1, 1), (100, 0, 7)."""
    V = pkg.Voxel
    for make, put, mips, raw in (
            (lambda: pkg.CpuOctree.new(0), lambda t, p, c, d: t.put_in_voxel(p, V(*c), d), lambda t: (lambda v: (v.r, v.g, v.b))(t.generate_mip_tree()), lambda t: t.raw()),
            (lambda: O.Tree.new(0), lambda t, p, c, d: t.put_in_voxel(p, c, d), lambda t: tuple(t.generate_mips()), lambda t: t.raw())):
        t = make()
        put(t, (-0.75, -0.75, -0.75), (10, 20, 30), 2)   # root child 0 -> group 8, its child 0
        put(t, (-0.75, -0.75, -0.25), (20, 40, 61), 2)   # its child 1 (z >= -0.5)
        put(t, (0.5, -0.5, 0.5), (100, 0, 7), 1)         # root child 4 + 0 + 1 = 5
        put(t, (-0.75, 0.75, 0.75), (0, 0, 0), 2)        # root child 3 subdivided into eight black leaves -> group 16
        top = mips(t)
        ptrs, rgb = raw(t)
        assert ptrs[0] == 8 and ptrs[3] == 16 and len(ptrs) == 24
        assert rgb[8].tolist() == [10, 20, 30] and rgb[9].tolist() == [20, 40, 61] and rgb[5].tolist() == [100, 0, 7]
        assert rgb[0].tolist() == [15, 30, 45]
        assert rgb[3].tolist() == [1, 1, 1]
        # the root: children 0 (15, 30, 45), 3 (1, 1, 1), 5 (100, 0, 7): (116 / 3 = 38.67 -> 38, 31 / 3 = 10.33 -> 10, 53 / 3 = 17.67 -> 17)
        assert top == (38, 10, 17)


def test_camera_matrices_by_hand(pkg, O):
    """Render::update's matrices (render.rs:191-206, main.rs:139-162) for a pose that can be done on paper: eye (0, 0, -2) looking
    along +z, 200 x 100 pixels, fov 90.  cgmath's look_at_rh: f = (0, 0, 1), s = normalize(f x up) = (-1, 0, 0), u = s x f = (0, 1, 0);
    view columns (s.x, u.x, -f.x, 0), (s.y, u.y, -f.y, 0), (s.z, u.z, -f.z, 0), (-eye.s, -eye.u, eye.f, 1) = (-1,0,0,0), (0,1,0,0), (0,0,-1,0),
    (0,0,-2,1).  proj = diag(a k, k, -1, 1) with k = 1 / tan(45 degrees) and -- the reference's quirk -- a = HEIGHT / WIDTH = 0.5.
    camera = proj * view: columns (-a k, 0, 0, 0), (0, k, 0, 0), (0, 0, 1, 0), (0, 0, 2, 1); its inverse: (-1 / (a k), 0, 0, 0),
    (0, 1 / k, 0, 0), (0, 0, 1, 0), (0, 0, -2, 1).  Everything but k is exact; k is 1 up to the rounding of tan."""
    for cam, inv in (pkg.camera_matrices((0.0, 0.0, -2.0), (0.0, 0.0, 1.0), 90.0, 200, 100), O.camera((0.0, 0.0, -2.0), (0.0, 0.0, 1.0), 90.0, 200.0, 100.0)):
        k = float(cam[5])
        assert abs(k - 1.0) < 1e-6
        assert cam.tolist() == [np.float32(-0.5 * k), 0, 0, 0, 0, np.float32(k), 0, 0, 0, 0, 1, 0, 0, 0, 2, 1]
        assert inv[[1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13]].tolist() == [0] * 11 and inv[[10, 14, 15]].tolist() == [1, -2, 1]
        assert abs(float(inv[0]) * (-0.5 * k) - 1.0) < 1e-6 and abs(float(inv[5]) * k - 1.0) < 1e-6
        # the ray through the centre of the frame: origin = camera_inverse * (0, 0, 0, 1) = the eye, direction +z (shader.wgsl:54-59)
        assert inv[12:15].tolist() == [0, 0, -2]


def test_octree_subdivide_unsubdivide(pkg):
    """octree.rs:51-110: free-list reuse, panics as exceptions, pos_offset KAT 2."""
    V = pkg.Voxel
    o = pkg.Octree.new([V(i + 1, 0, 0) for i in range(8)])
    assert len(o) == 8 and o.nodes[3] == V(4, 0, 0).to_value()
    assert pkg.Octree.pos_offset(5, 2) == (0.25, -0.25, 0.25)   # ((1,0,1) * 2 - 1) / 4
    o.subdivide(2, [V(0, 9, i) for i in range(8)], 2)
    assert len(o) == 16 and o.nodes[2] == pkg.create_node(8) and o.get_node(2) == 8
    with pytest.raises(RuntimeError, match="already subdivided"):
        o.subdivide(2, [V(0, 0, 0)] * 8, 2)
    idx, depth, pos = o.find_voxel((-0.9, 0.9, -0.9))
    assert depth == 2 and idx == 8 + 2  # child 2 of the root (x-,y+,z-), then its child (x-,y+,z-) = 2
    assert o.unsubdivide(2) is True
    assert o.nodes[2] == V(255, 0, 0).to_value() and o.hole_count() == 1
    assert o.unsubdivide(2) is False  # "not subdivided": reference prints and returns
    o.subdivide(5, [V(7, 7, 7)] * 8, 2)  # reuses the hole at 8
    assert len(o) == 16 and o.get_node(5) == 8 and o.hole_count() == 0
    ex = o.expanded(40)
    assert ex.size == 40 and (ex[16:] == 0).all() and np.array_equal(ex[:16], o.raw_data())
    # round trip through words keeps find_voxel working
    o2 = pkg.Octree.from_words(o.raw_data())
    assert o2.find_voxel((0.9, -0.9, 0.9))[0] == o.find_voxel((0.9, -0.9, 0.9))[0]


def test_camera_matrices_match_oracle(pkg, O):
    """render.rs:191-206 + main.rs:139-162: both host restatements agree; inverse really inverts."""
    for pos, look, fov, w, h in (((0.1, 0.2, -1.5), (0, 0, 1.5), 90.0, 1920, 1080),
                                 ((1.3, 0.9, 1.2), (-1.0, -0.6, -1.0), 70.0, 640, 480)):
        cam, inv = pkg.camera_matrices(pos, look, fov, w, h)
        ocam, oinv = O.camera(pos, look, fov, w, h)
        assert np.array_equal(cam, ocam) and np.array_equal(inv, oinv)
        m = cam.reshape(4, 4).T.astype(np.float64) @ inv.reshape(4, 4).T.astype(np.float64)
        assert np.allclose(m, np.eye(4), atol=1e-5)
    cam, _ = pkg.camera_matrices((0, 0, -2), (0, 0, 1), 90.0, 200, 100)
    assert cam[0] == pytest.approx(-0.5) and cam[5] == pytest.approx(1.0)  # aspect passed as H/W (render.rs:200); right = -x when looking down +z


def test_generators_are_deterministic_and_wellformed(pkg):
    a = pkg.scenes.random_tree(seed=4, max_depth=6, p_split=0.5, p_solid=0.4)
    b = pkg.scenes.random_tree(seed=4, max_depth=6, p_split=0.5, p_solid=0.4)
    assert np.array_equal(a, b) and a.size % 8 == 0 and pkg.scenes.max_depth(a) <= 6
    cam, look = pkg.scenes.terrain_camera(0, 12)
    t1 = pkg.scenes.terrain(seed=0, max_depth=12, cam=cam, lod_c=200.0, max_words=2_000_000)
    t2 = pkg.scenes.terrain(seed=0, max_depth=12, cam=cam, lod_c=200.0, max_words=2_000_000)
    assert np.array_equal(t1, t2) and pkg.scenes.max_depth(t1) == 12
    ptr = t1 >> 4
    interior = ptr < pkg.VOXEL_OFFSET
    assert (ptr[interior] % 8 == 0).all() and (ptr[interior] + 8 <= t1.size).all() and (ptr[interior] > 0).all()
    # BFS layout: children always come after their parent word
    assert (ptr[interior] > np.flatnonzero(interior)).all()
    f = pkg.scenes.fractal(seed=1, max_depth=10, cam=(0.3, 0.3, 0.3), lod_c=400.0, max_words=2_000_000)
    assert pkg.scenes.max_depth(f) == 10
    capped = pkg.scenes.terrain(seed=0, max_depth=12, cam=cam, lod_c=200.0, max_words=80_000)
    assert capped.size <= 80_000 and pkg.scenes.max_depth(capped) >= 1


def test_host_find_voxel_matches_oracle_descent(pkg, O, monu9_words):
    """Octree::find_voxel (`>=`, octree.rs:113-141) equals the shader's find_voxel with misc_bool set."""
    o = pkg.Octree.from_words(monu9_words)
    rng = np.random.default_rng(3)
    for p in rng.uniform(-1, 1, (200, 3)).astype(np.float32):
        idx, depth, pos = o.find_voxel(p.tolist())
        v, vpos, d = O.find_voxel(monu9_words, p.tolist(), misc_bool=True)
        assert (idx, depth, pos) == (v, d, vpos)


def test_vox_without_rgba_chunk_uses_the_default_palette(pkg, O):
    """dot_vox falls back to MagicaVoxel's default palette for files saved without an RGBA chunk and the
    reference loads them (cpu_octree.rs:177-193).  Product and oracle (independent codings of the format's table) agree,
    and known entries come out: file index 1 = white, 216 = the first step of the red ramp, 255 = grey 0x11."""
    import struct
    size = 4
    xyzi = np.array([[0, 0, 0, 1], [1, 0, 0, 216], [2, 0, 0, 255], [3, 3, 3, 37]], dtype=np.uint8)
    blob = pkg.cpu_octree.vox_write(size, xyzi, np.arange(256, dtype=np.uint32))
    at = blob.index(b"RGBA")
    n = struct.unpack_from("<I", blob, at + 4)[0]
    cut = blob[:at] + blob[at + 12 + n:]
    cut = cut[:16] + struct.pack("<I", struct.unpack_from("<I", cut, 16)[0] - (12 + n)) + cut[20:]  # MAIN's children size
    ours = pkg.CpuOctree.load_vox(cut).to_octree_words()
    theirs = O.Tree.from_vox(cut).to_octree()
    assert np.array_equal(ours, theirs)
    leaves = sorted(int(w >> 4) - (1 << 27) for w in ours if (w >> 4) > (1 << 27))
    # colour value = R << 16 | G << 8 | B (cpu_octree.rs:204 via Voxel::to_value)
    j = 36  # file index 37 -> cube entry j = 36: R = 0xff - 0x33, G = B = 0xff
    assert leaves == sorted([0xFFFFFF, 0xEE0000, 0x111111, (0xFF - 0x33) << 16 | 0xFF << 8 | 0xFF])


def test_relayout_keeps_the_tree_and_translates_indices_back(pkg, O, monu9_words):
    """svo_nodes_relayout (round 5): the child groups of a tree in another order -- upper levels breadth-first, every subtree below
    block_level as one contiguous block -- is the same tree: every ray's record is the same but for the voxel index, and the permutation
    it returns translates that back.  Trees of three builders, several block levels; unreachable padding is dropped."""
    terrain = pkg.scenes.terrain(seed=2, max_depth=12, cam=(0.1, 0.3, -0.2), lod_c=300.0, max_words=3_000_000)
    random9 = pkg.scenes.random_tree(seed=5, max_depth=9, p_split=0.55, p_solid=0.25, max_words=1 << 20)
    u = O.make_uniforms((0.1, 0.3, -0.2), (0.0, -0.3, 1.0), 90.0, 160, 90)
    for words, levels in ((terrain, (1, 6, 10, 31)), (random9, (3, 8)), (np.asarray(monu9_words), (2, 5))):
        want = O.trace_frame(words, u, threads=4).reshape(-1)
        for b in levels:
            out, perm = pkg.scenes.relayout(words, b, with_perm=True)
            assert out.size == words.size and pkg.scenes.max_depth(out) == pkg.scenes.max_depth(words)
            assert np.array_equal(np.sort(perm), np.arange(words.size)), "a permutation of the words"
            got = O.trace_frame(out, u, threads=4).reshape(-1)
            assert np.array_equal(got["t"], want["t"]) and np.array_equal(got["info"], want["info"]) and np.array_equal(got["normal_bits"], want["normal_bits"])
            real = want["value"] < words.size
            assert np.array_equal(perm[got["value"][real]], want["value"][real]) and np.array_equal(got["value"][~real], want["value"][~real])
    padded = np.concatenate([terrain, np.zeros(64, dtype=np.uint32)])
    assert pkg.scenes.relayout(padded, 6).size == terrain.size
    bad = terrain.copy()
    bad[0] = (bad.size + 8) << 4  # a pointer past the array
    with pytest.raises(ValueError):
        pkg.scenes.relayout(bad, 6)
