"""Chunk table with block instancing, chunk dumps and the CPU half of the streaming loop (world.rs, adaptive.rs,
cpu_octree.rs:262-272): the native host model (svo_world_*, svo_adaptive_*) against the pure-Python restatement
in oracle/world_oracle.py.  CPU only."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import world_oracle as WO

BLOCKS = ("stone", "dirt", "grass", "wood", "leaf", "slate", "crystal", "glass")


def block_voxels():
    z = np.load(os.path.join(GOLDEN, "blocks_vox.npz"))
    return [(z[n + "_xyzi"], z[n + "_palette"]) for n in BLOCKS]


def shell_rsvo(pkg, depth, radius=0.7):
    """A sphere shell at `depth` as an .rsvo stream (the format of the reference's statuette model)."""
    tree = pkg.CpuOctree.new(0)
    n = 1 << depth
    ax = (np.arange(n) + 0.5) / n * 2 - 1
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z)
    for i, j, k in np.argwhere(np.abs(r - radius) < 1.0 / n):
        tree.put_in_voxel((float(ax[i]), float(ax[j]), float(ax[k])), pkg.Voxel(9, 9, 9), depth)
    return tree.to_rsvo()


def make_worlds(pkg, O, depth=2):
    """The same world on both sides: chunk 0 = .rsvo shell whose leaves reference blocks 1..8 (cpu_octree.rs:37),
    chunks 1..8 = the block models with their mip colours (World::new, world.rs:19-58)."""
    world, oworld = pkg.World.new(""), WO.World()
    for i, (xyzi, pal) in enumerate(block_voxels()):
        world.insert(i + 1, pkg.CpuOctree.from_voxels(16, xyzi, pal))
        world.generate_mip_tree(i + 1)
        ptr, rgb = O.Tree.from_voxels(16, xyzi, pal).raw()
        oworld.chunks[i + 1] = WO.Chunk(ptr, rgb)
        oworld.generate_mip_tree(i + 1)
    blob = shell_rsvo(pkg, depth)
    world.insert(0, pkg.CpuOctree.load_octree(blob, depth))
    ptr, rgb = O.Tree.from_rsvo(blob, depth).raw()
    oworld.chunks[0] = WO.Chunk(ptr, rgb)
    return world, oworld


def assert_chunk_equal(chunk, ochunk):
    ptr, rgb = chunk.raw()
    assert ptr.tolist() == ochunk.ptr
    assert [tuple(c) for c in rgb.tolist()] == ochunk.rgb


def test_block_mips_and_world_find_voxel(pkg, O):
    world, oworld = make_worlds(pkg, O)
    top = world.generate_mip_tree(0)
    assert (top.r, top.g, top.b) == oworld.generate_mip_tree(0)
    for cid in range(9):
        assert_chunk_equal(world.chunk(cid), oworld.chunks[cid])
    # block leaves of the root chunk now show the referenced block's top_mip (world.rs:273-279)
    ptr, rgb = world.chunk(0).raw()
    refs = np.flatnonzero(ptr > pkg.CHUNK_OFFSET)
    assert refs.size > 8 and set((ptr[refs] - pkg.CHUNK_OFFSET).tolist()) <= set(range(1, 9))
    assert ((ptr[refs] - pkg.CHUNK_OFFSET) == refs % 8 + 1).all()  # add_voxels: id = index % 8 + 1
    for i in refs[:16]:
        assert tuple(rgb[i]) == oworld.chunks[int(ptr[i] - pkg.CHUNK_OFFSET)].top_mip
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (400, 3)).astype(np.float32)
    pts[:40] = np.round(pts[:40] * 8) / 8  # points on cell boundaries: the `>=` side matters
    crossed = 0
    for p in pts.tolist():
        for md in (None, 1, 2, 3, 5):
            got = world.find_voxel(p, md)
            assert got == oworld.find_voxel(p, md), (p, md)
            crossed += got[0] != 0
    assert crossed > 50  # walks that continue inside a block chunk


def test_world_missing_chunk_is_an_error_not_a_crash(pkg):
    world = pkg.World.new("")
    root = pkg.CpuOctree.new(0b00000100)  # child 2 references block 3, which is not loaded
    world.insert(0, root)
    with pytest.raises(RuntimeError, match="chunk 3"):
        world.find_voxel((-0.5, 0.5, -0.5))
    with pytest.raises(RuntimeError, match="block 3"):
        world.generate_mip_tree(0)
    assert world.find_voxel((0.5, 0.5, 0.5))[:3] == (0, 7, 1)
    with pytest.raises(KeyError):
        world.chunk(9)


def test_streaming_lists_against_restatement(pkg, O):
    """process_subdivision / process_unsubdivision (adaptive.rs): same lists -> same device words, positions
    and hole reuse as the restatement, including block-reference leaves (:49-58)."""
    world, oworld = make_worlds(pkg, O)
    world.generate_mip_tree(0)
    oworld.generate_mip_tree(0)
    octree = world.root_octree()
    ooctree = WO.Octree(oworld.chunks[0].get_node_mask(0))
    assert octree.raw_data().tolist() == ooctree.nodes
    rng = np.random.default_rng(11)
    for rnd in range(7):
        words = octree.raw_data()
        leaves = np.flatnonzero((words >> 4) > pkg.VOXEL_OFFSET)
        pick = rng.permutation(leaves)[:max(4, leaves.size // 3)].astype(np.uint32)
        if rnd % 2:
            pick = np.concatenate([pick, pick[:3]])  # duplicates: "Doubleup!" (:32-35)
        before = octree.raw_data()
        n = pkg.adaptive.process_subdivision(pick, octree, world)
        WO.process_subdivision(pick.tolist(), ooctree, oworld)
        assert octree.raw_data().tolist() == ooctree.nodes, f"round {rnd}"
        assert n > 0
        if rnd == 0:
            octree.take_dirty()  # (words written while the root tree was built)
        else:
            # the dirty set is exactly what an incremental upload must send: applying it to the old array gives the new one
            idx, val = octree.take_dirty()
            assert np.unique(idx).size == idx.size
            patched = np.concatenate([before, np.zeros(len(octree) - before.size, dtype=np.uint32)])
            patched[idx] = val
            assert np.array_equal(patched, octree.raw_data())
            assert octree.take_dirty()[0].size == 0
        if rnd in (2, 4):  # collapse some interior nodes again; the freed groups are reused next round
            interior = np.flatnonzero(((octree.raw_data() >> 4) < pkg.VOXEL_OFFSET))
            drop = rng.permutation(interior)[:5].astype(np.uint32)
            before = octree.raw_data()
            pkg.adaptive.process_unsubdivision(drop, octree, world)
            WO.process_unsubdivision(drop.tolist(), ooctree, oworld)
            assert octree.raw_data().tolist() == ooctree.nodes
            idx, val = octree.take_dirty()
            before[idx] = val
            assert np.array_equal(before, octree.raw_data()) and set(idx.tolist()) == set(drop.tolist())
            assert octree.hole_count() == len(ooctree.hole_stack) > 0
    for i in rng.integers(0, len(octree), 64).tolist():
        assert octree.position(i) == ooctree.positions[i]
    # depth reached: shell depth 2 + some block levels
    depths = [octree.find_voxel(p)[1] for p in rng.uniform(-1, 1, (200, 3)).tolist()]
    assert max(depths) >= 5


def test_expand_is_the_fixed_point_of_sorted_passes(pkg, O):
    world, oworld = make_worlds(pkg, O)
    world.generate_mip_tree(0)
    oworld.generate_mip_tree(0)
    full = world.root_octree()
    n_sub = world.expand(full, max_depth=6)
    ooctree = WO.Octree(oworld.chunks[0].get_node_mask(0))
    while True:  # process_subdivision fed every leaf in index order, until nothing changes
        before = len(ooctree.nodes)
        leaves = [i for i, w in enumerate(ooctree.nodes) if (w >> 4) >= WO.VOXEL_OFFSET]
        leaves = [i for i in leaves if ooctree.find_voxel(ooctree.positions[i])[1] < 6]
        WO.process_subdivision(leaves, ooctree, oworld)
        if len(ooctree.nodes) == before:
            break
    assert full.raw_data().tolist() == ooctree.nodes
    assert n_sub == (len(full) - 8) // 8
    # shell depth 2 + 16^3 blocks (4 levels) = depth 6 everywhere a block is not uniform
    words = full.raw_data()
    assert pkg._lib.lib().svo_nodes_max_depth(words.ctypes.data, words.size) == 6
    # distance rule: refinement stops away from the camera, the array is a prefix-closed tree
    near = world.root_octree()
    world.expand(near, max_depth=6, cam=(0.7, 0.0, 0.0), lod_c=8.0)
    assert 8 < len(near) < len(full)
    capped = world.root_octree()
    world.expand(capped, max_depth=6, max_words=4096)
    assert len(capped) <= 4096 and len(capped) % 8 == 0


def test_chunk_dumps_and_on_demand_loading(pkg, O, tmp_path):
    """<id>.bin (world.rs:159-198, cpu_octree.rs:262-272): 8 bytes per node; load_world reads 0.bin; a listed
    leaf whose chunk is missing triggers a load and is subdivided when listed again (adaptive.rs:49-58)."""
    world, oworld = make_worlds(pkg, O, depth=1)
    streamed = pkg.CHUNK_OFFSET // 2 + 3  # ids >= CHUNK_OFFSET / 2 are streamed terrain chunks (world.rs:104)
    xyzi, pal = block_voxels()[2]
    world.insert(streamed, pkg.CpuOctree.from_voxels(16, xyzi, pal))
    top = world.generate_mip_tree(streamed)
    root = pkg.CpuOctree.new(0)
    root.put_in_block((-0.5, -0.5, -0.5), streamed, 1)
    root.put_in_block((0.5, 0.5, 0.5), 2, 1)
    world.insert(0, root)
    world.generate_mip_tree(0)
    w2 = pkg.World(str(tmp_path))
    # save through a world rooted at tmp_path
    for cid in (0, streamed):
        blob = world.chunk(cid).bin()
        (tmp_path / f"{cid}.bin").write_bytes(blob)
    blob = (tmp_path / "0.bin").read_bytes()
    assert len(blob) == 8 * 8
    ptr0, r, g, b, pad = struct.unpack_from("<IBBBB", blob, 0)
    assert ptr0 == pkg.CHUNK_OFFSET + streamed and (r, g, b) == (top.r, top.g, top.b) and pad == 0
    assert WO.Chunk.from_bin(blob).bin() == blob
    back = pkg.CpuOctree.from_bin(blob)
    assert back.bin() == blob
    with pytest.raises(ValueError):
        pkg.CpuOctree.from_bin(blob[:-3])
    # save_chunk writes the same bytes
    w2.insert(0, pkg.CpuOctree.from_bin(blob))
    os.remove(tmp_path / "0.bin")
    w2.save_chunk(0)
    assert (tmp_path / "0.bin").read_bytes() == blob
    # load_world: only the root chunk is resident
    with pytest.raises(ValueError, match="doesn't exist"):
        pkg.World.load_world(str(tmp_path / "nope"))
    w3 = pkg.World.load_world(str(tmp_path))
    assert w3.chunk_ids() == [0]
    octree = w3.root_octree()
    lst = np.array([0], dtype=np.uint32)  # child 0 = the streamed chunk's leaf
    assert pkg.adaptive.process_subdivision(lst, octree, w3) == 0 and len(octree) == 8
    assert w3.contains(streamed)  # loaded by the first request ...
    assert pkg.adaptive.process_subdivision(lst, octree, w3) == 1 and len(octree) == 16  # ... used by the second
    want = [WO.to_value(tuple(c)) for c in world.chunk(streamed).raw()[1][:8].tolist()]
    assert octree.raw_data()[8:16].tolist() == want
    # block 2 is neither resident nor on disk: the request fails quietly, as the reference's load would
    assert pkg.adaptive.process_subdivision(np.array([7], dtype=np.uint32), octree, w3) == 0
    # collapsing the streamed chunk's node drops the chunk from the table (adaptive.rs:104-110)
    assert pkg.adaptive.process_unsubdivision(lst, octree, w3) == 1
    assert not w3.contains(streamed) and octree.hole_count() == 1
    assert octree.raw_data()[0] == WO.to_value((top.r, top.g, top.b))
