"""Chunk table with block instancing: reference src/world.rs (World)."""
import ctypes as C
import os

import numpy as np

from ._lib import lib
from .cpu_octree import CpuOctree
from .octree import Voxel

BLOCK_NAMES = ("stone", "dirt", "grass", "wood", "leaf", "slate", "crystal", "glass")  # ids 1..8, world.rs:19-58


class World:
    """world.rs:5-9.  Chunk 0 is the root tree; a leaf whose pointer is CHUNK_OFFSET + id continues in chunk
    `id` (ids 1..8: the 16^3 block models; ids >= CHUNK_OFFSET / 2: streamed terrain chunks)."""

    def __init__(self, path="", _handle=None):
        self._h = _handle if _handle else lib().svo_world_new(str(path).encode())
        self.path = str(path)

    @classmethod
    def new(cls, path="", blocks_dir=None):
        """World::new.  The reference loads blocks/<name>.vox as chunks 1..8 from its working directory
        (world.rs:19-58); here that happens only when the directory is given."""
        w = cls(path)
        if blocks_dir is not None:
            for i, name in enumerate(BLOCK_NAMES):
                w.insert(i + 1, CpuOctree.load_file(os.path.join(blocks_dir, name + ".vox"), 0))
                w.generate_mip_tree(i + 1)
        return w

    @classmethod
    def load_world(cls, path):
        """world.rs:159-174"""
        if not os.path.exists(path):
            raise ValueError("File doesn't exist!")
        err = C.create_string_buffer(256)
        h = lib().svo_world_load(str(path).encode(), err, 256)
        if not h:
            raise ValueError(err.value.decode())
        return cls(path, _handle=h)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter exit)
            lib().svo_world_free(self._h)
        self._h = None

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError(lib().svo_world_last_error(self._h).decode())  # the reference panics (unwrap)
        return rc

    def insert(self, chunk_id, chunk: CpuOctree):
        """chunks.insert(id, chunk): the world owns the chunk from here on."""
        self._check(lib().svo_world_insert(self._h, chunk_id, chunk._h))
        chunk._owned = False

    def remove(self, chunk_id):
        return lib().svo_world_remove(self._h, chunk_id) == 0

    def contains(self, chunk_id):
        return bool(lib().svo_world_chunk(self._h, chunk_id))

    def chunk(self, chunk_id):
        h = lib().svo_world_chunk(self._h, chunk_id)
        if not h:
            raise KeyError(chunk_id)
        return CpuOctree(_handle=h, _owned=False)  # borrowed: valid while the world holds the chunk

    def chunk_ids(self):
        n = lib().svo_world_chunk_ids(self._h, None, 0)
        ids = np.empty(n, dtype=np.uint32)
        lib().svo_world_chunk_ids(self._h, ids.ctypes.data, n)
        return ids.tolist()

    def find_voxel(self, pos, max_depth=None):
        """world.rs:201-232 -> (chunk, index, depth, pos)"""
        ch, idx, d, out = C.c_uint32(), C.c_uint64(), C.c_uint32(), (C.c_float * 3)()
        self._check(lib().svo_world_find_voxel(self._h, (C.c_float * 3)(*pos), -1 if max_depth is None else max_depth,
                                               C.byref(ch), C.byref(idx), C.byref(d), out))
        return ch.value, idx.value, d.value, tuple(out)

    def generate_mip_tree(self, chunk_id):
        """world.rs:234-336; returns the chunk's top_mip"""
        top = (C.c_uint8 * 3)()
        self._check(lib().svo_world_generate_mip_tree(self._h, chunk_id, top))
        return Voxel(*top)

    def save_chunk(self, chunk_id):
        self._check(lib().svo_world_save_chunk(self._h, chunk_id))

    def load_chunk(self, chunk_id):
        self._check(lib().svo_world_load_chunk(self._h, chunk_id))

    def get_node_mask(self, first_child, chunk_id=0):
        return self.chunk(chunk_id).get_node_mask(first_child)

    def root_octree(self):
        """App::new (app.rs:47-48): the device tree starts as the root chunk's 8 mip-coloured children."""
        from .octree import Octree
        return Octree.new(self.get_node_mask(0))

    def expand(self, octree, max_depth, cam=None, lod_c=0.0, max_words=1 << 27):
        """Fixed point of the subdivision loop for a distance rule (svo_world_expand); returns #subdivisions."""
        camv = (C.c_float * 3)(*(cam if cam is not None else (0.0, 0.0, 0.0)))
        return lib().svo_world_expand(self._h, octree._h, max_depth, camv, float(lod_c if cam is not None else 0.0),
                                      max_words)
