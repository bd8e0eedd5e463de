"""CPU-side tree and loaders: reference src/cpu_octree.rs (CpuOctree)."""
import ctypes as C

import numpy as np

from ._lib import lib
from .octree import Octree, Voxel

CHUNK_OFFSET = 2147483648  # cpu_octree.rs:3


class CpuOctree:
    def __init__(self, mask=0, _handle=None, _owned=True):
        self._h = _handle if _handle else lib().svo_cpu_octree_new(mask)
        self._owned = _owned  # False once a World holds the chunk (world.py)

    @classmethod
    def new(cls, mask=0):
        return cls(mask)

    @staticmethod
    def _wrap(h, err):
        if not h:
            raise ValueError(err.value.decode() or "load failed")  # the reference returns Err(String)
        return CpuOctree(_handle=h)

    @classmethod
    def load_file(cls, file, octree_depth=0):
        """cpu_octree.rs:113-125: dispatch on the extension (.rsvo | .vox)."""
        err = C.create_string_buffer(256)
        return cls._wrap(lib().svo_cpu_octree_load_file(str(file).encode(), octree_depth, err, 256), err)

    @classmethod
    def load_vox(cls, data: bytes):
        err = C.create_string_buffer(256)
        return cls._wrap(lib().svo_cpu_octree_load_vox(data, len(data), err, 256), err)

    @classmethod
    def load_octree(cls, data: bytes, octree_depth):
        """.rsvo, cpu_octree.rs:128-175"""
        err = C.create_string_buffer(256)
        return cls._wrap(lib().svo_cpu_octree_load_rsvo(data, len(data), octree_depth, err, 256), err)

    @classmethod
    def from_voxels(cls, size, xyzi, palette):
        xyzi = np.ascontiguousarray(xyzi, dtype=np.uint8)
        palette = np.ascontiguousarray(palette, dtype=np.uint32)
        err = C.create_string_buffer(256)
        return cls._wrap(lib().svo_cpu_octree_from_voxels(size, xyzi.ctypes.data, xyzi.shape[0],
                                                          palette.ctypes.data, err, 256), err)

    def __del__(self):
        if getattr(self, "_h", None) and self._owned and lib is not None:  # (module globals are gone at interpreter exit)
            lib().svo_cpu_octree_free(self._h)
        self._h = None

    def __len__(self):
        return lib().svo_cpu_octree_len(self._h)

    def put_in_voxel(self, pos, voxel: Voxel, depth):
        lib().svo_cpu_octree_put_in_voxel(self._h, (C.c_float * 3)(*pos), (C.c_uint8 * 3)(voxel.r, voxel.g, voxel.b), depth)

    def put_in_block(self, pos, block_id, depth):
        lib().svo_cpu_octree_put_in_block(self._h, (C.c_float * 3)(*pos), block_id, depth)

    def find_voxel(self, pos, max_depth=None):
        idx, d, out = C.c_uint64(), C.c_uint32(), (C.c_float * 3)()
        lib().svo_cpu_octree_find_voxel(self._h, (C.c_float * 3)(*pos), -1 if max_depth is None else max_depth,
                                        C.byref(idx), C.byref(d), out)
        return idx.value, d.value, tuple(out)

    def get_node_mask(self, node):
        buf = (C.c_uint8 * 24)()
        lib().svo_cpu_octree_get_node_mask(self._h, node, buf)
        return [Voxel(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(8)]

    def to_octree_words(self):
        out = np.empty(len(self), dtype=np.uint32)
        lib().svo_cpu_octree_to_octree(self._h, out.ctypes.data)
        return out

    def to_octree(self):
        """cpu_octree.rs:233-252"""
        return Octree.from_words(self.to_octree_words())

    def raw(self):
        ptrs = np.empty(len(self), dtype=np.uint32)
        rgb = np.empty((len(self), 3), dtype=np.uint8)
        lib().svo_cpu_octree_raw(self._h, ptrs.ctypes.data, rgb.ctypes.data)
        return ptrs, rgb

    def generate_mip_tree(self):
        """World::generate_mip_tree for a single chunk (world.rs:234-336); returns top_mip."""
        top = (C.c_uint8 * 3)()
        lib().svo_cpu_octree_generate_mips(self._h, top)
        return Voxel(*top)

    def bin(self):
        """CpuOctree::bin (cpu_octree.rs:262-264): the <id>.bin image, 8 bytes per node."""
        n = lib().svo_cpu_octree_bin(self._h, None, 0)
        buf = np.empty(n, dtype=np.uint8)
        lib().svo_cpu_octree_bin(self._h, buf.ctypes.data, n)
        return buf.tobytes()

    @classmethod
    def from_bin(cls, data: bytes):
        """CpuOctree::from_bin (cpu_octree.rs:266-272)"""
        err = C.create_string_buffer(256)
        return cls._wrap(lib().svo_cpu_octree_from_bin(data, len(data), err, 256), err)

    def to_rsvo(self):
        n = lib().svo_rsvo_write(self._h, None, 0)
        if n == 0:
            raise ValueError("tree is not representable as .rsvo (non-empty leaf above the last level)")
        buf = np.empty(n, dtype=np.uint8)
        lib().svo_rsvo_write(self._h, buf.ctypes.data, n)
        return buf.tobytes()


def vox_parse(data: bytes):
    size = (C.c_uint32 * 3)()
    err = C.create_string_buffer(256)
    n = lib().svo_vox_parse(data, len(data), size, None, 0, None, err, 256)
    if n < 0:
        raise ValueError(err.value.decode())
    xyzi = np.empty((n, 4), dtype=np.uint8)
    pal = np.empty(256, dtype=np.uint32)
    lib().svo_vox_parse(data, len(data), size, xyzi.ctypes.data, xyzi.nbytes, pal.ctypes.data, err, 256)
    return tuple(size), xyzi, pal


def vox_write(size, xyzi, palette):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.uint8)
    palette = np.ascontiguousarray(palette, dtype=np.uint32)
    n = lib().svo_vox_write(size, xyzi.ctypes.data, xyzi.shape[0], palette.ctypes.data, None, 0)
    buf = np.empty(n, dtype=np.uint8)
    lib().svo_vox_write(size, xyzi.ctypes.data, xyzi.shape[0], palette.ctypes.data, buf.ctypes.data, n)
    return buf.tobytes()
