"""Host mirror of the device node array: reference src/octree.rs (Octree, Voxel, create_node)."""
import ctypes as C

import numpy as np

from ._lib import lib

VOXEL_OFFSET = 134217728  # octree.rs:5


class Voxel:
    """octree.rs:7-35"""
    __slots__ = ("r", "g", "b")

    def __init__(self, r, g, b):
        self.r, self.g, self.b = int(r), int(g), int(b)

    @staticmethod
    def from_value(value):
        return Voxel((value >> 16) & 0xFF, (value >> 8) & 0xFF, value & 0xFF)

    def to_cpu_value(self):
        return (self.r << 16) | (self.g << 8) | self.b

    def to_value(self):
        return ((VOXEL_OFFSET + self.to_cpu_value()) << 4) & 0xFFFFFFFF

    def __eq__(self, o):
        return (self.r, self.g, self.b) == (o.r, o.g, o.b)

    def __repr__(self):
        return f"({self.r}, {self.g}, {self.b})"


def create_node(value):
    """octree.rs:164-166"""
    return (int(value) << 4) & 0xFFFFFFFF


def _mask_bytes(mask):
    arr = (C.c_uint8 * 24)()
    for i, v in enumerate(mask):
        arr[3 * i], arr[3 * i + 1], arr[3 * i + 2] = v.r, v.g, v.b
    return arr


class Octree:
    """octree.rs:43-162.  `nodes` is the word array uploaded to the device node buffer."""

    def __init__(self, mask=None, _handle=None):
        self._h = _handle if _handle else lib().svo_octree_new(_mask_bytes(mask))

    @classmethod
    def new(cls, mask):
        return cls(mask)

    @classmethod
    def from_words(cls, words):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        return cls(_handle=lib().svo_octree_from_words(words.ctypes.data, words.size))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter exit)
            lib().svo_octree_free(self._h)
        self._h = None

    def __len__(self):
        return lib().svo_octree_len(self._h)

    @property
    def nodes(self):
        return self.raw_data()

    def raw_data(self):
        n = len(self)
        p = lib().svo_octree_raw_data(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy()

    def get_node(self, index):
        return lib().svo_octree_get_node(self._h, index)

    def subdivide(self, node, mask, depth):
        if lib().svo_octree_subdivide(self._h, node, _mask_bytes(mask), depth) != 0:
            raise RuntimeError("Node already subdivided!")  # the reference panics, octree.rs:73-75

    def unsubdivide(self, node):
        rc = lib().svo_octree_unsubdivide(self._h, node)
        if rc < 0:
            raise RuntimeError("Tried to unsubdivide a node without position!")  # octree.rs:104-107
        return rc == 0

    def find_voxel(self, pos, max_depth=None):
        p = (C.c_float * 3)(*pos)
        idx, d, out = C.c_uint64(), C.c_uint32(), (C.c_float * 3)()
        lib().svo_octree_find_voxel(self._h, p, -1 if max_depth is None else max_depth, C.byref(idx), C.byref(d), out)
        return idx.value, d.value, tuple(out)

    def expanded(self, size):
        out = np.empty(size, dtype=np.uint32)
        if lib().svo_octree_expanded(self._h, size, out.ctypes.data) != 0:
            raise ValueError("expanded(size) smaller than the octree")
        return out

    def set_node(self, index, word):
        lib().svo_octree_set_node(self._h, index, int(word) & 0xFFFFFFFF)

    def position(self, index):
        out = (C.c_float * 3)()
        lib().svo_octree_position(self._h, index, out)
        return tuple(out)

    def take_dirty(self):
        """(indices, words) written since the previous call, each index once: the input of Render.scatter_nodes."""
        n = lib().svo_octree_take_dirty(self._h, None, None, 0)
        idx = np.empty(n, dtype=np.uint32)
        val = np.empty(n, dtype=np.uint32)
        if n:
            lib().svo_octree_take_dirty(self._h, idx.ctypes.data, val.ctypes.data, n)
        return idx, val

    def hole_count(self):
        return lib().svo_octree_holes(self._h)

    @staticmethod
    def pos_offset(child_index, depth):
        out = (C.c_float * 3)()
        lib().svo_octree_pos_offset(child_index, depth, out)
        return tuple(out)
