"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's chunk table and streaming-loop list processing.

Pure Python over numpy arrays, for SMALL cases (tests/ only; never imported by the product).  Parity unpinned:
the reference holds no tests or golden vectors for this code and cannot be built here (no Rust toolchain), so
this restates the algorithm from the source text:
    World::find_voxel            src/world.rs:201-232
    World::generate_mip_tree     src/world.rs:234-336
    Octree::{new,subdivide,unsubdivide,find_voxel,pos_offset}   src/octree.rs:51-162
    process_subdivision / process_unsubdivision                 src/adaptive.rs:29-61, 93-121
    CpuOctree::bin / from_bin    src/cpu_octree.rs:262-272 (Node = u32 pointer + 3 colour bytes, 8-byte stride)
All coordinates are dyadic rationals with small exponents, so Python floats reproduce the f32 arithmetic exactly.
"""
import collections
import struct

CHUNK_OFFSET = 2147483648
VOXEL_OFFSET = 134217728


def pos_offset(child, depth):
    d = float(1 << depth)
    return (((child >> 2) & 1) * 2.0 - 1.0) / d, (((child >> 1) & 1) * 2.0 - 1.0) / d, ((child & 1) * 2.0 - 1.0) / d


def to_value(rgb):
    return ((VOXEL_OFFSET + ((int(rgb[0]) << 16) | (int(rgb[1]) << 8) | int(rgb[2]))) << 4) & 0xFFFFFFFF


class Chunk:
    """A CpuOctree: parallel lists of pointers and colours, plus top_mip."""

    def __init__(self, pointers, rgb, top_mip=(50, 255, 50)):
        self.ptr = [int(p) for p in pointers]
        self.rgb = [tuple(int(c) for c in v) for v in rgb]
        self.top_mip = tuple(top_mip)

    def get_node_mask(self, first):
        return [self.rgb[first + i] for i in range(8)]

    def bin(self):
        return b"".join(struct.pack("<IBBBx", p, *c) for p, c in zip(self.ptr, self.rgb))

    @classmethod
    def from_bin(cls, data):
        n = len(data) // 8
        rec = [struct.unpack_from("<IBBBx", data, 8 * i) for i in range(n)]
        return cls([r[0] for r in rec], [r[1:] for r in rec], top_mip=(0, 0, 0))


class World:
    def __init__(self):
        self.chunks = {}

    def find_voxel(self, pos, max_depth=None):
        node_index, chunk, depth = 0, 0, 0
        c = [0.0, 0.0, 0.0]
        while True:
            depth += 1
            child = (pos[0] >= c[0]) * 4 + (pos[1] >= c[1]) * 2 + (pos[2] >= c[2])
            o = pos_offset(child, depth)
            c = [c[0] + o[0], c[1] + o[1], c[2] + o[2]]
            tnipt = self.chunks[chunk].ptr[node_index + child]
            if tnipt == CHUNK_OFFSET or depth == max_depth:
                return chunk, node_index + child, depth, tuple(c)
            if tnipt > CHUNK_OFFSET:
                chunk, node_index = tnipt - CHUNK_OFFSET, 0
            else:
                node_index = tnipt

    def generate_mip_tree(self, cid):
        ch = self.chunks[cid]
        levels = [[0]]
        queue = collections.deque()
        for child in range(8):
            p = ch.ptr[child]
            if p < CHUNK_OFFSET:
                queue.append((child, 1))
            elif p > CHUNK_OFFSET:
                ch.rgb[child] = self.chunks[p - CHUNK_OFFSET].top_mip
        while queue:
            node, depth = queue.popleft()
            while len(levels) <= depth:
                levels.append([])
            levels[depth].append(node)
            first = ch.ptr[node]
            for child in range(8):
                p = ch.ptr[first + child]
                if p < CHUNK_OFFSET:
                    queue.append((first + child, depth + 1))
                elif p > CHUNK_OFFSET:
                    ch.rgb[first + child] = self.chunks[p - CHUNK_OFFSET].top_mip
        for i in reversed(range(len(levels))):
            for node in levels[i]:
                first = ch.ptr[node] if i != 0 else 0
                kids = [ch.rgb[first + k] for k in range(8) if ch.rgb[first + k] != (0, 0, 0)]
                div = float(len(kids))
                out = []
                for axis in range(3):
                    s = float(sum(k[axis] for k in kids))
                    m = s / div if div else float("nan")
                    q = 0 if m != m else max(0, min(255, int(m)))  # `as u8`: saturating, NaN -> 0
                    out.append(max(q, 1))
                if i != 0:
                    ch.rgb[node] = tuple(out)
                else:
                    ch.top_mip = tuple(out)
        return ch.top_mip


class Octree:
    def __init__(self, mask):
        self.nodes = [to_value(v) for v in mask]
        self.positions = [pos_offset(i, 1) for i in range(8)]
        self.hole_stack = []

    def get_node(self, i):
        return self.nodes[i] >> 4

    def subdivide(self, node, mask, depth):
        assert self.get_node(node) >= VOXEL_OFFSET, "Node already subdivided!"
        pos = self.positions[node]
        if self.hole_stack:
            first = self.hole_stack.pop()
        else:
            first = len(self.nodes)
            self.nodes.extend([0] * 8)
            self.positions.extend([None] * 8)
        self.nodes[node] = (first << 4) & 0xFFFFFFFF
        for i in range(8):
            o = pos_offset(i, depth)
            self.nodes[first + i] = to_value(mask[i])
            self.positions[first + i] = (pos[0] + o[0], pos[1] + o[1], pos[2] + o[2])

    def unsubdivide(self, node):
        t = self.get_node(node)
        if t >= VOXEL_OFFSET:
            return
        self.hole_stack.append(t)
        assert self.positions[node] != (0.0, 0.0, 0.0)
        self.nodes[node] = to_value((255, 0, 0))

    def find_voxel(self, pos, max_depth=None):
        node_index, depth = 0, 0
        c = [0.0, 0.0, 0.0]
        while True:
            depth += 1
            child = (pos[0] >= c[0]) * 4 + (pos[1] >= c[1]) * 2 + (pos[2] >= c[2])
            o = pos_offset(child, depth)
            c = [c[0] + o[0], c[1] + o[1], c[2] + o[2]]
            t = self.get_node(node_index + child)
            if t >= VOXEL_OFFSET or depth == max_depth:
                return node_index + child, depth, tuple(c)
            node_index = t


def process_subdivision(node_list, octree, world, load_chunk=None):
    for node_index in node_list:
        if octree.get_node(node_index) < VOXEL_OFFSET:
            continue
        pos = octree.positions[node_index]
        _, voxel_depth, _ = octree.find_voxel(pos)
        cpu_chunk, cpu_index, _, _ = world.find_voxel(pos, voxel_depth)
        p = world.chunks[cpu_chunk].ptr[cpu_index]
        if p < CHUNK_OFFSET:
            octree.subdivide(node_index, world.chunks[cpu_chunk].get_node_mask(p), voxel_depth + 1)
        elif p > CHUNK_OFFSET:
            cid = p - CHUNK_OFFSET
            if cid in world.chunks:
                octree.subdivide(node_index, world.chunks[cid].get_node_mask(0), voxel_depth + 1)
            elif load_chunk is not None:
                load_chunk(cid)


def process_unsubdivision(node_list, octree, world):
    for node_index in node_list:
        octree.unsubdivide(node_index)
        pos = octree.positions[node_index]
        _, voxel_depth, _ = octree.find_voxel(pos)
        cpu_chunk, cpu_index, _, _ = world.find_voxel(pos, voxel_depth)
        ch = world.chunks[cpu_chunk]
        p = ch.ptr[cpu_index]
        if p > CHUNK_OFFSET and p - CHUNK_OFFSET >= CHUNK_OFFSET // 2:
            world.chunks.pop(p - CHUNK_OFFSET, None)
        octree.nodes[node_index] = to_value(ch.rgb[cpu_index])
