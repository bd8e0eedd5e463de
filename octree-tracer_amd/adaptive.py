"""View-dependent LOD streaming on top of the device scan: reference src/adaptive.rs
(process_subdivision :6-68, process_unsubdivision :70-126) and the frame update of src/app.rs:94-118,
restricted to a world made of one chunk without block references (the reference's block/chunk
indirection, world.rs:201-232, is asset management outside the GPU path)."""
from .cpu_octree import CHUNK_OFFSET
from .octree import VOXEL_OFFSET, Octree


class World:
    """world.rs:5-9 with a single chunk 0 (a CpuOctree whose mip colours were generated)."""

    def __init__(self, chunk0):
        self.chunk = chunk0
        self.top_mip = chunk0.generate_mip_tree()
        self._ptrs, self._rgb = chunk0.raw()

    def find_voxel(self, pos, max_depth=None):
        """world.rs:201-232 -> (chunk, index, depth, pos); no chunk indirection in a one-chunk world."""
        idx, depth, p = self.chunk.find_voxel(pos, max_depth)
        return 0, idx, depth, p

    def node(self, index):
        return int(self._ptrs[index]), tuple(int(c) for c in self._rgb[index])

    def get_node_mask(self, first_child):
        return self.chunk.get_node_mask(first_child)

    def root_octree(self):
        """App::new (app.rs:47-48): the device tree starts as the root's 8 mip-coloured children."""
        return Octree.new(self.get_node_mask(0))


def process_subdivision(compute_lists, octree, world):
    """adaptive.rs:29-61 over the subdivide list: hot leaves get their 8 children from the CPU world."""
    done = 0
    for node_index in compute_lists:
        node_index = int(node_index)
        if octree.get_node(node_index) < VOXEL_OFFSET:  # "Doubleup!" :32-35
            continue
        pos = octree.position(node_index)
        _, voxel_depth, _ = octree.find_voxel(pos)
        _, cpu_index, _, _ = world.find_voxel(pos, voxel_depth)
        pointer, _ = world.node(cpu_index)
        if pointer < CHUNK_OFFSET:                       # :42-48
            octree.subdivide(node_index, world.get_node_mask(pointer), voxel_depth + 1)
            done += 1
    return done


def process_unsubdivision(compute_lists, octree, world):
    """adaptive.rs:93-121 over the unsubdivide list: cold interior nodes collapse to their mip colour."""
    done = 0
    for node_index in compute_lists:
        node_index = int(node_index)
        pos = octree.position(node_index)
        if pos == (0.0, 0.0, 0.0):
            continue  # the reference panics here (octree.rs:104-107); the root group carries real positions
        if not octree.unsubdivide(node_index):           # :95
            continue
        _, voxel_depth, _ = octree.find_voxel(pos)
        _, cpu_index, _, _ = world.find_voxel(pos, voxel_depth)
        _, rgb = world.node(cpu_index)
        value = (VOXEL_OFFSET + ((rgb[0] << 16) | (rgb[1] << 8) | rgb[2])) << 4  # tnipt.value.to_value() :117
        octree.set_node(node_index, value)
        done += 1
    return done


class AdaptiveLoop:
    """App::update (app.rs:94-118): uniforms -> trace (counters live) -> scan -> CPU (un)subdivide -> re-upload."""

    def __init__(self, gpu, render, compute, octree, world):
        self.gpu, self.render, self.compute, self.octree, self.world = gpu, render, compute, octree, world

    def frame(self, settings, character, deterministic=False):
        self.render.update(settings, character)
        hits = self.render.render()
        if self.render.uniforms.flags & 1:  # pause_adaptive (app.rs:97)
            return hits, 0, 0
        self.compute.update(len(self.octree))
        sub, unsub = self.compute.read_lists()
        if deterministic:  # the device appends in no particular order; sorted lists make runs repeatable
            sub.sort()
            unsub.sort()
        n_sub = process_subdivision(sub, self.octree, self.world)
        n_unsub = process_unsubdivision(unsub, self.octree, self.world)
        # app.rs:113-118: the whole array goes back (host words carry counter 0, which also clears the counters)
        self.render.write_nodes(self.octree.raw_data())
        return hits, n_sub, n_unsub
