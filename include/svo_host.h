/*
 * svo_host.h -- host-side data model of the drop-in: the octree containers the reference keeps
 * on the CPU and feeds to the node buffer, plus camera maths and deterministic scene generators.
 * C ABI (the C++ classes behind it live in octree-tracer_amd/csrc/svo_host.cpp).
 *
 * Reference counterparts (file:line under /root/reference/src):
 *   svo_cpu_octree_*      CpuOctree (cpu_octree.rs:17-273): new/add_voxels :23-45, find_voxel
 *                         :48-76, put_in_voxel :100-111, load_file :113-125, .rsvo :128-175,
 *                         .vox :177-210, to_octree :233-252; World::generate_mip_tree
 *                         (world.rs:234-336)
 *   svo_octree_*          Octree (octree.rs:43-162): new :51-66, subdivide :72-93, unsubdivide
 *                         :95-110, find_voxel :113-141, expanded :143-148, pos_offset :154-161
 *   svo_world_*           World (world.rs:5-9): chunk table with block references; find_voxel :201-232,
 *                         generate_mip_tree :234-336, save_chunk / load_chunk / load_world :159-198;
 *                         svo_cpu_octree_bin / _from_bin: CpuOctree::bin / from_bin (cpu_octree.rs:262-272)
 *   svo_adaptive_*        process_subdivision / process_unsubdivision (adaptive.rs:6-126), the list
 *                         processing after the read-back that svo_scan_read performs
 *   svo_camera_matrices   Render::update (render.rs:191-206) + create_proj_matrix (main.rs:139-162)
 *   svo_gen_*             no counterpart: the reference's generator (procedual.wgsl) is racy and
 *                         nondeterministic; these are deterministic CPU scene builders for the
 *                         benchmark configs.
 */
#ifndef SVO_HOST_H
#define SVO_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_CHUNK_OFFSET 2147483648u /* cpu_octree.rs:3 */

typedef struct svo_cpu_octree svo_cpu_octree;
typedef struct svo_octree svo_octree;

/* ---- CpuOctree ---- */
svo_cpu_octree *svo_cpu_octree_new(uint8_t mask);
void svo_cpu_octree_free(svo_cpu_octree *t);
size_t svo_cpu_octree_len(const svo_cpu_octree *t);
/* load_file: dispatch on the extension ("rsvo" | "vox"), else "Unknown file type". NULL on error. */
svo_cpu_octree *svo_cpu_octree_load_file(const char *path, uint32_t octree_depth, char *err, size_t errlen);
svo_cpu_octree *svo_cpu_octree_load_vox(const uint8_t *data, size_t len, char *err, size_t errlen);
svo_cpu_octree *svo_cpu_octree_load_rsvo(const uint8_t *data, size_t len, uint32_t octree_depth, char *err,
                                         size_t errlen);
/* voxel list in file order (x, y, z, file colour index), palette = 256 LE RGBA words */
svo_cpu_octree *svo_cpu_octree_from_voxels(uint32_t size, const uint8_t *xyzi, size_t n_voxels,
                                           const uint32_t *palette256, char *err, size_t errlen);
void svo_cpu_octree_put_in_voxel(svo_cpu_octree *t, const float pos[3], const uint8_t rgb[3], uint32_t depth);
void svo_cpu_octree_put_in_block(svo_cpu_octree *t, const float pos[3], uint32_t block_id, uint32_t depth);
/* max_depth < 0: None */
void svo_cpu_octree_find_voxel(const svo_cpu_octree *t, const float pos[3], int64_t max_depth, uint64_t *index,
                               uint32_t *depth, float node_pos[3]);
void svo_cpu_octree_get_node_mask(const svo_cpu_octree *t, size_t node, uint8_t rgb_out[24]);
/* GPU words (to_octree); out holds svo_cpu_octree_len words */
void svo_cpu_octree_to_octree(const svo_cpu_octree *t, uint32_t *out);
void svo_cpu_octree_raw(const svo_cpu_octree *t, uint32_t *pointers, uint8_t *rgb);
void svo_cpu_octree_generate_mips(svo_cpu_octree *t, uint8_t top_mip[3]);
/* parse a .vox: returns voxel count (or -1), size[3]; xyzi/palette optional */
int64_t svo_vox_parse(const uint8_t *data, size_t len, uint32_t size[3], uint8_t *xyzi, size_t xyzi_cap,
                      uint32_t *palette256, char *err, size_t errlen);
/* write a minimal .vox (SIZE, XYZI, RGBA); returns bytes needed/written */
size_t svo_vox_write(uint32_t size, const uint8_t *xyzi, size_t n_voxels, const uint32_t *palette256, uint8_t *out,
                     size_t cap);
/* serialise a CpuOctree as an .rsvo child-mask stream (inverse of load_rsvo); returns bytes */
size_t svo_rsvo_write(const svo_cpu_octree *t, uint8_t *out, size_t cap);

/* ---- Octree (host mirror of the device node array) ---- */
svo_octree *svo_octree_new(const uint8_t mask_rgb[24]);
svo_octree *svo_octree_from_words(const uint32_t *words, size_t n);
void svo_octree_free(svo_octree *o);
size_t svo_octree_len(const svo_octree *o);
const uint32_t *svo_octree_raw_data(const svo_octree *o);
uint32_t svo_octree_get_node(const svo_octree *o, size_t index);
/* 0 ok; -1 "Node already subdivided!" (the reference panics) */
int svo_octree_subdivide(svo_octree *o, size_t node, const uint8_t mask_rgb[24], uint32_t depth);
/* 0 ok; 1 node was not subdivided (reference prints and returns); -1 no position (reference panics) */
int svo_octree_unsubdivide(svo_octree *o, size_t node);
void svo_octree_find_voxel(const svo_octree *o, const float pos[3], int64_t max_depth, uint64_t *index,
                           uint32_t *depth, float node_pos[3]);
/* expanded(size): zero-padded copy; out holds size words; returns -1 if size < len */
int svo_octree_expanded(const svo_octree *o, size_t size, uint32_t *out);
void svo_octree_pos_offset(uint32_t child_index, uint32_t depth, float out[3]);
size_t svo_octree_holes(const svo_octree *o);
void svo_octree_set_node(svo_octree *o, size_t index, uint32_t word);   /* octree.nodes[i] = word (adaptive.rs:117) */
void svo_octree_position(const svo_octree *o, size_t index, float out[3]); /* octree.positions[i] */
/* Words written (subdivide, unsubdivide, set_node, svo_adaptive_*) since the previous call, each index once, with
 * their current values: the input of svo_nodes_scatter.  Returns the count; with NULL outputs or cap too small nothing
 * is consumed. */
size_t svo_octree_take_dirty(svo_octree *o, uint32_t *indices, uint32_t *words, size_t cap);

/* ---- World: chunks, block instancing, chunk dumps ---- */
typedef struct svo_world svo_world;
/* World::new(path) WITHOUT the eight block .vox loads of world.rs:19-58: the caller inserts block chunks
 * 1..8 (svo_cpu_octree_load_file + svo_world_insert + svo_world_generate_mip_tree), see INTEGRATION.md. */
svo_world *svo_world_new(const char *path);
void svo_world_free(svo_world *w);
const char *svo_world_last_error(const svo_world *w);
/* chunks.insert(id, chunk): the world takes ownership of the chunk (an existing chunk of that id is freed) */
int svo_world_insert(svo_world *w, uint32_t id, svo_cpu_octree *chunk);
int svo_world_remove(svo_world *w, uint32_t id);                      /* 0 removed, 1 not present */
svo_cpu_octree *svo_world_chunk(const svo_world *w, uint32_t id);     /* borrowed; NULL if absent */
size_t svo_world_chunk_ids(const svo_world *w, uint32_t *ids, size_t cap);
/* (chunk, index, depth, pos); -1 where the reference would panic on a chunk that is not loaded */
int svo_world_find_voxel(const svo_world *w, const float pos[3], int64_t max_depth, uint32_t *chunk,
                         uint64_t *index, uint32_t *depth, float node_pos[3]);
/* block leaves take the referenced chunk's top_mip, interior nodes the mean of their children */
int svo_world_generate_mip_tree(svo_world *w, uint32_t id, uint8_t top_mip[3]);
/* <path>/<id>.bin: 8 bytes per node (LE u32 pointer, r, g, b, one pad byte) */
int svo_world_save_chunk(svo_world *w, uint32_t id);
int svo_world_load_chunk(svo_world *w, uint32_t id);                  /* synchronous (reference: tokio task) */
svo_world *svo_world_load(const char *path, char *err, size_t errlen); /* load_world: reads <path>/0.bin */
size_t svo_cpu_octree_bin(const svo_cpu_octree *t, uint8_t *out, size_t cap);
svo_cpu_octree *svo_cpu_octree_from_bin(const uint8_t *data, size_t len, char *err, size_t errlen);

/* ---- streaming loop, CPU half (adaptive.rs) ---- */
/* lists as svo_scan_read returns them (node indices). Returns nodes (un)subdivided, -1 on error
 * (svo_world_last_error).  A listed leaf whose block/chunk is not loaded triggers a load and is skipped
 * this time, as in the reference (*chunks_loaded counts them). */
int64_t svo_adaptive_subdivide(svo_world *w, svo_octree *o, const uint32_t *list, size_t n, uint64_t *chunks_loaded);
int64_t svo_adaptive_unsubdivide(svo_world *w, svo_octree *o, const uint32_t *list, size_t n);
/* Fixed point of the subdivision loop without a device: every leaf with world children is subdivided, pass by
 * pass in index order, while depth < max_depth and (lod_c <= 0 or 2^depth * distance(cam, cube) < lod_c) and
 * the array stays <= max_words.  Returns the number of subdivisions. */
uint64_t svo_world_expand(svo_world *w, svo_octree *o, uint32_t max_depth, const float cam[3], float lod_c,
                          uint64_t max_words);

/* ---- camera ---- */
void svo_camera_matrices(const float pos[3], const float look[3], float fov_deg, float width, float height,
                         float camera[16], float camera_inverse[16]);

/* ---- deterministic scene generators (benchmark inputs) ---- */
typedef struct svo_terrain_params {
    uint32_t seed;
    uint32_t max_depth;      /* finest level (<= 20) */
    float cam[3];            /* LOD centre, world coordinates */
    float lod_c;             /* a mixed node at distance r is refined while 2^level < lod_c / r */
    uint32_t min_depth;      /* always refine to at least this level */
    uint64_t max_words;      /* hard cap on the node array (<= 2^27) */
} svo_terrain_params;
/* Returns the number of words written (BFS order, root group at 0); out holds cap words. */
uint64_t svo_gen_terrain(const svo_terrain_params *p, uint32_t *out, uint64_t cap);
/* terrain height (in finest-voxel units, 0..2^max_depth) of column (X, Z) at the finest level */
int32_t svo_gen_terrain_height(uint32_t seed, uint32_t max_depth, uint32_t X, uint32_t Z);
/* Sierpinski-tetrahedron octree (children of even coordinate parity survive), refined by the same
 * distance rule, so a depth-20 tree stays under the 2^27-word layout cap.  Words returned. */
uint64_t svo_gen_fractal(const svo_terrain_params *p, uint32_t *out, uint64_t cap);
/* random sparse tree for tests: each child is interior with probability p_split (until max_depth),
 * else solid with probability p_solid */
uint64_t svo_gen_random(uint32_t seed, uint32_t max_depth, float p_split, float p_solid, uint64_t max_words,
                        uint32_t *out, uint64_t cap);
/* maximum leaf depth of a node array (BFS over the array; 0 for malformed) */
uint32_t svo_nodes_max_depth(const uint32_t *words, uint64_t n);
/* Re-linearisation (round 5): the same tree with its child groups reordered -- levels 1 .. block_level breadth-first, then every subtree
 * below a level-block_level interior word as one contiguous block.  out: >= n words; perm (optional, >= n words): perm[new word] = old
 * word.  Returns the words written (unreachable words are dropped), 0 for a malformed tree.  The reference has no counterpart: its
 * loaders emit insertion order (.vox, cpu_octree.rs:100-111) or breadth-first order (.rsvo, :160-172). */
uint64_t svo_nodes_relayout(const uint32_t *words, uint64_t n, uint32_t block_level, uint32_t *out, uint32_t *perm);

#ifdef __cplusplus
}
#endif
#endif
