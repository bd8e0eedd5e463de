/*
 * svo_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference's sparse-voxel-octree path
 * (ria8651/octree-tracer): tree construction (src/cpu_octree.rs), the GPU
 * node-word layout (LAYOUT.md, src/octree.rs), ray generation and the
 * octree_ray / find_voxel traversal (src/shader.wgsl), the counter scan
 * (src/compute.wgsl) and the camera maths (src/render.rs, src/main.rs).
 *
 * PARITY STATUS: "parity unpinned" by the reference -- the reference has no
 * tests, no golden vectors, and cannot be built or run here (Rust + WGSL,
 * no cargo/rustc, no wgpu device).  This oracle is pinned only by the
 * hand-derivable known-answer tests listed in SURVEY.md section 8c
 * (tests/test_oracle_kat.py) and by the asset-derived node counts.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (octree-tracer_amd/) never links or calls it.
 */
#ifndef SVO_ORACLE_H
#define SVO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_VOXEL_OFFSET 134217728u  /* src/octree.rs:5, src/shader.wgsl:30 */
#define ORACLE_CHUNK_OFFSET 2147483648u /* src/cpu_octree.rs:3 */

/* uniform flags (explicit u32 bits instead of the reference's 1-byte bools,
 * src/render.rs:287-301, src/shader.wgsl:2-13) */
#define ORACLE_F_PAUSE_ADAPTIVE 1u
#define ORACLE_F_SHOW_STEPS 2u
#define ORACLE_F_SHOW_HITS 4u
#define ORACLE_F_SHADOWS 8u
#define ORACLE_F_MISC_BOOL 16u

typedef struct oracle_uniforms {
    float camera[16];         /* column-major, src/render.rs:205 */
    float camera_inverse[16]; /* column-major, src/render.rs:206 */
    float dimensions[4];      /* (W, H, 0, 0), src/render.rs:204 */
    float sun_dir[4];         /* default (-1.7,-1.0,0.8,0), src/render.rs:312 */
    uint32_t flags;
    float misc_value;
} oracle_uniforms;

/* HitInfo (src/shader.wgsl:182-189) packed into 16 bytes.
 * info: bits 0..7 steps, bits 8..15 depth, bit 16 hit,
 *       bits 17..18 normal.x code, 19..20 normal.y, 21..22 normal.z
 *       (code 0 -> 0, 1 -> +1, 2 -> -1, 3 -> other). */
typedef struct oracle_hit {
    uint32_t value;
    float t;
    uint32_t info;
    uint32_t normal_bits; /* same 6-bit normal code, kept in its own word */
} oracle_hit;

/* ---- tree construction (src/cpu_octree.rs) ---- */
typedef struct oracle_tree oracle_tree;

oracle_tree *oracle_tree_new(uint8_t mask);                 /* CpuOctree::new :23-30 */
void oracle_tree_free(oracle_tree *t);
size_t oracle_tree_len(const oracle_tree *t);
void oracle_tree_put_in_voxel(oracle_tree *t, float x, float y, float z, uint8_t r, uint8_t g,
                              uint8_t b, uint32_t depth);   /* :100-111 */
void oracle_tree_find_voxel(const oracle_tree *t, float x, float y, float z, int64_t max_depth,
                            uint64_t *index, uint32_t *depth, float pos[3]); /* :48-76 */
/* .vox (MagicaVoxel) -> CpuOctree, src/cpu_octree.rs:177-210 over dot_vox 4.1.0 */
oracle_tree *oracle_tree_from_vox(const uint8_t *data, size_t len, char *err, size_t errlen);
/* explicit voxel list in file order (x,y,z,i file-index; palette as 256 LE RGBA u32, entry k
 * colours file index k+1) -- same insertion rules as load_vox */
oracle_tree *oracle_tree_from_voxels(uint32_t size, const uint8_t *xyzi, size_t n_voxels,
                                     const uint32_t *palette256);
/* .rsvo BFS child-mask stream -> CpuOctree, src/cpu_octree.rs:128-175 */
oracle_tree *oracle_tree_from_rsvo(const uint8_t *data, size_t len, uint32_t octree_depth,
                                   char *err, size_t errlen);
/* CpuOctree::to_octree :233-252 -> GPU words; out must hold oracle_tree_len words */
void oracle_tree_to_octree(const oracle_tree *t, uint32_t *out);
/* raw CPU nodes: pointer[i], rgb[3*i..] */
void oracle_tree_raw(const oracle_tree *t, uint32_t *pointers, uint8_t *rgb);
/* World::generate_mip_tree for a single chunk without block references, src/world.rs:234-336 */
void oracle_tree_generate_mips(oracle_tree *t, uint8_t top_mip[3]);

/* parse only: returns voxel count, fills size[3]; xyzi/palette may be NULL */
int64_t oracle_vox_parse(const uint8_t *data, size_t len, uint32_t size[3], uint8_t *xyzi,
                         size_t xyzi_cap, uint32_t *palette256, char *err, size_t errlen);

/* ---- camera maths (src/render.rs:191-206, src/main.rs:139-162, cgmath 0.18) ---- */
void oracle_camera(const float pos[3], const float look[3], float fov_deg, float width,
                   float height, float camera[16], float camera_inverse[16]);

/* ---- traversal (src/shader.wgsl) ---- */
/* Point location, shader.wgsl:130-175 (no counter writes). */
void oracle_find_voxel(const uint32_t *nodes, size_t n_nodes, const float pos[3], int misc_bool,
                       uint32_t *value, float vpos[3], uint32_t *depth);

/* Trace explicit rays (pos.xyz, dir.xyz per ray).  stats (optional): 2 words per ray:
 * w_restart = sum of descent depths, w_reuse = words read with perfect per-ray reuse. */
void oracle_trace_rays(const uint32_t *nodes, size_t n_nodes, uint32_t flags, const float *rays,
                       size_t n_rays, oracle_hit *out, uint32_t *stats, int n_threads);

/* Trace the primary rays of a tile [x0,x0+w) x [y0,y0+h) of a W x H frame
 * (W,H come from u->dimensions), shader.wgsl:250-261.  out/stats are tile-local row-major.
 * counters (optional, n_nodes entries): per-word visit counts of primary-ray descents when
 * pause_adaptive is off (shader.wgsl:157-161), saturated by the caller. */
void oracle_trace_frame(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0,
                        int y0, int w, int h, oracle_hit *out, uint32_t *stats, int n_threads);

/* Full fs_main: colour per pixel (RGBA f32, before surface-format conversion),
 * shader.wgsl:250-304. */
void oracle_shade_frame(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0,
                        int y0, int w, int h, float *rgba, int n_threads);

/* Sequential semantic of the hit-counter side effect (shader.wgsl:157-161): every word visited
 * by a primary-ray descent (including shadow rays, which pass primary=true, :276) gets +1,
 * saturating at 15.  Applies the increments of the given tile to nodes in place. */
/* svo_render_secondary's definition on the CPU: primary records + n_secondary * w * h records, ray-major */
void oracle_secondary_frame(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0, int y0, int w,
                            int h, uint32_t n_secondary, oracle_hit *primary, oracle_hit *secondary, int n_threads);
void oracle_count_frame(uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0, int y0,
                        int w, int h);

/* ---- counter scan (src/compute.wgsl:26-47, src/adaptive.rs:22-23) ---- */
/* sub / unsub hold capacity words each: slot 0 = count, slots 1.. = node indices
 * (ascending order; the reference's order is nondeterministic). */
void oracle_scan(const uint32_t *nodes, size_t n_nodes, uint32_t node_length, uint32_t *sub,
                 uint32_t *unsub, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
