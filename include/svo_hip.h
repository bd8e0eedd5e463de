/*
 * svo_hip.h -- the drop-in boundary: a C ABI over the MI355X (gfx950) HIP implementation of
 * ria8651/octree-tracer's GPU path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * What each entry point replaces in the reference (file:line under /root/reference/src):
 *   svo_ctx_create / destroy      Gpu::new (gpu.rs:11-49): instance/adapter/device/queue
 *   svo_nodes_alloc               Render::new node buffer, `octree.expanded(10_000_000)`
 *                                 create_buffer_init STORAGE|COPY_DST (render.rs:53-61)
 *   svo_nodes_write               queue.write_buffer(&render.node_buffer, 0, nodes)
 *                                 (app.rs:113-118, :163-168, :194-199, :239-244)
 *   svo_nodes_scatter             the same upload restricted to the words that changed (no reference counterpart)
 *   svo_set_uniforms              Render::update -> queue.write_buffer(uniform_buffer)
 *                                 (render.rs:191-212); struct Uniforms (render.rs:287-322,
 *                                 shader.wgsl:2-13)
 *   svo_render / svo_render_tiles Render::render: one pass, draw(0..4, 0..1) running fs_main once
 *                                 per pixel (render.rs:217-284, shader.wgsl:250-305)
 *   svo_trace_rays                octree_ray on caller-supplied rays (shader.wgsl:191-248; the
 *                                 shadow ray of shader.wgsl:276 is such a ray)
 *   svo_render_secondary /        no reference entry point: benchmark config 5 (BASELINE.json configs[4]) --
 *   svo_render_tiles_secondary    primary rays plus up to 4 secondary rays per hit pixel; ray 0 is fs_main's
 *                                 shadow ray (shader.wgsl:275-280), the others reuse its origin
 *   svo_scan_dispatch             Compute::update: dispatch(ceil(n/16/256), 256, 1) of
 *                                 compute.wgsl main (compute.rs:99-127, compute.wgsl:26-47)
 *   svo_scan_read                 map_async + device.poll(Wait) + counter reset
 *                                 (adaptive.rs:12-23, :76-87)
 *   svo_sync                      device.poll(Maintain::Wait)
 *   svo_comm_* / svo_gather_frame no reference counterpart (the reference drives one device, main.rs:40-88): the
 *                                 frame-end exchange of the tile-sharded multi-GPU frame, RCCL behind the boundary
 *                                 (SURVEY.md 8b "Threading", 8e): communicator set-up for one process per GPU
 *                                 (svo_comm_init_rank) or one process driving all GPUs (svo_comm_init_all), and the
 *                                 ONE gather of hit records to rank 0 per frame
 * Errors: the reference unwraps/panics (gpu.rs:24,39; adaptive.rs:66,124); here every call
 * returns 0 on success or a negative svo_status, and svo_last_error() gives the text.
 * Threading: like the reference (all device calls from one thread, main.rs:40-88) a ctx is not
 * thread-safe; one ctx per GPU, each ctx launches on one HIP stream.
 */
#ifndef SVO_HIP_H
#define SVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_VOXEL_OFFSET 134217728u /* octree.rs:5, shader.wgsl:30 */

typedef enum svo_status {
    SVO_OK = 0,
    SVO_ERR_ARG = -1,     /* bad argument */
    SVO_ERR_HIP = -2,     /* a HIP runtime call failed; see svo_last_error */
    SVO_ERR_STATE = -3,   /* call order (e.g. render before nodes_alloc) */
    SVO_ERR_NO_DEVICE = -4,
    SVO_ERR_COMM = -5     /* RCCL missing or a collective call failed; see svo_last_error */
} svo_status;

/* Uniform flags: the reference's five 1-byte bools (render.rs:294-299) as explicit bits. */
#define SVO_F_PAUSE_ADAPTIVE 1u
#define SVO_F_SHOW_STEPS 2u
#define SVO_F_SHOW_HITS 4u
#define SVO_F_SHADOWS 8u
#define SVO_F_MISC_BOOL 16u

typedef struct svo_uniforms {
    float camera[16];         /* column-major */
    float camera_inverse[16]; /* column-major */
    float dimensions[4];      /* (W, H, 0, 0) */
    float sun_dir[4];
    uint32_t flags;
    float misc_value;
} svo_uniforms;

/* HitInfo (shader.wgsl:182-189) as a 16-byte record.
 * steps_depth_hit: bits 0..7 steps, 8..15 depth, bit 16 hit, bits 17..22 normal code;
 * packed_normal: 2 bits per axis (x: bits 0..1, y: 2..3, z: 4..5), 0 -> 0, 1 -> +1, 2 -> -1.
 * voxel_index is HitInfo.value: the node-array index of the leaf word, or the reference's
 * sentinels 0 (never entered the cube), 0x20202000 (left the cube), 0xFF000000 (>100 steps).
 * t = ray_box_dist entry distance + t_current of the last DDA step. */
typedef struct svo_hit {
    uint32_t voxel_index;
    float t;
    uint32_t steps_depth_hit;
    uint32_t packed_normal;
} svo_hit;

typedef struct svo_ctx svo_ctx;

/* kernel variants (svo_set_option SVO_OPT_VARIANT) */
#define SVO_VARIANT_RESTART 0 /* reference-shaped: float-compare descent from the root every step */
#define SVO_VARIANT_STACK 1   /* integer path codes + per-ray ancestor stack in LDS + LDS top table + refill (default) */
/* (Round 3's two experiments over a device-built child-mask table -- one ray per lane, and two rays per lane software-pipelined --
 * measured slower than SVO_VARIANT_STACK on every scene and left the library in round 4; the source is kept under
 * tools/experiments/ with its logs in profiles/r03_*, DESIGN.md Appendix A.  Values 2 and 3 are refused.) */

typedef enum svo_option {
    SVO_OPT_VARIANT = 0,
    SVO_OPT_TIMING = 1,      /* n > 0: bracket trace launches with HIP events on the launch stream (ring of n) */
    SVO_OPT_GRID_BLOCKS = 2, /* persistent grid size override (0 = auto) */
    SVO_OPT_REFILL_MIN = 3,  /* idle lanes needed before a wave refills (default 32 since round 5) */
    SVO_OPT_STRIP_ITEMS = 4, /* pixel slots a wave claims at a time (multiple of 64) */
    SVO_OPT_DYNAMIC_STRIPS = 5, /* accepted and ignored: the STACK kernel always claims its strips from device counters (the static
                                round-robin deal of round 1 cost the hot loop scalar registers) */
    SVO_OPT_SCHEDULE = 8,    /* n > 0 (default 2): trace the strips whose rays took the most steps in an earlier frame of
                                the same layout first, rebuilding that schedule every n frames; 0: screen order.
                                Results do not depend on it. */
    SVO_OPT_TREE_DEPTH = 9,  /* upper bound of the octree depth (Settings.octree_depth, app.rs:24); default 16.
                                <= 16: default kernel; <= 22: deep-stack kernel (the STACK variant's 23-bit path codes resolve 22 levels; it was 23 up to
                                round 3); above: the general RESTART kernel.  A tree deeper than the bound given here is refused: the launch
                                completes, svo_sync returns SVO_ERR_STATE (svo_last_error says so) and the frame's records are not to be used
                                (tests/test_parity_gpu.py: test_tree_deeper_than_declared_is_refused) */
    SVO_OPT_BLOCK_SHAPE = 10, /* log2 of the width of the 64-pixel blocks a wave works on (3: 8x8, 4: 16x4, ...) */
    SVO_OPT_DEBUG_BUFFER = 7, /* device pointer to 2 x 16384 x 16 words.  First region, 16 words per wave: start, queue-dry, end (10 ns
                                 ticks), rounds, ..., shader cycles per phase (refill, descent, step), descent-loop shape; second region
                                 (round 5), 16 words per wave: loop entry, the first 14 ray generations (10 ns ticks), XCC / hardware id
                                 (tools/wave_timeline.py, tools/timeline_by_xcd.py) */
    SVO_OPT_SCAN_CLEARS_COUNTERS = 11, /* 1: svo_scan_dispatch also zeroes the hit counters it has scanned, so that the
                                          host need not re-upload the whole array to reset them (svo_nodes_scatter) */
    SVO_OPT_FUSED_SHADOWS = 12, /* shaded frames with shadows (svo_render* with rgba_out), STACK variant: 1 = the lane that finds a hit goes
                                   on with that pixel's shadow ray inside the primary launch; 0 = shadow rays are a second launch;
                                   2 (default) = automatic, which fuses whenever the sun direction is one the fast arithmetic covers
                                   (1080p: 0.65 -> 0.58 ms, 4K: 2.04 -> 1.78 ms on the benchmark tree).  The image is the same either way. */
    SVO_OPT_PAIR_TABLE = 13, /* accepted and ignored: the two-levels-per-load table of round 2 measured a net loss and left the library */
    SVO_OPT_CULL = 14,       /* pixel frames, STACK variant: 64-pixel blocks whose rays all miss the cube (decided conservatively from
                                the block's corner rays) get their all-zero records from a pre-pass and are never claimed by the
                                trace.  0 off, 1 whenever the camera is outside the cube, 2 (default) when in addition at least 40 % of a
                                coarse grid of rays tested on the host look past the cube.  Results do not depend on it. */
    SVO_OPT_CAMERA_SHORTCUT = 15, /* pixel frames, camera inside the cube, STACK variant: every primary ray starts in the camera's leaf, so a wave
                                walks from the root to it once and its lanes copy that walk when they pick up a ray.  1 (default) on, 0 off.
                                Results do not depend on it. */
    SVO_OPT_SCHEDULE_MOTION = 16, /* strip schedule while the camera moves (pixel frames of one rectangle): strips that had at least
                                `min_count` strips with a step-limit ray among their (2 radius + 1)^2 neighbours are scheduled as at least
                                class `floor`.  value = floor (1..12; 0 = off) | radius << 8 | min_count << 12; default 0x1204 (at least class 4
                                for strips with one such strip among their 24 neighbours: 3 - 5 % off a moving camera's frame).  When the
                                camera comes to rest the lists are rebuilt once more from the classes as measured.
                                Results do not depend on it. */
    SVO_OPT_PRIO_STEPS = 6   /* accepted and ignored: raising the issue priority of waves with old rays measured no effect and
                                left the kernel */
} svo_option;

/* Number of HIP devices visible to the process (0 without a GPU). */
int svo_device_count(void);
int svo_ctx_create(int hip_device, svo_ctx **out);
int svo_ctx_destroy(svo_ctx *ctx);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream; NULL is HIP's default
 * stream), or, with use_own != 0, on the private non-blocking stream the ctx was created with. */
int svo_ctx_set_stream(svo_ctx *ctx, void *hip_stream, int use_own);
int svo_set_option(svo_ctx *ctx, int option, int64_t value);
const char *svo_last_error(const svo_ctx *ctx);
int svo_sync(svo_ctx *ctx);

int svo_nodes_alloc(svo_ctx *ctx, size_t capacity_words);
/* Use caller-owned device memory as the node buffer instead (no copy).  The library cannot see writes the caller makes to
 * that memory: call svo_nodes_invalidate after each of them, or the STACK kernel descends from a stale top table. */
int svo_nodes_bind_device(svo_ctx *ctx, uint32_t *device_words, size_t capacity_words);
/* Trace from the SAME node buffer as `owner` (same device; frames in flight on several contexts / streams, one buffer).
 * The contexts share the words, a generation counter and the stream ordering of writes: after svo_nodes_write /
 * svo_nodes_scatter / svo_nodes_invalidate through ANY of them, every one rebuilds its top table and strip schedule
 * before its next trace, and that trace waits (on the device) for the write.  A write does not wait for traces other
 * contexts still have in flight: that ordering stays with the caller (svo_sync them first), as with any shared buffer.
 * The buffer lives until the last context bound to it lets go. */
int svo_nodes_share(svo_ctx *ctx, svo_ctx *owner);
/* The words of the node buffer were changed behind the library's back (a kernel or copy of the caller's, enqueued on
 * this context's stream before this call): bump the generation, like a write. */
int svo_nodes_invalidate(svo_ctx *ctx);
int svo_nodes_write(svo_ctx *ctx, size_t word_offset, const uint32_t *host_words, size_t n);
/* Incremental form of the reference's per-frame `queue.write_buffer(&node_buffer, 0, nodes)` (app.rs:113-118): write
 * host_words[i] to word indices[i] (host pointers, n pairs, indices unique), e.g. the words the streaming loop changed
 * (svo_octree_take_dirty).  Asynchronous on the ctx stream like svo_nodes_write: both arrays must stay valid until the
 * next blocking call. */
int svo_nodes_scatter(svo_ctx *ctx, const uint32_t *indices, const uint32_t *host_words, size_t n);
int svo_nodes_read(svo_ctx *ctx, size_t word_offset, uint32_t *host_words, size_t n);
/* Device pointer of the node buffer (for zero-copy consumers). */
int svo_nodes_device_ptr(svo_ctx *ctx, uint32_t **out, size_t *capacity_words);

/* Device memory for hit records, wire records and gathered frames, for hosts that carry no HIP binding of their own (the
 * `*_out` parameters below are device pointers; any device pointer of the ctx's device will do).  svo_buffer_read copies
 * to host memory behind everything enqueued on the ctx stream and blocks until done. */
int svo_buffer_alloc(svo_ctx *ctx, size_t bytes, void **device_out);
int svo_buffer_free(svo_ctx *ctx, void *device_ptr);
int svo_buffer_read(svo_ctx *ctx, const void *device_ptr, void *host_out, size_t bytes);

int svo_set_uniforms(svo_ctx *ctx, const svo_uniforms *u);

/* Trace the primary rays of the pixel rectangle [x0,x0+tile_w) x [y0,y0+tile_h) of a
 * width x height frame (must equal uniforms.dimensions).  hits_out (tile_w*tile_h records,
 * row-major within the rectangle) and rgba_out (tile_w*tile_h RGBA8, optional, the shaded colour
 * of fs_main) are DEVICE pointers; either may be NULL.  Asynchronous on the ctx stream. */
int svo_render(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
               uint32_t tile_w, uint32_t tile_h, svo_hit *hits_out, uint32_t *rgba_out);
/* Same with HOST output pointers; blocking (staging buffer + D2H). */
int svo_render_host(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                    uint32_t tile_w, uint32_t tile_h, svo_hit *hits_out, uint32_t *rgba_out);
/* Multi-GPU sharding: the frame is cut into tile_w x tile_h tiles numbered row-major; this call
 * traces tiles first_tile, first_tile + tile_stride, ... and writes them contiguously in that
 * order (tile k of this rank at hits_out[k * tile_w * tile_h], row-major inside the tile).
 * width and height must be multiples of the tile size. */
int svo_render_tiles(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t tile_w,
                     uint32_t tile_h, uint32_t first_tile, uint32_t tile_stride,
                     svo_hit *hits_out, uint32_t *rgba_out);
/* Primary rays of the rectangle / tile set as above, then n_secondary (1..4) rays from every pixel whose
 * primary ray hit: origin hit.pos + normal * 2.5e-6 (shader.wgsl:276); ray 0 towards -normalize(sun_dir) (the
 * shadow ray), ray k >= 1 along normalize(e) with e_i = float((h >> 10 i) & 1023) - 511.5,
 * h = mix32((py * width + px) * 4 + k + 0x9E3779B9) (mix32: xorshift-multiply, see oracle/svo_oracle.c),
 * negated if dot(normal, e) < 0.  secondary_out holds n_secondary * n records, ray-major
 * (secondary_out[k * n + i] belongs to primary record i); pixels without a primary hit get the miss record
 * {0, 0, 0, 0}.  Like the shadow ray, secondary rays bump the hit counters unless pause_adaptive.
 * primary_out may be NULL.  Device pointers; asynchronous on the ctx stream. */
int svo_render_secondary(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                         uint32_t tile_w, uint32_t tile_h, uint32_t n_secondary, svo_hit *primary_out,
                         svo_hit *secondary_out);
int svo_render_tiles_secondary(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t tile_w,
                               uint32_t tile_h, uint32_t first_tile, uint32_t tile_stride,
                               uint32_t n_secondary, svo_hit *primary_out, svo_hit *secondary_out);
/* Rank 0 after the frame-end gather: `gathered` holds world x n_pad tiles (rank r's k-th tile = tile r + k * world, the
 * layout svo_render_tiles writes); frame_out receives the row-major width x height frame.  Device pointers. */
int svo_assemble_tiles(svo_ctx *ctx, const svo_hit *gathered, uint32_t world, uint32_t n_pad, uint32_t width,
                       uint32_t height, uint32_t tile_w, uint32_t tile_h, svo_hit *frame_out);
/* The same with 12-byte wire records, and the packing that produces them: the fourth word of svo_hit (packed_normal)
 * repeats bits 17..22 of the third, so a rank can send {voxel_index, t, steps_depth_hit} only -- a quarter fewer bytes
 * on the links -- and rank 0 rebuilds full records while it un-permutes.  Device pointers. */
int svo_pack_records(svo_ctx *ctx, const svo_hit *records, size_t n, uint32_t *wire_out);
int svo_assemble_tiles_packed(svo_ctx *ctx, const uint32_t *gathered_wire, uint32_t world, uint32_t n_pad, uint32_t width,
                              uint32_t height, uint32_t tile_w, uint32_t tile_h, svo_hit *frame_out);
/* ---- multi-GPU frame end: one gather of hit records to the root rank per frame, RCCL over xGMI.  librccl is
 * loaded on first use (dlopen); without it these return SVO_ERR_COMM and everything else keeps working. ----
 * One process per GPU: rank 0 makes an id (svo_comm_unique_id), hands its 128 bytes to the other ranks by any means (the
 * launcher's rendezvous, a file, MPI, torch.distributed), and every rank calls svo_comm_init_rank on its context:
 * collective, blocks until all `world` ranks have called it. */
#define SVO_COMM_ID_BYTES 128
int svo_comm_unique_id(uint8_t id_out[SVO_COMM_ID_BYTES]);
int svo_comm_init_rank(svo_ctx *ctx, const uint8_t id[SVO_COMM_ID_BYTES], int world, int rank);
/* One process (one thread) driving n contexts on n different devices, the reference's threading model (main.rs:40-88):
 * ncclCommInitAll; ctxs[r] becomes rank r. */
int svo_comm_init_all(int n, svo_ctx *const *ctxs);
int svo_comm_destroy(svo_ctx *ctx);
/* This rank's contribution to the frame-end gather: `bytes` bytes (a multiple of 4, the same on every rank) from `send` to
 * `recv_on_root` + rank * bytes on rank `root` (device pointers; recv_on_root is ignored elsewhere and may be NULL).  Every
 * rank of the communicator must call it once per frame, in the same order of frames.  Asynchronous: the exchange runs on a
 * communication stream of the context, ordered BEHIND everything enqueued on the ctx stream so far (the trace / pack that
 * produced `send`), so the ctx stream is free to start the next frame while records travel.  svo_gather_wait makes the ctx
 * stream wait for the gathers issued so far: call it before work on the ctx stream reads recv_on_root (svo_assemble_tiles on
 * the root) or overwrites `send`; svo_sync waits for both streams. */
int svo_gather_frame(svo_ctx *ctx, const void *send, size_t bytes, void *recv_on_root, int root);
int svo_gather_wait(svo_ctx *ctx);
/* The same for the single-process form, all ranks in one call: send[r] is rank r's buffer (on ctxs[r]'s device);
 * recv_on_root lives on ctxs[root]'s device. */
int svo_gather_frame_all(int n, svo_ctx *const *ctxs, const void *const *send, size_t bytes, void *recv_on_root, int root);

/* The same un-permute for a gathered COLOUR frame: when the consumer of the sharded frame is a display -- the reference's
 * output is the RGBA image of fs_main -- the ranks gather rgba_out of svo_render_tiles, 4 bytes per ray on the links instead
 * of 12 (the gather into one GPU is what bounds an 8-GPU frame).  Device pointers. */
int svo_assemble_tiles_rgba(svo_ctx *ctx, const uint32_t *gathered_rgba, uint32_t world, uint32_t n_pad, uint32_t width,
                            uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t *rgba_frame_out);

/* octree_ray over n explicit rays (6 floats each: pos.xyz, dir.xyz; device pointers). */
int svo_trace_rays(svo_ctx *ctx, const float *rays, size_t n_rays, svo_hit *hits_out);

/* Duration of the most recent trace launch in ms (needs SVO_OPT_TIMING > 0); blocks on it. */
int svo_last_render_ms(svo_ctx *ctx, float *ms);
/* Durations (ms, oldest first) of the trace launches recorded since the last collect, at most the
 * ring size; blocks until they have finished and resets the record. */
int svo_timing_collect(svo_ctx *ctx, float *ms_out, size_t cap, size_t *n_out);

/* Diagnostic for profiling: n_loads single-dword buffer loads over the node buffer, lane i at byte i * stride_bytes
 * (the trace kernels' access shape), to calibrate hardware byte counters on a known line count. */
int svo_diag_gather(svo_ctx *ctx, uint32_t stride_bytes, uint32_t n_loads);

/* Diagnostic for the tests of the culling pass (SVO_OPT_CULL): the class byte of each of the first n_strips 64-pixel blocks of
 * the last pixel frame that ran the pass (block b of a width x height rectangle, 8x8 blocks row-major; 0xFF = culled: the pass
 * wrote the block's all-zero records and the trace never claimed it).  Blocking; SVO_ERR_STATE when that frame ran without it. */
int svo_diag_strip_classes(svo_ctx *ctx, uint8_t *host_out, size_t n_strips);

/* Counter scan (compute.wgsl).  Lists hold `capacity` words: slot 0 = count, slots 1.. = indices. */
int svo_scan_dispatch(svo_ctx *ctx, uint32_t node_length);
int svo_scan_read(svo_ctx *ctx, uint32_t *sub, uint32_t *n_sub, uint32_t *unsub, uint32_t *n_unsub,
                  size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
