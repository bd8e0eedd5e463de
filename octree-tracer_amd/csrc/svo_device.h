// svo_device.h -- internal launch interface between the C ABI (svo_abi.cpp) and the gfx950
// kernels (svo_kernels.hip).  Not part of the public boundary (include/svo_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svo_hip.h"

namespace svo {

constexpr uint32_t kVoxelOffset = 134217728u;  // octree.rs:5
constexpr uint32_t kMaxDescent = 31;           // descent guard (see oracle/svo_oracle.c)
constexpr int kTopLevels = 3;                  // K: octree levels folded into the LDS top table
constexpr int kTopEntries = 1 << (3 * kTopLevels);
constexpr int kTopAuxEntries = 8 + 64;  // level-1 and level-2 cells -> child group of the next level (kTopLevels = 3)
constexpr int kPathBits = 23;                  // D: path-code bits per axis (grid units: 2^(D-1) per unit of the reference's cube, so that
                                               // 2^D + code is an exact f32 whose mantissa IS the code: DESIGN.md 4.2)
// Words between two claim counters.  Rounds 2 - 4 kept them 128 B apart (all 64 within 8 KB); round 5: 64 KB + 128 B apart.  Every claim is
// an atomic executed at the memory side, and while one is in flight the L2 channel it went through answers everything else more slowly
// (profiles/r05_pmc_claim_timing_ab.json) -- 8 KB of counters sit behind one or two channels, 4 MB of them behind all
// (profiles/r05_counter_stride_ab*.log: 1080p -1.1 / -3.4 %, 4K -1.6 / -0.3 % in two runs; 4 KB + 128 B: -0.5 / -2.5 %, -0.7 / -1.2 %)
#ifndef SVO_COUNTER_STRIDE
#define SVO_COUNTER_STRIDE 16416
#endif
constexpr int kCounterStride = SVO_COUNTER_STRIDE;
constexpr int kCounterWords = 64 * kCounterStride;  // 64 claim counters (8 lists x 8 counters)

// Top-table entry (one per level-K cell, index = cx << 2K | cy << K | cz): level << 27 | child group index.
// The level is K+1 below an interior cell, or the shallower level at which a leaf covers the whole cell
// (the descent then reads that leaf word itself).

// Work decomposition: a list of equally sized pixel rectangles, each cut into 64-pixel blocks (8x8 by default).
// Item q (one pixel slot): rect = q / (64 * blocks_per_rect); block and lane follow.
struct WorkDesc {
    uint32_t mode;        // 0: one rectangle at (x0, y0); 1: tiles first_tile + k * tile_stride; 2: explicit rays
    uint32_t x0, y0;      // mode 0 origin
    uint32_t w, h;        // rectangle size in pixels
    uint32_t bw_log2;     // pixel blocks are 2^bw_log2 wide and 64 / that high (8x8: 3)
    uint32_t bpr;         // blocks per rectangle row
    uint32_t bprect;      // blocks per rectangle
    uint32_t n_rects;
    uint32_t tiles_x;     // mode 1: tiles per frame row
    uint32_t first_tile, tile_stride;
    uint32_t n_items;     // mode 0/1: n_rects * bprect * 64; mode 2: n_rays
    uint32_t magic_bpr, magic_bprect, magic_tiles_x;  // floor(2^32 / d) + 1 (0 when d == 1), see fast_div
};

struct TraceArgs {
    const uint32_t *nodes;
    uint32_t n_words;
    const uint32_t *top_table;  // kTopEntries words (device) or nullptr
    svo_uniforms u;
    WorkDesc work;
    const float *rays;          // mode 2
    svo_hit *hits;              // may be nullptr
    uint32_t *rgba;             // may be nullptr
    float *aux_t;               // optional: t_current of the last step per record (the shading pass rebuilds HitInfo.pos)
    uint32_t *status;           // device word: bit 0 set when a STACK-variant descent exceeded kPathBits
    uint32_t refill_min;
    uint32_t cam_shortcut;      // STACK: 1 = rays that start at the camera's own position inside the cube share the wave's one first walk
    uint32_t *count_nodes;      // writable alias of nodes when hit counters are live (pause_adaptive off), else nullptr
    const uint32_t *order;      // STACK, optional: schedule built by strip_order_kernel (8 lengths + 8 lists)
    uint32_t order_cap;         // entries reserved per list
    uint32_t *debug;            // optional: 16 words per wave (start, queue-dry, end ticks of 10 ns, rounds, active-lane sum, ..., phase cycles)
    const uint8_t *skip;        // mode 2, optional: one byte per ray; non-zero = no ray here, its (all-zero) record is already written
    uint32_t *balance;          // STACK, optional: kBalanceWords words of the schedule's list-share feedback (see launch_post); the trace kernel
                                // records when its first wave started ([9], min) and when a sample of every list's workgroups ended, 10 ns ticks
    svo_hit *shadow_hits;       // STACK, optional (fused shadow rays): the lane that finishes a primary ray with a hit goes on with that
                                // pixel's shadow ray (shader.wgsl:275-280) and writes its record here; pixels without one get zeros
};

struct LaunchInfo {
    int variant;
    int grid_blocks;         // 0 = auto
    int num_cus;
    int *occupancy;          // STACK: the context's cache of resident workgroups per CU, one slot per instantiation
    bool deep_stack;         // STACK: 19-level ancestor stack (trees deeper than 16 levels)
    uint32_t strip_items;    // STACK: pixel slots a wave claims at a time (multiple of 64)
    uint32_t *counters;      // STACK: kCounterWords claim-counter words, zero when a frame starts
    uint32_t *work_counter;  // STACK: counters + 0 for dynamic strip claiming, or nullptr (static round-robin)
    uint32_t *defer;         // STACK: this frame's deferred list: count, then one slot per item handed to RESTART
    uint32_t *next_defer_count;  // STACK: count word of the other list (lists alternate between frames)
};

hipError_t launch_build_top_table(const uint32_t *nodes, uint32_t n_words, uint32_t *top_table,
                                  hipStream_t stream);
hipError_t launch_trace(const TraceArgs &args, const LaunchInfo &li, hipStream_t stream);
int stack_max_depth(bool deep);
// after a STACK trace: deferred rays, per-strip costs (cost != nullptr) and the next schedule; re-arms the counters
constexpr uint32_t kMaxScheduledStrips = 1u << 20;  // (2^26 items / 64)
// cost class of a strip = (largest step count among its rays) >> kCostShift: 4 steps per class -- with 8 the cheapest two classes of the
// benchmark view hold 3 439 of 32 400 strips, fewer than the grid has waves, so every wave's last strip came from the 16..23-step
// class in screen order; the finer classes end the lists with the strips that really are the shortest
#ifndef SVO_COST_SHIFT
#define SVO_COST_SHIFT 2
#endif
constexpr uint32_t kCostShift = SVO_COST_SHIFT, kCostClasses = 128u >> SVO_COST_SHIFT;
constexpr uint32_t kOrderHistWords = 64 * kCostClasses;  // chunk histograms of the schedule builder, stored behind the class bytes
// list-share feedback of a schedule: [0..8] cumulative shares of the 8 lists (16-bit fractions, [0] = 0, [8] = 65536), [9] the start
// stamp of the last frame's first workgroup, [18] updates so far, [19] the last trace launch's workgroups; behind them (kBalanceHead) one end stamp per workgroup of the trace
// launch (workgroup b started on list b % 8)
constexpr uint32_t kBalanceHead = 32, kBalanceSlots = 4096, kBalanceWords = kBalanceHead + kBalanceSlots;
// entries reserved per list of a schedule: a list holds its share of every cost class -- an eighth, or what the feedback gives it
// (at most twice that)
inline uint32_t order_list_cap(const WorkDesc &, uint32_t n_strips) { return (n_strips + 3u) / 4u + kCostClasses; }
hipError_t launch_post(const TraceArgs &args, const LaunchInfo &li, uint8_t *cost, uint32_t *sched, uint32_t n_strips,
                       uint32_t cap, bool build_schedule, hipStream_t stream, uint8_t *moved = nullptr, uint32_t motion_floor = 0,
                       uint32_t balance_update = 0, bool reuse_cost = false);

// explicit rays with a skip mask: this frame's strip lists without the strips that hold no ray (classes from `prev`, or screen order)
hipError_t launch_schedule_skipping(const uint8_t *skip, uint32_t n_items, const uint8_t *prev, uint8_t *cls, uint32_t *sched,
                                    uint32_t n_strips, uint32_t cap, hipStream_t stream);
// pixel frames seen from outside the cube: this frame's strip lists without the strips whose rays all miss it (their all-zero
// records are written here); args carries the work layout, the uniforms and the output buffers of the trace that follows
hipError_t launch_schedule_culling(const TraceArgs &args, const uint8_t *prev, uint8_t *cls, uint32_t *sched, uint32_t n_strips,
                                   uint32_t cap, hipStream_t stream);
// secondary rays of the hit pixels of args.hits; pixels without a hit get skip = 1 and a zero record in `out` instead of a ray
// (rays k_first .. n_secondary - 1; slot of ray k of record r: (k - k_first) * n_records + r)
hipError_t launch_secondary_gen(const TraceArgs &args, const float *aux_t, float *rays, uint8_t *skip, svo_hit *out, uint32_t k_first,
                                uint32_t n_secondary, uint32_t n_records, hipStream_t stream);
hipError_t launch_shade(const TraceArgs &args, const svo_hit *shadow_hits, uint32_t *rgba, hipStream_t stream);
hipError_t launch_diag_gather(const uint32_t *buf, uint32_t n_words, uint32_t stride_words, uint32_t n_loads, uint32_t *sink,
                              hipStream_t stream);
hipError_t launch_scan(uint32_t *nodes, uint32_t n_words, uint32_t node_length, uint32_t *sub, uint32_t *unsub,
                       uint32_t capacity, bool clear_counters, hipStream_t stream);
hipError_t launch_assemble_tiles(const void *gathered, bool packed, svo_hit *frame, uint32_t world, uint32_t n_pad, uint32_t width,
                                 uint32_t height, uint32_t tile_w, uint32_t tile_h, hipStream_t stream);
hipError_t launch_assemble_tiles_rgba(const uint32_t *gathered, uint32_t *frame, uint32_t world, uint32_t n_pad, uint32_t width,
                                      uint32_t height, uint32_t tile_w, uint32_t tile_h, hipStream_t stream);
hipError_t launch_pack_records(const svo_hit *records, uint32_t *wire, uint32_t n, hipStream_t stream);
hipError_t launch_scatter(uint32_t *nodes, uint32_t n_words, const uint32_t *indices, const uint32_t *words, uint32_t n,
                          hipStream_t stream);

}  // namespace svo
