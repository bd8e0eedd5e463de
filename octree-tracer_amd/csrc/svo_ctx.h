// svo_ctx.h -- the device context behind the opaque `svo_ctx` of include/svo_hip.h; shared by svo_abi.cpp (trace / scan
// dispatch) and svo_comm.cpp (RCCL frame gather).  Internal: not part of the boundary.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "svo_device.h"
#include "svo_hip.h"

// The device node buffer (render.rs:53-61) and what is derived from it.  Several contexts can trace from one store
// (svo_nodes_share: frames in flight on several streams): they share the words, the generation counter that tells every
// one of them when its own top table and strip schedule are stale, and the event that orders their reads behind the
// last write, whichever context issued it.
struct svo_node_store {
    int device = 0;
    uint32_t *nodes = nullptr;
    size_t capacity = 0;
    bool owned = false;          // allocated by svo_nodes_alloc (freed with the last reference)
    int refs = 1;
    uint64_t version = 1;        // bumped whenever the words may have changed
    hipEvent_t last_write = nullptr;   // recorded on the writing context's stream after every write
    hipStream_t last_writer = nullptr; // that stream: other streams wait for the event before they read
};

struct svo_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // node buffer (render.rs:53-61): `nodes` / `capacity` mirror the store this context is bound to
    svo_node_store *store = nullptr;
    uint32_t *nodes = nullptr;
    size_t capacity = 0;
    uint64_t top_version = 0;    // store version this context's top table was built from (0: none)
    uint32_t *top_table = nullptr;
    int cull_mode = 2;           // SVO_OPT_CULL
    bool cam_shortcut = true;    // SVO_OPT_CAMERA_SHORTCUT
    void *comm = nullptr;        // ncclComm_t (svo_comm.cpp); world size and rank of this context in it
    int comm_world = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;  // the gathers run here, ordered against `stream` by the two events
    hipEvent_t comm_ready = nullptr, comm_done = nullptr;
    bool gathers_issued = false;
    uint32_t *status = nullptr;        // device error word
    uint32_t *defer_buf = nullptr;     // {strip counter, deferred count, deferred item indices...}
    size_t defer_items = 0;
    // scan lists (compute.rs:46-64): slot 0 = count
    uint32_t *scan_sub = nullptr, *scan_unsub = nullptr;
    size_t scan_capacity = 0;
    // host staging for svo_render_host
    void *stage = nullptr;
    size_t stage_bytes = 0;
    svo_uniforms uniforms{};
    bool have_uniforms = false;
    // options
    int variant = SVO_VARIANT_STACK;
    int grid_blocks = 0;
    int occupancy[16] = {};  // resident workgroups per CU of each STACK instantiation on this device (0: not asked yet)
    uint32_t refill_min = 32;  // (round 5: 16 until the schedule's locality work; profiles/r05_refill_sweep.log)
    bool scan_clears = false;
    int fused_shadows = 2;  // 0: off, 1: on, 2: by frame size and tree depth (see trace_common)
    void *scatter_buf = nullptr;
    size_t scatter_bytes = 0;
    uint32_t prio_steps = 0;
    uint32_t block_w_log2 = 3;  // 64-pixel blocks of 8x8
    uint32_t tree_depth = 16;  // caller's bound on the octree depth (the reference's Settings.octree_depth)
    // scheduling feedback (strip order from an earlier frame of the same work layout); slot 1: shadow rays
    struct Sched {
        uint8_t *cost = nullptr;
        uint8_t *cls_now = nullptr;  // launches with a skip mask: this frame's classes (0xFF = strip without a ray)
        uint32_t *order = nullptr;
        uint32_t *balance = nullptr;  // kBalanceWords: the lists' shares and the stamps they follow (svo_kernels.hip: balance_step)
        uint32_t balance_frames = 0;  // scheduled frames of this layout whose stamps have been fed back
        size_t cap = 0;
        bool valid = false;
        bool order_filtered = false;  // `order` was built for one frame without its empty / culled strips: not a general schedule
        uint32_t age = 0;
        svo::WorkDesc key{};
        // what the schedule was measured on: while camera and tree stay the same it stays exact and is not rebuilt
        svo_uniforms built_uniforms{};
        uint64_t built_nodes_version = 0;
        // camera motion (SVO_OPT_SCHEDULE_MOTION): the uniforms of the previous frame, and whether the lists in `order` were
        // built with the floor for strips near long ones (then they are rebuilt once more, exactly, when the camera rests)
        svo_uniforms prev_uniforms{};
        bool have_prev = false;
        bool floored = false;
    };
    Sched sched[2];
    bool list_balance = getenv("SVO_NO_LIST_BALANCE") == nullptr;  // (A/B switch of the list-share feedback, svo_kernels.hip: balance_step)
    uint32_t motion_floor = 0x1204;  // SVO_OPT_SCHEDULE_MOTION: class floor | radius << 8 | min_count << 12; 0 = off
    bool schedule = true;
    uint32_t sched_period = 2;  // frames between schedule rebuilds (tools/perf_probe.py --motion: 2 keeps the gain under camera motion)
    int frame_parity = 0;
    // shading pass scratch (svo_render with rgba_out)
    void *shade_hits = nullptr, *shade_aux = nullptr, *shade_rays = nullptr, *shade_shadow = nullptr, *shade_skip = nullptr;
    size_t shade_hits_bytes = 0, shade_aux_bytes = 0, shade_rays_bytes = 0, shade_shadow_bytes = 0, shade_skip_bytes = 0;
    uint32_t *debug_buf = nullptr;  // caller-provided device buffer for the per-wave timeline (diagnostics)
    uint32_t strip_items = 64;
    bool dynamic_strips = true;
    // launch timing: a ring of (start, stop) event pairs recorded around trace launches
    std::vector<hipEvent_t> ev;  // 2 per slot
    size_t ev_slots = 0, ev_count = 0;
    std::string err;
};


// svo_abi.cpp
int svo_fail(svo_ctx *ctx, int code, const char *what);
int svo_fail_hip(svo_ctx *ctx, hipError_t e, const char *what);
// svo_comm.cpp
void svo_comm_release(svo_ctx *ctx);
