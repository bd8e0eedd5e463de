// svo_abi.cpp -- the extern "C" boundary declared in include/svo_hip.h: device context, node
// buffer, uniforms, trace / scan dispatch.  Replaces the wgpu plumbing of the reference's gpu.rs,
// and the device halves of render.rs / compute.rs (citations in the header).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "svo_ctx.h"

int svo_fail(svo_ctx *ctx, int code, const char *what) {
    if (ctx) ctx->err = what;
    return code;
}

int svo_fail_hip(svo_ctx *ctx, hipError_t e, const char *what) {
    if (ctx) ctx->err = std::string(what) + ": " + hipGetErrorString(e);
    return SVO_ERR_HIP;
}

namespace {

constexpr size_t kScanCapacity = 1024000;  // adaptive.rs:3-4

int fail(svo_ctx *ctx, int code, const char *what) { return svo_fail(ctx, code, what); }
int fail_hip(svo_ctx *ctx, hipError_t e, const char *what) { return svo_fail_hip(ctx, e, what); }

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr); \
    } while (0)

int bind(svo_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return SVO_OK;
}

// ---- node store (svo_ctx.h) ----
// Reads issued by this context from now on come after the store's last write, whichever context's stream it ran on.
int order_after_last_write(svo_ctx *ctx) {
    svo_node_store *st = ctx->store;
    if (st->last_write && st->last_writer != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, st->last_write, 0));
    return SVO_OK;
}

// A write to the store's words has been enqueued on this context's stream (or, `on_stream` false, may happen behind
// the library's back through a pointer the caller holds): every context bound to the store rebuilds what it derived.
int note_write(svo_ctx *ctx, bool on_stream) {
    svo_node_store *st = ctx->store;
    st->version++;
    if (on_stream) {
        if (!st->last_write) HIP_TRY(ctx, hipEventCreateWithFlags(&st->last_write, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(st->last_write, ctx->stream));
        st->last_writer = ctx->stream;
    }
    return SVO_OK;
}

void release_store(svo_ctx *ctx) {
    svo_node_store *st = ctx->store;
    ctx->store = nullptr;
    ctx->nodes = nullptr;
    ctx->capacity = 0;
    ctx->top_version = 0;
    if (!st || --st->refs > 0) return;
    (void)hipSetDevice(st->device);
    if (st->nodes && st->owned) (void)hipFree(st->nodes);
    if (st->last_write) (void)hipEventDestroy(st->last_write);
    delete st;
}

void adopt_store(svo_ctx *ctx, svo_node_store *st) {
    ctx->store = st;
    ctx->nodes = st->nodes;
    ctx->capacity = st->capacity;
    ctx->top_version = 0;
}

int ensure_top_table(svo_ctx *ctx) {
    if (ctx->top_version == ctx->store->version) return SVO_OK;
    int rc = order_after_last_write(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, svo::launch_build_top_table(ctx->nodes, (uint32_t)ctx->capacity, ctx->top_table, ctx->stream));
    ctx->top_version = ctx->store->version;
    return SVO_OK;
}

// Is the pre-trace culling pass (strip_cull_kernel) worth its three small launches (about 15 us) for this frame?  Only a
// heuristic -- which strips are culled is decided on the device.  Yes when the camera stands outside the cube and a good part
// of the screen looks past it: a coarse 12 x 8 grid of rays, made the way the kernel makes them (camera_inverse only; the
// reference's forward matrix is affine and says nothing about the perspective), is tested against the cube here on the host.
bool cull_worthwhile(const svo_ctx *ctx) {
    if (ctx->cull_mode == 0) return false;
    const float *ci = ctx->uniforms.camera_inverse;
    const double w = ci[15];
    if (!(fabs(w) > 1e-20)) return false;
    const double o[3] = {ci[12] / w, ci[13] / w, ci[14] / w};  // camera_inverse * (0, 0, 0, 1)
    if (!(fabs(o[0]) > 1.001 || fabs(o[1]) > 1.001 || fabs(o[2]) > 1.001)) return false;  // inside (or NaN): nothing to cull
    if (ctx->cull_mode == 1) return true;
    constexpr int NX = 12, NY = 8;
    int miss = 0;
    for (int j = 0; j < NY; j++)
        for (int i = 0; i < NX; i++) {
            const double cx = (i + 0.5) / NX * 2.0 - 1.0, cy = -((j + 0.5) / NY * 2.0 - 1.0);
            double d4[4];
            for (int r = 0; r < 4; r++) d4[r] = ci[r] * cx + ci[4 + r] * cy + ci[8 + r] + ci[12 + r];
            if (!(fabs(d4[3]) > 1e-20)) return false;
            double t0 = -1e300, t1 = 1e300;
            for (int k = 0; k < 3; k++) {
                const double d = d4[k] / d4[3] - o[k];
                if (d == 0.0) {
                    if (fabs(o[k]) > 1.0) t1 = -1e300;
                    continue;
                }
                const double a = (-1.0 - o[k]) / d, b = (1.0 - o[k]) / d;
                t0 = fmax(t0, fmin(a, b));
                t1 = fmin(t1, fmax(a, b));
            }
            if (!(t1 >= 0.0 && t0 <= t1)) miss++;
        }
    return miss * 10 >= NX * NY * 4;  // at least 40 % of the sampled rays look past the cube
}

struct TraceOpts {
    float *aux_t = nullptr;    // t_current per record (shading pass)
    bool count_rays = false;   // explicit rays also bump hit counters (the shadow ray passes primary = true, shader.wgsl:276)
    int sched_slot = 0;        // which schedule history this launch feeds (0: primary frame, 1: shadow rays)
    const uint8_t *skip = nullptr;  // explicit rays: slots without a ray (secondary rays of pixels that hit nothing)
    svo_hit *shadow_out = nullptr;  // STACK: trace the shadow ray of every hit inside this launch, records here
};

int trace_launch(svo_ctx *ctx, const svo::WorkDesc &work, const float *rays, svo_hit *hits, const TraceOpts &opt) {
    int rc = bind(ctx);
    if (rc) return rc;
    // the kernel that runs: STACK resolves the levels its integer path codes cover (SVO_OPT_TREE_DEPTH says how deep the
    // caller's tree may be); deeper trees, and the debug view that reads counter bits, take the general RESTART kernel
    const bool want_stack = ctx->variant != SVO_VARIANT_RESTART && ctx->tree_depth <= (uint32_t)svo::stack_max_depth(true);
    if (want_stack) {
        rc = ensure_top_table(ctx);
        if (rc) return rc;
    } else {
        rc = order_after_last_write(ctx);
        if (rc) return rc;
    }
    svo::WorkDesc wd = work;
    auto magic = [](uint32_t d) -> uint32_t { return d <= 1u ? 0u : (uint32_t)(0x100000000ull / d) + 1u; };
    wd.magic_bpr = magic(wd.bpr);
    wd.magic_bprect = magic(wd.bprect);
    wd.magic_tiles_x = magic(wd.tiles_x);
    svo::TraceArgs a{};
    a.nodes = ctx->nodes;
    a.n_words = (uint32_t)ctx->capacity;
    a.top_table = ctx->top_table;
    a.u = ctx->uniforms;
    a.work = wd;
    a.rays = rays;
    a.hits = hits;
    a.aux_t = opt.aux_t;
    a.status = ctx->status;
    a.refill_min = ctx->refill_min;
    a.cam_shortcut = ctx->cam_shortcut ? 1u : 0u;
    a.debug = ctx->debug_buf;
    a.skip = opt.skip;
    // shader.wgsl:159: counters are live unless pause_adaptive; rays handed in by the caller (svo_trace_rays) never count
    const bool counting = (work.mode != 2 || opt.count_rays) && !(ctx->uniforms.flags & SVO_F_PAUSE_ADAPTIVE);
    a.count_nodes = counting ? ctx->nodes : nullptr;
    const bool debug_hits = (ctx->uniforms.flags & SVO_F_PAUSE_ADAPTIVE) && (ctx->uniforms.flags & SVO_F_SHOW_HITS);
    const bool stack = want_stack && !debug_hits;
    a.shadow_hits = stack ? opt.shadow_out : nullptr;
    const uint32_t n_strips = (wd.n_items + 63u) / 64u;
    const bool schedule = ctx->schedule && n_strips <= svo::kMaxScheduledStrips;
    svo_ctx::Sched &sc = ctx->sched[opt.sched_slot & 1];
    const bool cull = stack && schedule && wd.mode != 2 && hits != nullptr && cull_worthwhile(ctx);
    const bool filtered = stack && schedule && (opt.skip != nullptr || cull);
    if (stack && schedule) {
        if (sc.cap < n_strips) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (sc.cost) (void)hipFree(sc.cost);
            if (sc.cls_now) (void)hipFree(sc.cls_now);
            if (sc.order) (void)hipFree(sc.order);
            if (sc.balance) (void)hipFree(sc.balance);
            sc = svo_ctx::Sched{};
            size_t want = n_strips < 4096 ? 4096 : n_strips;
            HIP_TRY(ctx, hipMalloc((void **)&sc.cost, want + 32 + svo::kOrderHistWords * sizeof(uint32_t)));
            HIP_TRY(ctx, hipMalloc((void **)&sc.cls_now, want + 32 + svo::kOrderHistWords * sizeof(uint32_t)));
            HIP_TRY(ctx, hipMalloc((void **)&sc.order, (2 * want + 8 * (svo::kCostClasses + 8) + 8) * sizeof(uint32_t)));  // (8 lists of order_list_cap entries)
            HIP_TRY(ctx, hipMalloc((void **)&sc.balance, svo::kBalanceWords * sizeof(uint32_t)));
            sc.cap = want;
        }
        // costs and order are only meaningful for the same work layout (same pixels behind every strip)
        if (sc.valid && memcmp(&sc.key, &wd, sizeof(wd)) != 0) sc.valid = false;
        if (!sc.valid) {  // a new layout starts from equal shares of the lists
            uint32_t init[svo::kBalanceWords] = {};
            for (uint32_t k = 0; k <= 8; k++) init[k] = k * 8192u;
            HIP_TRY(ctx, hipMemcpyAsync(sc.balance, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
            sc.balance_frames = 0;
        }
        a.balance = ctx->list_balance ? sc.balance : nullptr;
        a.order_cap = svo::order_list_cap(wd, n_strips);
        if (filtered) {
            // slots without a ray (secondary rays of pixels that hit nothing): this frame's lists leave out the strips
            // that consist of nothing else, ordered by the costs of an earlier frame when there are any
            // (likewise for pixel frames seen from outside the cube: strips of sky are culled before the trace, their zero
            // records written by the culling pass)
            if (opt.skip)
                HIP_TRY(ctx, svo::launch_schedule_skipping(opt.skip, wd.n_items, sc.valid ? sc.cost : nullptr, sc.cls_now, sc.order, n_strips,
                                                           a.order_cap, ctx->stream));
            else
                HIP_TRY(ctx, svo::launch_schedule_culling(a, sc.valid ? sc.cost : nullptr, sc.cls_now, sc.order, n_strips, a.order_cap,
                                                          ctx->stream));
            a.order = sc.order;
            sc.order_filtered = true;  // these lists leave strips out: good for this frame only
        } else {
            a.order = (sc.valid && !sc.order_filtered) ? sc.order : nullptr;
        }
    }
    svo::LaunchInfo li{};
    li.variant = stack ? SVO_VARIANT_STACK : SVO_VARIANT_RESTART;
    li.grid_blocks = ctx->grid_blocks;
    li.num_cus = ctx->num_cus;
    li.strip_items = ctx->strip_items;
    li.deep_stack = ctx->tree_depth > (uint32_t)svo::stack_max_depth(false);
    li.occupancy = ctx->occupancy;
    if (stack && ctx->defer_items < wd.n_items) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->defer_buf) (void)hipFree(ctx->defer_buf);
        ctx->defer_buf = nullptr;
        ctx->frame_parity = 0;
        ctx->defer_items = 0;
        size_t want = wd.n_items < (1u << 16) ? (1u << 16) : wd.n_items;
        // layout: claim counters | list A: count + items | list B: count + items
        HIP_TRY(ctx, hipMalloc((void **)&ctx->defer_buf, (svo::kCounterWords + 2 * (want + 1)) * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->defer_buf, 0, (svo::kCounterWords + 2 * (want + 1)) * sizeof(uint32_t), ctx->stream));
        ctx->defer_items = want;
    }
    li.counters = ctx->defer_buf;
    li.work_counter = ctx->defer_buf;  // (strips are always claimed dynamically; SVO_OPT_DYNAMIC_STRIPS is accepted and ignored)
    if (ctx->defer_buf) {
        uint32_t *lists = ctx->defer_buf + svo::kCounterWords;
        const size_t stride = ctx->defer_items + 1;
        li.defer = lists + (ctx->frame_parity ? stride : 0);
        li.next_defer_count = lists + (ctx->frame_parity ? 0 : stride);
    }
    const size_t slot = ctx->ev_slots ? (ctx->ev_count % ctx->ev_slots) : 0;
    const bool timed = ctx->ev_slots && opt.sched_slot == 0;  // the timing ring records the primary trace launches
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[2 * slot], ctx->stream));
    HIP_TRY(ctx, svo::launch_trace(a, li, ctx->stream));
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[2 * slot + 1], ctx->stream));
        ctx->ev_count++;
    }
    if (stack) {
        // deferred rays, scheduling feedback for the next frames, counter re-arm
        // Rebuild the schedule when there is none, and every sched_period frames while the input moves.  A ray's step
        // count does not depend on the order it was traced in, so a schedule measured on this camera and tree stays
        // exact as long as both stay put (caller-supplied rays cannot be compared: they always count as moving; 64
        // frames is a backstop for node buffers written behind this context's back).
        const bool same_input = (wd.mode != 2 || opt.sched_slot == 1) && sc.built_nodes_version == ctx->store->version &&
                                memcmp(&sc.built_uniforms, &ctx->uniforms, sizeof(svo_uniforms)) == 0 && sc.age < 64;
        const bool moving = sc.have_prev && memcmp(&sc.prev_uniforms, &ctx->uniforms, sizeof(svo_uniforms)) != 0;
        // (a frame traced with complete lists also feeds the time its lists took back into their shares: while camera and tree stay
        // put the lists are rebuilt for the first sixteen such frames, then the shares have settled)
        const bool fed_back = a.balance != nullptr && a.order != nullptr && !filtered;
        const bool rebuild = schedule && (!sc.valid || (sc.order_filtered && !filtered) || (!same_input && sc.age + 1 >= ctx->sched_period) ||
                                          (same_input && sc.floored && !moving) || (same_input && fed_back && sc.balance_frames < 16u));
        // (a resting view rebuilt for the shares' sake only: its strips' cost classes are the ones measured last time, the pass is skipped)
        const bool shares_only = rebuild && sc.valid && same_input && fed_back && sc.balance_frames < 16u && !(sc.order_filtered && !filtered) &&
                                 !(sc.floored && !moving);
        // a camera in motion: strips near the long ones of this frame are not scheduled as cheap (strip_danger_kernel)
        const bool floor_now = rebuild && !filtered && moving && ctx->motion_floor != 0u && wd.mode == 0 && wd.n_rects == 1u;
        // (a launch with a skip mask builds its lists before the trace, every frame: here only the costs are measured)
        HIP_TRY(ctx, svo::launch_post(a, li, rebuild ? sc.cost : nullptr, sc.order, n_strips, svo::order_list_cap(wd, n_strips),
                                      rebuild && !filtered, ctx->stream, floor_now ? sc.cls_now : nullptr, ctx->motion_floor,
                                      (fed_back && (!same_input || sc.balance_frames < 16u)) ? sc.balance_frames + 1u : 0u, shares_only));
        if (fed_back) sc.balance_frames++;
        if (rebuild && !filtered) sc.floored = floor_now;
        sc.prev_uniforms = ctx->uniforms;
        sc.have_prev = true;
        ctx->frame_parity ^= 1;
        if (rebuild) {
            if (!filtered) sc.order_filtered = false;  // the post pass has just built complete lists
            sc.key = wd;
            sc.valid = true;
            sc.age = 0;
            sc.built_uniforms = ctx->uniforms;
            sc.built_nodes_version = ctx->store->version;
        } else if (schedule) {
            sc.age++;
        }
    }
    return SVO_OK;
}

// Shadow rays inside the primary launch?  STACK variant and a sun direction the fast arithmetic covers (it is the same for every
// shadow ray, so this is decided here; the origins are hit points inside the cube and always qualify).  Automatic mode fuses
// whenever that holds: the shadow ray restarts from the primary ray's ancestor stack instead of descending from the root, and
// there is one launch, one set of claims, one tail.  (Until the claim counters were split -- eight per list -- a 1080p frame
// of the depth-16 terrain was 5 % slower fused and the automatic mode fused only deep trees, 4K frames and counting frames;
// since then: 1080p 0.646 -> 0.581 ms, 4K 2.04 -> 1.78 ms, depth-20 fractal -24 %.)
bool fuse_shadow_rays(const svo_ctx *ctx, size_t n_pixels) {
    (void)n_pixels;
    const bool want = ctx->fused_shadows != 0;
    if (!want || ctx->variant == SVO_VARIANT_RESTART || ctx->tree_depth > (uint32_t)svo::stack_max_depth(true)) return false;
    const float *sd = ctx->uniforms.sun_dir;
    const float sl = sqrtf((sd[0] * sd[0] + sd[1] * sd[1]) + sd[2] * sd[2]);
    for (int k = 0; k < 3; k++) {
        float d = -(sd[k] / sl);
        if (d == 0.0f) d = 0.000001f;  // octree_ray's bias, shader.wgsl:193-194
        if (!(fabsf(d) >= 1.0e-11f && fabsf(d) <= 2.0f)) return false;  // (also NaN); the kernel needs 2^-40 <= |d| <= 2^40
    }
    return true;
}

int ensure_dev(svo_ctx *ctx, void **buf, size_t *have, size_t bytes) {
    if (*have >= bytes) return SVO_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    HIP_TRY(ctx, hipMalloc(buf, bytes));
    *have = bytes;
    return SVO_OK;
}

// Trace (and, when rgba is asked for, shade) the pixels of `work`: fs_main, shader.wgsl:250-304.
int trace_common(svo_ctx *ctx, const svo::WorkDesc &work, const float *rays, svo_hit *hits, uint32_t *rgba) {
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc / svo_nodes_bind_device not called");
    if (work.mode != 2 && !ctx->have_uniforms) return fail(ctx, SVO_ERR_STATE, "svo_set_uniforms not called");
    if (!hits && !rgba) return fail(ctx, SVO_ERR_ARG, "both hits_out and rgba_out are NULL");
    if (!rgba) return trace_launch(ctx, work, rays, hits, TraceOpts{});
    if (work.mode == 2) return fail(ctx, SVO_ERR_ARG, "explicit rays have no pixels to shade");
    int rc = bind(ctx);
    if (rc) return rc;
    const size_t n = (size_t)work.n_rects * work.w * work.h;  // records of this call
    const uint32_t f = ctx->uniforms.flags;
    const bool shadows = (f & SVO_F_SHADOWS) && !(f & SVO_F_SHOW_STEPS) && !(f & SVO_F_SHOW_HITS);
    if (!hits) {
        rc = ensure_dev(ctx, &ctx->shade_hits, &ctx->shade_hits_bytes, n * sizeof(svo_hit));
        if (rc) return rc;
        hits = (svo_hit *)ctx->shade_hits;
    }
    const bool fused = shadows && fuse_shadow_rays(ctx, n);
    TraceOpts primary;
    if (fused) {
        rc = ensure_dev(ctx, &ctx->shade_shadow, &ctx->shade_shadow_bytes, n * sizeof(svo_hit));
        if (rc) return rc;
        primary.shadow_out = (svo_hit *)ctx->shade_shadow;
    } else {
        rc = ensure_dev(ctx, &ctx->shade_aux, &ctx->shade_aux_bytes, n * sizeof(float));
        if (rc) return rc;
        primary.aux_t = (float *)ctx->shade_aux;
    }
    rc = trace_launch(ctx, work, nullptr, hits, primary);
    if (rc) return rc;
    svo::TraceArgs a{};
    a.nodes = ctx->nodes;
    a.n_words = (uint32_t)ctx->capacity;
    a.u = ctx->uniforms;
    a.work = work;
    a.hits = hits;
    if (shadows && !fused) {
        rc = ensure_dev(ctx, &ctx->shade_rays, &ctx->shade_rays_bytes, n * 6 * sizeof(float));
        if (rc) return rc;
        rc = ensure_dev(ctx, &ctx->shade_shadow, &ctx->shade_shadow_bytes, n * sizeof(svo_hit));
        if (rc) return rc;
        rc = ensure_dev(ctx, &ctx->shade_skip, &ctx->shade_skip_bytes, n);
        if (rc) return rc;
        HIP_TRY(ctx, svo::launch_secondary_gen(a, (const float *)ctx->shade_aux, (float *)ctx->shade_rays, (uint8_t *)ctx->shade_skip,
                                               (svo_hit *)ctx->shade_shadow, 0u, 1u, (uint32_t)n, ctx->stream));
        svo::WorkDesc rw{};
        rw.mode = 2;
        rw.n_items = (uint32_t)n;
        rw.bpr = rw.bprect = rw.tiles_x = 1;
        TraceOpts shadow;
        shadow.count_rays = true;
        shadow.sched_slot = 1;
        shadow.skip = (const uint8_t *)ctx->shade_skip;
        rc = trace_launch(ctx, rw, (const float *)ctx->shade_rays, (svo_hit *)ctx->shade_shadow, shadow);
        if (rc) return rc;
    }
    HIP_TRY(ctx, svo::launch_shade(a, shadows ? (const svo_hit *)ctx->shade_shadow : nullptr, rgba, ctx->stream));
    return SVO_OK;
}

// Primary rays of `work`, then n_secondary rays from every hit pixel (ray 0: fs_main's shadow ray), traced as one
// launch of explicit rays; records ray-major: secondary[k * n + record].
int trace_secondary(svo_ctx *ctx, const svo::WorkDesc &work, uint32_t n_secondary, svo_hit *primary, svo_hit *secondary) {
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc / svo_nodes_bind_device not called");
    if (!secondary) return fail(ctx, SVO_ERR_ARG, "secondary_out is NULL");
    if (n_secondary < 1 || n_secondary > 4) return fail(ctx, SVO_ERR_ARG, "n_secondary must be 1..4");
    int rc = bind(ctx);
    if (rc) return rc;
    const size_t n = (size_t)work.n_rects * work.w * work.h;
    if (n * n_secondary > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "too many secondary rays for one call");
    if (!primary) {
        rc = ensure_dev(ctx, &ctx->shade_hits, &ctx->shade_hits_bytes, n * sizeof(svo_hit));
        if (rc) return rc;
        primary = (svo_hit *)ctx->shade_hits;
    }
    rc = ensure_dev(ctx, &ctx->shade_aux, &ctx->shade_aux_bytes, n * sizeof(float));
    if (rc) return rc;
    rc = ensure_dev(ctx, &ctx->shade_rays, &ctx->shade_rays_bytes, n * n_secondary * 6 * sizeof(float));
    if (rc) return rc;
    // ray 0, the shadow ray, can run inside the primary launch (its records go straight to the first set)
    const bool debug_view = (ctx->uniforms.flags & SVO_F_PAUSE_ADAPTIVE) && (ctx->uniforms.flags & SVO_F_SHOW_HITS);
    const bool fused = !debug_view && fuse_shadow_rays(ctx, n);
    const uint32_t k_first = fused ? 1u : 0u;
    TraceOpts popt;
    popt.aux_t = (float *)ctx->shade_aux;
    if (fused) popt.shadow_out = secondary;
    rc = trace_launch(ctx, work, nullptr, primary, popt);
    if (rc) return rc;
    if (k_first == n_secondary) return SVO_OK;
    svo::TraceArgs a{};
    a.nodes = ctx->nodes;
    a.n_words = (uint32_t)ctx->capacity;
    a.u = ctx->uniforms;
    a.work = work;
    a.hits = primary;
    rc = ensure_dev(ctx, &ctx->shade_skip, &ctx->shade_skip_bytes, n * n_secondary);
    if (rc) return rc;
    svo_hit *rest = secondary + (size_t)k_first * n;
    HIP_TRY(ctx, svo::launch_secondary_gen(a, (const float *)ctx->shade_aux, (float *)ctx->shade_rays, (uint8_t *)ctx->shade_skip, rest,
                                           k_first, n_secondary, (uint32_t)n, ctx->stream));
    svo::WorkDesc rw{};
    rw.mode = 2;
    rw.n_items = (uint32_t)(n * (n_secondary - k_first));
    rw.bpr = rw.bprect = rw.tiles_x = 1;
    TraceOpts sopt;
    sopt.count_rays = true;  // like the shadow ray, which passes primary = true (shader.wgsl:276)
    sopt.sched_slot = 1;
    sopt.skip = (const uint8_t *)ctx->shade_skip;
    return trace_launch(ctx, rw, (const float *)ctx->shade_rays, rest, sopt);
}

int make_tiles_work(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t first_tile,
                    uint32_t tile_stride, svo::WorkDesc &work) {
    if (!ctx->have_uniforms) return fail(ctx, SVO_ERR_STATE, "svo_set_uniforms not called");
    if ((float)width != ctx->uniforms.dimensions[0] || (float)height != ctx->uniforms.dimensions[1])
        return fail(ctx, SVO_ERR_ARG, "width/height differ from uniforms.dimensions");
    if (tile_w == 0 || tile_h == 0 || width % tile_w || height % tile_h)
        return fail(ctx, SVO_ERR_ARG, "frame must be a whole number of tiles");
    if (tile_stride == 0) return fail(ctx, SVO_ERR_ARG, "tile_stride must be >= 1");
    uint32_t tiles_x = width / tile_w, tiles = tiles_x * (height / tile_h);
    work = svo::WorkDesc{};
    work.mode = 1;
    work.w = tile_w; work.h = tile_h;
    work.bw_log2 = ctx->block_w_log2;
    {
        const uint32_t bw = 1u << work.bw_log2, bh = 64u >> work.bw_log2;
        work.bpr = (tile_w + bw - 1) / bw;
        work.bprect = work.bpr * ((tile_h + bh - 1) / bh);
    }
    work.n_rects = first_tile < tiles ? (tiles - first_tile + tile_stride - 1) / tile_stride : 0;
    work.tiles_x = tiles_x;
    work.first_tile = first_tile;
    work.tile_stride = tile_stride;
    uint64_t items = (uint64_t)work.n_rects * work.bprect * 64u;
    if (items > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "too many pixels for one call");
    work.n_items = (uint32_t)items;
    return SVO_OK;
}

int make_rect_work(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                   svo::WorkDesc &work) {
    if (!ctx->have_uniforms) return fail(ctx, SVO_ERR_STATE, "svo_set_uniforms not called");
    if ((float)width != ctx->uniforms.dimensions[0] || (float)height != ctx->uniforms.dimensions[1])
        return fail(ctx, SVO_ERR_ARG, "width/height differ from uniforms.dimensions");
    if (w == 0 || h == 0 || x0 + (uint64_t)w > width || y0 + (uint64_t)h > height)
        return fail(ctx, SVO_ERR_ARG, "tile rectangle outside the frame");
    if ((uint64_t)w * h > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "tile too large");
    work = svo::WorkDesc{};
    work.mode = 0;
    work.x0 = x0; work.y0 = y0; work.w = w; work.h = h;
    work.bw_log2 = ctx->block_w_log2;
    {
        const uint32_t bw = 1u << work.bw_log2, bh = 64u >> work.bw_log2;
        work.bpr = (w + bw - 1) / bw;
        work.bprect = work.bpr * ((h + bh - 1) / bh);
    }
    work.n_rects = 1;
    work.tiles_x = 1;
    work.n_items = work.bprect * 64u;
    return SVO_OK;
}

int ensure_stage(svo_ctx *ctx, size_t bytes) {
    if (ctx->stage_bytes >= bytes) return SVO_OK;
    if (ctx->stage) (void)hipFree(ctx->stage);
    ctx->stage = nullptr;
    ctx->stage_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    return SVO_OK;
}

}  // namespace

extern "C" {

int svo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int svo_buffer_alloc(svo_ctx *ctx, size_t bytes, void **device_out) {
    if (!ctx || !device_out || bytes == 0) return SVO_ERR_ARG;
    *device_out = nullptr;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMalloc(device_out, bytes));
    return SVO_OK;
}

int svo_buffer_free(svo_ctx *ctx, void *device_ptr) {
    if (!ctx) return SVO_ERR_ARG;
    if (!device_ptr) return SVO_OK;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // work that still uses the buffer finishes first
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    HIP_TRY(ctx, hipFree(device_ptr));
    return SVO_OK;
}

int svo_buffer_read(svo_ctx *ctx, const void *device_ptr, void *host_out, size_t bytes) {
    if (!ctx || ((!device_ptr || !host_out) && bytes)) return SVO_ERR_ARG;
    int rc = bind(ctx);
    if (rc || bytes == 0) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(host_out, device_ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

int svo_ctx_create(int hip_device, svo_ctx **out) {
    if (!out) return SVO_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SVO_ERR_NO_DEVICE;
    if (hip_device < 0 || hip_device >= n) return SVO_ERR_ARG;
    svo_ctx *ctx = new (std::nothrow) svo_ctx();
    if (!ctx) return SVO_ERR_HIP;
    ctx->device = hip_device;
    hipError_t e = hipSetDevice(hip_device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, hip_device);
    if (e == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->top_table, (svo::kTopEntries + svo::kTopAuxEntries) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->status, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(ctx->status, 0, sizeof(uint32_t));
    if (e != hipSuccess) {
        svo_ctx_destroy(ctx);
        return SVO_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return SVO_OK;
}

int svo_ctx_destroy(svo_ctx *ctx) {
    if (!ctx) return SVO_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    svo_comm_release(ctx);
    release_store(ctx);
    if (ctx->top_table) (void)hipFree(ctx->top_table);
    if (ctx->status) (void)hipFree(ctx->status);
    if (ctx->defer_buf) (void)hipFree(ctx->defer_buf);
    for (auto &sc : ctx->sched) {
        if (sc.cost) (void)hipFree(sc.cost);
        if (sc.cls_now) (void)hipFree(sc.cls_now);
        if (sc.order) (void)hipFree(sc.order);
    }
    for (void *p : {ctx->shade_hits, ctx->shade_aux, ctx->shade_rays, ctx->shade_shadow, ctx->shade_skip, ctx->scatter_buf})
        if (p) (void)hipFree(p);
    if (ctx->scan_sub) (void)hipFree(ctx->scan_sub);  // scan_unsub is the second half of the same allocation
    if (ctx->stage) (void)hipFree(ctx->stage);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SVO_OK;
}

int svo_ctx_set_stream(svo_ctx *ctx, void *hip_stream, int use_own) {
    if (!ctx) return SVO_ERR_ARG;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // work queued on the old stream finishes first
    ctx->stream = use_own ? ctx->own_stream : (hipStream_t)hip_stream;
    return SVO_OK;
}

int svo_set_option(svo_ctx *ctx, int option, int64_t value) {
    if (!ctx) return SVO_ERR_ARG;
    switch (option) {
        case SVO_OPT_VARIANT:
            if (value != SVO_VARIANT_RESTART && value != SVO_VARIANT_STACK) return fail(ctx, SVO_ERR_ARG, "unknown variant");
            ctx->variant = (int)value;
            return SVO_OK;
        case SVO_OPT_TIMING: {
            if (value < 0 || value > 65536) return fail(ctx, SVO_ERR_ARG, "timing ring size out of range");
            int rc = bind(ctx);
            if (rc) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            while (ctx->ev.size() < 2 * (size_t)value) {
                hipEvent_t e;
                HIP_TRY(ctx, hipEventCreate(&e));
                ctx->ev.push_back(e);
            }
            ctx->ev_slots = (size_t)value;
            ctx->ev_count = 0;
            return SVO_OK;
        }
        case SVO_OPT_GRID_BLOCKS:
            if (value < 0 || value > 65535) return fail(ctx, SVO_ERR_ARG, "grid_blocks out of range");
            ctx->grid_blocks = (int)value;
            return SVO_OK;
        case SVO_OPT_REFILL_MIN:
            if (value < 1 || value > 64) return fail(ctx, SVO_ERR_ARG, "refill_min must be 1..64");
            ctx->refill_min = (uint32_t)value;
            return SVO_OK;
        case SVO_OPT_SCAN_CLEARS_COUNTERS:
            ctx->scan_clears = value != 0;
            return SVO_OK;
        case SVO_OPT_FUSED_SHADOWS:
            if (value < 0 || value > 2) return fail(ctx, SVO_ERR_ARG, "fused shadows: 0 (off), 1 (on) or 2 (automatic)");
            ctx->fused_shadows = (int)value;
            return SVO_OK;
        case SVO_OPT_STRIP_ITEMS:
            if (value < 64 || value > 2048 || (value & 63)) return fail(ctx, SVO_ERR_ARG, "strip_items must be a multiple of 64 in [64, 2048]");
            ctx->strip_items = (uint32_t)value;
            return SVO_OK;
        case SVO_OPT_DYNAMIC_STRIPS:
            ctx->dynamic_strips = value != 0;
            return SVO_OK;
        case SVO_OPT_SCHEDULE:
            if (value < 0 || value > 1024) return fail(ctx, SVO_ERR_ARG, "schedule period out of range");
            ctx->schedule = value != 0;
            if (value) ctx->sched_period = (uint32_t)value;
            ctx->sched[0].valid = ctx->sched[1].valid = false;
            return SVO_OK;
        case SVO_OPT_TREE_DEPTH:
            if (value < 1 || value > 31) return fail(ctx, SVO_ERR_ARG, "tree depth must be 1..31");
            ctx->tree_depth = (uint32_t)value;  // (which kernel that means is decided per launch, see trace_launch)
            return SVO_OK;
        case SVO_OPT_SCHEDULE_MOTION:
            if (value < 0 || (value & 15) > 12 || ((value >> 8) & 15) > 4 || (value >> 12) > 80)
                return fail(ctx, SVO_ERR_ARG, "schedule motion: class floor (0..12, 0 = off) | radius (0..4) << 8 | min_count (0..80) << 12");
            ctx->motion_floor = (uint32_t)value;
            return SVO_OK;
        case SVO_OPT_CAMERA_SHORTCUT:
            ctx->cam_shortcut = value != 0;
            return SVO_OK;
        case SVO_OPT_CULL:
            if (value < 0 || value > 2) return fail(ctx, SVO_ERR_ARG, "cull: 0 (off), 1 (whenever the camera is outside the cube) or 2 (automatic)");
            ctx->cull_mode = (int)value;
            return SVO_OK;
        case SVO_OPT_PAIR_TABLE:  // (the table left the library in round 3)
            return SVO_OK;
        case SVO_OPT_DEBUG_BUFFER:
            ctx->debug_buf = (uint32_t *)(uintptr_t)value;  // device pointer, 64 B per wave of the grid; 0 = off
            return SVO_OK;
        case SVO_OPT_BLOCK_SHAPE:
            if (value < 0 || value > 6) return fail(ctx, SVO_ERR_ARG, "block width log2 must be 0..6");
            ctx->block_w_log2 = (uint32_t)value;
            return SVO_OK;
        case SVO_OPT_PRIO_STEPS:
            if (value < 0 || value > 255) return fail(ctx, SVO_ERR_ARG, "prio_steps must be 0..255");
            ctx->prio_steps = (uint32_t)value;
            return SVO_OK;
        default:
            return fail(ctx, SVO_ERR_ARG, "unknown option");
    }
}

const char *svo_last_error(const svo_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int svo_sync(svo_ctx *ctx) {
    if (!ctx) return SVO_ERR_ARG;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));  // frame-end gathers in flight
    // surface device-side errors raised by trace kernels
    uint32_t st = 0;
    HIP_TRY(ctx, hipMemcpy(&st, ctx->status, sizeof(st), hipMemcpyDeviceToHost));
    if (st & 2u) {
        HIP_TRY(ctx, hipMemset(ctx->status, 0, sizeof(uint32_t)));
        return fail(ctx, SVO_ERR_STATE, "fused shadow ray outside the range of the fast arithmetic (set SVO_OPT_FUSED_SHADOWS to 0)");
    }
    if (st & 1u) {
        HIP_TRY(ctx, hipMemset(ctx->status, 0, sizeof(uint32_t)));
        return fail(ctx, SVO_ERR_STATE,
                    "octree deeper than SVO_OPT_TREE_DEPTH declares (the frame's records are not valid); "
                    "raise SVO_OPT_TREE_DEPTH (above 22: the general kernel) or use SVO_VARIANT_RESTART");
    }
    return SVO_OK;
}

int svo_nodes_alloc(svo_ctx *ctx, size_t capacity_words) {
    if (!ctx) return SVO_ERR_ARG;
    if (capacity_words < 8 || capacity_words > (size_t)SVO_VOXEL_OFFSET)
        return fail(ctx, SVO_ERR_ARG, "capacity_words must be in [8, 2^27]");
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    release_store(ctx);  // (contexts that share the old store keep it alive)
    uint32_t *words = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&words, capacity_words * sizeof(uint32_t)));
    svo_node_store *st = new (std::nothrow) svo_node_store();
    if (!st) {
        (void)hipFree(words);
        return fail(ctx, SVO_ERR_HIP, "out of host memory");
    }
    st->device = ctx->device;
    st->nodes = words;
    st->capacity = capacity_words;
    st->owned = true;
    adopt_store(ctx, st);
    // Octree::expanded zero-fills the tail (octree.rs:143-148)
    HIP_TRY(ctx, hipMemsetAsync(ctx->nodes, 0, capacity_words * sizeof(uint32_t), ctx->stream));
    return note_write(ctx, true);
}

int svo_nodes_bind_device(svo_ctx *ctx, uint32_t *device_words, size_t capacity_words) {
    if (!ctx || !device_words) return SVO_ERR_ARG;
    if (capacity_words < 8 || capacity_words > (size_t)SVO_VOXEL_OFFSET)
        return fail(ctx, SVO_ERR_ARG, "capacity_words must be in [8, 2^27]");
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    release_store(ctx);
    svo_node_store *st = new (std::nothrow) svo_node_store();
    if (!st) return fail(ctx, SVO_ERR_HIP, "out of host memory");
    st->device = ctx->device;
    st->nodes = device_words;
    st->capacity = capacity_words;
    st->owned = false;
    adopt_store(ctx, st);
    return SVO_OK;
}

int svo_nodes_share(svo_ctx *ctx, svo_ctx *owner) {
    if (!ctx || !owner) return SVO_ERR_ARG;
    if (!owner->store) return fail(ctx, SVO_ERR_STATE, "the owner has no node buffer (svo_nodes_alloc / svo_nodes_bind_device)");
    if (owner->device != ctx->device) return fail(ctx, SVO_ERR_ARG, "contexts on different devices cannot share a node buffer");
    if (ctx->store == owner->store) return SVO_OK;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    release_store(ctx);
    owner->store->refs++;
    adopt_store(ctx, owner->store);
    return SVO_OK;
}

int svo_nodes_invalidate(svo_ctx *ctx) {
    if (!ctx) return SVO_ERR_ARG;
    if (!ctx->store) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc / svo_nodes_bind_device not called");
    int rc = bind(ctx);
    if (rc) return rc;
    return note_write(ctx, true);
}

int svo_nodes_write(svo_ctx *ctx, size_t word_offset, const uint32_t *host_words, size_t n) {
    if (!ctx || (!host_words && n)) return SVO_ERR_ARG;
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc not called");
    if (word_offset > ctx->capacity || n > ctx->capacity - word_offset)
        return fail(ctx, SVO_ERR_ARG, "write past the node buffer capacity");
    int rc = bind(ctx);
    if (rc) return rc;
    if (n) {
        rc = order_after_last_write(ctx);  // a shared store: writes land in the order they were issued, whichever context issued them
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->nodes + word_offset, host_words, n * sizeof(uint32_t), hipMemcpyHostToDevice,
                                    ctx->stream));
    }
    return note_write(ctx, true);
}

int svo_nodes_scatter(svo_ctx *ctx, const uint32_t *indices, const uint32_t *host_words, size_t n) {
    if (!ctx || ((!indices || !host_words) && n)) return SVO_ERR_ARG;
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc not called");
    if (n > (1u << 28)) return fail(ctx, SVO_ERR_ARG, "too many words for one scatter");
    for (size_t i = 0; i < n; i++)
        if (indices[i] >= ctx->capacity) return fail(ctx, SVO_ERR_ARG, "scatter index past the node buffer capacity");
    int rc = bind(ctx);
    if (rc || n == 0) return rc;
    rc = order_after_last_write(ctx);  // a shared store: writes land in the order they were issued, whichever context issued them
    if (rc) return rc;
    rc = ensure_dev(ctx, &ctx->scatter_buf, &ctx->scatter_bytes, 2 * n * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *d_idx = (uint32_t *)ctx->scatter_buf, *d_val = d_idx + n;
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, indices, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_val, host_words, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, svo::launch_scatter(ctx->nodes, (uint32_t)ctx->capacity, d_idx, d_val, (uint32_t)n, ctx->stream));
    return note_write(ctx, true);
}

int svo_nodes_read(svo_ctx *ctx, size_t word_offset, uint32_t *host_words, size_t n) {
    if (!ctx || (!host_words && n)) return SVO_ERR_ARG;
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc not called");
    if (word_offset > ctx->capacity || n > ctx->capacity - word_offset)
        return fail(ctx, SVO_ERR_ARG, "read past the node buffer capacity");
    int rc = bind(ctx);
    if (rc) return rc;
    if (n) {
        rc = order_after_last_write(ctx);  // (a write another context of a shared store has enqueued is read, not raced)
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(host_words, ctx->nodes + word_offset, n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                    ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

int svo_nodes_device_ptr(svo_ctx *ctx, uint32_t **out, size_t *capacity_words) {
    if (!ctx || !out) return SVO_ERR_ARG;
    *out = ctx->nodes;
    if (capacity_words) *capacity_words = ctx->capacity;
    if (ctx->store) return note_write(ctx, false);  // the caller may write through the pointer
    return SVO_OK;
}

int svo_set_uniforms(svo_ctx *ctx, const svo_uniforms *u) {
    if (!ctx || !u) return SVO_ERR_ARG;
    ctx->uniforms = *u;
    ctx->have_uniforms = true;
    return SVO_OK;
}

int svo_render(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t tile_w,
               uint32_t tile_h, svo_hit *hits_out, uint32_t *rgba_out) {
    if (!ctx) return SVO_ERR_ARG;
    svo::WorkDesc work;
    int rc = make_rect_work(ctx, width, height, x0, y0, tile_w, tile_h, work);
    if (rc) return rc;
    return trace_common(ctx, work, nullptr, hits_out, rgba_out);
}

int svo_render_host(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t tile_w,
                    uint32_t tile_h, svo_hit *hits_out, uint32_t *rgba_out) {
    if (!ctx) return SVO_ERR_ARG;
    if (!hits_out && !rgba_out) return fail(ctx, SVO_ERR_ARG, "both hits_out and rgba_out are NULL");
    svo::WorkDesc work;
    int rc = make_rect_work(ctx, width, height, x0, y0, tile_w, tile_h, work);
    if (rc) return rc;
    size_t n = (size_t)tile_w * tile_h;
    rc = bind(ctx);
    if (rc) return rc;
    rc = ensure_stage(ctx, n * (sizeof(svo_hit) + sizeof(uint32_t)));
    if (rc) return rc;
    svo_hit *dh = (svo_hit *)ctx->stage;
    uint32_t *dc = (uint32_t *)((char *)ctx->stage + n * sizeof(svo_hit));
    rc = trace_common(ctx, work, nullptr, dh, rgba_out ? dc : nullptr);
    if (rc) return rc;
    if (hits_out) HIP_TRY(ctx, hipMemcpyAsync(hits_out, dh, n * sizeof(svo_hit), hipMemcpyDeviceToHost, ctx->stream));
    if (rgba_out) HIP_TRY(ctx, hipMemcpyAsync(rgba_out, dc, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    return svo_sync(ctx);
}

int svo_render_tiles(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                     uint32_t first_tile, uint32_t tile_stride, svo_hit *hits_out, uint32_t *rgba_out) {
    if (!ctx) return SVO_ERR_ARG;
    svo::WorkDesc work;
    int rc = make_tiles_work(ctx, width, height, tile_w, tile_h, first_tile, tile_stride, work);
    if (rc) return rc;
    return trace_common(ctx, work, nullptr, hits_out, rgba_out);
}

int svo_render_secondary(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t tile_w,
                         uint32_t tile_h, uint32_t n_secondary, svo_hit *primary_out, svo_hit *secondary_out) {
    if (!ctx) return SVO_ERR_ARG;
    svo::WorkDesc work;
    int rc = make_rect_work(ctx, width, height, x0, y0, tile_w, tile_h, work);
    if (rc) return rc;
    return trace_secondary(ctx, work, n_secondary, primary_out, secondary_out);
}

int svo_render_tiles_secondary(svo_ctx *ctx, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                               uint32_t first_tile, uint32_t tile_stride, uint32_t n_secondary, svo_hit *primary_out,
                               svo_hit *secondary_out) {
    if (!ctx) return SVO_ERR_ARG;
    svo::WorkDesc work;
    int rc = make_tiles_work(ctx, width, height, tile_w, tile_h, first_tile, tile_stride, work);
    if (rc) return rc;
    return trace_secondary(ctx, work, n_secondary, primary_out, secondary_out);
}

static int assemble_common(svo_ctx *ctx, const void *gathered, bool packed, uint32_t world, uint32_t n_pad, uint32_t width,
                           uint32_t height, uint32_t tile_w, uint32_t tile_h, svo_hit *frame_out) {
    if (!ctx || !gathered || !frame_out) return SVO_ERR_ARG;
    if (world == 0 || tile_w == 0 || tile_h == 0 || width % tile_w || height % tile_h)
        return fail(ctx, SVO_ERR_ARG, "frame must be a whole number of tiles");
    const uint64_t tiles = (uint64_t)(width / tile_w) * (height / tile_h);
    if ((uint64_t)n_pad * world < tiles) return fail(ctx, SVO_ERR_ARG, "gathered buffer holds fewer tiles than the frame");
    if ((uint64_t)width * height > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "frame too large");
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, svo::launch_assemble_tiles(gathered, packed, frame_out, world, n_pad, width, height, tile_w, tile_h, ctx->stream));
    return SVO_OK;
}

int svo_assemble_tiles(svo_ctx *ctx, const svo_hit *gathered, uint32_t world, uint32_t n_pad, uint32_t width, uint32_t height,
                       uint32_t tile_w, uint32_t tile_h, svo_hit *frame_out) {
    return assemble_common(ctx, gathered, false, world, n_pad, width, height, tile_w, tile_h, frame_out);
}

int svo_assemble_tiles_packed(svo_ctx *ctx, const uint32_t *gathered_wire, uint32_t world, uint32_t n_pad, uint32_t width,
                              uint32_t height, uint32_t tile_w, uint32_t tile_h, svo_hit *frame_out) {
    return assemble_common(ctx, gathered_wire, true, world, n_pad, width, height, tile_w, tile_h, frame_out);
}

int svo_assemble_tiles_rgba(svo_ctx *ctx, const uint32_t *gathered_rgba, uint32_t world, uint32_t n_pad, uint32_t width, uint32_t height,
                            uint32_t tile_w, uint32_t tile_h, uint32_t *rgba_frame_out) {
    if (!ctx || !gathered_rgba || !rgba_frame_out) return SVO_ERR_ARG;
    if (world == 0 || tile_w == 0 || tile_h == 0 || width % tile_w || height % tile_h)
        return fail(ctx, SVO_ERR_ARG, "frame must be a whole number of tiles");
    const uint64_t tiles = (uint64_t)(width / tile_w) * (height / tile_h);
    if ((uint64_t)n_pad * world < tiles) return fail(ctx, SVO_ERR_ARG, "gathered buffer holds fewer tiles than the frame");
    if ((uint64_t)width * height > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "frame too large");
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, svo::launch_assemble_tiles_rgba(gathered_rgba, rgba_frame_out, world, n_pad, width, height, tile_w, tile_h, ctx->stream));
    return SVO_OK;
}

int svo_pack_records(svo_ctx *ctx, const svo_hit *records, size_t n, uint32_t *wire_out) {
    if (!ctx || ((!records || !wire_out) && n)) return SVO_ERR_ARG;
    if (n > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "too many records for one call");
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, svo::launch_pack_records(records, wire_out, (uint32_t)n, ctx->stream));
    return SVO_OK;
}

int svo_trace_rays(svo_ctx *ctx, const float *rays, size_t n_rays, svo_hit *hits_out) {
    if (!ctx || (!rays && n_rays)) return SVO_ERR_ARG;
    if (n_rays > (1u << 26)) return fail(ctx, SVO_ERR_ARG, "too many rays for one call");
    svo::WorkDesc work{};
    work.mode = 2;
    work.n_items = (uint32_t)n_rays;
    work.bpr = work.bprect = work.tiles_x = 1;
    return trace_common(ctx, work, rays, hits_out, nullptr);
}

int svo_last_render_ms(svo_ctx *ctx, float *ms) {
    if (!ctx || !ms) return SVO_ERR_ARG;
    if (!ctx->ev_slots || !ctx->ev_count) return fail(ctx, SVO_ERR_STATE, "no timed launch (set SVO_OPT_TIMING first)");
    int rc = bind(ctx);
    if (rc) return rc;
    const size_t slot = (ctx->ev_count - 1) % ctx->ev_slots;
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2 * slot + 1]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev[2 * slot], ctx->ev[2 * slot + 1]));
    return SVO_OK;
}

int svo_timing_collect(svo_ctx *ctx, float *ms_out, size_t cap, size_t *n_out) {
    if (!ctx || !n_out || (!ms_out && cap)) return SVO_ERR_ARG;
    *n_out = 0;
    if (!ctx->ev_slots) return fail(ctx, SVO_ERR_STATE, "timing is off (set SVO_OPT_TIMING first)");
    int rc = bind(ctx);
    if (rc) return rc;
    size_t n = ctx->ev_count < ctx->ev_slots ? ctx->ev_count : ctx->ev_slots;
    if (n > cap) n = cap;
    const size_t first = ctx->ev_count - n;
    for (size_t i = 0; i < n; i++) {
        const size_t slot = (first + i) % ctx->ev_slots;
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2 * slot + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms_out[i], ctx->ev[2 * slot], ctx->ev[2 * slot + 1]));
    }
    *n_out = n;
    ctx->ev_count = 0;
    return SVO_OK;
}

int svo_diag_gather(svo_ctx *ctx, uint32_t stride_bytes, uint32_t n_loads) {
    if (!ctx || stride_bytes < 4 || (stride_bytes & 3)) return SVO_ERR_ARG;
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc not called");
    if ((uint64_t)n_loads * (stride_bytes / 4) > ctx->capacity) return fail(ctx, SVO_ERR_ARG, "pattern exceeds the node buffer");
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, svo::launch_diag_gather(ctx->nodes, (uint32_t)ctx->capacity, stride_bytes / 4, n_loads, ctx->status, ctx->stream));
    return SVO_OK;
}

int svo_diag_strip_classes(svo_ctx *ctx, uint8_t *host_out, size_t n_strips) {
    if (!ctx || (!host_out && n_strips)) return SVO_ERR_ARG;
    const svo_ctx::Sched &sc = ctx->sched[0];
    if (!sc.cls_now || !sc.order_filtered) return fail(ctx, SVO_ERR_STATE, "the last pixel frame was traced without a culling pass");
    if (n_strips > sc.cap) return fail(ctx, SVO_ERR_ARG, "more strips than the last frame had");
    int rc = bind(ctx);
    if (rc || n_strips == 0) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(host_out, sc.cls_now, n_strips, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

int svo_scan_dispatch(svo_ctx *ctx, uint32_t node_length) {
    if (!ctx) return SVO_ERR_ARG;
    if (!ctx->nodes) return fail(ctx, SVO_ERR_STATE, "svo_nodes_alloc not called");
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ctx->scan_sub) {
        // Compute::new: two lists of 1 024 000 words, zero-initialised (compute.rs:46-64)
        // (one allocation for both, so that a failure cannot leave half of the pair behind)
        uint32_t *lists = nullptr;
        HIP_TRY(ctx, hipMalloc((void **)&lists, 2 * kScanCapacity * sizeof(uint32_t)));
        ctx->scan_sub = lists;
        ctx->scan_unsub = lists + kScanCapacity;
        ctx->scan_capacity = kScanCapacity;
        HIP_TRY(ctx, hipMemsetAsync(ctx->scan_sub, 0, sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->scan_unsub, 0, sizeof(uint32_t), ctx->stream));
    }
    uint32_t n = node_length < ctx->capacity ? node_length : (uint32_t)ctx->capacity;
    rc = order_after_last_write(ctx);  // (the scan reads -- and with SCAN_CLEARS_COUNTERS writes -- the words of a possibly shared store)
    if (rc) return rc;
    HIP_TRY(ctx, svo::launch_scan(ctx->nodes, n, node_length, ctx->scan_sub, ctx->scan_unsub,
                                  (uint32_t)ctx->scan_capacity, ctx->scan_clears, ctx->stream));
    return SVO_OK;
}

int svo_scan_read(svo_ctx *ctx, uint32_t *sub, uint32_t *n_sub, uint32_t *unsub, uint32_t *n_unsub, size_t capacity) {
    if (!ctx || !sub || !unsub || !n_sub || !n_unsub || capacity < 1) return SVO_ERR_ARG;
    if (!ctx->scan_sub) return fail(ctx, SVO_ERR_STATE, "svo_scan_dispatch not called");
    int rc = bind(ctx);
    if (rc) return rc;
    uint32_t counts[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(&counts[0], ctx->scan_sub, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&counts[1], ctx->scan_unsub, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // adaptive.rs:22,86: len = min(count, MAX - 1)
    uint32_t ls = counts[0] < ctx->scan_capacity - 1 ? counts[0] : (uint32_t)ctx->scan_capacity - 1;
    uint32_t lu = counts[1] < ctx->scan_capacity - 1 ? counts[1] : (uint32_t)ctx->scan_capacity - 1;
    if (ls + 1 > capacity) ls = (uint32_t)capacity - 1;
    if (lu + 1 > capacity) lu = (uint32_t)capacity - 1;
    sub[0] = ls;
    unsub[0] = lu;
    if (ls) HIP_TRY(ctx, hipMemcpyAsync(sub + 1, ctx->scan_sub + 1, ls * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (lu) HIP_TRY(ctx, hipMemcpyAsync(unsub + 1, ctx->scan_unsub + 1, lu * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    // adaptive.rs:23,87: reset the atomic counter
    HIP_TRY(ctx, hipMemsetAsync(ctx->scan_sub, 0, sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->scan_unsub, 0, sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_sub = ls;
    *n_unsub = lu;
    return SVO_OK;
}

}  // extern "C"
