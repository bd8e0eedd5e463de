// svo_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the SVO ray-traversal path.
//
// What is computed is defined by the reference's WGSL (src/shader.wgsl:54-80,119-248 for the
// traversal, src/compute.wgsl:26-47 for the counter scan); HOW it is computed is MI355X-first:
//   * octree_ray's stepping arithmetic (shader.wgsl:227-235) is kept operation for operation
//     (strict IEEE f32: compiled with -ffp-contract=off, correctly rounded division/sqrt), because
//     which leaf a ray visits next depends on those roundings;
//   * find_voxel's root descent (shader.wgsl:130-171) is replaced, in the STACK variant, by an
//     exact integer formulation: node centres are dyadic rationals, so the chain of `pos > centre`
//     comparisons equals the bits of ceil(pos * 2^23) - 1 + 2^23 (floor(..) + 2^23 for the `>=`
//     form selected by misc_bool).  A step then re-descends only below the deepest ancestor shared
//     by the old and the new position (clz of the xor of the path codes), from a per-ray ancestor
//     stack kept in LDS; the top kTopLevels levels of the tree are folded into a 2 KiB table staged
//     in LDS; waves are persistent, generate rays 64 at a time into an LDS pool and refill finished
//     lanes from it (ballot + mbcnt compaction); strips of work are claimed longest-rays-first from
//     8 XCD-local lists built from the previous frames' step counts.
//   * node words are read with buffer loads (hardware range check: an index past the buffer
//     reads 0, the semantics the oracle defines for out-of-range words).
#include <hip/hip_runtime.h>

#include "svo_device.h"
#include "svo_trace_fn.h"

// Build-time switches kept for same-box A/B builds (-DSVO_...=n); the defaults are what measured fastest.
#ifndef SVO_WALK_STOP
#define SVO_WALK_STOP 1
#endif
#ifndef SVO_TOP_IN_LDS   // 0: the top table is read from global memory (2 KiB less LDS per workgroup)
#define SVO_TOP_IN_LDS 1
#endif
#ifndef SVO_WALK_TRIPS   // 1: the walk's depth limit is a count of its iterations (wave-uniform) instead of a test of every lane's level
#define SVO_WALK_TRIPS 1
#endif
#ifndef SVO_WALK_ASM   // 1: the walk's loop hand-written (all but the timeline build): what follows a word's arrival is a compare, one scalar
#define SVO_WALK_ASM 1  // instruction on the execute mask and the branch
#endif
#ifndef SVO_WALK_LUT   // 1: the walk takes the child indices of six levels at a time from one register built with three LDS look-ups (a 64-entry
#define SVO_WALK_LUT 2  // bit-spreading table) instead of three bit-field extracts and two shift-ors per level (trees up to depth 16; the counting kernels too: SVO_WALK_LUT_CNT)
#endif
#ifndef SVO_WALK_LUT_CNT   // the same walk in the counting instantiations (hit counters live), the saturation notes as one carry bit per level
#define SVO_WALK_LUT_CNT 1
#endif
#ifndef SVO_DBG_LIGHT   // 1: the timeline instantiation records only its time stamps (start, loop entry, generations, dry, end) and runs the
#define SVO_DBG_LIGHT 0  // product's hand-written walk at the product's occupancy: a progress curve that is not distorted by the phase clocks
#endif
#ifndef SVO_CAM_SCALAR
#define SVO_CAM_SCALAR 0
#endif
#ifndef SVO_WAVES_PER_SIMD_CNT   // of the counting instantiations (hit counters live)
#define SVO_WAVES_PER_SIMD_CNT 6
#endif
#ifndef SVO_WAVES_PER_SIMD   // of the default instantiation (static tree, trees up to depth 16)
#define SVO_WAVES_PER_SIMD 7
#endif

namespace svo {


// ---------------------------------------------------------------------------------------------
// Variant RESTART: the reference's algorithm shape -- float-compare descent from the root on
// every step (shader.wgsl:130-171 inside :213-245), one ray per lane, grid-stride over items.
// ---------------------------------------------------------------------------------------------
// Hit-counter side effect of find_voxel (shader.wgsl:157-161): `if (primary && cnt < 15 && !pause_adaptive)
// n.data[p] = value + 1`.  The reference's plain read-modify-write races between rays; here the increment is a
// compare-and-swap that stops at 15, so the counters after a frame are min(15, old + visits) whatever the
// order (the oracle's oracle_count_frame).  Words saturate after 15 visits and are then only read.
// Returns the counter as this call leaves or finds it.
__device__ __forceinline__ uint32_t count_add(uint32_t *nodes, uint32_t p, uint32_t word, uint32_t n) {
    while ((word & 15u) < 15u) {
        const uint32_t add = min(n, 15u - (word & 15u));
        const uint32_t seen = atomicCAS(&nodes[p], word, word + add);
        if (seen == word) return (word & 15u) + add;
        word = seen;
    }
    return 15u;
}

// Called by all lanes that are at the same level of their descent.  Rays of a wave are coherent: near the root all 64
// lanes visit the same word, and 64 lanes racing their compare-and-swaps on one word cost 64 atomics per increment
// (a 1080p frame took 400 ms that way).  Lanes with the same word are therefore grouped and the first lane of a group
// adds the size of the group; all groups issue their compare-and-swap together, after the grouping.  Neighbouring lanes
// trace neighbouring pixels, so equal addresses mostly sit in runs of adjacent lanes: with few runs the groups are
// exact (one ballot per distinct address merges runs that share a word), with many runs every run is its own group
// (no loop; two runs on one word then cost two atomics, which is still correct).
// visit_groups: the number of visits this lane reports for word p -- the size of its group on the group's first lane, 0 on
// every other lane (and on lanes whose word is saturated or outside the buffer).
// RUNS_ONLY: every run is its own group whatever their number (the STACK kernel's queue merges equal words again before it
// is flushed, so the exact grouping here would be paid twice).
template <bool RUNS_ONLY = false>
__device__ __forceinline__ uint32_t visit_groups(uint32_t n_words, uint32_t p, uint32_t word) {
    const bool need = p < n_words && (word & 15u) < 15u;
    const uint64_t needmask = __ballot(need);
    if (!needmask) return 0u;
    const uint32_t lane = __lane_id();
    if (RUNS_ONLY) {
        // (round 5, half the instructions of the general form below: lanes with nothing to report carry an index no word has, so "the
        // lane below reports another word" is one compare; a run ends at the next head or idle lane above -- or at lane 64: the mask
        // of those, shifted down to this lane, always has a lowest bit)
        const uint32_t pm = need ? p : 0xFFFFFFFFu;
        const uint32_t under = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)pm, 0x138, 0xF, 0xF, false);  // (DPP wave_shr:1; lane 0 keeps the -1)
        const bool head = need && under != pm;
        const uint64_t stop = ((((uint64_t)__ballot(head) | ~needmask) >> 1) | (1ull << 63)) >> lane;  // bit j: the run ends below lane + 1 + j
        return head ? (uint32_t)__builtin_ctzll(stop) + 1u : 0u;
    }
    // p of lane - 1 (DPP wave_shr:1: a shuffle through the LDS pipe costs its latency); only looked at when lane - 1 is in needmask
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x138, 0xF, 0xF, false);
    const bool head = need && (lane == 0u || !((needmask >> (lane - 1u)) & 1ull) || prev != p);
    const uint64_t headmask = __ballot(head);
    uint32_t n = 0u;  // visits this lane reports: the size of its group on the group's first lane, 0 elsewhere
    if (!RUNS_ONLY && __popcll(headmask) <= 8) {
        uint64_t todo = headmask;
        while (todo) {
            const uint32_t leader = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
            const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)leader);
            const uint64_t grp = __ballot(need && p == p0);
            if (lane == leader) n = (uint32_t)__popcll(grp);
            todo &= ~grp;
        }
    } else {
        const uint64_t above = lane == 63u ? 0ull : (~0ull << (lane + 1u));
        const uint64_t stop = (headmask | ~needmask) & above;  // where this lane's run ends
        const uint32_t end = stop ? (uint32_t)__ffsll((unsigned long long)stop) - 1u : 64u;
        if (head) n = end - lane;
    }
    return n;
}

__device__ __forceinline__ void count_visit(uint32_t *nodes, uint32_t n_words, uint32_t p, uint32_t word) {
    const uint32_t n = visit_groups(n_words, p, word);
    if (n) count_add(nodes, p, word, n);
}

// One ray, the reference's way.  `hits[out]` receives the record, `aux` (optional) t_current of the last step, and the
// return value is that t_current as well (the fused-shadow post pass builds the shadow ray of a deferred primary ray from it).
__device__ __forceinline__ float trace_ray_restart(const TraceArgs &a, rsrc_t rs, bool misc_bool, bool counter_hits, const RayIn &r,
                                                   svo_hit *hits, uint32_t out, float *aux) {
    const bool count = a.count_nodes != nullptr;
    float pos[3], dir[3], dist;
    if (!ray_enter(r, pos, dir, dist)) {
        write_hit(hits, out, 0u, 0.0f, 0u, 0u, 0u, 0u);
        if (aux) aux[out] = 0.0f;
        return 0.0f;
    }
    float rs0 = sign_w(dir[0]), rs1 = sign_w(dir[1]), rs2 = sign_w(dir[2]);
    float vp0 = pos[0], vp1 = pos[1], vp2 = pos[2];
    float n0 = truncf(pos[0] * 1.000001f), n1 = truncf(pos[1] * 1.000001f), n2 = truncf(pos[2] * 1.000001f);
    uint32_t steps = 0;
    float t_current = 0.0f;
    for (;;) {
        // find_voxel
        uint32_t node_index = 0, depth = 0, p, word;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        bool overflow = false;
        for (;;) {
            depth += 1;
            uint32_t bx, by, bz;
            if (misc_bool) { bx = vp0 >= c0; by = vp1 >= c1; bz = vp2 >= c2; }
            else           { bx = vp0 >  c0; by = vp1 >  c1; bz = vp2 >  c2; }
            float d = (float)(1u << depth);
            c0 = c0 + ((float)bx * 2.0f - 1.0f) / d;
            c1 = c1 + ((float)by * 2.0f - 1.0f) / d;
            c2 = c2 + ((float)bz * 2.0f - 1.0f) / d;
            p = node_index + bx * 4u + by * 2u + bz;
            word = load_word(rs, p);
            // (a stale cached word only costs the group one failed compare-and-swap, which returns the fresh one;
            // device-scope loads measured 5 % slower)
            if (count) count_visit(a.count_nodes, a.n_words, p, word);
            uint32_t tn = word >> 4;
            if (tn >= kVoxelOffset) break;
            if (depth >= kMaxDescent) { overflow = true; break; }
            node_index = tn;
        }
        uint32_t nc = normal_code(n0) | (normal_code(n1) << 2) | (normal_code(n2) << 4);
        if (overflow) {
            write_hit(hits, out, 0xFF000000u, dist + t_current, steps, 100u, 1u, nc);
            if (aux) aux[out] = t_current;
            return t_current;
        }
        bool solid = counter_hits ? ((word & 15u) > 0u) : (((word >> 4) - kVoxelOffset) > 0u);
        if (solid) {
            write_hit(hits, out, p, dist + t_current, steps, depth, 1u, nc);
            if (aux) aux[out] = t_current;
            return t_current;
        }
        float voxel_size = 2.0f / (float)(1u << depth);
        float t0 = (c0 - pos[0] + rs0 * voxel_size / 2.0f) / dir[0];
        float t1 = (c1 - pos[1] + rs1 * voxel_size / 2.0f) / dir[1];
        float t2 = (c2 - pos[2] + rs2 * voxel_size / 2.0f) / dir[2];
        float m0 = (t0 <= fmin_w(t1, t2)) ? 1.0f : 0.0f;
        float m1 = (t1 <= fmin_w(t2, t0)) ? 1.0f : 0.0f;
        float m2 = (t2 <= fmin_w(t0, t1)) ? 1.0f : 0.0f;
        n0 = m0 * -rs0; n1 = m1 * -rs1; n2 = m2 * -rs2;
        t_current = fmin_w(fmin_w(t0, t1), t2);
        vp0 = pos[0] + dir[0] * t_current - n0 * 0.000002f;
        vp1 = pos[1] + dir[1] * t_current - n1 * 0.000002f;
        vp2 = pos[2] + dir[2] * t_current - n2 * 0.000002f;
        if (!in_bounds(vp0, vp1, vp2)) {
            write_hit(hits, out, 0x20202000u, dist + t_current, steps, depth, 0u, 0u);
            if (aux) aux[out] = t_current;
            return t_current;
        }
        steps += 1;
        if (steps > 100u) {
            write_hit(hits, out, 0xFF000000u, dist + t_current, steps, 100u, 1u,
                      normal_code(n0) | (normal_code(n1) << 2) | (normal_code(n2) << 4));
            if (aux) aux[out] = t_current;
            return t_current;
        }
    }
}


__device__ __forceinline__ float trace_one_restart(const TraceArgs &a, rsrc_t rs, bool misc_bool, bool counter_hits, uint32_t q) {
    Item it = decode_item(a.work, q);
    if (!it.valid) return 0.0f;
    if (a.work.mode == 2 && a.skip && a.skip[q]) return 0.0f;  // no ray: the producer wrote the record
    return trace_ray_restart(a, rs, misc_bool, counter_hits, item_ray(a, it), a.hits, it.out, a.aux_t);
}

// list == nullptr: every item of the work description; else the items list[1 .. list[0]] (rays the
// STACK variant deferred because its fast arithmetic does not cover them).
__global__ __launch_bounds__(256) void trace_restart_kernel(TraceArgs a, const uint32_t *list) {
    const rsrc_t rs = make_rsrc(a.nodes, a.n_words);
    const bool misc_bool = (a.u.flags & SVO_F_MISC_BOOL) != 0;
    const bool counter_hits = (a.u.flags & SVO_F_PAUSE_ADAPTIVE) && (a.u.flags & SVO_F_SHOW_HITS);
    const uint32_t n = list ? list[0] : a.work.n_items;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        trace_one_restart(a, rs, misc_bool, counter_hits, list ? list[1u + i] : i);
}

// ---------------------------------------------------------------------------------------------
// Top table: fold the first kTopLevels levels into one word per level-K cell.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_top_table_kernel(const uint32_t *nodes, uint32_t n_words,
                                                              uint32_t *table, int top_levels) {
    const rsrc_t rs = make_rsrc(nodes, n_words);
    uint32_t cell = blockIdx.x * 256u + threadIdx.x;
    if (cell >= (1u << (3 * top_levels))) return;
    uint32_t cx = (cell >> (2 * top_levels)) & ((1u << top_levels) - 1u);
    uint32_t cy = (cell >> top_levels) & ((1u << top_levels) - 1u);
    uint32_t cz = cell & ((1u << top_levels) - 1u);
    // entry = level << 27 | index of the child group to read at that level: level K+1 below an interior cell,
    // or the (shallower) level whose word is the leaf covering the whole cell
    uint32_t node_index = 0, entry = ((uint32_t)top_levels + 1u) << 27;
    for (int lvl = 1; lvl <= top_levels; lvl++) {
        int sh = top_levels - lvl;
        uint32_t child = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        uint32_t tn = load_word(rs, node_index + child) >> 4;
        if (tn >= kVoxelOffset) {
            entry = ((uint32_t)lvl << 27) | (node_index & 0x07FFFFFFu);
            break;
        }
        node_index = tn;
        entry = (((uint32_t)top_levels + 1u) << 27) | (tn & 0x07FFFFFFu);
    }
    table[cell] = entry;
    // Behind the table: for every level-k cell (k = 1 .. K-1) the child group to read at level k+1, or all ones below a
    // leaf.  Only the hit-counting instantiation of the STACK kernel uses them: it needs the address of EVERY word on a
    // ray's path, also above the levels the table skips.
    uint32_t at = (uint32_t)kTopEntries;
    for (int k = 1; k < top_levels; k++) {
        if (cell < (1u << (3 * k))) {
            const uint32_t kx = (cell >> (2 * k)) & ((1u << k) - 1u), ky = (cell >> k) & ((1u << k) - 1u), kz = cell & ((1u << k) - 1u);
            uint32_t g = 0;
            for (int lvl = 1; lvl <= k && g != 0xFFFFFFFFu; lvl++) {
                const int sh = k - lvl;
                const uint32_t child = (((kx >> sh) & 1u) << 2) | (((ky >> sh) & 1u) << 1) | ((kz >> sh) & 1u);
                const uint32_t tn = load_word(rs, g + child) >> 4;
                g = tn >= kVoxelOffset ? 0xFFFFFFFFu : tn;
            }
            table[at + cell] = g;
        }
        at += 1u << (3 * k);
    }
}

// ---------------------------------------------------------------------------------------------
// Variant STACK (see file header).
// ---------------------------------------------------------------------------------------------

// The traversal runs in GRID UNITS: positions and directions are pre-multiplied by 2^(D-1) = 2^22 (the path-code
// scale).  Scaling by a power of two commutes with every IEEE rounding involved (no overflow/underflow on
// a clean ray), so  A = (C - P) + H,  t = A / Dr,  G = (P + Dr * t) +- K  are exactly 2^22 times the
// reference's  a,  the same t,  and  2^22 * voxel_pos  -- and G is what the path codes need.
//
// Round 4: the step is written for the way gfx950 issues vector instructions (tools/issue_rate.hip, profiles/r04_issue_rate*.log):
// plain f32 add / mul / fma (also with the clamp modifier) take 2 cycles of a SIMD and run BESIDE the other group -- compares,
// selects, min/max, conversions, bit-field and shift-or forms, every three-operand integer instruction: 4 cycles each -- in which the
// kernel of rounds 1-3 did five sixths of its work.  So everything the DDA step needs is now derived in f32, exactly:
//   * a path code U (D = 23 bits) lives in a register as the bits of the float 2^23 + U, which IS 0x4B000000 | U: the walk
//     extracts child bits from it, the step subtracts 2^23 and has U as a float;
//   * the leaf centre: the code with its low s bits replaced by 1 0 .. 0 (one and-or) is, read as a float, 2^23 + centre + 2^22;
//   * the tie mask  t_i == min  selects the nudge k or 0 (compare + select; as  1 - clamp(clamp((t_i - min) 2^126) 2^24)  it needs no
//     compare, same time), the nudge  k_i * copysign(1, dir)  is one fma with it, the mask bits a sum of the k_i;
//   * in_bounds is a product of six clamps (a coordinate is a float: below 2^22 it is at most 2^22 - 1/4, so
//     clamp(4 (2^22 - g)) is exactly 1 or 0), the step limit one more factor, and ONE compare decides who goes on;
//   * steps, mask, lane state and the leaf's level (sh) are floats; the state changes by multiplication (svo_trace_fn.h).
// What stays in the other group: floor() of the three new coordinates, the xor / or3 / count-leading-zeros that finds the
// restart level, three and-ors, min3, the tie compares and selects -- 22 instead of 75 per round.
// And the walk is written for where its instructions stand: a wave that has waited for a word re-enters the SIMD's
// rotation among seven, so every instruction between the word's arrival and the next load costs about one rotation (2 % of the
// frame each), those issued while a load is in flight nearly nothing.  Its loop is hand-written: compare, s_andn2 exec, branch,
// shift, shift-add, load; child index of the next level, push, level counter and the depth limit (an iteration count) in the shadow.
// The kernel's first argument, re-read from the kernarg segment where it is needed: ray generation and the record writes use
// some sixty scalar values (the uniforms, the work description, five pointers) that the hot loop never touches; read through
// the by-value parameter they would be loaded once and stay in scalar registers for the whole kernel, which has none to spare --
// the compiler then parks wave-uniform values and constants in VECTOR registers, which is what bounds the occupancy.  The empty
// asm makes the pointer opaque, so the loads stay where they are written (scalar loads: the pointer is uniform).
__device__ __forceinline__ const TraceArgs &fresh_args() {
    auto p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const TraceArgs *)p;
}

// LDS of a trace workgroup, in words: ONE definition for the kernel and for the launch's lds_bytes (VERDICT r4 / ADVICE r4: the two
// must not drift apart).  The ancestor stacks are the LAST region and nothing may be added behind them: the walk does not check the
// level it has reached, so in a tree deeper than the caller declared a lane pushes rows that do not exist -- beyond the workgroup's
// allocation, where gfx950 drops the write and reads return 0 (tools/experiments/lds_oob_probe.hip, tests/test_lds_oob_gpu.py) --
// instead of into another region's data.  A new region goes IN FRONT of `stacks`.
template <int BLOCK, int NS, int K, bool CNT>
struct StackLds {
    static constexpr int top = SVO_TOP_IN_LDS != 0 ? (1 << (3 * K)) : 0;                              // top table
    static constexpr int lut = top;                                                                   // the walk's bit-spreading table, 64 words
    static constexpr int pools = lut + 64;                                                            // [BLOCK / 64][kPoolWords][64]
    static constexpr int counting = pools + (BLOCK / 64) * (kPoolWords * 64);                         // CNT: queues, saturation tags, per-cell flags
    static constexpr int stacks = counting + (CNT ? (BLOCK / 64) * kCountQueue + kSatTags + (1 << (3 * K)) / 8 : 0);  // [NS][BLOCK], last
    static constexpr int total = stacks + NS * BLOCK;
};
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "svo_kernels.hip is written for gfx950: the walk relies on its handling of LDS accesses beyond the allocation and on v_rcp_f32 + one Newton step being the IEEE reciprocal (tools/rcptest_gpu.hip)"
#endif

// CNT: hit counters live (adaptive mode, shader.wgsl:157-161), see step 3a in the loop.
template <int BLOCK, int NS, int K, bool GE, bool DBG, bool CNT, bool SHD>
__global__ __launch_bounds__(BLOCK, NS > 12 ? 4 : (CNT ? SVO_WAVES_PER_SIMD_CNT : ((SHD || (DBG && SVO_DBG_LIGHT == 0)) ? 6 : SVO_WAVES_PER_SIMD))) void trace_stack_kernel(TraceArgs a, uint32_t strip_items,
                                                               uint32_t *work_counter, uint32_t *defer) {
    constexpr bool DBGH = DBG && SVO_DBG_LIGHT == 0;  // the timeline build's phase clocks and tallies (the light build keeps the time stamps only)
    constexpr int D = kPathBits;
    constexpr int SBASE = K + 2;       // first level kept on the LDS stack
    constexpr int SMAX = K + 1 + NS;   // last level kept on the LDS stack = deepest level resolved
    static_assert(SMAX <= D - 1, "stack deeper than the path codes");
    constexpr int TBL = 1 << (3 * K);
    constexpr bool kTopInLds = SVO_TOP_IN_LDS != 0;
    using Lds = StackLds<BLOCK, NS, K, CNT>;  // (the layout: one definition for kernel and launch)
    constexpr int kLutWords = 64;
    constexpr int LOFF = Lds::lut;
    constexpr int TOFF = Lds::pools;  // LDS words in front of the ray pools
    constexpr bool kLut = SVO_WALK_LUT != 0 && NS <= 12 && (!CNT || SVO_WALK_LUT_CNT != 0);
    constexpr int SOFF = Lds::stacks;
    static_assert(Lds::total == SOFF + NS * BLOCK, "the ancestor stacks are the last LDS region (see StackLds)");
    static_assert(D == 23, "2^D + code must be an f32 with unit spacing");
    constexpr float kScale = (float)(1 << (D - 1));  // 2^22: grid units per unit of the cube
    constexpr float kInvScale = 1.0f / kScale;
    constexpr float kMagic = (float)(1 << D);        // 2^23: as_uint(kMagic + U) = 0x4B000000 | U for an integer 0 <= U < 2^23
    constexpr float kNudge = 0.000002f * kScale;     // |voxel_pos nudge| (shader.wgsl:235) in grid units (exact: a power-of-two multiple)
    // SHD: direction of every shadow ray, -normalize(sun_dir) (wave-uniform, kept in scalar registers)
    float shadow_d0 = 0.0f, shadow_d1 = 0.0f, shadow_d2 = 0.0f;
    float sDr0 = 1.0f, sDr1 = 1.0f, sDr2 = 1.0f, sY0 = 1.0f, sY1 = 1.0f, sY2 = 1.0f;  // what ray_enter and the pick-up make of it
    float sS0 = 1.0f, sS1 = 1.0f, sS2 = 1.0f;
    if (SHD) {
        auto uni = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
        float sun[3];
        sun_direction(a.u, sun);
        shadow_d0 = uni(-sun[0]); shadow_d1 = uni(-sun[1]); shadow_d2 = uni(-sun[2]);
        sDr0 = uni((shadow_d0 + ((shadow_d0 == 0.0f) ? 1.0f : 0.0f) * 0.000001f) * kScale);  // octree_ray's bias (ray_enter), then grid units
        sDr1 = uni((shadow_d1 + ((shadow_d1 == 0.0f) ? 1.0f : 0.0f) * 0.000001f) * kScale);
        sDr2 = uni((shadow_d2 + ((shadow_d2 == 0.0f) ? 1.0f : 0.0f) * 0.000001f) * kScale);
        sY0 = uni(1.0f / sDr0); sY1 = uni(1.0f / sDr1); sY2 = uni(1.0f / sDr2);
        sS0 = uni(copysign_bits(1.0f, sDr0)); sS1 = uni(copysign_bits(1.0f, sDr1)); sS2 = uni(copysign_bits(1.0f, sDr2));
    }
    extern __shared__ uint32_t lds[];
    const uint32_t *tbl = kTopInLds ? lds : a.top_table;  // TBL entries
    uint32_t *pool_all = lds + TOFF;  // [BLOCK / 64][kPoolWords][64]; behind them the counting queues and tables (CNT), then the stacks [NS][BLOCK]
    // CNT: level-1 / level-2 cell -> child group (kTopAuxEntries words behind the top table): read from global memory, where the
    // 288 bytes stay in the L1 -- in LDS they cost the sixth workgroup per CU (the allocation granule)
    const uint32_t *aux = a.top_table + TBL;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t *pool = pool_all + (tid >> 6) * (kPoolWords * 64);
    const rsrc_t rs = make_rsrc(a.nodes, a.n_words);
    // (the same descriptor as four scalar words, for the hand-written loop of the walk: raw buffer, stride 0, size in bytes)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 rs_asm = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)a.nodes),
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)a.nodes >> 32)) & 0xFFFFu,
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.n_words << 2)), 0x00020000u};
    uint32_t c_refill = 0, c_desc = 0, c_step = 0, c_gen = 0, dbg_desc_iters = 0, dbg_desc_lanes = 0, dbg_desc_rounds = 0, dbg_desc_start = 0;
    // CNT: the visits of a round are not added to the counters on the spot -- a compare-and-swap is executed at the memory
    // side (the L2s of the 8 XCDs are not coherent with each other), its answer takes 0.4 - 1.3 us under this kernel's load,
    // and a round that waits for one answer per level took 8 us instead of 2.  Final counters are min(15, old + visits) in
    // whatever order the visits arrive, so every wave queues its (word, visits) pairs in LDS and adds them 64 at a time,
    // all lanes in parallel: one wait per 64 - 192 pairs instead of one per level and round.
    // What a lane learns only there -- that a word has reached 15 -- goes into a direct-mapped table of word indices shared
    // by the workgroup (exact: an entry is the index itself); step 3a looks a word up before it queues a visit, so the
    // hundreds of later visits of a word near the root cost one LDS read each, on whichever lane and ray they happen.
    uint32_t *cq = pool_all + (BLOCK / 64) * (kPoolWords * 64) + (tid >> 6) * kCountQueue;  // (CNT only: the launch allocates both)
    uint32_t *sat_tags = pool_all + (BLOCK / 64) * (kPoolWords * 64) + (BLOCK / 64) * kCountQueue;
    auto sat_slot = [](uint32_t p) -> uint32_t { return (p ^ (p >> 9)) & (uint32_t)(kSatTags - 1); };  // (no multiply: quarter rate)
    // ... and what the table said about the words of levels 1..K -- which the walk never reads -- is kept per level-K cell (4 bits
    // each: bit l = the level-l word on the way to this cell is saturated): every ray picked up and every ray that crosses a
    // level-K boundary would otherwise look those words up one level per loop iteration (half of the loop's iterations), and a
    // direct-mapped table forgets
    uint32_t *top_sat = sat_tags + kSatTags;  // TBL / 8 words
    if (CNT) {
        for (uint32_t i = tid; i < (uint32_t)kSatTags; i += BLOCK) sat_tags[i] = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < (uint32_t)(TBL / 8); i += BLOCK) top_sat[i] = 0u;
        __syncthreads();
    }
    uint32_t cq_n = 0u;
    // A flush takes the whole queue, two records per lane.  Four in ten of 64 consecutive records name a word that another
    // of them names too (the same leaf or ancestor visited in consecutive rounds, or by lanes that were not neighbours): as
    // two compare-and-swaps of one wave on one word, the second is bound to fail and costs the wave another round trip.
    // So the records are merged first, in a hash table of the wave that lives in the queue's own 128 words (the records
    // are in registers by then; more LDS would cost the fifth workgroup per CU): the first record of a word claims an entry,
    // the others add their visits to it (saturating at 15, which is all a counter can take) and drop out; a record that
    // finds neither its word nor a free entry within four probes goes alone -- correct, just not merged.  The device-scope
    // loads of both halves travel together, and so do their compare-and-swaps.
    static_assert(kCountQueue == 128, "the flush holds the queue in two registers per lane");
    auto cq_flush = [&]() {
        constexpr uint32_t kFree = 0xFFFFFFFFu, kWord = 0x07FFFFFFu;
        const bool v0 = lane < cq_n, v1 = 64u + lane < cq_n;
        const uint32_t rec0 = v0 ? cq[lane] : 0u, rec1 = v1 ? cq[64u + lane] : 0u;
        cq[lane] = kFree;
        cq[64u + lane] = kFree;
        // returns the visits to report for this record (0: another record of the wave reports them)
        auto merge_in = [&](bool valid, uint32_t rec, uint32_t &slot) -> bool {  // true: this lane reports the word
            const uint32_t p = rec & kWord;
            slot = (p * 2654435761u) >> 25;
            uint32_t role = valid ? 0u : 3u;  // 0: alone so far, 1: claimed an entry, 2: its word has an entry, 3: no record
            uint32_t seen = 0u;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (role == 0u) {
                    seen = atomicCAS(&cq[slot], kFree, rec);
                    if (seen == kFree) role = 1u;
                    else if ((seen & kWord) == p) role = 2u;
                    else slot = (slot + 1u) & 127u;
                }
            }
            if (role == 2u) {
                for (;;) {
                    const uint32_t sum = min(15u, (seen >> 27) + (rec >> 27));
                    const uint32_t want = p | (sum << 27);
                    if (want == seen) break;
                    const uint32_t got = atomicCAS(&cq[slot], seen, want);
                    if (got == seen) break;
                    seen = got;
                }
            }
            if (role == 0u) slot = 0xFFFFFFFFu;  // alone: its own visits
            return role <= 1u;
        };
        uint32_t slot0, slot1;
        const bool i0 = merge_in(v0, rec0, slot0), i1 = merge_in(v1, rec1, slot1);
        // (every lane's merges are done before any lane reads an entry back: the retry loops above diverge -- ADVICE r3)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t p0 = rec0 & kWord, p1 = rec1 & kWord;
        const uint32_t n0 = (i0 && slot0 != 0xFFFFFFFFu) ? cq[slot0] >> 27 : rec0 >> 27;
        const uint32_t n1 = (i1 && slot1 != 0xFFFFFFFFu) ? cq[slot1] >> 27 : rec1 >> 27;
        // (device-scope loads: the copy in this XCD's L2 is as old as its last miss, and a stale word costs a failed
        // compare-and-swap -- the memory side's atomic rate, 24 G/s for the whole device, is what this mode runs against)
        const uint32_t f0 = i0 ? __hip_atomic_load(a.count_nodes + p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 15u;
        const uint32_t f1 = i1 ? __hip_atomic_load(a.count_nodes + p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 15u;
        const bool t0 = (f0 & 15u) < 15u, t1 = (f1 & 15u) < 15u;
        const uint32_t a0 = min(n0, 15u - (f0 & 15u)), a1 = min(n1, 15u - (f1 & 15u));
        uint32_t s0 = f0, s1 = f1;
        if (t0) s0 = atomicCAS(&a.count_nodes[p0], f0, f0 + a0);
        if (t1) s1 = atomicCAS(&a.count_nodes[p1], f1, f1 + a1);
        uint32_t c0 = !t0 ? 15u : (s0 == f0 ? (f0 & 15u) + a0 : 16u), c1 = !t1 ? 15u : (s1 == f1 ? (f1 & 15u) + a1 : 16u);
        if (c0 == 16u) c0 = count_add(a.count_nodes, p0, s0, n0);  // (somebody else's visit came in between)
        if (c1 == 16u) c1 = count_add(a.count_nodes, p1, s1, n1);
        if (i0 && c0 == 15u) sat_tags[sat_slot(p0)] = p0;
        if (i1 && c1 == 15u) sat_tags[sat_slot(p1)] = p1;
        cq_n = 0u;
    };
    // all lanes call this (uniform control flow); `mine` = the lane has a visit of word p (value `word`) to report
    auto cq_push = [&](bool mine, uint32_t p, uint32_t word) {
        const uint32_t n = visit_groups<true>(a.n_words, p, mine ? word : 15u);
        const uint64_t m = __ballot(n != 0u);
        if (!m) return;
#ifndef SVO_COUNT_FLUSH_AT   // (A/B: flush once the queue holds more than this many records; at most kCountQueue - 64, a push adds up to 64)
#define SVO_COUNT_FLUSH_AT (kCountQueue - 64)
#endif
        if (cq_n > (uint32_t)(SVO_COUNT_FLUSH_AT)) {
            if (DBGH) dbg_desc_rounds += 1u;  // (CNT: slot 14 = queue flushes, slot 13 = cycles in them)
            const uint64_t c_f0 = DBGH ? __builtin_amdgcn_s_memtime() : 0ull;
            cq_flush();
            if (DBGH && lane == 0u) dbg_desc_lanes += (uint32_t)(__builtin_amdgcn_s_memtime() - c_f0);
        }
        if (n) cq[cq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = p | (min(n, 15u) << 27);
        cq_n += (uint32_t)__popcll(m);
    };

    if (kTopInLds) {
        for (uint32_t i = tid; i < (uint32_t)TBL; i += BLOCK) lds[i] = a.top_table[i];
    }
    if (kLut && tid < (uint32_t)kLutWords) {  // entry v: bit j of v moved to bit 3 j, as a float (the three axes are combined by two fmas)
        uint32_t sp3 = 0u;
#pragma unroll
        for (int j = 0; j < 6; j++) sp3 |= ((tid >> j) & 1u) << (3 * j);
        lds[LOFF + tid] = __float_as_uint((float)sp3);
    }
    if (kTopInLds || kLut) __syncthreads();

    const uint32_t n_items = a.work.n_items;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + tid) >> 6);

    // wave-uniform work cursor: a wave claims strips [next, strip_end) of strip_items items.
    // Dynamic mode (work_counter): the strips are cut into kShards contiguous screen regions, each with its
    // own claim counter on its own cache line.  A workgroup starts in region blockIdx % 8 -- workgroups are
    // dealt round-robin over the 8 XCDs, so the waves sharing an L2 work on one screen region and walk it in
    // order -- and moves on to the next region when its own is exhausted (work stealing keeps the tail short;
    // a single counter would serialise at ~88 claims/us).  Static mode: strips dealt round-robin to waves.
    constexpr uint32_t kShards = 8, kShardStride = (uint32_t)kCounterStride;  // (svo_device.h: 64 KB + 128 B apart)
    // Every list has kSubs claim counters, each handing out every kSubs-th of the list's entries (counter j: entries
    // reserved + j, reserved + j + kSubs, ...: every counter walks the whole list, longest strips first); a wave starts
    // with counter (blockIdx / kShards) % kSubs of its list, takes the list's counters in turn, and moves on to the next
    // list when they have run out.  Atomics are executed at the memory side, one address at a time (11 ns each): on a frame of short
    // rays the waves finish their strips in step, 768 claims arrive at one counter together, and the last of them
    // waited 8 us -- half of such a frame was spent waiting for claims (wave_timeline: 90 k of 187 k cycles per wave on
    // config 2).
    constexpr uint32_t kSubs = 8;
    static_assert(kShards * kSubs == 64, "the stealing probe reads one counter per lane");
    const uint32_t n_strips = (n_items + strip_items - 1) / strip_items;
    // a.order (optional): strip numbers sorted by the cost they had in the previous frame, longest rays first
    // (LPT scheduling: a ray is a serial chain of up to 101 rounds, so the long ones must start early or the
    // whole chip waits for them at the end); layout and rationale at strip_order_kernel.
    const uint32_t *order = a.order;  // 8 list lengths, then 8 lists of a.order_cap strip numbers
    // The first entries of every list are RESERVED, one per wave that starts on that list: a wave's first strip is
    // the entry with its own rank, without an atomic (6144 waves claiming at t = 0 queue up for ~10 us on 8
    // counters otherwise); the counters hand out what comes after the reserved part.
    // A claim is issued as soon as a strip has been turned into pool rays, but its answer is only looked at when the pool
    // is empty again: the atomic's round trip overlaps the traversal instead of stalling the wave.
    // (Wave-uniform state, kept small -- the kernel sits at the limit of its scalar registers, and what does not fit there
    // ends up in scratch memory, inside the hot loop: `next` == 0xFFFFFFFE says that a claim is in flight, 0xFFFFFFFF that
    // the frame has no more strips for this wave.  The stealing code exists once, in the refill step.)
    uint32_t home = (blockIdx.x % kShards) * kSubs + (blockIdx.x / kShards) % kSubs;  // the claim counter (list * kSubs + counter) this wave draws from
    uint32_t pend = 0;                // lane 0: what the atomic returned
    uint32_t next, strip_end;
    auto entry_of = [=](uint32_t sh, uint32_t k) -> uint32_t {  // strip behind entry k of list sh (0xFFFFFFFF: past the end)
        if (order) return k < order[sh] ? order[kShards + sh * a.order_cap + k] : 0xFFFFFFFFu;
        // no schedule yet (the first frame of a layout): runs of 16 consecutive strips dealt round-robin to the 8 lists, so that the
        // lists are equally long in work whatever part of the screen is expensive (8 contiguous regions left whole lists of sky
        // idle early and their waves stealing, one probe and one synchronous claim per strip)
        const uint32_t cand = (((k >> 4) * kShards + sh) << 4) | (k & 15u);
        return cand < n_strips ? cand : 0xFFFFFFFFu;
    };
    {
        const uint32_t my_rank = (blockIdx.x / kShards) * (uint32_t)(BLOCK / 64) + __builtin_amdgcn_readfirstlane(tid >> 6);  // among the waves that start on my list
        const uint32_t s0 = __builtin_amdgcn_readfirstlane(entry_of(blockIdx.x % kShards, my_rank));
        if (s0 != 0xFFFFFFFFu) {
            next = s0 * strip_items;
            strip_end = min(next + strip_items, n_items);
        } else {  // more waves than entries on this list: as if a claim had come back empty (the refill step looks elsewhere)
            pend = 0x00FFFFFFu;
            next = strip_end = 0xFFFFFFFEu;
        }
    }
    uint32_t pool_n = 0, pool_i = 0;  // wave-uniform: rays waiting in the pool, index of the first
    // optional per-wave timeline (diagnostic builds of the host set a.debug): start, queue-dry, end in 10 ns ticks
    uint32_t t_begin = 0, t_dry = 0;
    uint32_t n_rounds = 0, dbg_active = 0, dbg_iters = 0, dbg_refills = 0, dbg_gens = 0;
    // phase clocks of the timeline build (shader cycles, s_memtime): refill / descent / step, and the descent's shape
    uint64_t c_mark = 0;
    if (DBG) t_begin = (uint32_t)__builtin_amdgcn_s_memrealtime();

    // per-lane ray state
    float stf = ST_IDLE;              // see ST_* (svo_trace_fn.h); negative: a shadow ray (SHD)
    uint32_t out = 0;                 // bits 0..25 output index, 26..31 entry normal code
    float P0 = 0, P1 = 0, P2 = 0, Dr0 = 1, Dr1 = 1, Dr2 = 1, Y0 = 1, Y1 = 1, Y2 = 1;
    float S0 = 1, S1 = 1, S2 = 1;     // copysign(1, dir): r_sign of shader.wgsl:211
    float dist = 0.0f, tcur = 0.0f;
    float stepsf = 0.0f;              // steps taken (shader.wgsl:240), an integer 0 .. 101
    float nmf = 0.0f;                 // the last step's mask (axes whose t_i was the minimum) as a number 0 .. 7, times kNudge
    uint32_t mu0 = __float_as_uint(kMagic), mu1 = mu0, mu2 = mu0;  // path codes, as the bits of kMagic + code
    // sh: at a leaf, the bit of the path codes that selects the child at the leaf's level, D - L (the leaf's cell is 2^sh grid
    // units wide); before a walk, one more than the bit of the first level to read.  nidx: the child group that level lives in.
    // (sh lives in its register as the bits of the float 2^23 + sh, like the path codes: the walk's decrement is then an f32 add -- which
    // issues beside the bit-field extracts for nothing, tools/issue_rate.hip -- and the instructions that take sh as a bit position or a
    // shift amount read the low five bits only.  sh_of() where the number itself is wanted.)
    uint32_t sh = __float_as_uint(kMagic + 1.0f), nidx = 0;
    auto sh_of = [](uint32_t shbits) -> uint32_t { return shbits & 31u; };
    constexpr uint32_t kShMagic = 0x4B000000u;  // bits of 2^23
    // the lane's column of the ancestor stacks as ONE per-lane constant: byte address of row -(kRb + SBASE), so that row r - SBASE is
    // colbase + (r + kRb) * 1024; every other per-lane LDS address is written as this one plus a constant (the counting instantiation
    // has no register for a second one)
    constexpr uint32_t kRb = (uint32_t)(31 - D);
    constexpr uint32_t kRow = (uint32_t)(BLOCK * 4);
    const uint32_t colbase = ((uint32_t)SOFF + threadIdx.x) * 4u - (kRb + (uint32_t)SBASE) * kRow;
    uint32_t sp = colbase + (kRb + (uint32_t)SBASE) * kRow;  // LDS slot (byte offset) one row BELOW the walk's next push
    uint32_t leaf_off = 0, leaf_w = 0;  // current leaf: byte offset of its word, and the word
    auto sabs = [](float x) -> float { return SHD ? __builtin_fabsf(x) : x; };  // (only the SHD instantiation has negative states)
    constexpr bool kWalkStops = SVO_WALK_STOP != 0 && !CNT;
    constexpr bool kWalkTrips = SVO_WALK_TRIPS != 0;
    constexpr bool kWalkAsm = SVO_WALK_ASM != 0 && kWalkTrips && !DBGH;
    uint32_t satm = 0;                // CNT: bit l = the word of level l on the lane's current path is known to be saturated (step 3a)

    // (re)start a descent: from the LDS top table when the restart level r is at most K+1 (the table also
    // knows leaves that cover a whole level-K cell), else from the lane's ancestor stack.  One LDS read.
    // LDS by plain byte address: through `lds` -- a symbol whose address the linker fills in -- every access costs a v_add of that
    // address, which is 0 (the kernel has no static LDS, so the dynamic array starts at the bottom: checked, at compile time in effect)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    auto lds_at = [&](uint32_t byte_offset) -> lds_u32 & { return *(lds_u32 *)(uintptr_t)byte_offset; };
    if (__builtin_amdgcn_groupstaticsize() != 0u) __builtin_trap();
    // rb = r + (31 - D): the step has r as a count of leading zeros, and every use of it here takes a constant anyway
    auto restart_at_rb = [&](uint32_t rb) {
        const bool top = rb <= (uint32_t)(K + 1) + kRb;
        sp = colbase + rb * kRow;  // stack[r - SBASE][lane], in bytes
        if (__ballot(top)) {  // wave-uniform: most rounds no lane crosses a level-(K+1) boundary
            const uint32_t cell = (__builtin_amdgcn_ubfe(mu0, D - K, K) << (2 * K)) | (__builtin_amdgcn_ubfe(mu1, D - K, K) << K) |
                                  __builtin_amdgcn_ubfe(mu2, D - K, K);
            const uint32_t e = kTopInLds ? lds_at(top ? cell * 4u : sp) : (top ? a.top_table[cell] : lds_at(sp));
            sh = (kShMagic + (uint32_t)(D + 1)) - (top ? (e >> 27) : rb - kRb);
            nidx = e & 0x07FFFFFFu;
            // a walk from the table starts at level K+1 or above: its first push (if any) is the group of level K+2, row 0
            sp = top ? colbase + (kRb + (uint32_t)SBASE - 1u) * kRow : sp;
        } else {  // a stack entry is the child group itself (< 2^27): nothing to unpack; the walk's first push goes to the row of level r + 1
            sh = (kShMagic + (uint32_t)(D + 1) + kRb) - rb;
            nidx = lds_at(sp);
        }
    };
    auto restart_at = [&](uint32_t r) { restart_at_rb(r + kRb); };

    // write the record of a finished ray (deferred to the next refill so that it runs for many lanes at once)
    auto flush_record = [&](bool may_continue) {
        const TraceArgs &a = fresh_args();  // (shadows the parameter: see fresh_args)
        // (the leaf word the ray ended on is still in the lane's register: an interior word = the walk gave up at level SMAX)
        const uint32_t steps = (uint32_t)stepsf;
        // a ray that ends without stopping in its leaf left the cube -- its count is at most 100 -- or ran into the step limit,
        // which is the one way to count 101 (shader.wgsl:237-244)
        const bool too_deep = leaf_w < (kVoxelOffset << 4) || (kWalkTrips && __uint_as_float(sh) < kMagic + (float)(D - SMAX)), solid = (leaf_w >> 4) != kVoxelOffset, inb = steps == 101u;
        const bool stop_here = too_deep || solid;
        if (too_deep) atomicOr(a.status, 1u);  // reported by svo_sync
        const uint32_t L = (uint32_t)D - sh_of(sh), nm = (uint32_t)__builtin_fmaf(nmf, 1.0f / kNudge, 0.5f);  // (nmf is 0 .. 7 times k, up to rounding)
        uint32_t c0n = (Dr0 > 0.0f) ? 2u : 1u, c1n = (Dr1 > 0.0f) ? 2u : 1u, c2n = (Dr2 > 0.0f) ? 2u : 1u;
        uint32_t ncode = ((nm & 1u) ? c0n : 0u) | ((nm & 2u) ? (c1n << 2) : 0u) | ((nm & 4u) ? (c2n << 4) : 0u);
        if (steps == 0u) ncode = out >> 26;               // no step taken: the entry normal
        if (!stop_here && !inb) ncode = 0u;               // left the cube: the miss record carries no normal
        const uint32_t leaf_index = leaf_off >> 2;
        const uint32_t value = too_deep ? 0xFF000000u : (solid ? leaf_index : (!inb ? 0x20202000u : 0xFF000000u));
        const uint32_t depth = (too_deep || (!solid && inb)) ? 100u : L;
        const uint32_t hit = (stop_here || inb) ? 1u : 0u;
        const bool was_shadow = SHD && stf < 0.0f;
        write_hit(was_shadow ? a.shadow_hits : a.hits, out & 0x03FFFFFFu, value, dist + tcur, steps, depth, hit, ncode);
        if (a.aux_t && !was_shadow) a.aux_t[out & 0x03FFFFFFu] = tcur;
        if (SHD && !was_shadow) {
            // fs_main's shadow ray (shader.wgsl:275-280), on the lane that found the hit: same arithmetic as
            // secondary_gen_kernel (the primary ray's entry point and direction are P, Dr scaled by 2^23, exactly), and the
            // descent restarts from the ancestors this lane already holds instead of from the root.
            bool go = false;
            float pos2[3], dir2[3], d2 = 0.0f;
            if (may_continue && hit) {
                const float pw[3] = {P0 * kInvScale, P1 * kInvScale, P2 * kInvScale};
                const float dw[3] = {Dr0 * kInvScale, Dr1 * kInvScale, Dr2 * kInvScale};
                float org[3], nrm[3];
                secondary_origin(pw, dw, steps, ncode, tcur, org, nrm);
                if (in_bounds(org[0], org[1], org[2])) {  // ray_enter's common case: the ray starts where it is
                    pos2[0] = org[0]; pos2[1] = org[1]; pos2[2] = org[2];
                    go = true;
                } else {  // a hit on the cube's surface with the normal pointing out: the whole of ray_enter
                    const RayIn r = {org[0], org[1], org[2], shadow_d0, shadow_d1, shadow_d2};
                    go = ray_enter(r, pos2, dir2, d2);
                }
                if (go && !(fabsf(pos2[0]) <= 2.0f && fabsf(pos2[1]) <= 2.0f && fabsf(pos2[2]) <= 2.0f && fabsf(d2) <= 1.0e30f)) {
                    // cannot happen for an origin next to a hit inside the cube (the host checked the direction); reported by svo_sync
                    atomicOr(a.status, 2u);
                    go = false;
                }
            }
            if (go) {
                const uint32_t L_old = L;
                P0 = pos2[0] * kScale; P1 = pos2[1] * kScale; P2 = pos2[2] * kScale;
                Dr0 = sDr0; Dr1 = sDr1; Dr2 = sDr2;  // the same for every shadow ray
                Y0 = sY0; Y1 = sY1; Y2 = sY2;
                S0 = sS0; S1 = sS1; S2 = sS2;
                dist = d2;
                tcur = 0.0f;
                stepsf = 0.0f;
                nmf = 0.0f;
                const uint32_t ecode = normal_code(truncf(pos2[0] * 1.000001f)) | (normal_code(truncf(pos2[1] * 1.000001f)) << 2) |
                                       (normal_code(truncf(pos2[2] * 1.000001f)) << 4);
                out = (out & 0x03FFFFFFu) | (ecode << 26);
                const uint32_t j0 = entry_magic<GE>(P0), j1 = entry_magic<GE>(P1), j2 = entry_magic<GE>(P2);
                const uint32_t diff = (mu0 ^ j0) | (mu1 ^ j1) | (mu2 ^ j2);  // (the exponent bits cancel)
                const uint32_t c = (uint32_t)__clz((int)((diff << (32 - D)) | (1u << (31 - D))));  // levels the old and the new path share
                mu0 = j0; mu1 = j1; mu2 = j2;
                const uint32_t r = max(min(min(c + 1u, L_old), (uint32_t)SMAX), 1u);
                stf = -ST_DESC;
                if (CNT) satm &= (1u << r) - 1u;
                restart_at(r);
                return;
            }
            reinterpret_cast<uint4 *>(a.shadow_hits)[out & 0x03FFFFFFu] = make_uint4(0u, 0u, 0u, 0u);  // no shadow ray: the record of a ray that never enters
        }
        stf = ST_IDLE;
    };

    // the one-level walk: one dependent word per level below (sh, nidx), ancestors pushed on the lane's LDS stack
    auto descend = [&]() {
            uint32_t off, w, key;
            // sh -= 1, then the child the position selects at the level of that bit: x << 2 | y << 1 | z.  Three-operand forms the
            // compiler does not pick by itself, in ONE asm statement (the compiler pads every inline-asm statement with an s_nop, and
            // keeping the decrement inside saves a copy of the counter)
            // kLut (round 5): those five instructions of the 4-cycle group, once per level and lane, were a third of the walk's issue cycles.
            // The child indices of SIX levels are taken at once instead: the six code bits of every axis below the restart level index a
            // 64-entry LDS table that spreads them to every third bit (as a float), m = 4 sx + 2 sy + sz + 2^23 is two fmas and an add of
            // the 2-cycle group, and a level's child index is ONE bit-field extract of m at a wave-uniform position (the lanes of a walk
            // start at different levels but advance together).  Six levels later (one walk in a hundred) the look-up is repeated.
            // First code bit wanted = sh - 1 >= D - SMAX = 7, so the table offset (code >> (sh - 8)) & 0xFC needs no left shift.
            uint32_t m = 0u;
            int32_t pos = 0;  // wave-uniform: bit position in m of the child index worked out last
            auto lut_fetch = [&](float first_bit_minus_7) {  // (as the bits of 2^23 + that number: a shift amount is read from the low five bits)
                const uint32_t k = __float_as_uint(first_bit_minus_7);
                const float fx = __uint_as_float(lds_at((uint32_t)LOFF * 4u + ((mu0 >> (k & 31u)) & 0xFCu)));
                const float fy = __uint_as_float(lds_at((uint32_t)LOFF * 4u + ((mu1 >> (k & 31u)) & 0xFCu)));
                const float fz = __uint_as_float(lds_at((uint32_t)LOFF * 4u + ((mu2 >> (k & 31u)) & 0xFCu)));
                m = __float_as_uint(__builtin_fmaf(fx, 4.0f, __builtin_fmaf(fy, 2.0f, fz + kMagic)));
            };
            constexpr bool kLutLate = SVO_WALK_LUT == 2;  // the first word's child the old way, the look-up in the shadow of the first load
            if (kLut && !kLutLate) {
                lut_fetch(__uint_as_float(sh) - 8.0f);
                pos = 18;
            }
            bool first_child = true;
            auto child_below = [&]() -> uint32_t {
                if (kLut && !(kLutLate && first_child)) {
                    sh = __float_as_uint(__uint_as_float(sh) - 1.0f);
                    pos -= 3;
                    if (pos < 0) {
                        lut_fetch(__uint_as_float(sh) - 7.0f);  // (sh is the bit of the level whose child is wanted now)
                        pos = 15;
                    }
                    return __builtin_amdgcn_ubfe(m, (uint32_t)pos, 3u);
                }
                uint32_t c, tmp;
                asm("v_add_f32 %2, -1.0, %2\n\t"
                    "v_bfe_u32 %0, %3, %2, 1\n\t"
                    "v_bfe_u32 %1, %4, %2, 1\n\t"
                    "v_lshl_or_b32 %0, %0, 1, %1\n\t"
                    "v_bfe_u32 %1, %5, %2, 1\n\t"
                    "v_lshl_or_b32 %0, %0, 1, %1"
                    : "=&v"(c), "=&v"(tmp), "+v"(sh)
                    : "v"(mu0), "v"(mu1), "v"(mu2));
                return c;
            };
            auto tally = [&]() {
                if (DBGH) {
                    const uint64_t in_loop = __ballot(true);
                    if (lane == (uint32_t)__ffsll((unsigned long long)in_loop) - 1u) {  // (tallies summed over lanes at the end)
                        if (!CNT) dbg_desc_iters += 1u;
                        if (!CNT) dbg_desc_lanes += (uint32_t)__popcll(in_loop);
                    }
                }
            };
            // The loop is rotated twice.  (1) A word is read, then looked at: a leaf, or level SMAX, ends the walk BEFORE anything is
            // pushed for it.  (2) While a word travels the child index of the NEXT level is worked out -- it depends on the position
            // only -- so that between a word's arrival and the next load stand a shift and an add instead of eight instructions.
            uint32_t c = child_below();
            first_child = false;
            bool first_word = true;
            auto read_word = [&]() {
                tally();
                off = (nidx + c) << 2;
                w = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
                if (kLut && kLutLate && first_word) {
                    __builtin_amdgcn_sched_barrier(0);  // (the look-up's address arithmetic behind the load, not in front of it)
                    lut_fetch(__uint_as_float(sh) - 7.0f);  // (sh is the bit of the word in flight: the first of the six)
                    asm volatile("" : "+v"(m));  // (here, under the load -- the compiler would sink it behind the wait for the word)
                    pos = 15;
                }
                first_word = false;
                // (a counter never goes down within a frame; the hand-written loop of the look-up-table walk takes its notes itself, see satb)
                if (CNT && !(kWalkAsm && kLut)) satm |= ((w & 15u) == 15u ? 1u : 0u) << (((uint32_t)D - sh_of(sh)) & 31u);
                c = child_below();  // (sh now points one level below the word in flight)
                if (kLut) asm volatile("" : "+v"(c));  // (before the wait for the word, not behind it)
                // sign bit: a leaf (word >= VOXEL_OFFSET << 4), or level SMAX reached (deeper trees are refused)
                key = kWalkTrips ? w : (w | (sh_of(sh) - (uint32_t)(D - SMAX)));
            };
            read_word();
            // kWalkTrips: a walk reads at most NS words below its first one in a tree of the declared depth -- the last of them, at level
            // SMAX, a leaf.  In a deeper tree the count (one scalar subtraction per iteration instead of two vector instructions per lane)
            // ends the walk; lanes that started low have gone past level SMAX by then: their pushes fell off the end of the LDS
            // allocation (see SOFF).  A word found below level SMAX is handed on as an interior word, which is what the walk of rounds
            // 1-3 stopped at: the ray ends there with the "too deep" record and svo_sync reports the frame.
            uint32_t trips = (uint32_t)NS;
            uint32_t satb = 0u;  // CNT, look-up-table walk: one saturation bit per word the hand-written loop has gone through, the last one lowest
            if (kWalkAsm) {
                // The same loop as below, instruction for instruction, but for its control: the compiler keeps the lanes that have
                // left in a second mask and folds the count's flag into it -- four scalar instructions between a word's arrival
                // and the next load; every instruction there costs the walk its latency (profiles/r04_walk_critical_path_ab.log).
                if (CNT && kLut && (int32_t)w >= 0) {
                    // (hit counters live, look-up-table walk: the loop of the kLut branch below plus the saturation note of the word that has
                    // just arrived -- three instructions: its counter bits, a compare, and satb = 2 satb + carry.  After n iterations bit
                    // n - 1 - i of satb belongs to the word of iteration i; the walk turns them into level bits of satm when it is over.)
                    uint32_t tmp, cnt4;
                    uint64_t saved;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n"
                        "1:\n\t"
                        "v_and_b32 %[t2], 15, %[w]\n\t"
                        "v_lshrrev_b32 %[nidx], 4, %[w]\n\t"
                        "v_add_lshl_u32 %[off], %[nidx], %[c], 2\n\t"
                        "buffer_load_dword %[w], %[off], %[rs], 0 offen\n\t"
                        "s_add_i32 %[trips], %[trips], -1\n\t"
                        "v_cmp_eq_u32 vcc, 15, %[t2]\n\t"
                        "v_addc_co_u32 %[satb], vcc, %[satb], %[satb], vcc\n\t"
                        "v_add_u32 %[sp], %[row], %[sp]\n\t"
                        "v_add_f32 %[sh], -1.0, %[sh]\n\t"
                        "s_add_i32 %[pos], %[pos], -3\n\t"
                        "s_cmp_lt_i32 %[pos], 0\n\t"
                        "s_cbranch_scc1 3f\n"
                        "4:\n\t"
                        "v_bfe_u32 %[c], %[m], %[pos], 3\n\t"
                        "ds_write_b32 %[sp], %[nidx]\n\t"
                        "s_cmp_eq_u32 %[trips], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "v_cmp_gt_i32 vcc, 0, %[w]\n\t"
                        "s_andn2_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execnz 1b\n\t"
                        "s_branch 2f\n"
                        "3:\n\t"
                        "v_add_f32 %[m], 0xc0e00000, %[sh]\n\t"
                        "v_lshrrev_b32 %[c], %[m], %[m0]\n\t"
                        "v_lshrrev_b32 %[t], %[m], %[m1]\n\t"
                        "v_lshrrev_b32 %[m], %[m], %[m2]\n\t"
                        "v_and_b32 %[c], 0xfc, %[c]\n\t"
                        "v_and_b32 %[t], 0xfc, %[t]\n\t"
                        "v_and_b32 %[m], 0xfc, %[m]\n\t"
                        "ds_read_b32 %[c], %[c] offset:%[lut]\n\t"
                        "ds_read_b32 %[t], %[t] offset:%[lut]\n\t"
                        "ds_read_b32 %[m], %[m] offset:%[lut]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_add_f32 %[m], 0x4b000000, %[m]\n\t"
                        "v_fmac_f32 %[m], 2.0, %[t]\n\t"
                        "v_fmac_f32 %[m], 4.0, %[c]\n\t"
                        "s_mov_b32 %[pos], 15\n\t"
                        "s_branch 4b\n"
                        "2:\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [w] "+v"(w), [nidx] "+v"(nidx), [off] "+v"(off), [c] "+v"(c), [t] "=&v"(tmp), [t2] "=&v"(cnt4), [sp] "+v"(sp), [sh] "+v"(sh),
                          [m] "+v"(m), [satb] "+v"(satb), [trips] "+s"(trips), [pos] "+s"(pos), [sv] "=&s"(saved)
                        : [m0] "v"(mu0), [m1] "v"(mu1), [m2] "v"(mu2), [rs] "s"(rs_asm), [row] "s"((uint32_t)(BLOCK * 4)), [lut] "n"(LOFF * 4)
                        : "vcc", "scc", "memory");
                } else if (CNT && !kLut && (int32_t)w >= 0) {
                    // (hit counters live: an interior word whose counter has reached 15 is noted in satm, bit = its level, while the
                    // next word travels -- only the counter bits are taken before the load is issued; the word that ends the walk
                    // needs no note: step 3a looks at the leaf's own counter, and bit L is cleared by every restart before it is read)
                    uint32_t tmp, cnt4;
                    uint64_t saved;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n"
                        "1:\n\t"
                        "v_and_b32 %[t2], 15, %[w]\n\t"
                        "v_lshrrev_b32 %[nidx], 4, %[w]\n\t"
                        "v_add_lshl_u32 %[off], %[nidx], %[c], 2\n\t"
                        "buffer_load_dword %[w], %[off], %[rs], 0 offen\n\t"
                        "s_add_i32 %[trips], %[trips], -1\n\t"
                        "v_cmp_eq_u32 vcc, 15, %[t2]\n\t"
                        "v_sub_u32 %[t], %[dm1], %[sh]\n\t"
                        "v_cndmask_b32_e64 %[t2], 0, 1, vcc\n\t"
                        "v_lshlrev_b32 %[t2], %[t], %[t2]\n\t"
                        "v_or_b32 %[satm], %[satm], %[t2]\n\t"
                        "v_add_u32 %[sp], %[row], %[sp]\n\t"
                        "v_add_f32 %[sh], -1.0, %[sh]\n\t"
                        "v_bfe_u32 %[c], %[m0], %[sh], 1\n\t"
                        "v_bfe_u32 %[t], %[m1], %[sh], 1\n\t"
                        "v_lshl_or_b32 %[c], %[c], 1, %[t]\n\t"
                        "v_bfe_u32 %[t], %[m2], %[sh], 1\n\t"
                        "v_lshl_or_b32 %[c], %[c], 1, %[t]\n\t"
                        "ds_write_b32 %[sp], %[nidx]\n\t"
                        "s_cmp_eq_u32 %[trips], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "v_cmp_gt_i32 vcc, 0, %[w]\n\t"
                        "s_andn2_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execnz 1b\n"
                        "2:\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [w] "+v"(w), [nidx] "+v"(nidx), [off] "+v"(off), [c] "+v"(c), [t] "=&v"(tmp), [t2] "=&v"(cnt4), [sp] "+v"(sp), [sh] "+v"(sh),
                          [satm] "+v"(satm), [trips] "+s"(trips), [sv] "=&s"(saved)
                        : [m0] "v"(mu0), [m1] "v"(mu1), [m2] "v"(mu2), [rs] "s"(rs_asm), [row] "s"((uint32_t)(BLOCK * 4)), [dm1] "s"((uint32_t)(D - 1))
                        : "vcc", "scc", "memory");
                } else if (!CNT && kLut && (int32_t)w >= 0) {
                    // (the loop of the next branch with the child index taken from m: one bit-field extract at the scalar position pos;
                    // label 3 is the table look-up for the next six levels, taken when pos runs out -- in the shadow of the load like the rest)
                    uint32_t tmp;
                    uint64_t saved;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n"
                        "1:\n\t"
                        "v_lshrrev_b32 %[nidx], 4, %[w]\n\t"
                        "v_add_lshl_u32 %[off], %[nidx], %[c], 2\n\t"
                        "buffer_load_dword %[w], %[off], %[rs], 0 offen\n\t"
                        "s_add_i32 %[trips], %[trips], -1\n\t"
                        "v_add_u32 %[sp], %[row], %[sp]\n\t"
                        "v_add_f32 %[sh], -1.0, %[sh]\n\t"
                        "s_add_i32 %[pos], %[pos], -3\n\t"
                        "s_cmp_lt_i32 %[pos], 0\n\t"
                        "s_cbranch_scc1 3f\n"
                        "4:\n\t"
                        "v_bfe_u32 %[c], %[m], %[pos], 3\n\t"
                        "ds_write_b32 %[sp], %[nidx]\n\t"
                        "s_cmp_eq_u32 %[trips], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "v_cmp_gt_i32 vcc, 0, %[w]\n\t"
                        "s_andn2_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execnz 1b\n\t"
                        "s_branch 2f\n"
                        "3:\n\t"
                        "v_add_f32 %[m], 0xc0e00000, %[sh]\n\t"
                        "v_lshrrev_b32 %[c], %[m], %[m0]\n\t"
                        "v_lshrrev_b32 %[t], %[m], %[m1]\n\t"
                        "v_lshrrev_b32 %[m], %[m], %[m2]\n\t"
                        "v_and_b32 %[c], 0xfc, %[c]\n\t"
                        "v_and_b32 %[t], 0xfc, %[t]\n\t"
                        "v_and_b32 %[m], 0xfc, %[m]\n\t"
                        "ds_read_b32 %[c], %[c] offset:%[lut]\n\t"
                        "ds_read_b32 %[t], %[t] offset:%[lut]\n\t"
                        "ds_read_b32 %[m], %[m] offset:%[lut]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_add_f32 %[m], 0x4b000000, %[m]\n\t"
                        "v_fmac_f32 %[m], 2.0, %[t]\n\t"
                        "v_fmac_f32 %[m], 4.0, %[c]\n\t"
                        "s_mov_b32 %[pos], 15\n\t"
                        "s_branch 4b\n"
                        "2:\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [w] "+v"(w), [nidx] "+v"(nidx), [off] "+v"(off), [c] "+v"(c), [t] "=&v"(tmp), [sp] "+v"(sp), [sh] "+v"(sh), [m] "+v"(m),
                          [trips] "+s"(trips), [pos] "+s"(pos), [sv] "=&s"(saved)
                        : [m0] "v"(mu0), [m1] "v"(mu1), [m2] "v"(mu2), [rs] "s"(rs_asm), [row] "s"((uint32_t)(BLOCK * 4)), [lut] "n"(LOFF * 4)
                        : "vcc", "scc", "memory");
                } else if (!CNT && (int32_t)w >= 0) {
                    uint32_t tmp;
                    uint64_t saved;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n"
                        "1:\n\t"
                        "v_lshrrev_b32 %[nidx], 4, %[w]\n\t"
                        "v_add_lshl_u32 %[off], %[nidx], %[c], 2\n\t"
                        "buffer_load_dword %[w], %[off], %[rs], 0 offen\n\t"
                        "s_add_i32 %[trips], %[trips], -1\n\t"
                        "v_add_u32 %[sp], %[row], %[sp]\n\t"
                        "v_add_f32 %[sh], -1.0, %[sh]\n\t"
                        "v_bfe_u32 %[c], %[m0], %[sh], 1\n\t"
                        "v_bfe_u32 %[t], %[m1], %[sh], 1\n\t"
                        "v_lshl_or_b32 %[c], %[c], 1, %[t]\n\t"
                        "v_bfe_u32 %[t], %[m2], %[sh], 1\n\t"
                        "v_lshl_or_b32 %[c], %[c], 1, %[t]\n\t"
                        "ds_write_b32 %[sp], %[nidx]\n\t"
                        "s_cmp_eq_u32 %[trips], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "v_cmp_gt_i32 vcc, 0, %[w]\n\t"
                        "s_andn2_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execnz 1b\n"
                        "2:\n\t"
                        "s_waitcnt vmcnt(0)\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [w] "+v"(w), [nidx] "+v"(nidx), [off] "+v"(off), [c] "+v"(c), [t] "=&v"(tmp), [sp] "+v"(sp), [sh] "+v"(sh),
                          [trips] "+s"(trips), [sv] "=&s"(saved)
                        : [m0] "v"(mu0), [m1] "v"(mu1), [m2] "v"(mu2), [rs] "s"(rs_asm), [row] "s"((uint32_t)(BLOCK * 4))
                        : "vcc", "scc", "memory");
                }
            } else
            while ((int32_t)key >= 0 && (!kWalkTrips || trips != 0u)) {
                if (kWalkTrips) trips -= 1u;
                nidx = w >> 4;
                sp += (uint32_t)(BLOCK * 4);
                if (kWalkTrips) asm("" : "+v"(sp));  // (keeps the compiler from turning sp into base + a scalar: one more v_add per iteration)
                lds_at(sp) = nidx;
#if defined(SVO_DUMMY_WALK_S) || defined(SVO_DUMMY_WALK_F)
                {
#ifdef SVO_DUMMY_WALK_S
                    for (int i_ = 0; i_ < SVO_DUMMY_WALK_S; i_++) { uint32_t t_; asm volatile("v_bfe_u32 %0, %1, 1, 30" : "=v"(t_) : "v"(mu0)); }
#endif
#ifdef SVO_DUMMY_WALK_F
                    for (int i_ = 0; i_ < SVO_DUMMY_WALK_F; i_++) { uint32_t t_; asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(t_) : "v"(mu0)); }
#endif
                }
#endif
                read_word();
            }
            sh = __float_as_uint(__uint_as_float(sh) + 1.0f);  // back to the leaf's own bit
            // (the loop's notes: bit j of satb = the word j + 1 levels above the leaf, i.e. level L - 1 - j with L = D - sh; 32 - L = 9 + sh)
            if (CNT && kWalkAsm && kLut) satm |= __builtin_bitreverse32(satb) >> ((sh + (uint32_t)(32 - D)) & 31u);
            // (a word found below level SMAX -- the tree is deeper than declared -- ends the ray like the interior word the walk of rounds 1-3
            // stopped at; compared as floats: sh may have passed 0)
            const bool below = kWalkTrips && __uint_as_float(sh) < kMagic + (float)(D - SMAX);
            if (kWalkTrips && !kWalkStops) w = below ? 0u : w;
            leaf_off = off;
            leaf_w = w;
            // ST_DESC -> ST_LEAF; or straight to ST_PENDING when the ray ends in this leaf (anything but an empty leaf: a solid one, or
            // an interior word at level SMAX) -- with live hit counters the step section has to see the lane once more (step 3a)
            if (kWalkStops) stf *= ((w >> 4) != kVoxelOffset || below) ? 0.25f : 0.5f;
            else stf *= 0.5f;
            };

    // Camera shortcut.  Every primary ray of a camera that stands INSIDE the cube starts at the same point, hence in the
    // same leaf, with the same ancestors: the first walk of a ray -- the longest it ever makes, root to leaf, while the
    // other lanes of the wave wait for it -- is the same for all of them.  The wave makes it once, here, all lanes alike,
    // and keeps what it found in one register (lane l holds word l: the stack rows, then the leaf's word offset, the
    // word and its level); a lane that picks up a ray copies it into its stack column instead of walking, and takes its
    // first step in the same round.
    // (Not with live hit counters: the copied leaf word would carry the counter bits it had when the wave started, and
    // every ray's compare-and-swap on that one word would fail once -- two million serialised atomics on one address.)
    constexpr bool CAM = !CNT;
    // (kCamScalar: the camera's walk in eighteen scalar registers instead of the lanes of one vector register -- a build-time trial:
    // the compiler parks the array in vector registers and spills, 52 bytes of scratch, so it stays off)
    constexpr bool kCamScalar = SVO_CAM_SCALAR != 0 && NS <= 12;
    uint32_t camv = 0;
    uint32_t cam_rows[NS] = {}, cam_leaf_off = 0, cam_leaf_w = 0, cam_sh = 0, cam_mu0 = 0, cam_mu1 = 0, cam_mu2 = 0;
    bool cam_ok = false;  // wave-uniform
    if (CAM && a.cam_shortcut && a.work.mode != 2) {
        const RayIn r0 = gen_ray(a.u, 0u, 0u);  // (the position does not depend on the pixel)
        if (in_bounds(r0.px, r0.py, r0.pz)) {   // ray_enter: such rays start where they are, dist = 0
            mu0 = entry_magic<GE>(r0.px * kScale);
            mu1 = entry_magic<GE>(r0.py * kScale);
            mu2 = entry_magic<GE>(r0.pz * kScale);
            stf = ST_DESC;
            restart_at(1u);
            descend();
            uint32_t v = lane < (uint32_t)NS ? (uint32_t)lds_at(colbase + (kRb + (uint32_t)SBASE + lane) * kRow) : 0u;  // row `lane` of this lane's own column
            v = lane == (uint32_t)NS ? leaf_off : v;
            v = lane == (uint32_t)NS + 1u ? leaf_w : v;
            v = lane == (uint32_t)NS + 2u ? sh : v;  // (D - the leaf's level)
            camv = v;
            if (kCamScalar) {
#pragma unroll
                for (int l = 0; l < NS; l++) cam_rows[l] = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
                cam_leaf_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf_off);
                cam_leaf_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf_w);
                cam_sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh);
                cam_mu0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mu0);
                cam_mu1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mu1);
                cam_mu2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mu2);
            }
            // worth it when the camera's leaf is deep (the copy at pick-up costs about two walk iterations)
            cam_ok = (uint32_t)D - sh_of((uint32_t)__builtin_amdgcn_readfirstlane(sh)) >= (uint32_t)(SBASE + 2);
            stf = ST_IDLE;
        }
    }

    // timeline build, second region of the debug buffer (16 words per wave behind the first 16384 x 16): slot 0 = the moment the wave
    // enters its main loop (top table staged, the camera's walk made), slots 1..14 = the moments of its first fourteen ray generations, slot 15 = where it runs
    // list-share feedback (launch_post): when the frame started -- the first workgroup's stamp (plain stores, here and at the end) --
    {
        uint32_t *const bal = fresh_args().balance;
        if (bal != nullptr && blockIdx.x == 0u && tid == 0u) {
            bal[9] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        }
    }
    if (DBG) {
        uint32_t *const ev = fresh_args().debug + 16u * 16384u + 16u * wave_id;
        if (lane == 0u) ev[0] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        // slot 15: where the wave runs -- HW_REG_XCC_ID[3:0] << 16 | HW_REG_HW_ID[15:0] (wave, SIMD, pipe, CU, SH, SE)
        if (lane == 0u) ev[15] = (((uint32_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20)) << 16) | (uint32_t)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
    }
    for (;;) {
        if (DBGH) c_mark = __builtin_amdgcn_s_memtime();
        if (DBG) n_rounds += 1;
        if (DBGH) {  // per-wave timeline build only (SVO_OPT_DEBUG_BUFFER): the default instantiation carries none of this
            const uint32_t nd = (uint32_t)__popcll(__ballot(sabs(stf) == ST_DESC));
            (void)nd;
            if (!CNT) dbg_desc_rounds += nd ? 1u : 0u;
        }
        // ---- 1. descent: one dependent word per level below the restart level ----
        if (sabs(stf) == ST_DESC) {
            descend();
        }
        if (DBGH) {
            const uint64_t now = __builtin_amdgcn_s_memtime();
            c_desc += (uint32_t)(now - c_mark);
            c_mark = now;
        }
        // ---- 2. refill idle lanes from the ray pool (ballot compaction) ----
        // (Between the descent and the step: the stores and atomics issued here -- records of finished rays, records of
        // rays that miss the cube, the next claim -- are counted in vmcnt together with the loads, in order, so a descent
        // right behind them would wait for their acknowledgements; the step's arithmetic runs meanwhile.  Rays picked up
        // here descend in the next round.)
        // Common case first and cheap: fewer than refill_min idle lanes -> straight on.  (refill_min <= 64, so a wave
        // without active lanes always takes the slow path, where the exit test lives.)
        uint64_t act = __ballot(sabs(stf) >= ST_LEAF);  // lanes with a ray in flight
        const uint32_t n_idle = 64u - (uint32_t)__popcll(act);
        if (n_idle >= a.refill_min) {
            if (next == 0xFFFFFFFEu && pool_n == 0u) {  // now the answer of the early claim is needed
                const uint64_t c_w0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                if (DBG && !DBGH) {  // (light timeline build, slot 14: cycles until the claim's answer itself is there; slot 15 adds the list look-ups)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    dbg_desc_rounds += (uint32_t)(__builtin_amdgcn_s_memtime() - c_w0);
                }
                uint32_t lst = home / kSubs;
                uint32_t s = entry_of(lst, __builtin_amdgcn_readfirstlane(pend) * kSubs + home % kSubs + ((gridDim.x + kShards - 1u - lst) / kShards) * (uint32_t)(BLOCK / 64));
                // The home counter ran out: lane i looks at counter i -- one load for all 64 -- and the wave draws from the first
                // counter that still has entries, starting behind its own (the other counters of its list, then list by list), and
                // makes that counter its home: later claims from it are issued ahead of time again (frames whose lists differ
                // much in length -- secondary rays of the pixels that hit something -- would otherwise pay two round trips per
                // strip).  Probing the counters one atomic after the other would cost a finished wave 64 round trips to find
                // out that the frame is over.
                while (s == 0xFFFFFFFFu) {
                    if (DBG && !DBGH && lane == 0u) dbg_desc_iters += 1u;  // (light timeline build, slot 12: probes of the 64 claim counters)
                    const uint32_t l = lane / kSubs;
                    const uint32_t cv = __hip_atomic_load(work_counter + lane * kShardStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t res_l = ((gridDim.x + kShards - 1u - l) / kShards) * (uint32_t)(BLOCK / 64);
                    uint32_t len_l;
                    if (order) {
                        len_l = order[l];
                    } else {  // (entry_of's deal: runs of 16 strips, run g belongs to list g mod 8; the last run may be short)
                        const uint32_t runs = (n_strips + 15u) >> 4, mine = runs > l ? (runs - l + kShards - 1u) / kShards : 0u;
                        len_l = (mine << 4) - ((runs != 0u && (runs - 1u) % kShards == l) ? (runs << 4) - n_strips : 0u);
                    }
                    const uint64_t has = __ballot((uint64_t)cv * kSubs + lane % kSubs + res_l < (uint64_t)len_l);
                    if (!has) break;
                    const uint64_t rot = home ? ((has >> home) | (has << (64u - home))) : has;
                    home = (home + (uint32_t)__ffsll((unsigned long long)rot) - 1u) & 63u;
                    lst = home / kSubs;
                    uint32_t k = 0u;
                    if (lane == 0) k = atomicAdd(work_counter + home * kShardStride, 1u);
                    s = entry_of(lst, __builtin_amdgcn_readfirstlane(k) * kSubs + home % kSubs + ((gridDim.x + kShards - 1u - lst) / kShards) * (uint32_t)(BLOCK / 64));
                }
                // the next claim goes to the list's next counter: every wave of a list draws from all of its counters in turn, so
                // the counters advance together and the list is consumed front to back, as with one counter (waves tied to one
                // counter let the counters drift apart: a group of waves that got long strips leaves its share of the list's
                // front -- long strips -- for the end; 4K frames with fused shadow rays lost 25 %)
                home = (home & ~(kSubs - 1u)) | ((home + 1u) & (kSubs - 1u));
                s = __builtin_amdgcn_readfirstlane(s);
                next = s != 0xFFFFFFFFu ? s * strip_items : 0xFFFFFFFFu;
                strip_end = s != 0xFFFFFFFFu ? min(next + strip_items, n_items) : 0xFFFFFFFFu;
                if (DBG && next == 0xFFFFFFFFu && t_dry == 0) t_dry = (uint32_t)__builtin_amdgcn_s_memrealtime() | 1u;
                if (DBG && !CNT) dbg_desc_start += (uint32_t)(__builtin_amdgcn_s_memtime() - c_w0);  // (slot 15: cycles spent waiting for claims)
            }
            const bool more = (pool_n != 0u) || (next != 0xFFFFFFFFu);
            if (more) {
                if (DBGH) dbg_refills += 1;
                if (pool_n == 0u) {
                    if (DBG) dbg_gens += 1;
                    if (DBG && dbg_gens <= 14u) {
                        uint32_t *const ev = fresh_args().debug + 16u * 16384u + 16u * wave_id + dbg_gens;
                        if (lane == 0u) ev[0] = (uint32_t)__builtin_amdgcn_s_memrealtime();
                    }
                    const uint64_t c_g0 = DBGH ? __builtin_amdgcn_s_memtime() : 0ull;
                    // -- generate the next (up to) 64 rays, all lanes --
                    const TraceArgs &a = fresh_args();  // (shadows the parameter: see fresh_args)
                    const uint32_t q = next + lane;
                    bool alive = false;
                    float gp0 = 0, gp1 = 0, gp2 = 0, gd0 = 1, gd1 = 1, gd2 = 1, gdist = 0;
                    uint32_t gout = 0;
                    if (q < strip_end) {
                        ItemFast it = decode_item_fast(a.work, q);  // (decode_item_wave's scalar divisions cost this loop registers it does not have)
                        if (a.work.mode == 2 && a.skip && a.skip[q]) it.valid = false;  // no ray: the producer wrote the record
                        if (it.valid) {
                            RayIn r;
                            if (a.work.mode == 2) {
                                const float *p = a.rays + 6ull * it.out;
                                r = RayIn{p[0], p[1], p[2], p[3], p[4], p[5]};
                            } else {
                                r = gen_ray(a.u, it.px, it.py);
                            }
                            float pos[3], dir[3];
                            if (!ray_enter(r, pos, dir, gdist)) {
                                write_hit(a.hits, it.out, 0u, 0.0f, 0u, 0u, 0u, 0u);
                                if (a.aux_t) a.aux_t[it.out] = 0.0f;
                                if (SHD) reinterpret_cast<uint4 *>(a.shadow_hits)[it.out] = make_uint4(0u, 0u, 0u, 0u);
                            } else if (!(clean_component(pos[0], dir[0]) && clean_component(pos[1], dir[1]) &&
                                         clean_component(pos[2], dir[2]) && fabsf(gdist) <= 1.0e30f)) {
                                // outside the proven range of the fast arithmetic: hand the ray to the
                                // reference-shaped kernel that runs right after this one
                                uint32_t slot = atomicAdd(&defer[0], 1u);
                                defer[1u + slot] = q;
                            } else {
                                alive = true;
                                gp0 = pos[0]; gp1 = pos[1]; gp2 = pos[2];
                                gd0 = dir[0]; gd1 = dir[1]; gd2 = dir[2];
                                uint32_t ncode = normal_code(truncf(gp0 * 1.000001f)) |
                                                 (normal_code(truncf(gp1 * 1.000001f)) << 2) |
                                                 (normal_code(truncf(gp2 * 1.000001f)) << 4);
                                gout = it.out | (ncode << 26);
                            }
                        }
                    }
                    const uint64_t am = __ballot(alive);
                    if (alive) {
                        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                        gd0 *= kScale; gd1 *= kScale; gd2 *= kScale;  // exact: power-of-two scaling
                        pool[0 * 64 + slot] = __float_as_uint(gp0 * kScale);
                        pool[1 * 64 + slot] = __float_as_uint(gp1 * kScale);
                        pool[2 * 64 + slot] = __float_as_uint(gp2 * kScale);
                        pool[3 * 64 + slot] = __float_as_uint(gd0);
                        pool[4 * 64 + slot] = __float_as_uint(gd1);
                        pool[5 * 64 + slot] = __float_as_uint(gd2);
                        pool[6 * 64 + slot] = __float_as_uint(gdist);
                        pool[7 * 64 + slot] = gout;
                    }
                    pool_n = (uint32_t)__popcll(am);
                    pool_i = 0u;
                    next += min(64u, strip_end - next);
                    if (next >= strip_end) {
                        if (lane == 0) pend = atomicAdd(work_counter + home * kShardStride, 1u);
#ifdef SVO_PROBE_CLAIM_LATENCY   // (light timeline build: how long a claim's answer takes under the kernel's own load, slot 15)
                        if (DBG) {
                            const uint64_t c_a0 = __builtin_amdgcn_s_memtime();
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            dbg_desc_start += (uint32_t)(__builtin_amdgcn_s_memtime() - c_a0);
                        }
#endif
                        next = strip_end = 0xFFFFFFFEu;  // not known yet (and not "dry")
                    }
                    if (DBG && next == 0xFFFFFFFFu && t_dry == 0) t_dry = (uint32_t)__builtin_amdgcn_s_memrealtime() | 1u;
                    if (DBGH) c_gen += (uint32_t)(__builtin_amdgcn_s_memtime() - c_g0);
                }
                // -- idle lanes first write the record of the ray they finished, then take rays pool_i .. --
                if (sabs(stf) == ST_PENDING) flush_record(true);
                if (SHD) act = __ballot(sabs(stf) >= ST_LEAF);  // lanes that went on with a shadow ray are not idle
                if (!(sabs(stf) >= ST_LEAF)) {
                    const uint64_t idle = ~act;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (rank < pool_n) {
                        const uint32_t e = pool_i + rank;
                        P0 = __uint_as_float(pool[0 * 64 + e]);
                        P1 = __uint_as_float(pool[1 * 64 + e]);
                        P2 = __uint_as_float(pool[2 * 64 + e]);
                        Dr0 = __uint_as_float(pool[3 * 64 + e]);
                        Dr1 = __uint_as_float(pool[4 * 64 + e]);
                        Dr2 = __uint_as_float(pool[5 * 64 + e]);
                        // RN(1 / Dr), what div_by_recip wants, recomputed here instead of three more words per pooled ray (8 words keep a
                        // workgroup at 22 KiB of LDS): v_rcp_f32 and one Newton step give the correctly rounded reciprocal for every
                        // f32 of a clean ray's range (tools/rcptest_gpu.hip: all 1.43e9 of them, profiles/r04_rcptest.log; tests/test_lds_oob_gpu.py re-runs it)
                        Y0 = recip_rn(Dr0);
                        Y1 = recip_rn(Dr1);
                        Y2 = recip_rn(Dr2);
                        S0 = copysign_bits(1.0f, Dr0);
                        S1 = copysign_bits(1.0f, Dr1);
                        S2 = copysign_bits(1.0f, Dr2);
                        dist = __uint_as_float(pool[6 * 64 + e]);
                        out = pool[7 * 64 + e];
                        tcur = 0.0f;
                        stepsf = 0.0f;
                        nmf = 0.0f;
                        if (CAM && cam_ok && dist == 0.0f) {
                            // a ray from the camera's own position: the walk the wave made at the start (see above), path codes included
                            if (kCamScalar) {
#pragma unroll
                                for (int l = 0; l < NS; l++) lds_at(colbase + (kRb + (uint32_t)SBASE + (uint32_t)l) * kRow) = cam_rows[l];  // (rows below the leaf: never read)
                                leaf_off = cam_leaf_off;
                                leaf_w = cam_leaf_w;
                                sh = cam_sh;
                                mu0 = cam_mu0; mu1 = cam_mu1; mu2 = cam_mu2;
                            } else {
                                const uint32_t sh0 = (uint32_t)__builtin_amdgcn_readlane((int)camv, NS + 2), L0 = (uint32_t)D - sh_of(sh0);
                                for (uint32_t l = 0; l < (uint32_t)NS && l + (uint32_t)SBASE <= L0; l++)
                                    lds_at(colbase + (kRb + (uint32_t)SBASE + l) * kRow) = (uint32_t)__builtin_amdgcn_readlane((int)camv, (int)l);
                                leaf_off = (uint32_t)__builtin_amdgcn_readlane((int)camv, NS);
                                leaf_w = (uint32_t)__builtin_amdgcn_readlane((int)camv, NS + 1);
                                sh = sh0;
                                mu0 = entry_magic<GE>(P0); mu1 = entry_magic<GE>(P1); mu2 = entry_magic<GE>(P2);
                            }
                            stf = (kWalkStops && (leaf_w >> 4) != kVoxelOffset) ? ST_PENDING : ST_LEAF;  // at its leaf, no step taken
                        } else {
                            // entry path codes (the position may sit a rounding error outside the cube: clamp)
                            mu0 = entry_magic<GE>(P0);
                            mu1 = entry_magic<GE>(P1);
                            mu2 = entry_magic<GE>(P2);
                            stf = ST_DESC;
                            if (CNT) satm = 0u;
                            restart_at(1u);
                        }
                    }
                }
                const uint32_t took = min(SHD ? 64u - (uint32_t)__popcll(act) : n_idle, pool_n);
                pool_i += took;
                pool_n -= took;
                act = __ballot(sabs(stf) >= ST_LEAF);
            }
            if (SHD && !more) {  // nothing left to hand out, but finished primary rays still have their shadow rays to trace
                if (sabs(stf) == ST_PENDING) flush_record(true);
                act = __ballot(sabs(stf) >= ST_LEAF);
            }
            // the only exit: nothing in flight, nothing pooled, nothing left to claim.  (No lane active but work left --
            // e.g. a strip whose rays all miss the cube: the traversal below is skipped lane-wise and the loop comes back.)
            if (act == 0ull && pool_n == 0u && next == 0xFFFFFFFFu) break;
        }

        if (DBGH) {
            dbg_active += (uint32_t)__popcll(__ballot(sabs(stf) == ST_LEAF));
            const uint64_t now = __builtin_amdgcn_s_memtime();
            c_refill += (uint32_t)(now - c_mark);
            c_mark = now;
        }
        if (CNT) {
            // ---- 3a. hit counters.  The reference's find_voxel bumps every word from the root to the leaf, once per
            // call, i.e. once per round here (shader.wgsl:157-161); this kernel does not walk those words, but it
            // knows their addresses: child group of level l (the top tables up to level K+1, the lane's stack below)
            // + child index from the path codes.  A counter stops at 15, and near the root that happens within the
            // first rounds of a frame, so each lane remembers which words of its current path it has seen saturated
            // (satm: set by the walk, which reads every word from level K+1 down anyway; a restart at level r keeps the
            // bits of the levels above r; and the workgroup's table of saturated words, see the queue above) and only
            // reports the others.  Nothing is loaded here: the words of levels 1..K, which the walk never reads (the top
            // table stands for them), are reported until the table knows them -- a compare-and-swap that finds a 15 does nothing.
            // Final counters = min(15, old + visits), like the RESTART kernel.
            // (Wave-uniform control flow around per-lane predicates: the queue cursor is a scalar.)
            // (a lane whose walk ended below level SMAX -- a tree deeper than declared: its ray ends with the "too deep" record and the frame
            // is refused -- reports no visits: ADVICE r4)
            const bool at_leaf = sabs(stf) == ST_LEAF && !(kWalkTrips && __uint_as_float(sh) < kMagic + (float)(D - SMAX));
            const uint64_t c_c0 = DBGH ? __builtin_amdgcn_s_memtime() : 0ull;
            if (__ballot(at_leaf) != 0ull) {
                // (a word below level SMAX ends the ray: the tree is deeper than declared -- with kWalkTrips such a lane is not at_leaf)
                const uint32_t L = kWalkTrips ? (uint32_t)D - sh_of(sh) : min((uint32_t)D - sh_of(sh), (uint32_t)SMAX);
                uint32_t todo = at_leaf ? (~satm & ((1u << L) - 2u)) : 0u;  // levels 1 .. L-1 not known to be saturated
                constexpr uint32_t kTopLv = (2u << K) - 2u;  // levels 1 .. K
                auto cell_k = [&](uint32_t x0, uint32_t x1, uint32_t x2) -> uint32_t {
                    return (__builtin_amdgcn_ubfe(x0, D - K, K) << (2 * K)) | (__builtin_amdgcn_ubfe(x1, D - K, K) << K) | __builtin_amdgcn_ubfe(x2, D - K, K);
                };
                if (__ballot((todo & kTopLv) != 0u) != 0ull) {  // (rays picked up, rays that crossed a top-level boundary)
                    const uint32_t cellK = cell_k(mu0, mu1, mu2);
                    const uint32_t known = (todo & kTopLv) ? ((top_sat[cellK >> 3] >> ((cellK & 7u) * 4u)) & todo & kTopLv) : 0u;
                    satm |= known;
                    todo &= ~known;
                }
                // (bit 0: the leaf itself -- word and address are at hand.  It takes its turn in the loop like a level: lanes
                // report different words in one pass anyway, and a pass of its own for the leaf cost a third of the step)
                todo |= (at_leaf && (leaf_w & 15u) < 15u) ? 1u : 0u;
                while (__ballot(todo != 0u) != 0ull) {
                    bool mine = todo != 0u;
                    const uint32_t l = mine ? (uint32_t)__builtin_ctz(todo) : 1u;
                    todo &= todo - 1u;
                    if (DBGH && lane == 0u) dbg_desc_iters += 1u;  // (CNT: slot 12 = iterations of this loop)
                    const uint32_t kk = l - 1u, shc = (uint32_t)D - kk;
                    uint32_t g = 0u;
                    if (l >= (uint32_t)SBASE) {
                        g = lds_at(colbase + (l + kRb) * kRow);
                    } else if (l >= 2u) {
                        const uint32_t cell = (__builtin_amdgcn_ubfe(mu0, shc, kk) << (2u * kk)) | (__builtin_amdgcn_ubfe(mu1, shc, kk) << kk) | __builtin_amdgcn_ubfe(mu2, shc, kk);
                        g = kk == (uint32_t)K ? (tbl[cell] & 0x07FFFFFFu) : aux[(kk == 1u ? 0u : 8u) + cell];
                    }
                    const uint32_t bit = (uint32_t)D - l;  // (l = 0: bit D, which belongs to the exponent of kMagic -- and p is replaced below)
                    uint32_t p = g + ((__builtin_amdgcn_ubfe(mu0, bit, 1u) << 2) | (__builtin_amdgcn_ubfe(mu1, bit, 1u) << 1) |
                                      __builtin_amdgcn_ubfe(mu2, bit, 1u));
                    p = l == 0u ? leaf_off >> 2 : p;
                    if (mine && sat_tags[sat_slot(p)] == p) {  // some lane of the workgroup has seen it reach 15
                        satm |= (1u << l) & ~1u;
                        mine = false;
                        if (l - 1u < (uint32_t)K) {
                            uint32_t x0 = mu0, x1 = mu1, x2 = mu2;
                            // (the cell's arithmetic stays in this rare branch: the compiler had moved it in front of the loop, eight instructions a round)
                            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
                            const uint32_t cellK = cell_k(x0, x1, x2);
                            atomicOr(&top_sat[cellK >> 3], (1u << l) << ((cellK & 7u) * 4u));
                        }
                    }
                    cq_push(mine, p, 0u);
                }
            }
            if (DBGH) dbg_desc_start += (uint32_t)(__builtin_amdgcn_s_memtime() - c_c0);  // (CNT: slot 15 = cycles spent counting)
        }
        // ---- 3. hit test / DDA step (shader.wgsl:215-244), clean rays, grid units; see the header of this kernel ----
        if (sabs(stf) == ST_LEAF) {  // at a leaf (rays picked up above descend first)
            // (leaf words 0x8000000c, c the counter bits, are the empty leaves: everything else ends the ray here -- a solid
            // leaf, or an interior word at level SMAX -- with the normal, distance and count the last step left)
            if (!kWalkStops && (leaf_w >> 4) != kVoxelOffset) {
                stf *= 0.5f;  // ST_LEAF -> ST_PENDING
            } else {
                const float Hm = __uint_as_float((sh << 23) + 0x3F000000u);  // 2^(sh-1) = voxel_size / 2 in grid units (sh = D - L >= 1; the shift drops the magic's bits)
                // leaf centre along one axis and a = (centre - pos) + r_sign * voxel_size / 2 (shader.wgsl:229): the code with its low sh
                // bits replaced by 1 0 .. 0 is, read as a float, 2^23 + (lower corner + 2^(sh-1)); minus (2^23 + 2^22) that is the centre in
                // grid units, exactly (the reference sums +-2^-k, which is exact too).  One instruction of the 4-cycle group per axis;
                // the all-f32 form -- floor taken by the rounding of (U - (2^(sh-1) - 1/2)) + 3 * 2^(22+sh) -- costs four more of the
                // other group and measured 2 % slower (profiles/r04_step_variants_ab.log)
                const uint32_t keepm = 0xFFFFFFFFu << sh_of(sh), hbit = __float_as_uint(kMagic + Hm);
                auto t_axis = [&](uint32_t mu, float P, float Dr, float Y, float S) -> float {
                    const float c = __uint_as_float((mu & keepm) | hbit) - (kMagic + kScale);
                    return div_by_recip(__builtin_fmaf(S, Hm, c - P), Dr, Y);
                };
                const float t0 = t_axis(mu0, P0, Dr0, Y0, S0), t1 = t_axis(mu1, P1, Dr1, Y1, S1), t2 = t_axis(mu2, P2, Dr2, Y2, S2);
                // no NaNs here, so IEEE minNum equals the oracle's (b < a) ? b : a up to the sign of a zero, which no later value
                // depends on, and "t_i <= min(t_j, t_k)" is "t_i equals the minimum of the three"
                const float tnew = __builtin_fminf(__builtin_fminf(t0, t1), t2);
                // voxel_pos = pos + dir * t - normal * 2e-6, normal = mask * -sign(dir): k where t_i is the minimum, else 0, times the
                // sign, in one fma.  (The mask as 1 - clamp(clamp((t_i - min) 2^126) 2^24) needs no compare; same time, three more
                // instructions.)
                const float k0 = t0 == tnew ? kNudge : 0.0f, k1 = t1 == tnew ? kNudge : 0.0f, k2 = t2 == tnew ? kNudge : 0.0f;
                const float G0 = __builtin_fmaf(k0, S0, P0 + Dr0 * tnew);
                const float G1 = __builtin_fmaf(k1, S1, P1 + Dr1 * tnew);
                const float G2 = __builtin_fmaf(k2, S2, P2 + Dr2 * tnew);
                // in_bounds (shader.wgsl:177-180), -2^22 <= G < 2^22: G is a float, so below the bound it is at most 2^22 - 1/4 and
                // 4 (2^22 - G) >= 1, at or above it <= 0; likewise 2 G + (2^23 + 1) >= 1 from -2^22 up and <= 0 from -2^22 - 1/2 down
                // (six clamps and five products of the 2-cycle group; max3 / min3 first: three instructions fewer, not faster)
                auto inside = [&](float G) -> float {
                    return __builtin_amdgcn_fmed3f(__builtin_fmaf(G, -4.0f, 4.0f * kScale), 0.0f, 1.0f) *
                           __builtin_amdgcn_fmed3f(__builtin_fmaf(G, 2.0f, 2.0f * kScale + 1.0f), 0.0f, 1.0f);
                };
                const float inbf = (inside(G0) * inside(G1)) * inside(G2);
                // the ray goes on unless the step leaves the cube or the count after it exceeds 100 (iff it is 100 now)
                const float contf = inbf * __builtin_amdgcn_fmed3f(100.0f - stepsf, 0.0f, 1.0f);
                tcur = tnew;
                stepsf += inbf;                                          // (shader.wgsl:237-240: counted only inside the cube)
                nmf = __builtin_fmaf(k2, 4.0f, __builtin_fmaf(k1, 2.0f, k0));  // k * (mask as a number 0 .. 7), decoded when the record is written
                stf *= __builtin_fmaf(contf, 1.5f, 0.5f);                // ST_LEAF -> ST_DESC (goes on) or ST_PENDING (ended)
                if (contf > 0.5f) {
                    // new path codes: the position is inside the cube.  `>` mode: ceil(G) - 1 + 2^22 = (2^22 - 1) - floor(-G), which
                    // is -1 on the face G = -2^22: scaled by 2^-23 the code lies in [0, 1) and the clamp does that for nothing
                    uint32_t n0, n1, n2;
                    if (GE) {
                        n0 = __float_as_uint(__builtin_floorf(G0) + (kScale + kMagic));
                        n1 = __float_as_uint(__builtin_floorf(G1) + (kScale + kMagic));
                        n2 = __float_as_uint(__builtin_floorf(G2) + (kScale + kMagic));
                    } else {
                        auto code_gt = [&](float G) -> uint32_t {
                            const float us = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_floorf(-G), -1.0f / kMagic, (kScale - 1.0f) / kMagic), 0.0f, 1.0f);
                            return __float_as_uint(__builtin_fmaf(us, kMagic, kMagic));
                        };
                        n0 = code_gt(G0); n1 = code_gt(G1); n2 = code_gt(G2);
                    }
                    // first level at which the old and the new path differ (the exponent bits cancel in the xor); the leaf's own
                    // bit is or-ed in, so that level is at most L -- also when nothing differs (the walk stops at L <= SMAX)
                    const uint32_t lbit = __float_as_uint(__builtin_fmaf(Hm, 2.0f, kMagic)) & ((1u << D) - 1u);  // 1 << sh
                    const uint32_t diff = ((mu0 ^ n0) | (mu1 ^ n1) | (mu2 ^ n2)) | lbit;
                    mu0 = n0; mu1 = n1; mu2 = n2;
#ifdef SVO_DUMMY_STEP_S
                    for (int i_ = 0; i_ < SVO_DUMMY_STEP_S; i_++) { uint32_t t_; asm volatile("v_bfe_u32 %0, %1, 1, 30" : "=v"(t_) : "v"(mu0)); }
#endif
                    const uint32_t rb = (uint32_t)__builtin_clz(diff);  // r + (31 - D); diff != 0
                    if (CNT) {  // levels r and below belong to a new path
                        satm &= (1u << (rb - kRb)) - 1u;
                    }
                    restart_at_rb(rb);
                }
            }
        }
        if (DBGH) c_step += (uint32_t)(__builtin_amdgcn_s_memtime() - c_mark);
    }
    if (CNT) cq_flush();
    if (DBGH) {  // wave totals of the per-lane tallies
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dbg_desc_iters += (uint32_t)__shfl_xor((int)dbg_desc_iters, o);
            dbg_desc_lanes += (uint32_t)__shfl_xor((int)dbg_desc_lanes, o);
        }
    }
    if (DBGH) {  // the rays that finish last: their step counts tell whether the tail is long rays or late starts
        uint32_t last = sabs(stf) == ST_PENDING ? (uint32_t)stepsf : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, (uint32_t)__shfl_xor((int)last, o));
        dbg_iters = last;
    }
    if (sabs(stf) == ST_PENDING) flush_record(false);
    // -- and when the waves that started on list l (the XCD the list belongs to) were done
    {
        uint32_t *const bal = fresh_args().balance;
        // (every workgroup stores its stamp in a slot of its own -- plain stores: an atomic on one address is executed at the memory
        // side, 11 ns apiece, and 7168 waves adding to 8 words at their exit held the END of the kernel up by 10 us)
        if (bal != nullptr && tid == 0u && blockIdx.x < kBalanceSlots) bal[kBalanceHead + blockIdx.x] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        if (bal != nullptr && tid == 0u && blockIdx.x == 0u) bal[19] = gridDim.x;  // (how many stamps this launch leaves)
    }
    if (DBG && lane == 0) {
        uint64_t t_end = __builtin_amdgcn_s_memrealtime();
        uint32_t *d = a.debug + 16u * wave_id;
        d[8] = c_refill;         // shader cycles in the refill section (ray generation included)
        d[9] = c_desc;           // ... in the descent loop
        d[10] = c_step;          // ... in the step section
        d[11] = c_gen;           // ... generating rays (part of d[8])
        d[12] = dbg_desc_iters;  // descent-loop iterations of the wave (one dependent load each)
        d[13] = dbg_desc_lanes;  // lanes inside the loop, summed over its iterations
        d[14] = dbg_desc_rounds; // rounds in which any lane descended
        d[15] = dbg_desc_start;  // cycles spent waiting for the answers of strip claims (CNT: cycles spent counting)
        d[0] = t_begin;
        d[1] = t_dry ? t_dry : (uint32_t)t_end;
        d[2] = (uint32_t)t_end;
        d[3] = n_rounds;
        d[4] = dbg_active;   // sum over rounds of active lanes
        d[5] = dbg_iters;    // largest step count among the rays the wave finished last
        d[6] = dbg_refills;
        d[7] = dbg_gens;
    }
}

// ---------------------------------------------------------------------------------------------
// Counter scan, compute.wgsl:26-47 (the reference appends with one atomicAdd per node).
// ---------------------------------------------------------------------------------------------
// A workgroup takes chunks of kScanChunk words (2048 per wave, 4 words per lane and iteration, 16-byte loads) and
// makes two passes over a chunk: count the candidates, reserve the list space of the whole chunk with ONE atomic
// per list (one per wave and 64 words serialised at ~90 atomics/us: 4.8 ms for the 428 MB benchmark tree, 1 % of the
// HBM roofline), then re-read the chunk (it is still in L2) and write the indices.  A list that is already full --
// on a large tree nearly every interior node is an unsubdivide candidate -- is not touched any more: only the first
// `capacity - 1` entries are ever used (adaptive.rs:22,86), so the count stops being maintained beyond that.
constexpr uint32_t kScanChunk = 8192;

__device__ __forceinline__ void scan_classify(uint32_t node, bool &is_sub, bool &is_unsub) {
    const uint32_t counter = node & 15u;
    is_unsub = node != 0u && counter == 0u && (node >> 4) < kVoxelOffset;            // compute.wgsl:40-42
    is_sub = node != 0u && !is_unsub && counter >= 4u && (node >> 4) > kVoxelOffset;  // :43-46
}

// clear_counters (SVO_OPT_SCAN_CLEARS_COUNTERS): the second pass also zeroes the hit counters it has just classified.
// The reference gets that effect from re-uploading the whole array with counter-free host words every frame
// (app.rs:113-118); with this flag the host only sends the words it changed (svo_nodes_scatter).
__global__ __launch_bounds__(256) void scan_kernel(uint32_t *nodes, uint32_t n_words, uint32_t node_length,
                                                   uint32_t *sub, uint32_t *unsub, uint32_t capacity, bool clear_counters) {
    __shared__ uint32_t tot[2][4];
    __shared__ uint32_t base[2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t limit = min(n_words, node_length);  // `id < node_length` (compute.wgsl:41,44)
    auto load4 = [&](uint32_t id, uint32_t (&v)[4]) {
        if (id + 4u <= limit) {
            const uint4 q = *reinterpret_cast<const uint4 *>(nodes + id);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (id + (uint32_t)k < limit) ? nodes[id + (uint32_t)k] : 0u;
        }
    };
    for (uint32_t chunk = blockIdx.x * kScanChunk; chunk < limit; chunk += gridDim.x * kScanChunk) {
        const uint32_t w0 = chunk + wave * (kScanChunk / 4u);
        uint32_t cs = 0, cu = 0;
        for (uint32_t it = 0; it < kScanChunk / 4u / 256u; it++) {
            uint32_t v[4];
            load4(w0 + it * 256u + lane * 4u, v);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bool a, b;
                scan_classify(v[k], a, b);
                cs += a ? 1u : 0u;
                cu += b ? 1u : 0u;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cs += (uint32_t)__shfl_xor((int)cs, o);
            cu += (uint32_t)__shfl_xor((int)cu, o);
        }
        if (lane == 0) { tot[0][wave] = cs; tot[1][wave] = cu; }
        __syncthreads();
        if (threadIdx.x < 2u) {
            uint32_t *list = threadIdx.x == 0u ? sub : unsub;
            const uint32_t n = tot[threadIdx.x][0] + tot[threadIdx.x][1] + tot[threadIdx.x][2] + tot[threadIdx.x][3];
            uint32_t b = capacity;  // "full": nothing is written
            if (n != 0u && __hip_atomic_load(&list[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u < capacity)
                b = atomicAdd(&list[0], n);
            base[threadIdx.x] = b;
        }
        __syncthreads();
        uint32_t bs = base[0], bu = base[1];
        for (uint32_t w = 0; w < wave; w++) { bs += tot[0][w]; bu += tot[1][w]; }
        const bool write_s = cs != 0u && (uint64_t)bs + 1u < capacity, write_u = cu != 0u && (uint64_t)bu + 1u < capacity;
        if (write_s || write_u || clear_counters) {  // wave-uniform
            for (uint32_t it = 0; it < kScanChunk / 4u / 256u; it++) {
                uint32_t v[4];
                const uint32_t id = w0 + it * 256u + lane * 4u;
                load4(id, v);
                if (clear_counters && ((v[0] | v[1] | v[2] | v[3]) & 15u)) {
                    if (id + 4u <= limit) {
                        *reinterpret_cast<uint4 *>(nodes + id) = make_uint4(v[0] & ~15u, v[1] & ~15u, v[2] & ~15u, v[3] & ~15u);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (id + (uint32_t)k < limit) nodes[id + (uint32_t)k] = v[k] & ~15u;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bool a, b;
                    scan_classify(v[k], a, b);
                    const uint64_t ms = __ballot(a), mu = __ballot(b);
                    if (write_s && ms) {
                        const uint32_t r = bs + __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
                        if (a && (uint64_t)r + 1u < capacity) sub[1u + r] = id + (uint32_t)k;
                        bs += (uint32_t)__popcll(ms);
                    }
                    if (write_u && mu) {
                        const uint32_t r = bu + __builtin_amdgcn_mbcnt_hi((uint32_t)(mu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mu, 0u));
                        if (b && (uint64_t)r + 1u < capacity) unsub[1u + r] = id + (uint32_t)k;
                        bu += (uint32_t)__popcll(mu);
                    }
                }
            }
        }
        __syncthreads();  // tot / base are reused by the next chunk
    }
}

// ---------------------------------------------------------------------------------------------
// Post pass of a STACK frame (one launch): trace the rays the fast kernel deferred, compute the cost of
// every 64-item strip (the largest step count among its rays) as scheduling feedback for the next frame,
// and re-arm the claim counters.  A strip whose deferred rays are written by another workgroup of this
// launch may see their old records; that only perturbs the schedule, never a result.
// ---------------------------------------------------------------------------------------------
// List shares (round 5).  Every list is one XCD's, and the XCDs do not take the same time over the same number of strips: a list's part of
// the screen decides how deep its rays walk and how many lines its L2 has to fetch (the middle columns of the benchmark frame: 50 k
// lines, the outer ones 10 k), and with claims running ahead no list has anything left to steal when the fast XCDs are done -- they
// end 13 - 15 us before the slow ones (profiles/r05_xcd_imbalance.txt).  So the trace kernel stamps when the waves of every list
// ended (the mean over a sample of its workgroups: the latest wave is a noisy figure), and the lists' shares of every cost class
// follow: share *= 1 + gain (mean / time - 1), clamped per step and in total.
// (called by all 256 threads of one workgroup, only for frames that are fed back: every workgroup of a frame overwrites its stamp, so
// nothing has to be cleared in between)
// The shares are a proposal, not a proof: on the depth-20 fractal at 4K the frame they converge to is a tenth SLOWER than the one with equal
// shares (profiles/r05_balance_probe.log).  So the step also notes how long the frame took with the shares it was traced with (the latest
// of the sampled end stamps), remembers the best shares seen since the layout was new -- the equal shares of its first frame included --
// and the last of the 16 steps of a resting view puts those back: the schedule that is then kept is never worse than the unweighted one.
__device__ __forceinline__ void balance_step(uint32_t *bal, float gain, uint32_t n_slots, uint32_t update) {
    __shared__ float t_sum[8], t_n[8];
    __shared__ uint32_t t_last;
    const uint32_t t0 = bal[9];
    if (threadIdx.x < 8u) t_sum[threadIdx.x] = t_n[threadIdx.x] = 0.0f;
    if (threadIdx.x == 0u) t_last = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += 256u) {  // slot i: workgroup i, of list i % 8
        const float t = (float)(bal[kBalanceHead + i] - t0);  // (10 ns ticks; the difference survives a wrap of the counter)
        if (t > 0.0f && t < 1.0e8f) {
            atomicAdd(&t_sum[i & 7u], t);
            atomicAdd(&t_n[i & 7u], 1.0f);
            atomicMax(&t_last, (uint32_t)t);
        }
    }
    __syncthreads();
    if (threadIdx.x != 0u) return;
    // [20] the shortest frame so far (ticks; 0: none yet), [21..29] the shares it was traced with
    if (update <= 16u && t_last != 0u && (update == 1u || bal[20] == 0u || t_last < bal[20])) {
        bal[20] = t_last;
        for (int k = 0; k <= 8; k++) bal[21 + k] = bal[k];
    }
    if (update == 16u && bal[20] != 0u) {  // the schedule built next is the one a resting view keeps
        for (int k = 0; k <= 8; k++) bal[k] = bal[21 + k];
        bal[18] += 1u;
        return;
    }
    float T[8], w[8], mean = 0.0f;
    bool ok = true;
    for (int k = 0; k < 8; k++) {
        ok = ok && t_n[k] > 0.0f;
        T[k] = ok ? t_sum[k] / t_n[k] : 1.0f;
        w[k] = (float)(bal[k + 1] - bal[k]);
        mean += T[k] * 0.125f;
    }
    if (!ok) return;
    float sum = 0.0f;
    for (int k = 0; k < 8; k++) {
        const float r = fminf(fmaxf(mean / T[k], 0.8f), 1.25f);
        w[k] = fminf(fmaxf(w[k] * (1.0f + gain * (r - 1.0f)), 0.6f * 8192.0f), 1.6f * 8192.0f);
        sum += w[k];
    }
    uint32_t acc = 0u;
    float run = 0.0f;
    for (int k = 0; k < 8; k++) {
        bal[k] = acc;
        run += w[k];
        acc = k == 7 ? 65536u : (uint32_t)(run / sum * 65536.0f + 0.5f);
    }
    bal[8] = 65536u;
    bal[18] += 1u;
}

__global__ __launch_bounds__(256) void post_kernel(TraceArgs a, uint32_t *claim_counters, const uint32_t *list,
                                                   uint32_t *next_deferred_count, uint8_t *cost, uint32_t n_strips, uint32_t balance_update) {
    const rsrc_t rs = make_rsrc(a.nodes, a.n_words);
    const bool misc_bool = (a.u.flags & SVO_F_MISC_BOOL) != 0;
    // re-arm for the next frame: the claim counters (the trace kernel is done with them) and the deferred
    // count of the OTHER list (this frame's list is still being read by this launch; lists alternate)
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < 64u; i += 256u) claim_counters[i * (uint32_t)kCounterStride] = 0u;
        if (threadIdx.x == 0) *next_deferred_count = 0u;
        // (balance_update: n > 0 = the n-th frame fed back: the first steps are large, the later ones small)
        if (a.balance != nullptr && balance_update != 0u) balance_step(a.balance, balance_update <= 6u ? 0.6f : 0.25f, min(a.balance[19], kBalanceSlots), balance_update);
    }
    const uint32_t n_def = list[0];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_def; i += gridDim.x * 256u) {
        const uint32_t q = list[1u + i];
        const float t = trace_one_restart(a, rs, misc_bool, false, q);
        if (a.shadow_hits) {  // fused shadow rays: the deferred primary ray's shadow ray, the same way
            const Item it = decode_item(a.work, q);
            const uint4 rec = reinterpret_cast<const uint4 *>(a.hits)[it.out];
            bool traced = false;
            if ((rec.z >> 16) & 1u) {
                float pos[3], dir[3], dist, org[3], nrm[3], sun[3];
                if (ray_enter(item_ray(a, it), pos, dir, dist)) {
                    secondary_origin(pos, dir, rec.z & 0xFFu, rec.w, t, org, nrm);
                    sun_direction(a.u, sun);
                    const RayIn r = {org[0], org[1], org[2], -sun[0], -sun[1], -sun[2]};
                    trace_ray_restart(a, rs, misc_bool, false, r, a.shadow_hits, it.out, nullptr);
                    traced = true;
                }
            }
            if (!traced) reinterpret_cast<uint4 *>(a.shadow_hits)[it.out] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (cost) {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = gridDim.x * 4u;
        for (uint32_t s = wave; s < n_strips; s += n_waves) {
            ItemFast it = decode_item_wave(a.work, s * 64u, lane);
            uint32_t steps = 0;
            if (it.valid) {
                steps = reinterpret_cast<const uint4 *>(a.hits)[it.out].z & 0xFFu;
                // fused shadow rays: the lane's chain is the primary ray plus its shadow ray
                if (a.shadow_hits) steps += reinterpret_cast<const uint4 *>(a.shadow_hits)[it.out].z & 0xFFu;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) steps = max(steps, (uint32_t)__shfl_xor((int)steps, o));
            if (lane == 0) {
                uint32_t c = min(steps >> kCostShift, kCostClasses - 1u);  // cost class
                cost[s] = (uint8_t)c;
            }
        }
    }
}

// Schedule for the next frames.  Strips fall into 16 cost classes (steps / 8).  Classes are walked from the
// most expensive down; inside a class the strips stay in screen order and are cut into 8 contiguous
// segments, one per claim counter (= per XCD, see the kernel).  So every XCD starts its long rays first,
// gets an equal share of every class, and still walks screen-contiguous runs (node-cache locality).
// Output: sched[0..7] = entries per list, then 8 lists of `cap` strip numbers each.
// Two small launches over kOrderBlocks workgroups, each owning a contiguous chunk of strips: class histogram
// per chunk, then a stable counting sort (ballot + mbcnt ranks inside a wave, prefix over waves and chunks).
constexpr uint32_t kOrderBlocks = 64, kOrderThreads = 256, kOrderBins = kCostClasses, kOrderLists = 8;

__device__ __forceinline__ uint32_t order_chunk(uint32_t n_strips) {
    return (((n_strips + kOrderBlocks - 1) / kOrderBlocks) + 63u) & ~63u;  // whole 64-strip groups per workgroup
}

// Round 5: the strips of a class are ranked COLUMN by column when the frame is one rectangle of pixel blocks (bpr = blocks per row, else
// 0: strip order): a list's segment of a class is then a vertical slab of the screen, and the slabs of the different classes of one list
// overlap -- the lists (one per XCD, each with an L2 of its own) share fewer nodes: 225 k instead of 272 k cache lines fetched per
// frame by the eight L2s together on the benchmark frame (180 k distinct; row-major ranks cut every class into horizontal bands that
// lie elsewhere for every class), profiles/r05_footprint_by_partition.txt.
__device__ __forceinline__ uint32_t order_strip_at(uint32_t p, uint32_t n_strips, uint32_t bpr) {
    if (bpr == 0u) return p;
    const uint32_t rows = n_strips / bpr;  // (the caller passes bpr only when n_strips is rows * bpr)
    const uint32_t col = p / rows;
    return (p - col * rows) * bpr + col;
}

__global__ __launch_bounds__(kOrderThreads) void strip_hist_kernel(const uint8_t *cls, uint32_t n_strips, uint32_t *hist, uint32_t bpr) {
    __shared__ uint32_t tally[kOrderBins];
    if (threadIdx.x < kOrderBins) tally[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t chunk = order_chunk(n_strips);
    const uint32_t lo = min(blockIdx.x * chunk, n_strips), hi = min(lo + chunk, n_strips);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = lo + (threadIdx.x & ~63u); base < hi; base += kOrderThreads) {
        const uint32_t s = base + lane;
        const uint32_t c = s < hi ? cls[order_strip_at(s, n_strips, bpr)] : 0xFFu;
        uint64_t todo = __ballot(c != 0xFFu);
        while (todo) {  // one LDS atomic per (wave, class present) instead of one per strip
            const uint32_t b = __builtin_amdgcn_readlane(c, __ffsll((unsigned long long)todo) - 1);
            const uint64_t m = __ballot(c == b);
            if (lane == 0) atomicAdd(&tally[b], (uint32_t)__popcll(m));
            todo &= ~m;
        }
    }
    __syncthreads();
    if (threadIdx.x < kOrderBins) hist[blockIdx.x * kOrderBins + threadIdx.x] = tally[threadIdx.x];
}

__global__ __launch_bounds__(kOrderThreads) void strip_order_kernel(const uint8_t *cls, const uint32_t *hist, uint32_t *sched,
                                                                    uint32_t n_strips, uint32_t cap, uint32_t bpr, const uint32_t *shares) {
    constexpr uint32_t kWaves = kOrderThreads / 64;
    __shared__ uint32_t class_n[kOrderBins], before[kOrderBins];
    // bound[c][k]: rank (inside class c) of the first strip that goes to list k -- equal eighths, or the shares of the lists
    // (`shares`: nine cumulative 16-bit fractions 0 .. 65536, from the times the lists took in an earlier frame: see launch_post)
    __shared__ uint32_t bound[kOrderBins][kOrderLists + 1];
    __shared__ uint32_t list_base[kOrderBins][kOrderLists];
    __shared__ uint32_t wave_tot[kOrderBins][kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid < kOrderBins) {  // class totals and the count in the chunks before this one
        uint32_t total = 0, prior = 0;
        for (uint32_t b = 0; b < kOrderBlocks; b++) {
            const uint32_t c = hist[b * kOrderBins + tid];
            prior += b < blockIdx.x ? c : 0u;
            total += c;
        }
        class_n[tid] = total;
        before[tid] = prior;
        for (uint32_t k = 0; k <= kOrderLists; k++) {
            const uint32_t frac = shares ? shares[k] : (k << 16) / kOrderLists;
            bound[tid][k] = k == kOrderLists ? total : (uint32_t)(((uint64_t)total * frac + 32768u) >> 16);
        }
    }
    __syncthreads();
    if (tid < kOrderLists) {
        uint32_t acc = 0;
        for (int b = kOrderBins - 1; b >= 0; b--) {  // expensive classes first
            list_base[b][tid] = acc;
            acc += bound[b][tid + 1] - bound[b][tid];
        }
        if (blockIdx.x == 0) sched[tid] = min(acc, cap);
    }
    // each wave owns a contiguous run of 64-strip groups of the chunk; lanes 0..31 keep the class tallies
    const uint32_t chunk = order_chunk(n_strips);
    const uint32_t clo = min(blockIdx.x * chunk, n_strips), chi = min(clo + chunk, n_strips);
    const uint32_t per_wave = (((chunk / 64u) + kWaves - 1) / kWaves) * 64u;
    const uint32_t lo = min(clo + wv * per_wave, chi), hi = min(lo + per_wave, chi);
    uint32_t tally = 0;
    for (uint32_t base = lo; base < hi; base += 64u) {
        const uint32_t s = base + lane;
        const uint32_t c = s < hi ? cls[order_strip_at(s, n_strips, bpr)] : 0xFFu;
        uint64_t todo = __ballot(c != 0xFFu);
        while (todo) {
            const uint32_t b = __builtin_amdgcn_readlane(c, __ffsll((unsigned long long)todo) - 1);
            const uint64_t m = __ballot(c == b);
            if (lane == b) tally += (uint32_t)__popcll(m);
            todo &= ~m;
        }
    }
    if (lane < kOrderBins) wave_tot[lane][wv] = tally;
    __syncthreads();
    uint32_t run = 0;  // rank (inside its class) of the wave's next strip of class `lane`
    if (lane < kOrderBins) {
        run = before[lane];
        for (uint32_t w = 0; w < wv; w++) run += wave_tot[lane][w];
    }
    for (uint32_t base = lo; base < hi; base += 64u) {
        const uint32_t s = order_strip_at(base + lane, n_strips, bpr);  // (positions past the chunk's end are not looked at)
        const uint32_t c = base + lane < hi ? cls[s] : 0xFFu;
        uint64_t todo = __ballot(c != 0xFFu);
        uint32_t rank = 0;
        while (todo) {
            const uint32_t b = __builtin_amdgcn_readlane(c, __ffsll((unsigned long long)todo) - 1);
            const uint64_t m = __ballot(c == b);
            const uint32_t first = __builtin_amdgcn_readlane(run, b);
            if (c == b)
                rank = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == b) run += (uint32_t)__popcll(m);
            todo &= ~m;
        }
        if (c != 0xFFu) {
            uint32_t list = 0;
#pragma unroll
            for (uint32_t k = 1; k < kOrderLists; k++) list += rank >= bound[c][k] ? 1u : 0u;
            const uint32_t at = list_base[c][list] + (rank - bound[c][list]);
            if (at < cap) sched[kOrderLists + list * cap + at] = s;  // (cap: order_list_cap, which bounds a list's share)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fs_main's shading (shader.wgsl:261-304) on top of the hit records.  The shadow ray of :275-280 is traced
// by the same trace kernels as an explicit ray (svo_trace_rays path): secondary_gen_kernel writes one ray per
// pixel (a ray that starts outside the cube pointing away -- an immediate miss -- for pixels that need none),
// the trace kernel produces a second set of records, and shade_kernel combines both into RGBA8.
// ---------------------------------------------------------------------------------------------
// Directions of the extra secondary rays (benchmark config 5; the reference itself only has the shadow ray):
// three 10-bit fields of an integer hash of (frame pixel id, k) as half-integer components in (-512, 512),
// normalised like every other direction, mirrored into the hemisphere of the hit normal.  Integer -> float
// conversions are exact and the operation order is fixed, so the oracle produces the same bits.
__device__ __forceinline__ uint32_t mix32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

// Rays k_first .. n_secondary-1 of every pixel (ray 0 is the shadow ray; with k_first = 1 it was traced inside the primary launch);
// slot of ray k of record r: (k - k_first) * n_records + r, in `rays`, `skip` and `out` alike.
__global__ __launch_bounds__(256) void secondary_gen_kernel(TraceArgs a, const float *aux_t, float *rays, uint8_t *skip, svo_hit *out,
                                                            uint32_t k_first, uint32_t n_secondary, uint32_t n_records) {
    float sun[3];
    sun_direction(a.u, sun);
    const float s0 = sun[0], s1 = sun[1], s2 = sun[2];
    const uint32_t width = (uint32_t)a.u.dimensions[0];
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < a.work.n_items; q += gridDim.x * 256u) {
        Item it = decode_item(a.work, q);
        if (!it.valid) continue;
        const uint4 rec = reinterpret_cast<const uint4 *>(a.hits)[it.out];
        const bool hit = (rec.z >> 16) & 1u;
        if (!hit) {
            // No secondary rays from this pixel.  fs_main traces none; the record set still has a slot for them, which holds
            // what the trace kernels write for a ray that never enters the cube: all zeros.  The trace skips these slots.
            for (uint32_t k = k_first; k < n_secondary; k++) {
                const uint64_t slot = (uint64_t)(k - k_first) * n_records + it.out;
                skip[slot] = 1u;
                reinterpret_cast<uint4 *>(out)[slot] = make_uint4(0u, 0u, 0u, 0u);
            }
            continue;
        }
        RayIn r = gen_ray(a.u, it.px, it.py);
        float pos[3], dir[3], dist;
        ray_enter(r, pos, dir, dist);
        float org[3], nrm[3];
        secondary_origin(pos, dir, rec.z & 0xFFu, rec.w, (rec.z & 0xFFu) != 0u ? aux_t[it.out] : 0.0f, org, nrm);
        const float n0 = nrm[0], n1 = nrm[1], n2 = nrm[2];
        for (uint32_t k = k_first; k < n_secondary; k++) {
            float d0, d1, d2;
            if (k == 0u) {
                d0 = -s0; d1 = -s1; d2 = -s2;  // the shadow ray, shader.wgsl:276
            } else {
                const uint32_t hsh = mix32((it.py * width + it.px) * 4u + k + 0x9E3779B9u);
                float e0 = (float)(int)(hsh & 1023u) - 511.5f, e1 = (float)(int)((hsh >> 10) & 1023u) - 511.5f,
                      e2 = (float)(int)((hsh >> 20) & 1023u) - 511.5f;
                const float len = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
                e0 = e0 / len; e1 = e1 / len; e2 = e2 / len;
                const bool flip = (n0 * e0 + n1 * e1) + n2 * e2 < 0.0f;
                d0 = flip ? -e0 : e0; d1 = flip ? -e1 : e1; d2 = flip ? -e2 : e2;
            }
            const uint64_t slot = (uint64_t)(k - k_first) * n_records + it.out;
            skip[slot] = 0u;
            float *dst = rays + 6ull * slot;
            dst[0] = org[0]; dst[1] = org[1]; dst[2] = org[2];
            dst[3] = d0; dst[4] = d1; dst[5] = d2;
        }
    }
}

__global__ __launch_bounds__(256) void shade_kernel(TraceArgs a, const svo_hit *shadow_hits, uint32_t *rgba) {
    const rsrc_t rs = make_rsrc(a.nodes, a.n_words);
    const float sl = sqrtf((a.u.sun_dir[0] * a.u.sun_dir[0] + a.u.sun_dir[1] * a.u.sun_dir[1]) + a.u.sun_dir[2] * a.u.sun_dir[2]);
    const float s0 = a.u.sun_dir[0] / sl, s1 = a.u.sun_dir[1] / sl, s2 = a.u.sun_dir[2] / sl;
    const float gamma = ((a.u.flags & SVO_F_MISC_BOOL) ? 1.0f : 0.0f) * -1.2f + 2.2f;  // :304
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < a.work.n_items; q += gridDim.x * 256u) {
        Item it = decode_item(a.work, q);
        if (!it.valid) continue;
        const uint4 rec = reinterpret_cast<const uint4 *>(a.hits)[it.out];
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        const uint32_t word = rec.x < a.n_words ? load_word(rs, rec.x) : 0u;  // sentinel indices read 0
        if (a.u.flags & SVO_F_SHOW_STEPS) {
            c0 = c1 = c2 = (float)(rec.z & 0xFFu) / 64.0f;                               // :264
        } else if ((rec.z >> 16) & 1u) {
            if (a.u.flags & SVO_F_SHOW_HITS) {
                c0 = c1 = c2 = (float)(word & 15u) / 15.0f;                              // :268
            } else {
                const float n0 = code_to_normal(rec.w & 3u), n1 = code_to_normal((rec.w >> 2) & 3u), n2 = code_to_normal((rec.w >> 4) & 3u);
                float diffuse = fmax_w((n0 * -s0 + n1 * -s1) + n2 * -s2, 0.0f);         // :273
                if (shadow_hits && ((reinterpret_cast<const uint4 *>(shadow_hits)[it.out].z >> 16) & 1u)) diffuse = 0.0f;  // :277-279
                const uint32_t value = (word >> 4) - kVoxelOffset;                       // :282
                const float k = 0.3f + diffuse;                                          // ambient + diffuse
                c0 = k * ((float)((value >> 16) & 0xFFu) / 255.0f);
                c1 = k * ((float)((value >> 8) & 0xFFu) / 255.0f);
                c2 = k * ((float)(value & 0xFFu) / 255.0f);
            }
        } else {
            c0 = c1 = c2 = 0.2f;                                                         // :287
        }
        c0 = powf(fmin_w(fmax_w(c0, 0.0f), 1.0f), gamma);
        c1 = powf(fmin_w(fmax_w(c1, 0.0f), 1.0f), gamma);
        c2 = powf(fmin_w(fmax_w(c2, 0.0f), 1.0f), gamma);
        const uint32_t r8 = (uint32_t)(c0 * 255.0f + 0.5f), g8 = (uint32_t)(c1 * 255.0f + 0.5f), b8 = (uint32_t)(c2 * 255.0f + 0.5f);
        rgba[it.out] = r8 | (g8 << 8) | (b8 << 16) | (128u << 24);  // alpha 0.5
    }
}

hipError_t launch_secondary_gen(const TraceArgs &args, const float *aux_t, float *rays, uint8_t *skip, svo_hit *out, uint32_t k_first,
                                uint32_t n_secondary, uint32_t n_records, hipStream_t stream) {
    (void)hipGetLastError();
    uint32_t blocks = (args.work.n_items + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(secondary_gen_kernel, dim3(blocks), dim3(256), 0, stream, args, aux_t, rays, skip, out, k_first, n_secondary, n_records);
    return hipGetLastError();
}

hipError_t launch_shade(const TraceArgs &args, const svo_hit *shadow_hits, uint32_t *rgba, hipStream_t stream) {
    (void)hipGetLastError();
    uint32_t blocks = (args.work.n_items + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(shade_kernel, dim3(blocks), dim3(256), 0, stream, args, shadow_hits, rgba);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Counter calibration (diagnostic, see profiles/README.md): the trace kernels' access pattern -- one
// buffer_load_dword per lane, every lane on its own cache line -- over a buffer far larger than the caches, so
// that FETCH_SIZE can be compared with a known number of distinct lines at two strides.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_gather_kernel(const uint32_t *buf, uint32_t n_words, uint32_t stride_words,
                                                          uint32_t n_loads, uint32_t *sink) {
    const rsrc_t rs = make_rsrc(buf, n_words);
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_loads; i += gridDim.x * 256u)
        acc += load_word(rs, i * stride_words);
    if (acc == 0x12345678u) *sink = acc;  // keep the loads alive
}

hipError_t launch_diag_gather(const uint32_t *buf, uint32_t n_words, uint32_t stride_words, uint32_t n_loads, uint32_t *sink,
                              hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(diag_gather_kernel, dim3(2048), dim3(256), 0, stream, buf, n_words, stride_words, n_loads, sink);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_build_top_table(const uint32_t *nodes, uint32_t n_words, uint32_t *top_table, hipStream_t stream) {
    (void)hipGetLastError();  // drop a stale error left by other users of the runtime in this thread
    const int cells = 1 << (3 * kTopLevels);
    hipLaunchKernelGGL(build_top_table_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, nodes, n_words,
                       top_table, kTopLevels);
    return hipGetLastError();
}

constexpr int kStackBlock = 256;
constexpr int kStackLevels = 12;      // default: resolves levels up to 3 + 1 + 12 = 16
constexpr int kStackLevelsDeep = 18;  // deep trees: up to level 22 = kPathBits - 1

int stack_max_depth(bool deep) { return kTopLevels + 1 + (deep ? kStackLevelsDeep : kStackLevels); }

template <bool GE, int NS>
static hipError_t launch_stack(const TraceArgs &args, const LaunchInfo &li, hipStream_t stream) {
    const uint32_t strip_items = args.order ? 64u : (li.strip_items ? li.strip_items : 64u);
    const bool shd = args.shadow_hits != nullptr;  // fused shadow rays (no timeline build of that one)
    auto kern = shd ? (args.count_nodes ? trace_stack_kernel<kStackBlock, NS, kTopLevels, GE, false, true, true>
                                        : trace_stack_kernel<kStackBlock, NS, kTopLevels, GE, false, false, true>)
                    : (args.count_nodes
                           ? (args.debug ? trace_stack_kernel<kStackBlock, NS, kTopLevels, GE, true, true, false>
                                         : trace_stack_kernel<kStackBlock, NS, kTopLevels, GE, false, true, false>)
                           : (args.debug ? trace_stack_kernel<kStackBlock, NS, kTopLevels, GE, true, false, false>
                                         : trace_stack_kernel<kStackBlock, NS, kTopLevels, GE, false, false, false>));
    const size_t lds_bytes = sizeof(uint32_t) * (size_t)(args.count_nodes ? StackLds<kStackBlock, NS, kTopLevels, true>::total
                                                                               : StackLds<kStackBlock, NS, kTopLevels, false>::total);
    // cached per context (= per device): [deep stack?][fused shadows?][counting instantiation?]
    int &blocks_per_cu = li.occupancy[(args.debug ? 8 : 0) + (NS == kStackLevelsDeep ? 4 : 0) + (shd ? 2 : 0) + (args.count_nodes ? 1 : 0)];
    if (blocks_per_cu == 0) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kStackBlock, lds_bytes);
        if (e != hipSuccess) return e;
        blocks_per_cu = n > 0 ? n : 1;
        // The occupancy query does not round LDS up to the allocation granule; measured on MI355X, 336 bytes on top of
        // 26 KiB per workgroup already cost the sixth workgroup per CU.  Workgroups that are not resident at launch
        // start when the first ones end and bring their reserved (longest) strips with them, so overestimating is
        // far worse than underestimating.  The granule is 1280 bytes (320 dwords): 512 bytes on top of this kernel's 22 KiB keep the
        // seventh workgroup per CU, 528 lose it (profiles/r04_lds_granule_ab.log).
        const int by_lds = (int)((160u * 1024u) / (((lds_bytes + 1279u) / 1280u) * 1280u));
        if (by_lds >= 1 && by_lds < blocks_per_cu) blocks_per_cu = by_lds;
    }
    uint32_t blocks = (uint32_t)li.num_cus * (uint32_t)blocks_per_cu;
    if (li.grid_blocks > 0) blocks = (uint32_t)li.grid_blocks;
    uint32_t n_strips = (args.work.n_items + strip_items - 1) / strip_items;
    uint32_t need = (n_strips + (kStackBlock / 64) - 1) / (kStackBlock / 64);
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kStackBlock), lds_bytes, stream, args, strip_items, li.work_counter,
                       li.defer);
    return hipGetLastError();
}

hipError_t launch_trace(const TraceArgs &args, const LaunchInfo &li, hipStream_t stream) {
    (void)hipGetLastError();
    if (args.work.n_items == 0) return hipSuccess;
    const bool counter_hits = (args.u.flags & SVO_F_PAUSE_ADAPTIVE) && (args.u.flags & SVO_F_SHOW_HITS);
    // the debug hit test reads counter bits: it needs the reference-shaped walk
    if (li.variant == SVO_VARIANT_RESTART || counter_hits) {
        uint32_t blocks = (args.work.n_items + 255u) / 256u;
        uint32_t cap = (uint32_t)li.num_cus * 8u;
        if (li.grid_blocks > 0) cap = (uint32_t)li.grid_blocks;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(trace_restart_kernel, dim3(blocks), dim3(256), 0, stream, args, (const uint32_t *)nullptr);
        return hipGetLastError();
    }
    // li.counters = {64 claim counters (kCounterStride words apart), deferred-ray count, deferred items}: zero when a frame
    // starts (armed at allocation and re-armed by the last kernel of the previous frame)
    const bool ge = (args.u.flags & SVO_F_MISC_BOOL) != 0;
    if (li.deep_stack)  // trees deeper than kTopLevels + 1 + kStackLevels: more LDS per workgroup, fewer resident waves
        return ge ? launch_stack<true, kStackLevelsDeep>(args, li, stream) : launch_stack<false, kStackLevelsDeep>(args, li, stream);
    return ge ? launch_stack<true, kStackLevels>(args, li, stream) : launch_stack<false, kStackLevels>(args, li, stream);
}

// Explicit rays with a skip mask (secondary rays: most slots of a frame can be empty): cost classes for THIS frame's
// schedule -- 0xFF, which strip_order_kernel leaves out of the lists, for strips without a single ray, else the class the
// strip had when costs were last measured (0 when there is no measurement: screen order) -- and the lists built from
// them.  Strips that are not in a list are never claimed, so empty slots cost the trace nothing.
__global__ __launch_bounds__(256) void strip_classes_kernel(const uint8_t *skip, uint32_t n_items, const uint8_t *prev, uint8_t *cls,
                                                            uint32_t n_strips) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_strips) return;
    bool empty = true;
    if ((uint64_t)s * 64u + 64u <= n_items) {
        const uint4 *p = reinterpret_cast<const uint4 *>(skip + (uint64_t)s * 64u);  // the mask is 256-byte aligned (hipMalloc)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = p[i];  // the producer writes 0 or 1
            empty = empty && v.x == 0x01010101u && v.y == 0x01010101u && v.z == 0x01010101u && v.w == 0x01010101u;
        }
    } else {
        for (uint32_t q = s * 64u; q < n_items; q++) empty = empty && skip[q] != 0u;
    }
    cls[s] = empty ? (uint8_t)0xFFu : (prev ? prev[s] : (uint8_t)0u);
}

// Strips whose rays all miss the cube (sky): decided per 64-pixel block from the four corner rays, before the trace, so
// that such strips are never claimed, generated or refilled from -- on frames that are mostly sky that is most of the
// per-strip work.  The rays of a block are pos + s * (point(pixel) - pos), s > 0, where point() is the
// projective image of the pixel under camera_inverse: the block's points lie in the planar convex quadrilateral Q of its
// four corner pixels (same sign of w at the corners), so every ray lies in the cone over Q with apex pos.  If one side
// plane of that cone (through pos and an edge of Q, normal n pointing into the cone) has the whole cube strictly on its
// outer side -- max over the cube's corners of n.(v - pos) = |n.x| + |n.y| + |n.z| - n.pos < -margin -- no ray of the block
// meets the cube and ray_box_dist returns 0 for each of them (shader.wgsl:66-80: v7 > v8).  The margin (1e-3 of the
// plane function's scale, ~1e-3 rad) is four orders of magnitude above the rounding of either computation: blocks
// anywhere near the cube's silhouette are NOT culled and take the ordinary path.  A culled strip's 64 records are what
// the trace writes for rays that never enter the cube: all zeros.
// (One LANE per strip for the test -- the pixel bounds of a block follow from its position, no reduction over its pixels is
// needed -- and one wave-wide pass per culled strip for its 64 zero records: with one wave per strip, every lane repeating the
// four corner rays, the pass took 32 us per 1080p frame, a third of the trace it saves on.)
__global__ __launch_bounds__(256) void strip_cull_kernel(TraceArgs a, const uint8_t *prev, uint8_t *cls, uint32_t n_strips) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = gridDim.x * 4u;
    const WorkDesc &w = a.work;
    float p4[4];
    mat_vec(a.u.camera_inverse, 0.0f, 0.0f, 0.0f, 1.0f, p4);
    const float o0 = p4[0] / p4[3], o1 = p4[1] / p4[3], o2 = p4[2] / p4[3];
    for (uint32_t base = wave * 64u; base < n_strips; base += n_waves * 64u) {
        const uint32_t s = base + lane;
        bool culled = false;
        if (s < n_strips) {
            // the block's first pixel and how many of its columns and rows lie inside the rectangle
            const ItemFast it0 = decode_item_fast(w, s * 64u);
            uint32_t x_lo = 0u, x_hi = 0u, y_lo = 0u, y_hi = 0u;
            bool any = it0.valid;
            if (any) {
                const uint32_t blk = s, rect = w.n_rects > 1u ? fast_div(blk, w.bprect, w.magic_bprect) : 0u, b = blk - rect * w.bprect;
                const uint32_t by = fast_div(b, w.bpr, w.magic_bpr), bx = b - by * w.bpr;
                const uint32_t x = bx << w.bw_log2, y = by << (6u - w.bw_log2);
                x_lo = it0.px; y_lo = it0.py;
                x_hi = it0.px + min((1u << w.bw_log2) - 1u, w.w - 1u - x);
                y_hi = it0.py + min((1u << (6u - w.bw_log2)) - 1u, w.h - 1u - y);
            }
            if (any) {
                // corner k of the quadrilateral, in order around it: (lo,lo) (hi,lo) (hi,hi) (lo,hi)
                float q[4][3];
                bool ok = true;
                float wsign = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t px = (k == 1 || k == 2) ? x_hi : x_lo, py = (k >= 2) ? y_hi : y_lo;
                    const float cx = ((float)px + 0.5f) / a.u.dimensions[0] * 2.0f - 1.0f;
                    const float cy = -(((float)py + 0.5f) / a.u.dimensions[1] * 2.0f - 1.0f);
                    float d4[4];
                    mat_vec(a.u.camera_inverse, cx, cy, 1.0f, 1.0f, d4);
                    q[k][0] = d4[0] / d4[3] - o0; q[k][1] = d4[1] / d4[3] - o1; q[k][2] = d4[2] / d4[3] - o2;
                    ok = ok && fabsf(d4[3]) > 1.0e-20f && (k == 0 || (d4[3] > 0.0f) == (wsign > 0.0f));
                    wsign = d4[3];
                }
                ok = ok && fabsf(p4[3]) > 1.0e-20f;
                if (ok) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float *u = q[k], *v = q[(k + 1) & 3], *ww = q[(k + 2) & 3];
                        float n0 = u[1] * v[2] - u[2] * v[1], n1 = u[2] * v[0] - u[0] * v[2], n2 = u[0] * v[1] - u[1] * v[0];
                        const float inside = n0 * ww[0] + n1 * ww[1] + n2 * ww[2];  // the opposite corner is inside the cone
                        if (inside < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
                        const float reach = fabsf(n0) + fabsf(n1) + fabsf(n2);
                        const float at_pos = n0 * o0 + n1 * o1 + n2 * o2;
                        const float scale = reach + fabsf(n0 * o0) + fabsf(n1 * o1) + fabsf(n2 * o2);
                        // a degenerate quadrilateral (inside == 0: a one-pixel-wide strip) or NaNs leave every comparison false
                        if (fabsf(inside) > 0.0f && reach - at_pos < -1.0e-3f * scale) culled = true;
                    }
                }
            }
            cls[s] = culled ? (uint8_t)0xFFu : (prev ? prev[s] : (uint8_t)0u);
        }
        // the zero records of the culled strips, one strip at a time, one record per lane
        uint64_t todo = __ballot(culled);
        while (todo) {
            const uint32_t t = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
            todo &= todo - 1ull;
            const ItemFast it = decode_item_wave(w, (base + t) * 64u, lane);
            if (it.valid) {
                reinterpret_cast<uint4 *>(a.hits)[it.out] = make_uint4(0u, 0u, 0u, 0u);
                if (a.aux_t) a.aux_t[it.out] = 0.0f;
                if (a.shadow_hits) reinterpret_cast<uint4 *>(a.shadow_hits)[it.out] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

// blocks per row for the schedule builder's column-major ranks (0: rank in strip order): frames of one rectangle of whole pixel blocks
#ifndef SVO_ORDER_COLMAJOR
#define SVO_ORDER_COLMAJOR 1
#endif
static uint32_t order_bpr(const WorkDesc &w, uint32_t n_strips) {
    if (!SVO_ORDER_COLMAJOR || w.mode == 2 || w.n_rects > 1u || w.bpr == 0u || n_strips % w.bpr != 0u || n_strips * 64u != w.n_items) return 0u;
    return w.bpr;
}

// the two kernels of the schedule builder for the class bytes `cls` (the chunk histograms live behind them)
static void launch_order_pair(const WorkDesc &w, const uint8_t *cls, uint32_t *sched, uint32_t n_strips, uint32_t cap, hipStream_t stream,
                              const uint32_t *shares = nullptr) {
    uint32_t *hist = reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(cls) + ((n_strips + 15u) & ~15u));
    const uint32_t bpr = order_bpr(w, n_strips);
    hipLaunchKernelGGL(strip_hist_kernel, dim3(kOrderBlocks), dim3(kOrderThreads), 0, stream, cls, n_strips, hist, bpr);
    hipLaunchKernelGGL(strip_order_kernel, dim3(kOrderBlocks), dim3(kOrderThreads), 0, stream, cls, (const uint32_t *)hist, sched, n_strips, cap, bpr, shares);
}

// This frame's strip lists without the culled strips (classes from `prev`, or screen order); see launch_schedule_skipping.
hipError_t launch_schedule_culling(const TraceArgs &args, const uint8_t *prev, uint8_t *cls, uint32_t *sched, uint32_t n_strips,
                                   uint32_t cap, hipStream_t stream) {
    (void)hipGetLastError();
    uint32_t blocks = (n_strips + 255u) / 256u;  // a lane per strip
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(strip_cull_kernel, dim3(blocks), dim3(256), 0, stream, args, prev, cls, n_strips);
    launch_order_pair(args.work, cls, sched, n_strips, cap, stream);
    return hipGetLastError();
}

hipError_t launch_schedule_skipping(const uint8_t *skip, uint32_t n_items, const uint8_t *prev, uint8_t *cls, uint32_t *sched,
                                    uint32_t n_strips, uint32_t cap, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(strip_classes_kernel, dim3((n_strips + 255u) / 256u), dim3(256), 0, stream, skip, n_items, prev, cls, n_strips);
    uint32_t *hist = reinterpret_cast<uint32_t *>(cls + ((n_strips + 15u) & ~15u));  // same layout as the cost buffer
    hipLaunchKernelGGL(strip_hist_kernel, dim3(kOrderBlocks), dim3(kOrderThreads), 0, stream, (const uint8_t *)cls, n_strips, hist, 0u);
    hipLaunchKernelGGL(strip_order_kernel, dim3(kOrderBlocks), dim3(kOrderThreads), 0, stream, (const uint8_t *)cls,
                       (const uint32_t *)hist, sched, n_strips, cap, 0u, (const uint32_t *)nullptr);  // (explicit rays: strip order)
    return hipGetLastError();
}

// After the STACK kernel: deferred rays, per-strip cost classes (cost != nullptr), counter re-arm; and, when
// `build_schedule`, the strip lists for the next frames.
// Cost classes for a camera that MOVES.  The strips that decide a frame hold a ray that runs into the step limit; those rays
// are isolated pixels, and which pixels they are changes with a sub-pixel change of the view: the class a strip
// had one frame ago says little about whether it holds one now.  Where they can be does carry over -- they come in regions
// (grazing views of the terrain), other regions (sky, near surfaces seen head-on) have none.  A strip with at least
// `min_count` such strips among its (2 radius + 1)^2 neighbours is therefore given at least class `floor_class`: "cheap, but
// one in ten of its kind turns out to take the whole frame" sorts before "cheap" without getting ahead of the strips that
// were measured long.
__global__ __launch_bounds__(256) void strip_danger_kernel(const uint8_t *in, uint8_t *out, uint32_t n_strips, uint32_t bpr, int radius,
                                                           uint32_t min_count, uint32_t floor_class) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_strips) return;
    const int by = (int)(s / bpr), bx = (int)(s - (uint32_t)by * bpr), rows = (int)((n_strips + bpr - 1u) / bpr);
    uint32_t c = in[s], near = 0u;
    if (c < floor_class) {
        for (int dy = -radius; dy <= radius; dy++)
            for (int dx = -radius; dx <= radius; dx++) {
                const int y = by + dy, x = bx + dx;
                if (y < 0 || y >= rows || x < 0 || x >= (int)bpr) continue;
                const uint32_t t = (uint32_t)y * bpr + (uint32_t)x;
                if (t < n_strips && in[t] >= (64u >> kCostShift)) near++;  // a ray of 64 steps or more
            }
        if (near >= min_count) c = floor_class;
    }
    out[s] = (uint8_t)c;
}

hipError_t launch_post(const TraceArgs &args, const LaunchInfo &li, uint8_t *cost, uint32_t *sched, uint32_t n_strips,
                       uint32_t cap, bool build_schedule, hipStream_t stream, uint8_t *moved, uint32_t motion_floor, uint32_t balance_update,
                       bool reuse_cost) {
    // without the cost pass the launch only re-arms counters and traces deferred rays: normally none, but a frame
    // full of them (every ray NaN / extreme) must not crawl through 16 workgroups
    // (reuse_cost: the classes in `cost` are this input's already -- a resting view whose lists are rebuilt for the shares' sake only)
    uint8_t *const measure = reuse_cost ? nullptr : cost;
    uint32_t blocks = measure ? (n_strips + 3u) / 4u : 256u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks < 16u) blocks = 16u;
    hipLaunchKernelGGL(post_kernel, dim3(blocks), dim3(256), 0, stream, args, li.counters, (const uint32_t *)li.defer,
                       li.next_defer_count, measure, n_strips, balance_update);
    if (cost && build_schedule) {
        const uint8_t *cls = cost;
        if (moved && motion_floor) {  // (pixel frames of one rectangle: the ABI passes `moved` only then)
            hipLaunchKernelGGL(strip_danger_kernel, dim3((n_strips + 255u) / 256u), dim3(256), 0, stream, (const uint8_t *)cost, moved, n_strips,
                               args.work.bpr, (int)((motion_floor >> 8) & 15u), (motion_floor >> 12) & 255u, ((motion_floor & 15u) << 3) >> kCostShift);  // (the option counts the floor in units of 8 steps)
            cls = moved;
        }
        // (the chunk histograms live behind the class bytes: the ABI allocates kOrderHistWords extra words)
        launch_order_pair(args.work, cls, sched, n_strips, cap, stream, args.balance);
    }
    return hipGetLastError();
}

// Un-permute a gathered, tile-sharded frame (rank r, slot k holds tile r + k * world, row-major inside the tile) into
// the row-major frame: one 16-byte record per thread, 64-pixel tile rows stay contiguous on both sides.  PACKED: the
// gathered records are the 12-byte wire form (pack_records_kernel); the fourth word is rebuilt from the third.
template <bool PACKED>
__global__ __launch_bounds__(256) void assemble_tiles_kernel(const uint32_t *gathered, uint4 *frame, uint32_t world, uint32_t n_pad,
                                                             uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h) {
    const uint32_t tiles_x = width / tile_w, n = width * height;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t y = i / width, x = i - y * width;
        const uint32_t ty = y / tile_h, tx = x / tile_w;
        const uint32_t t = ty * tiles_x + tx, r = t % world, k = t / world;
        const uint64_t src = ((uint64_t)r * n_pad + k) * (tile_w * tile_h) + (y - ty * tile_h) * tile_w + (x - tx * tile_w);
        if (PACKED) {
            const uint32_t *p = gathered + 3u * src;
            const uint32_t info = p[2];
            frame[i] = make_uint4(p[0], p[1], info, (info >> 17) & 63u);
        } else {
            frame[i] = reinterpret_cast<const uint4 *>(gathered)[src];
        }
    }
}

// 16-byte records -> 12-byte wire records for the frame-end gather: the packed normal (word 3) is a copy of bits
// 17..22 of word 2 (write_hit), so dropping it loses nothing and saves a quarter of the bytes on the links.
__global__ __launch_bounds__(256) void pack_records_kernel(const uint4 *records, uint32_t *wire, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint4 r = records[i];
        wire[3u * i + 0u] = r.x;
        wire[3u * i + 1u] = r.y;
        wire[3u * i + 2u] = r.z;
    }
}

// The same un-permute for a gathered COLOUR frame (one RGBA8 word per pixel): the exchange of a sharded frame whose
// consumer wants the image fs_main produces, not hit records -- 4 bytes per ray on the links instead of 12.
__global__ __launch_bounds__(256) void assemble_tiles_rgba_kernel(const uint32_t *gathered, uint32_t *frame, uint32_t world, uint32_t n_pad,
                                                                  uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h) {
    const uint32_t tiles_x = width / tile_w, n = width * height;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t y = i / width, x = i - y * width;
        const uint32_t ty = y / tile_h, tx = x / tile_w;
        const uint32_t t = ty * tiles_x + tx, r = t % world, k = t / world;
        frame[i] = gathered[((uint64_t)r * n_pad + k) * (tile_w * tile_h) + (y - ty * tile_h) * tile_w + (x - tx * tile_w)];
    }
}

hipError_t launch_assemble_tiles_rgba(const uint32_t *gathered, uint32_t *frame, uint32_t world, uint32_t n_pad, uint32_t width,
                                      uint32_t height, uint32_t tile_w, uint32_t tile_h, hipStream_t stream) {
    (void)hipGetLastError();
    uint32_t blocks = (width * height + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(assemble_tiles_rgba_kernel, dim3(blocks), dim3(256), 0, stream, gathered, frame, world, n_pad, width, height, tile_w,
                       tile_h);
    return hipGetLastError();
}

hipError_t launch_assemble_tiles(const void *gathered, bool packed, svo_hit *frame, uint32_t world, uint32_t n_pad, uint32_t width,
                                 uint32_t height, uint32_t tile_w, uint32_t tile_h, hipStream_t stream) {
    (void)hipGetLastError();
    uint32_t blocks = (width * height + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    if (packed)
        hipLaunchKernelGGL(assemble_tiles_kernel<true>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(gathered),
                           reinterpret_cast<uint4 *>(frame), world, n_pad, width, height, tile_w, tile_h);
    else
        hipLaunchKernelGGL(assemble_tiles_kernel<false>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(gathered),
                           reinterpret_cast<uint4 *>(frame), world, n_pad, width, height, tile_w, tile_h);
    return hipGetLastError();
}

hipError_t launch_pack_records(const svo_hit *records, uint32_t *wire, uint32_t n, hipStream_t stream) {
    (void)hipGetLastError();
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(pack_records_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(records), wire, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scatter_kernel(uint32_t *nodes, uint32_t n_words, const uint32_t *indices,
                                                      const uint32_t *words, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t at = indices[i];
        if (at < n_words) nodes[at] = words[i];
    }
}

hipError_t launch_scatter(uint32_t *nodes, uint32_t n_words, const uint32_t *indices, const uint32_t *words, uint32_t n,
                          hipStream_t stream) {
    (void)hipGetLastError();
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, stream, nodes, n_words, indices, words, n);
    return hipGetLastError();
}

hipError_t launch_scan(uint32_t *nodes, uint32_t n_words, uint32_t node_length, uint32_t *sub, uint32_t *unsub,
                       uint32_t capacity, bool clear_counters, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_words == 0) return hipSuccess;
    uint32_t blocks = (n_words + kScanChunk - 1u) / kScanChunk;
    if (blocks > 1024u) blocks = 1024u;  // 1024 chunks of 32 KiB in flight = the L2 capacity (second pass re-reads them)
    hipLaunchKernelGGL(scan_kernel, dim3(blocks), dim3(256), 0, stream, nodes, n_words, node_length, sub, unsub, capacity,
                       clear_counters);
    return hipGetLastError();
}

}  // namespace svo
