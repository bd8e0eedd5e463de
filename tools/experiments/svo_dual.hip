// svo_dual.hip -- round 3's experiment (builds against commit 35128d4, the head of round 3): the STACK traversal with TWO rays per lane, software-pipelined, over a
// child-mask table, for static trees.  It does what it was built for -- waves wait at s_waitcnt half as long, a DDA step costs 1.05
// loads instead of 1.8 -- and is 2.5 x SLOWER than trace_stack_kernel on the benchmark frame: 1.75 x the instructions per frame, and a
// ray advances once per loop iteration of two turns, which stretches the serial chain of the 101-step rays past the whole frame.  It
// stays selectable (SVO_VARIANT_DUAL) under the parity tests and campaigns so that the measurement can be repeated; the default
// kernel does not use this file's table either (SVO_VARIANT_ETAB does: svo_kernels.hip, template flag ET).
//
// What bounds trace_stack_kernel (svo_kernels.hip) is one dependent load per octree level with nothing else for the
// wave to do: every lane waits for the deepest lane's walk, 550 cycles per level.  Two changes here,
// neither of which touches octree_ray's arithmetic (shader.wgsl:227-244 stays operation for operation):
//
//  1. A derived table E, one word per node word, built on the device when the node array is uploaded
//     (build_etab_kernel).  For an interior word p with child group G:  E[p] = (G >> 3) << 8 | empty-mask, where bit c
//     of the mask says that child c of G is an EMPTY LEAF (nodes[G + c] >> 4 == VOXEL_OFFSET); for a leaf word p:
//     E[p] = 0xFFFFFF00 | solid.  find_voxel (shader.wgsl:130-171) stops at the first leaf word on the path; with
//     the mask of a group in hand, a ray that steps into an empty child -- which is what nearly every DDA step does --
//     knows its new leaf (level, cell) without reading that leaf's word: one dependent load less per walk, and a step
//     between siblings needs none at all.  Same leaf as find_voxel by construction (the mask is the leaf test of the
//     same words); the node array itself, its layout and the host API are untouched (LAYOUT.md, octree.rs:5-35).
//     Requires child groups to be 8-aligned and inside the buffer, which Octree::subdivide guarantees
//     (octree.rs:72-90); the builder raises a flag otherwise and the one-ray kernel, which is general, takes over.
//
//  2. Every lane carries two rays, A and B, and the loop alternates between them: a ray's turn is
//     [take the word its load brought] -> [child of the next level: empty leaf, or a load] -> [DDA step if it is at an
//     empty leaf] -> [restart level from the ancestor stack, child, empty leaf or a load] -> [issue the load], and the
//     load travels while the lane works on its other ray.  No lane waits for another lane's walk (a ray that has to go
//     n levels down takes max(1, n) turns for that step), and no wave sits in s_waitcnt behind a single chain.
//
// Rays, records, strip claiming, schedule lists, deferred (unclean) rays: as in trace_stack_kernel.
#include <hip/hip_runtime.h>

#include "svo_device.h"
#include "svo_trace_fn.h"

namespace svo {

constexpr uint32_t kEMark = 0xFFFFFF00u;  // E word of a leaf (bit 0: solid); top-table entries add the leaf's level << 1

// ---------------------------------------------------------------------------------------------
// E table and its top table
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_etab_kernel(const uint32_t *nodes, uint32_t n_words, uint32_t *etab, uint32_t *flag) {
    bool bad = false;
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n_words; p += gridDim.x * 256u) {
        const uint32_t tn = nodes[p] >> 4;
        uint32_t e;
        if (tn >= kVoxelOffset) {
            e = kEMark | (tn != kVoxelOffset ? 1u : 0u);
        } else if ((tn & 7u) != 0u || (uint64_t)tn + 8u > n_words || (tn >> 3) >= 0x00FFFFFFu) {
            bad = true;  // a child group the 24-bit form cannot name: this array goes to the general kernel
            e = kEMark;
        } else {
            const uint4 lo = *reinterpret_cast<const uint4 *>(nodes + tn), hi = *reinterpret_cast<const uint4 *>(nodes + tn + 4u);
            const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t mask = 0u;
#pragma unroll
            for (int c = 0; c < 8; c++) mask |= ((w[c] >> 4) == kVoxelOffset ? 1u : 0u) << c;
            e = ((tn >> 3) << 8) | mask;
        }
        etab[p] = e;
    }
    if (bad) atomicOr(flag, 1u);
}

// One entry per level-K cell (index cx << 2K | cy << K | cz), then one leaf index per cell:
//   below an interior level-K word: that word's E entry (the child group of level K+1 and its empty-mask);
//   inside a leaf of level l <= K:  kEMark | l << 1 | solid, and the leaf's word index in the second half.
__global__ __launch_bounds__(256) void build_etop_kernel(const uint32_t *nodes, const uint32_t *etab, uint32_t n_words, uint32_t *etop,
                                                         int top_levels) {
    const rsrc_t rs = make_rsrc(nodes, n_words), re = make_rsrc(etab, n_words);
    const uint32_t cell = blockIdx.x * 256u + threadIdx.x, cells = 1u << (3 * top_levels);
    if (cell >= cells) return;
    const uint32_t cx = (cell >> (2 * top_levels)) & ((1u << top_levels) - 1u);
    const uint32_t cy = (cell >> top_levels) & ((1u << top_levels) - 1u);
    const uint32_t cz = cell & ((1u << top_levels) - 1u);
    uint32_t node_index = 0u, entry = 0u, leaf = 0u;
    for (int lvl = 1; lvl <= top_levels; lvl++) {
        const int sh = top_levels - lvl;
        const uint32_t child = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        const uint32_t p = node_index + child;
        const uint32_t tn = load_word(rs, p) >> 4;
        if (tn >= kVoxelOffset) {
            entry = kEMark | ((uint32_t)lvl << 1) | (tn != kVoxelOffset ? 1u : 0u);
            leaf = p;
            break;
        }
        node_index = tn;
        entry = load_word(re, p);
    }
    etop[cell] = entry;
    etop[cells + cell] = leaf;
}

hipError_t launch_build_etab(const uint32_t *nodes, uint32_t n_words, uint32_t *etab, uint32_t *etop, uint32_t *flag, hipStream_t stream) {
    (void)hipGetLastError();
    uint32_t blocks = (n_words + 255u) / 256u;
    if (blocks > 16384u) blocks = 16384u;
    if (blocks == 0u) blocks = 1u;
    hipLaunchKernelGGL(build_etab_kernel, dim3(blocks), dim3(256), 0, stream, nodes, n_words, etab, flag);
    const int cells = 1 << (3 * kTopLevels);
    hipLaunchKernelGGL(build_etop_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, nodes, (const uint32_t *)etab, n_words, etop,
                       kTopLevels);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------
// Per-ray state word: the bits of svo_trace_fn.h (steps, level of the child group / leaf, step mask, ENTRY, PENDING, F_*), plus
constexpr uint32_t D_WAIT = 1u << 18;     // a load of this ray is in flight (its answer is in `pe` next turn)
constexpr uint32_t D_LEAF = 1u << 19;     // at an empty leaf of level L: takes a DDA step this turn
constexpr uint32_t D_ADV = 1u << 20;      // picked up this turn: `cur` / L are set, the child of level L is still to be looked at
constexpr uint32_t D_TOPLEAF = 1u << 25;  // ended in a solid leaf above level K+1: its index is in the top table's second half

struct DRay {
    uint32_t st, out;   // out: bits 0..25 output index, 26..31 entry normal code
    float P0, P1, P2, Dr0, Dr1, Dr2, Y0, Y1, Y2, K0, K1, K2, dist, tcur;
    int32_t ix, iy, iz;
    uint32_t cur;       // E entry of the child group the ray is in at level L: (G >> 3) << 8 | empty-mask
    uint32_t poff;      // byte offset of that word (the leaf's index << 2 when it turns out to be a solid leaf)
};

template <int BLOCK, int NS, int K, bool GE, bool DBG>
__global__ __launch_bounds__(BLOCK, 4) void trace_dual_kernel(TraceArgs a, uint32_t strip_items, uint32_t *work_counter, uint32_t *defer) {
    constexpr int D = kPathBits;
    constexpr int SBASE = K + 2;      // first level whose child group is kept on the LDS stack
    constexpr int SMAX = K + 1 + NS;  // deepest level resolved
    static_assert(SMAX <= D - 1, "stack deeper than the path codes");
    constexpr int TBL = 1 << (3 * K);
    constexpr float kScale = 8388608.0f;  // 2^23
    constexpr uint32_t kNoLoad = 0x7FFFFFF0u;  // byte offset past any buffer: the load returns 0 without touching memory
    extern __shared__ uint32_t lds[];
    uint32_t *tbl = lds;                             // TBL entries
    // lds + TBL: the two ancestor stacks [2][NS][BLOCK], row l - SBASE = E entry of the ray's child group of level l
    uint32_t *pool_all = lds + TBL + 2 * NS * BLOCK;  // [BLOCK / 64][kPoolWords][64]

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t *pool = pool_all + (tid >> 6) * (kPoolWords * 64);
    const uint32_t col_a = (uint32_t)TBL + tid, col_b = (uint32_t)TBL + (uint32_t)(NS * BLOCK) + tid;  // row 0 of the lane's two stack columns (index into lds)
    const rsrc_t re = make_rsrc(a.etab, a.n_words);

    for (uint32_t i = tid; i < (uint32_t)TBL; i += BLOCK) tbl[i] = a.top_table[i];
    __syncthreads();

    const uint32_t n_items = a.work.n_items;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + tid) >> 6);

    // ---- strip claiming: exactly trace_stack_kernel's (8 lists x 8 counters, first strip reserved, claims issued early) ----
    constexpr uint32_t kShards = 8, kShardStride = 32, kSubs = 8;
    const uint32_t n_strips = (n_items + strip_items - 1) / strip_items;
    const uint32_t *order = a.order;
    uint32_t home = (blockIdx.x % kShards) * kSubs + (blockIdx.x / kShards) % kSubs;
    uint32_t pend = 0;
    uint32_t next, strip_end;
    auto entry_of = [=](uint32_t sh, uint32_t k) -> uint32_t {
        if (order) return k < order[sh] ? order[kShards + sh * a.order_cap + k] : 0xFFFFFFFFu;
        const uint32_t cand = (((k >> 4) * kShards + sh) << 4) | (k & 15u);
        return cand < n_strips ? cand : 0xFFFFFFFFu;
    };
    {
        const uint32_t my_rank = (blockIdx.x / kShards) * (uint32_t)(BLOCK / 64) + __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t s0 = __builtin_amdgcn_readfirstlane(entry_of(blockIdx.x % kShards, my_rank));
        if (s0 != 0xFFFFFFFFu) {
            next = s0 * strip_items;
            strip_end = min(next + strip_items, n_items);
        } else {
            pend = 0x00FFFFFFu;
            next = strip_end = 0xFFFFFFFEu;
        }
    }
    uint32_t pool_n = 0, pool_i = 0;
    uint64_t t_begin = 0, t_dry = 0, c_mark = 0;
    uint32_t n_iters = 0, dbg_slots = 0, dbg_refills = 0, dbg_gens = 0, c_refill = 0, c_ta = 0, c_tb = 0, c_gen = 0, dbg_steps = 0, dbg_loads = 0,
             dbg_done = 0, c_claim = 0;
    if (DBG) t_begin = __builtin_amdgcn_s_memrealtime();

    DRay A, B;
    auto clear = [](DRay &r) {
        r.st = 0u; r.out = 0u;
        r.P0 = r.P1 = r.P2 = 0.0f; r.Dr0 = r.Dr1 = r.Dr2 = 1.0f; r.Y0 = r.Y1 = r.Y2 = 1.0f; r.K0 = r.K1 = r.K2 = 0.0f;
        r.dist = r.tcur = 0.0f;
        r.ix = r.iy = r.iz = 0;
        r.cur = 0u; r.poff = 0u;
    };
    clear(A);
    clear(B);
    uint32_t pe_a = 0u, pe_b = 0u;

    auto top_cell = [&](const DRay &r) -> uint32_t {
        return ((uint32_t)(r.ix >> (D - K)) << (2 * K)) | ((uint32_t)(r.iy >> (D - K)) << K) | (uint32_t)(r.iz >> (D - K));
    };
    auto finish = [](DRay &r, uint32_t how) { r.st = (r.st & ~(ST_ACTIVE | D_WAIT | D_LEAF | D_ADV)) | ST_PENDING | how; };

    // A top-table entry for the ray's level-K cell: a leaf that covers the cell (empty: step from it; solid: the ray ends), or the
    // child group of level K+1.  true: `cur` / L are set and the child of level L is to be looked at.
    auto enter_top = [&](DRay &r, uint32_t e) -> bool {
        if (e >= kEMark) {
            r.st = (r.st & ~ST_L_MASK) | (((e >> 1) & 3u) << ST_L_SHIFT);
            if (e & 1u) finish(r, ST_F_SOLID | D_TOPLEAF);
            else r.st |= D_LEAF;
            return false;
        }
        r.cur = e;
        r.st = (r.st & ~ST_L_MASK) | ((uint32_t)(K + 1) << ST_L_SHIFT);
        return true;
    };

    // the child of level L the ray's position selects: an empty leaf (the ray is AT it: no load), or the word to read
    auto advance = [&](DRay &r, bool &want, uint32_t &off) {
        const uint32_t sh = (uint32_t)D - ((r.st >> ST_L_SHIFT) & 31u);
        const uint32_t c = ((((uint32_t)r.ix >> sh) & 1u) << 2) | ((((uint32_t)r.iy >> sh) & 1u) << 1) | (((uint32_t)r.iz >> sh) & 1u);
        if ((r.cur >> c) & 1u) {
            r.st |= D_LEAF;
        } else {
            want = true;
            off = ((r.cur >> 3) & ~31u) | (c << 2);  // ((G >> 3) << 3 + c) * 4
        }
    };

    // write the record of a finished ray (at the next refill of its set, for many lanes at once)
    auto flush_record = [&](DRay &r) {
        const uint32_t st = r.st;
        const bool too_deep = (st & ST_F_TOODEEP) != 0u, solid = (st & ST_F_SOLID) != 0u, inb = (st & ST_F_INB) != 0u;
        const bool stop_here = too_deep || solid;
        const uint32_t L = (st >> ST_L_SHIFT) & 31u, nm = (st >> ST_M_SHIFT) & 7u;
        const uint32_t c0n = (r.Dr0 > 0.0f) ? 2u : 1u, c1n = (r.Dr1 > 0.0f) ? 2u : 1u, c2n = (r.Dr2 > 0.0f) ? 2u : 1u;
        uint32_t ncode = ((nm & 1u) ? c0n : 0u) | ((nm & 2u) ? (c1n << 2) : 0u) | ((nm & 4u) ? (c2n << 4) : 0u);
        if (st & ST_ENTRY) ncode = r.out >> 26;    // no step taken: the entry normal
        if (!stop_here && !inb) ncode = 0u;         // left the cube: the miss record carries no normal
        uint32_t leaf_index = r.poff >> 2;
        if (st & D_TOPLEAF) leaf_index = a.top_table[(uint32_t)TBL + top_cell(r)];
        const uint32_t value = too_deep ? 0xFF000000u : (solid ? leaf_index : (!inb ? 0x20202000u : 0xFF000000u));
        const uint32_t depth = (too_deep || (!solid && inb)) ? 100u : L;
        const uint32_t hit = (stop_here || inb) ? 1u : 0u;
        write_hit(a.hits, r.out & 0x03FFFFFFu, value, r.dist + r.tcur, st & 0xFFu, depth, hit, ncode);
        if (a.aux_t) a.aux_t[r.out & 0x03FFFFFFu] = r.tcur;
        r.st = 0u;
    };

    // idle lanes of one set write the record of the ray they finished and take rays pool_i .. from the wave's pool
    auto take = [&](DRay &r, uint64_t act, uint32_t n_idle) {
        if (r.st & ST_PENDING) flush_record(r);
        if (!(r.st & ST_ACTIVE)) {
            const uint64_t idle = ~act;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (rank < pool_n) {
                const uint32_t e = pool_i + rank;
                r.P0 = __uint_as_float(pool[0 * 64 + e]);
                r.P1 = __uint_as_float(pool[1 * 64 + e]);
                r.P2 = __uint_as_float(pool[2 * 64 + e]);
                r.Dr0 = __uint_as_float(pool[3 * 64 + e]);
                r.Dr1 = __uint_as_float(pool[4 * 64 + e]);
                r.Dr2 = __uint_as_float(pool[5 * 64 + e]);
                r.Y0 = __uint_as_float(pool[6 * 64 + e]);
                r.Y1 = __uint_as_float(pool[7 * 64 + e]);
                r.Y2 = __uint_as_float(pool[8 * 64 + e]);
                r.K0 = copysign_bits(0.000002f * 8388608.0f, r.Dr0);
                r.K1 = copysign_bits(0.000002f * 8388608.0f, r.Dr1);
                r.K2 = copysign_bits(0.000002f * 8388608.0f, r.Dr2);
                r.dist = __uint_as_float(pool[9 * 64 + e]);
                r.out = pool[10 * 64 + e];
                // entry path codes (the position may sit a rounding error outside the cube: clamp)
                r.ix = entry_code<GE>(r.P0);
                r.iy = entry_code<GE>(r.P1);
                r.iz = entry_code<GE>(r.P2);
                r.tcur = 0.0f;
                r.st = ST_ACTIVE | ST_ENTRY;  // steps = 0
                if (enter_top(r, tbl[top_cell(r)])) r.st |= D_ADV;
            }
        }
        const uint32_t took = min(n_idle, pool_n);
        pool_i += took;
        pool_n -= took;
    };

    // One turn of one ray set (see the file header).
    auto turn = [&](DRay &r, uint32_t &pe, const uint32_t col) {
        const uint32_t e = pe;  // the E word the ray's load of the previous turn brought (looked at only when one was wanted)
        bool want = false;
        uint32_t off = kNoLoad;
        bool adv = (r.st & D_ADV) != 0u;
        r.st &= ~D_ADV;
        // -- 1. the word the ray's load brought --
        if (r.st & D_WAIT) {
            r.st &= ~D_WAIT;
            if (DBG) dbg_done += 1u;
            if (e >= kEMark) {  // the child of level L is a leaf; the mask said "not empty", so it is solid
                if (e & 1u) finish(r, ST_F_SOLID);
                else r.st |= D_LEAF;
            } else {
                const uint32_t l1 = ((r.st >> ST_L_SHIFT) & 31u) + 1u;
                if (l1 > (uint32_t)SMAX) {  // an interior word at level SMAX: deeper than this kernel resolves (reported by svo_sync)
                    atomicOr(a.status, 1u);
                    finish(r, ST_F_TOODEEP);
                } else {
                    r.cur = e;
                    lds[col + (l1 - (uint32_t)SBASE) * BLOCK] = e;
                    r.st += 1u << ST_L_SHIFT;
                    adv = true;
                }
            }
        }
        // -- 2. next level: empty leaf or a load --
        if (adv) advance(r, want, off);
        // -- 3. hit test / DDA step (shader.wgsl:215-244) from an empty leaf, clean rays, grid units (as trace_stack_kernel) --
        if (r.st & D_LEAF) {
            if (DBG) dbg_steps += 1u;
            const uint32_t L = (r.st >> ST_L_SHIFT) & 31u;
            const uint32_t sh = (uint32_t)D - L;
            const uint32_t keep = 0xFFFFFFFFu << sh, halfbit = 1u << (sh - 1u);
            const float C0 = (float)((int32_t)(((uint32_t)r.ix & keep) | halfbit) - 8388608);
            const float C1 = (float)((int32_t)(((uint32_t)r.iy & keep) | halfbit) - 8388608);
            const float C2 = (float)((int32_t)(((uint32_t)r.iz & keep) | halfbit) - 8388608);
            const float Hm = __uint_as_float((150u - L) << 23);  // 2^(23-L) = 2^23 * voxel_size / 2
            const float t0 = div_by_recip((C0 - r.P0) + copysign_bits(Hm, r.Dr0), r.Dr0, r.Y0);
            const float t1 = div_by_recip((C1 - r.P1) + copysign_bits(Hm, r.Dr1), r.Dr1, r.Y1);
            const float t2 = div_by_recip((C2 - r.P2) + copysign_bits(Hm, r.Dr2), r.Dr2, r.Y2);
            const float tnew = __builtin_fminf(__builtin_fminf(t0, t1), t2);
            const bool m0 = t0 == tnew, m1 = t1 == tnew, m2 = t2 == tnew;
            float G0 = r.P0 + r.Dr0 * tnew, G1 = r.P1 + r.Dr1 * tnew, G2 = r.P2 + r.Dr2 * tnew;
            G0 = m0 ? G0 + r.K0 : G0;
            G1 = m1 ? G1 + r.K1 : G1;
            G2 = m2 ? G2 + r.K2 : G2;
            const bool inb = (__builtin_fmaxf(__builtin_fmaxf(G0, G1), G2) < kScale) &&
                             (__builtin_fminf(__builtin_fminf(G0, G1), G2) >= -kScale);
            const uint32_t mbits = (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u);
            r.tcur = tnew;
            const uint32_t st1 = (r.st & ~(ST_M_MASK | ST_ENTRY | D_LEAF)) | (mbits << ST_M_SHIFT);
            if (!inb || (r.st & 0xFFu) >= 100u) {
                // the ray ends with this step: it leaves the cube, or its count passes 100 (shader.wgsl:236-244)
                r.st = ((st1 + (inb ? 1u : 0u)) & ~ST_ACTIVE) | ST_PENDING | (inb ? ST_F_INB : 0u);
            } else {
                r.st = st1 + 1u;
                int32_t jx, jy, jz;
                if (GE) {
                    jx = cvt_floor_i32(G0) + 8388608;
                    jy = cvt_floor_i32(G1) + 8388608;
                    jz = cvt_floor_i32(G2) + 8388608;
                } else {
                    jx = 8388607 - min(cvt_floor_neg_i32(G0), 8388607);
                    jy = 8388607 - min(cvt_floor_neg_i32(G1), 8388607);
                    jz = 8388607 - min(cvt_floor_neg_i32(G2), 8388607);
                }
                const uint32_t diff = (uint32_t)((r.ix ^ jx) | (r.iy ^ jy) | (r.iz ^ jz));
                const uint32_t c = (uint32_t)__clz((int)((diff << 8) | 0x80u));  // levels the old and the new path share
                r.ix = jx; r.iy = jy; r.iz = jz;
                const uint32_t rr = min(c + 1u, L);  // restart level (L <= SMAX)
                // the child group of level rr: from the top table up to level K+1, else from the lane's ancestor stack
                const bool top = rr <= (uint32_t)(K + 1);
                bool look = true;
                uint32_t addr = col + (rr - (uint32_t)SBASE) * BLOCK;
                if (__ballot(top)) {  // wave-uniform: most turns no lane crosses a level-(K+1) boundary
                    addr = top ? top_cell(r) : addr;
                    const uint32_t e = lds[addr];
                    if (top) {
                        look = enter_top(r, e);
                    } else {
                        r.cur = e;
                        r.st = (r.st & ~ST_L_MASK) | (rr << ST_L_SHIFT);
                    }
                } else {
                    r.cur = lds[addr];
                    r.st = (r.st & ~ST_L_MASK) | (rr << ST_L_SHIFT);
                }
                if (look) advance(r, want, off);
            }
        }
        // -- 4. the load (every turn, every lane: rays without one read past the buffer, which costs no memory access) --
        if (DBG) dbg_loads += want ? 1u : 0u;
        pe = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(re, (int)(want ? off : kNoLoad), 0, 0);
        if (want) {
            r.st |= D_WAIT;
            r.poff = off;
        }
    };

#ifdef SVO_DUAL_GUARD
    uint32_t guard = 0;  // bring-up builds: a wave that runs away ends with status bit 2 instead of hanging the device
#endif
    for (;;) {
#ifdef SVO_DUAL_GUARD
        if (++guard > (uint32_t)SVO_DUAL_GUARD) {
            if (lane == 0) atomicOr(a.status, 4u);
            break;
        }
#endif
        if (DBG) c_mark = __builtin_amdgcn_s_memtime();
        uint64_t act_a = __ballot((int32_t)A.st < 0), act_b = __ballot((int32_t)B.st < 0);
        const uint32_t idle_a = 64u - (uint32_t)__popcll(act_a), idle_b = 64u - (uint32_t)__popcll(act_b);
        if (DBG) {
            n_iters += 1u;
            dbg_slots += 128u - idle_a - idle_b;
        }
        // ---- refill idle lanes of either set from the ray pool (ballot compaction); trace_stack_kernel's step 2 ----
        if (idle_a >= a.refill_min || idle_b >= a.refill_min) {
            if (next == 0xFFFFFFFEu && pool_n == 0u) {  // now the answer of the early claim is needed
                const uint64_t c_w0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                uint32_t sh = home / kSubs;
                uint32_t s = entry_of(sh, __builtin_amdgcn_readfirstlane(pend) * kSubs + home % kSubs + ((gridDim.x + kShards - 1u - sh) / kShards) * (uint32_t)(BLOCK / 64));
                while (s == 0xFFFFFFFFu) {  // the home counter ran out: one probe of all 64 counters, draw from the first that has entries
                    const uint32_t l = lane / kSubs;
                    const uint32_t cv = __hip_atomic_load(work_counter + lane * kShardStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t res_l = ((gridDim.x + kShards - 1u - l) / kShards) * (uint32_t)(BLOCK / 64);
                    uint32_t len_l;
                    if (order) {
                        len_l = order[l];
                    } else {
                        const uint32_t runs = (n_strips + 15u) >> 4, mine = runs > l ? (runs - l + kShards - 1u) / kShards : 0u;
                        len_l = (mine << 4) - ((runs != 0u && (runs - 1u) % kShards == l) ? (runs << 4) - n_strips : 0u);
                    }
                    const uint64_t has = __ballot((uint64_t)cv * kSubs + lane % kSubs + res_l < (uint64_t)len_l);
                    if (!has) break;
                    const uint64_t rot = home ? ((has >> home) | (has << (64u - home))) : has;
                    home = (home + (uint32_t)__ffsll((unsigned long long)rot) - 1u) & 63u;
                    sh = home / kSubs;
                    uint32_t k = 0u;
                    if (lane == 0) k = atomicAdd(work_counter + home * kShardStride, 1u);
                    s = entry_of(sh, __builtin_amdgcn_readfirstlane(k) * kSubs + home % kSubs + ((gridDim.x + kShards - 1u - sh) / kShards) * (uint32_t)(BLOCK / 64));
                }
                home = (home & ~(kSubs - 1u)) | ((home + 1u) & (kSubs - 1u));
                s = __builtin_amdgcn_readfirstlane(s);
                next = s != 0xFFFFFFFFu ? s * strip_items : 0xFFFFFFFFu;
                strip_end = s != 0xFFFFFFFFu ? min(next + strip_items, n_items) : 0xFFFFFFFFu;
                if (DBG && next == 0xFFFFFFFFu && t_dry == 0) t_dry = __builtin_amdgcn_s_memrealtime();
                if (DBG) c_claim += (uint32_t)(__builtin_amdgcn_s_memtime() - c_w0);
            }
            const bool more = (pool_n != 0u) || (next != 0xFFFFFFFFu);
            if (more) {
                if (DBG) dbg_refills += 1;
                if (pool_n == 0u) {
                    if (DBG) dbg_gens += 1;
                    const uint64_t c_g0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                    // -- generate the next (up to) 64 rays, all lanes --
                    const uint32_t q = next + lane;
                    bool alive = false;
                    float gp0 = 0, gp1 = 0, gp2 = 0, gd0 = 1, gd1 = 1, gd2 = 1, gdist = 0;
                    uint32_t gout = 0;
                    if (q < strip_end) {
                        ItemFast it = decode_item_fast(a.work, q);
                        if (a.work.mode == 2 && a.skip && a.skip[q]) it.valid = false;  // no ray: the producer wrote the record
                        if (it.valid) {
                            RayIn ri;
                            if (a.work.mode == 2) {
                                const float *p = a.rays + 6ull * it.out;
                                ri = RayIn{p[0], p[1], p[2], p[3], p[4], p[5]};
                            } else {
                                ri = gen_ray(a.u, it.px, it.py);
                            }
                            float pos[3], dir[3];
                            if (!ray_enter(ri, pos, dir, gdist)) {
                                write_hit(a.hits, it.out, 0u, 0.0f, 0u, 0u, 0u, 0u);
                                if (a.aux_t) a.aux_t[it.out] = 0.0f;
                            } else if (!(clean_component(pos[0], dir[0]) && clean_component(pos[1], dir[1]) &&
                                         clean_component(pos[2], dir[2]) && fabsf(gdist) <= 1.0e30f)) {
                                // outside the proven range of the fast arithmetic: the reference-shaped pass that runs right after
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
                        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                        gd0 *= kScale; gd1 *= kScale; gd2 *= kScale;  // exact: power-of-two scaling
                        pool[0 * 64 + slot] = __float_as_uint(gp0 * kScale);
                        pool[1 * 64 + slot] = __float_as_uint(gp1 * kScale);
                        pool[2 * 64 + slot] = __float_as_uint(gp2 * kScale);
                        pool[3 * 64 + slot] = __float_as_uint(gd0);
                        pool[4 * 64 + slot] = __float_as_uint(gd1);
                        pool[5 * 64 + slot] = __float_as_uint(gd2);
                        pool[6 * 64 + slot] = __float_as_uint(1.0f / gd0);  // RN(1 / Dr) = 2^-23 * RN(1 / dir)
                        pool[7 * 64 + slot] = __float_as_uint(1.0f / gd1);
                        pool[8 * 64 + slot] = __float_as_uint(1.0f / gd2);
                        pool[9 * 64 + slot] = __float_as_uint(gdist);
                        pool[10 * 64 + slot] = gout;
                    }
                    pool_n = (uint32_t)__popcll(am);
                    pool_i = 0u;
                    next += min(64u, strip_end - next);
                    if (next >= strip_end) {
                        if (lane == 0) pend = atomicAdd(work_counter + home * kShardStride, 1u);
                        next = strip_end = 0xFFFFFFFEu;  // not known yet (and not "dry")
                    }
                    if (DBG) c_gen += (uint32_t)(__builtin_amdgcn_s_memtime() - c_g0);
                }
                if (idle_a >= a.refill_min) {
                    take(A, act_a, idle_a);
                    act_a = __ballot((int32_t)A.st < 0);
                }
                if (idle_b >= a.refill_min) {
                    take(B, act_b, idle_b);
                    act_b = __ballot((int32_t)B.st < 0);
                }
            }
            // the only exit: nothing in flight, nothing pooled, nothing left to claim
            if ((act_a | act_b) == 0ull && pool_n == 0u && next == 0xFFFFFFFFu) break;
        }
        if (DBG) {
            const uint64_t now = __builtin_amdgcn_s_memtime();
            c_refill += (uint32_t)(now - c_mark);
            c_mark = now;
        }
        turn(A, pe_a, col_a);
        if (DBG) {
            const uint64_t now = __builtin_amdgcn_s_memtime();
            c_ta += (uint32_t)(now - c_mark);
            c_mark = now;
        }
        turn(B, pe_b, col_b);
        if (DBG) c_tb += (uint32_t)(__builtin_amdgcn_s_memtime() - c_mark);
    }
    uint32_t last = 0u;
    if (DBG) {
        last = max((A.st & ST_PENDING) ? (A.st & 0xFFu) : 0u, (B.st & ST_PENDING) ? (B.st & 0xFFu) : 0u);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            last = max(last, (uint32_t)__shfl_xor((int)last, o));
            dbg_steps += (uint32_t)__shfl_xor((int)dbg_steps, o);
            dbg_loads += (uint32_t)__shfl_xor((int)dbg_loads, o);
            dbg_done += (uint32_t)__shfl_xor((int)dbg_done, o);
        }
    }
    if (A.st & ST_PENDING) flush_record(A);
    if (B.st & ST_PENDING) flush_record(B);
    if (DBG && lane == 0) {
        const uint64_t t_end = __builtin_amdgcn_s_memrealtime();
        uint32_t *d = a.debug + 16u * wave_id;
        d[0] = (uint32_t)t_begin;
        d[1] = (uint32_t)(t_dry ? t_dry : t_end);
        d[2] = (uint32_t)t_end;
        d[3] = n_iters;      // loop iterations (one turn of either ray set each)
        d[4] = dbg_slots;    // live rays (of 128 slots), summed over the iterations
        d[5] = last;         // largest step count among the rays the wave finished last
        d[6] = dbg_refills;
        d[7] = dbg_gens;
        d[8] = c_refill;     // shader cycles: refill section (ray generation included)
        d[9] = c_ta;         // ... turns of set A
        d[10] = c_tb;        // ... turns of set B
        d[11] = c_gen;       // ... generating rays (part of d[8])
        d[12] = dbg_steps;   // DDA steps taken by the wave's rays
        d[13] = dbg_loads;   // loads issued for them (lanes)
        d[14] = dbg_done;    // loads taken (lanes)
        d[15] = c_claim;     // cycles spent waiting for the answers of strip claims
    }
}

constexpr int kDualBlock = 256;
constexpr int kDualLevels = 12;  // levels K+2 .. K+1+12 = 5 .. 16 on the LDS stacks

int dual_max_depth() { return kTopLevels + 1 + kDualLevels; }

hipError_t launch_trace_dual(const TraceArgs &args, const LaunchInfo &li, hipStream_t stream) {
    (void)hipGetLastError();
    if (args.work.n_items == 0) return hipSuccess;
    const uint32_t strip_items = args.order ? 64u : (li.strip_items ? li.strip_items : 64u);
    const bool ge = (args.u.flags & SVO_F_MISC_BOOL) != 0;
    auto kern = args.debug ? (ge ? trace_dual_kernel<kDualBlock, kDualLevels, kTopLevels, true, true>
                                 : trace_dual_kernel<kDualBlock, kDualLevels, kTopLevels, false, true>)
                           : (ge ? trace_dual_kernel<kDualBlock, kDualLevels, kTopLevels, true, false>
                                 : trace_dual_kernel<kDualBlock, kDualLevels, kTopLevels, false, false>);
    const size_t lds_bytes = (size_t)((1 << (3 * kTopLevels)) + 2 * kDualLevels * kDualBlock + (kDualBlock / 64) * kPoolWords * 64) * sizeof(uint32_t);
    int &blocks_per_cu = li.occupancy[24 + (args.debug ? 1 : 0)];
    if (blocks_per_cu == 0) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kDualBlock, lds_bytes);
        if (e != hipSuccess) return e;
        blocks_per_cu = n > 0 ? n : 1;
        const int by_lds = (int)((160u * 1024u) / ((lds_bytes + 1023u) & ~(size_t)1023u));  // (see launch_stack)
        if (by_lds >= 1 && by_lds < blocks_per_cu) blocks_per_cu = by_lds;
    }
    uint32_t blocks = (uint32_t)li.num_cus * (uint32_t)blocks_per_cu;
    if (li.grid_blocks > 0) blocks = (uint32_t)li.grid_blocks;
    const uint32_t n_strips = (args.work.n_items + strip_items - 1) / strip_items;
    const uint32_t need = (n_strips + (kDualBlock / 64) - 1) / (kDualBlock / 64);
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kDualBlock), lds_bytes, stream, args, strip_items, li.work_counter, li.defer);
    return hipGetLastError();
}

}  // namespace svo
