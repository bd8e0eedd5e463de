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
//     stack kept in LDS; the top kTopLevels levels of the tree are folded into a 16 KiB table
//     staged in LDS; waves are persistent and refill finished lanes (ballot + mbcnt compaction).
//   * node words are read with buffer loads (hardware range check: an index past the buffer
//     reads 0, the semantics the oracle defines for out-of-range words).
#include <hip/hip_runtime.h>

#include "svo_device.h"

namespace svo {

// ---------------------------------------------------------------------------------------------
// strict-f32 helpers (definitions shared with oracle/svo_oracle.c)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fmin_w(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float fmax_w(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float sign_w(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

__device__ __forceinline__ uint32_t normal_code(float n) {
    return (n == 0.0f) ? 0u : ((n == 1.0f) ? 1u : ((n == -1.0f) ? 2u : 3u));
}

using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const uint32_t *p, uint32_t n_words) {
    // raw buffer, stride 0, num_records in bytes; 0x00020000 = DATA_FORMAT 32 (gfx9 raw dword)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p), 0, (int)(n_words << 2), 0x00020000);
}

__device__ __forceinline__ uint32_t load_word(rsrc_t rs, uint32_t idx) {
    // idx < 2^30 on every path that reaches here (descents use pointers < 2^27 + 8)
    return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(idx << 2), 0, 0);
}

struct RayIn {
    float px, py, pz, dx, dy, dz;
};

// mat4 * vec4, column-major, ((c0*x + c1*y) + c2*z) + c3*w
__device__ __forceinline__ void mat_vec(const float *m, float x, float y, float z, float w, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

// Ray generation, shader.wgsl:54-59,253-259
__device__ __forceinline__ RayIn gen_ray(const svo_uniforms &u, uint32_t px, uint32_t py) {
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float cx = fx / u.dimensions[0] * 2.0f;
    float cy = fy / u.dimensions[1] * 2.0f;
    cx = cx - 1.0f;
    cy = cy - 1.0f;
    cy = cy * -1.0f;
    float p4[4], d4[4];
    mat_vec(u.camera_inverse, 0.0f, 0.0f, 0.0f, 1.0f, p4);
    mat_vec(u.camera_inverse, cx, cy, 1.0f, 1.0f, d4);
    RayIn r;
    r.px = p4[0] / p4[3];
    r.py = p4[1] / p4[3];
    r.pz = p4[2] / p4[3];
    float dx = d4[0] / d4[3] - r.px, dy = d4[1] / d4[3] - r.py, dz = d4[2] / d4[3] - r.pz;
    float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    r.dx = dx / len;
    r.dy = dy / len;
    r.dz = dz / len;
    return r;
}

// in_bounds, shader.wgsl:177-180
__device__ __forceinline__ bool in_bounds(float x, float y, float z) {
    float s0 = ((-1.0f <= x) ? 1.0f : 0.0f) - ((1.0f <= x) ? 1.0f : 0.0f);
    float s1 = ((-1.0f <= y) ? 1.0f : 0.0f) - ((1.0f <= y) ? 1.0f : 0.0f);
    float s2 = ((-1.0f <= z) ? 1.0f : 0.0f) - ((1.0f <= z) ? 1.0f : 0.0f);
    return (s0 * s1 * s2) > 0.5f;
}

// ray_box_dist against [-1,1]^3, shader.wgsl:66-80 (uses the UNBIASED direction)
__device__ __forceinline__ float ray_box_dist(const RayIn &r) {
    float v1 = (-1.0f - r.px) / r.dx, v2 = (1.0f - r.px) / r.dx;
    float v3 = (-1.0f - r.py) / r.dy, v4 = (1.0f - r.py) / r.dy;
    float v5 = (-1.0f - r.pz) / r.dz, v6 = (1.0f - r.pz) / r.dz;
    float v7 = fmax_w(fmax_w(fmin_w(v1, v2), fmin_w(v3, v4)), fmin_w(v5, v6));
    float v8 = fmin_w(fmin_w(fmax_w(v1, v2), fmax_w(v3, v4)), fmax_w(v5, v6));
    if (v8 < 0.0f || v7 > v8) return 0.0f;
    return v7;
}

// octree_ray prologue, shader.wgsl:192-212.  false: the ray never enters the cube (value 0).
__device__ __forceinline__ bool ray_enter(const RayIn &r, float pos[3], float dir[3], float &dist) {
    dir[0] = r.dx + ((r.dx == 0.0f) ? 1.0f : 0.0f) * 0.000001f;
    dir[1] = r.dy + ((r.dy == 0.0f) ? 1.0f : 0.0f) * 0.000001f;
    dir[2] = r.dz + ((r.dz == 0.0f) ? 1.0f : 0.0f) * 0.000001f;
    pos[0] = r.px; pos[1] = r.py; pos[2] = r.pz;
    dist = 0.0f;
    if (!in_bounds(r.px, r.py, r.pz)) {
        dist = ray_box_dist(r);
        if (dist == 0.0f) return false;
        pos[0] = r.px + dir[0] * dist;
        pos[1] = r.py + dir[1] * dist;
        pos[2] = r.pz + dir[2] * dist;
    }
    return true;
}

__device__ __forceinline__ void write_hit(svo_hit *hits, uint32_t out, uint32_t value, float t, uint32_t steps,
                                          uint32_t depth, uint32_t hit, uint32_t ncode) {
    uint4 rec;
    rec.x = value;
    rec.y = __float_as_uint(t);
    rec.z = (steps & 0xFFu) | ((depth & 0xFFu) << 8) | (hit << 16) | (ncode << 17);
    rec.w = ncode;
    reinterpret_cast<uint4 *>(hits)[out] = rec;
}

struct Item {
    bool valid;
    uint32_t out, px, py;
};

__device__ __forceinline__ Item decode_item(const WorkDesc &w, uint32_t q) {
    Item it;
    if (w.mode == 2) {
        it.valid = q < w.n_items;
        it.out = q;
        it.px = it.py = 0;
        return it;
    }
    uint32_t blk = q >> 6, lane = q & 63u;
    uint32_t rect = blk / w.bprect;
    uint32_t b = blk - rect * w.bprect;
    uint32_t by = b / w.bpr, bx = b - by * w.bpr;
    uint32_t x = bx * 8u + (lane & 7u), y = by * 8u + (lane >> 3);
    it.valid = (q < w.n_items) && (x < w.w) && (y < w.h);
    it.out = rect * (w.w * w.h) + y * w.w + x;
    uint32_t ox = w.x0, oy = w.y0;
    if (w.mode == 1) {
        uint32_t t = w.first_tile + rect * w.tile_stride;
        uint32_t ty = t / w.tiles_x;
        ox = (t - ty * w.tiles_x) * w.w;
        oy = ty * w.h;
    }
    it.px = ox + x;
    it.py = oy + y;
    return it;
}

__device__ __forceinline__ RayIn item_ray(const TraceArgs &a, const Item &it) {
    if (a.work.mode == 2) {
        const float *p = a.rays + 6ull * it.out;
        RayIn r = {p[0], p[1], p[2], p[3], p[4], p[5]};
        return r;
    }
    return gen_ray(a.u, it.px, it.py);
}

// ---------------------------------------------------------------------------------------------
// Variant RESTART: the reference's algorithm shape -- float-compare descent from the root on
// every step (shader.wgsl:130-171 inside :213-245), one ray per lane, grid-stride over items.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trace_restart_kernel(TraceArgs a) {
    const rsrc_t rs = make_rsrc(a.nodes, a.n_words);
    const bool misc_bool = (a.u.flags & SVO_F_MISC_BOOL) != 0;
    const bool counter_hits = (a.u.flags & SVO_F_PAUSE_ADAPTIVE) && (a.u.flags & SVO_F_SHOW_HITS);
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < a.work.n_items; q += gridDim.x * 256u) {
        Item it = decode_item(a.work, q);
        if (!it.valid) continue;
        RayIn r = item_ray(a, it);
        float pos[3], dir[3], dist;
        if (!ray_enter(r, pos, dir, dist)) {
            write_hit(a.hits, it.out, 0u, 0.0f, 0u, 0u, 0u, 0u);
            continue;
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
                uint32_t tn = word >> 4;
                if (tn >= kVoxelOffset) break;
                if (depth >= kMaxDescent) { overflow = true; break; }
                node_index = tn;
            }
            if (overflow) {
                write_hit(a.hits, it.out, 0xFF000000u, dist + t_current, steps, 100u, 1u,
                          normal_code(n0) | (normal_code(n1) << 2) | (normal_code(n2) << 4));
                break;
            }
            bool solid = counter_hits ? ((word & 15u) > 0u) : (((word >> 4) - kVoxelOffset) > 0u);
            if (solid) {
                write_hit(a.hits, it.out, p, dist + t_current, steps, depth, 1u,
                          normal_code(n0) | (normal_code(n1) << 2) | (normal_code(n2) << 4));
                break;
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
                write_hit(a.hits, it.out, 0x20202000u, dist + t_current, steps, depth, 0u, 0u);
                break;
            }
            steps += 1;
            if (steps > 100u) {
                write_hit(a.hits, it.out, 0xFF000000u, dist + t_current, steps, 100u, 1u,
                          normal_code(n0) | (normal_code(n1) << 2) | (normal_code(n2) << 4));
                break;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Top table: fold the first kTopLevels levels into one word per level-K cell.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_top_table_kernel(const uint32_t *nodes, uint32_t n_words,
                                                              uint32_t *table) {
    const rsrc_t rs = make_rsrc(nodes, n_words);
    uint32_t cell = blockIdx.x * 256u + threadIdx.x;
    if (cell >= (uint32_t)kTopEntries) return;
    uint32_t cx = (cell >> (2 * kTopLevels)) & ((1u << kTopLevels) - 1u);
    uint32_t cy = (cell >> kTopLevels) & ((1u << kTopLevels) - 1u);
    uint32_t cz = cell & ((1u << kTopLevels) - 1u);
    uint32_t node_index = 0, entry = 0;
    for (int lvl = 1; lvl <= kTopLevels; lvl++) {
        int sh = kTopLevels - lvl;
        uint32_t child = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        uint32_t p = node_index + child;
        uint32_t tn = load_word(rs, p) >> 4;
        if (tn >= kVoxelOffset) {
            entry = kTopLeaf | ((tn != kVoxelOffset) ? kTopSolid : 0u) | ((uint32_t)lvl << 27) | (p & 0x07FFFFFFu);
            break;
        }
        node_index = tn;
        entry = tn;  // after the last level: child group of level K+1
    }
    table[cell] = entry;
}

// ---------------------------------------------------------------------------------------------
// Variant STACK (see file header).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t path_code(float v, bool ge_mode) {
    // exact: v * 2^23 only moves the exponent; fmaxf/fminf (IEEE maxNum/minNum) send NaN to the bound,
    // which reproduces "every comparison false" (code 0)
    float g = fminf(fmaxf(v * 8388608.0f, -8388608.0f), 8388608.0f);
    int32_t i = ge_mode ? (int32_t)floorf(g) : ((int32_t)ceilf(g) - 1);
    i += 8388608;
    i = i < 0 ? 0 : i;
    i = i > 0x00FFFFFF ? 0x00FFFFFF : i;
    return i;
}

template <int BLOCK, int NS>
__global__ __launch_bounds__(BLOCK) void trace_stack_kernel(TraceArgs a, uint32_t strip_items,
                                                            uint32_t *work_counter) {
    constexpr int K = kTopLevels;
    constexpr int D = kPathBits;
    constexpr int SBASE = K + 2;       // first level kept on the LDS stack
    constexpr int SMAX = K + 1 + NS;   // last level kept on the LDS stack
    extern __shared__ uint32_t lds[];
    uint32_t *tbl = lds;                 // kTopEntries
    uint32_t *stk = lds + kTopEntries;   // [NS][BLOCK]

    const uint32_t tid = threadIdx.x;
    const rsrc_t rs = make_rsrc(a.nodes, a.n_words);
    const bool ge_mode = (a.u.flags & SVO_F_MISC_BOOL) != 0;
    const bool counter_hits = (a.u.flags & SVO_F_PAUSE_ADAPTIVE) && (a.u.flags & SVO_F_SHOW_HITS);
    const bool use_table = (a.top_table != nullptr) && !counter_hits;

    if (use_table)
        for (uint32_t i = tid; i < (uint32_t)kTopEntries; i += BLOCK) tbl[i] = a.top_table[i];
    __syncthreads();

    const uint32_t n_items = a.work.n_items;
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + tid) >> 6);
    const uint32_t n_strips = (n_items + strip_items - 1) / strip_items;

    // wave-uniform work cursor
    uint32_t strip, next, strip_end;
    if (work_counter) {
        uint32_t s = 0;
        if ((tid & 63u) == 0) s = atomicAdd(work_counter, 1u);
        strip = __builtin_amdgcn_readfirstlane(s);
    } else {
        strip = wave_id;
    }
    if (strip < n_strips) {
        next = strip * strip_items;
        strip_end = min(next + strip_items, n_items);
    } else {
        next = strip_end = 0xFFFFFFFFu;
    }

    // per-lane ray state
    bool active = false;
    uint32_t out = 0;
    float pos[3] = {0, 0, 0}, dir[3] = {1, 1, 1};
    float dist = 0.0f, tcur = 0.0f;
    int32_t ix = 0, iy = 0, iz = 0;
    uint32_t steps = 0, ncode = 0;
    uint32_t lvl = 1, nidx = 0;       // next level to read and the child group it lives in
    uint32_t leaf_p = 0, L = 0;       // current leaf: word index and depth
    bool solid = false, desc = false;

    for (;;) {
        // ---- 1. refill idle lanes (ballot compaction) ----
        uint64_t act = __ballot(active);
        if (next != 0xFFFFFFFFu) {
            uint32_t n_idle = 64u - (uint32_t)__popcll(act);
            if (n_idle >= a.refill_min || act == 0ull) {
                uint32_t avail = strip_end - next;
                if (!active) {
                    uint64_t idle = ~act;
                    uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (rank < avail) {
                        Item it = decode_item(a.work, next + rank);
                        if (it.valid) {
                            RayIn r = item_ray(a, it);
                            out = it.out;
                            if (!ray_enter(r, pos, dir, dist)) {
                                write_hit(a.hits, out, 0u, 0.0f, 0u, 0u, 0u, 0u);
                            } else {
                                active = true;
                                steps = 0;
                                tcur = 0.0f;
                                ncode = normal_code(truncf(pos[0] * 1.000001f)) |
                                        (normal_code(truncf(pos[1] * 1.000001f)) << 2) |
                                        (normal_code(truncf(pos[2] * 1.000001f)) << 4);
                                ix = path_code(pos[0], ge_mode);
                                iy = path_code(pos[1], ge_mode);
                                iz = path_code(pos[2], ge_mode);
                                L = 0;
                                // start from the top
                                desc = true;
                                lvl = 1;
                                nidx = 0;
                                if (use_table) {
                                    uint32_t cell = ((uint32_t)(ix >> (D - K)) << (2 * K)) |
                                                    ((uint32_t)(iy >> (D - K)) << K) | (uint32_t)(iz >> (D - K));
                                    uint32_t e = tbl[cell];
                                    if (e & kTopLeaf) {
                                        desc = false;
                                        leaf_p = e & 0x07FFFFFFu;
                                        L = (e >> 27) & 7u;
                                        solid = (e & kTopSolid) != 0u;
                                    } else {
                                        lvl = K + 1;
                                        nidx = e;
                                    }
                                }
                            }
                        }
                    }
                }
                uint32_t taken = min(n_idle, avail);
                next += taken;
                if (next >= strip_end) {
                    if (work_counter) {
                        uint32_t s = 0;
                        if ((tid & 63u) == 0) s = atomicAdd(work_counter, 1u);
                        strip = __builtin_amdgcn_readfirstlane(s);
                    } else {
                        strip += n_waves;
                    }
                    if (strip < n_strips) {
                        next = strip * strip_items;
                        strip_end = min(next + strip_items, n_items);
                    } else {
                        next = strip_end = 0xFFFFFFFFu;
                    }
                }
                act = __ballot(active);
                if (act == 0ull) {
                    if (next == 0xFFFFFFFFu) break;
                    continue;
                }
            }
        } else if (act == 0ull) {
            break;
        }

        // ---- 2. descent: one dependent word per level below the restart level ----
        bool overflow = false;
        while (desc) {
            int sh = D - (int)lvl;
            uint32_t child = ((((uint32_t)ix >> sh) & 1u) << 2) | ((((uint32_t)iy >> sh) & 1u) << 1) |
                             (((uint32_t)iz >> sh) & 1u);
            uint32_t p = nidx + child;
            uint32_t w = load_word(rs, p);
            uint32_t tn = w >> 4;
            if (tn >= kVoxelOffset) {
                leaf_p = p;
                L = lvl;
                solid = counter_hits ? ((w & 15u) > 0u) : (tn != kVoxelOffset);
                desc = false;
            } else if (lvl >= (uint32_t)D) {
                overflow = true;  // deeper than the integer path codes resolve
                desc = false;
            } else {
                lvl += 1;
                nidx = tn;
                if (lvl >= (uint32_t)SBASE && lvl <= (uint32_t)SMAX) stk[(lvl - SBASE) * BLOCK + tid] = tn;
            }
        }

        // ---- 3. hit test / DDA step (shader.wgsl:215-244) ----
        if (active) {
            if (overflow) {
                atomicOr(a.status, 1u);
                write_hit(a.hits, out, 0xFF000000u, dist + tcur, steps, 100u, 1u, ncode);
                active = false;
            } else if (solid) {
                write_hit(a.hits, out, leaf_p, dist + tcur, steps, L, 1u, ncode);
                active = false;
            } else {
                // leaf centre from the path code: exact, equals the reference's accumulated node_pos
                int sh = D - (int)L;
                float inv = __uint_as_float((127u - L) << 23);  // 2^-L
                int32_t half = 1 << L;
                float c0 = (float)(2 * (ix >> sh) + 1 - half) * inv;
                float c1 = (float)(2 * (iy >> sh) + 1 - half) * inv;
                float c2 = (float)(2 * (iz >> sh) + 1 - half) * inv;
                float rs0 = sign_w(dir[0]), rs1 = sign_w(dir[1]), rs2 = sign_w(dir[2]);
                float voxel_size = inv * 2.0f;  // 2 / 2^L, exact
                float t0 = (c0 - pos[0] + rs0 * voxel_size / 2.0f) / dir[0];
                float t1 = (c1 - pos[1] + rs1 * voxel_size / 2.0f) / dir[1];
                float t2 = (c2 - pos[2] + rs2 * voxel_size / 2.0f) / dir[2];
                float m0 = (t0 <= fmin_w(t1, t2)) ? 1.0f : 0.0f;
                float m1 = (t1 <= fmin_w(t2, t0)) ? 1.0f : 0.0f;
                float m2 = (t2 <= fmin_w(t0, t1)) ? 1.0f : 0.0f;
                float n0 = m0 * -rs0, n1 = m1 * -rs1, n2 = m2 * -rs2;
                tcur = fmin_w(fmin_w(t0, t1), t2);
                float vp0 = pos[0] + dir[0] * tcur - n0 * 0.000002f;
                float vp1 = pos[1] + dir[1] * tcur - n1 * 0.000002f;
                float vp2 = pos[2] + dir[2] * tcur - n2 * 0.000002f;
                ncode = normal_code(n0) | (normal_code(n1) << 2) | (normal_code(n2) << 4);
                if (!in_bounds(vp0, vp1, vp2)) {
                    write_hit(a.hits, out, 0x20202000u, dist + tcur, steps, L, 0u, 0u);
                    active = false;
                } else {
                    steps += 1;
                    if (steps > 100u) {
                        write_hit(a.hits, out, 0xFF000000u, dist + tcur, steps, 100u, 1u, ncode);
                        active = false;
                    } else {
                        int32_t jx = path_code(vp0, ge_mode), jy = path_code(vp1, ge_mode), jz = path_code(vp2, ge_mode);
                        uint32_t diff = (uint32_t)((ix ^ jx) | (iy ^ jy) | (iz ^ jz));
                        uint32_t c = diff ? ((uint32_t)__clz((int)diff) - (32u - D)) : (uint32_t)D;  // shared levels
                        ix = jx; iy = jy; iz = jz;
                        uint32_t r = min(min(c + 1u, L), (uint32_t)SMAX);
                        desc = true;
                        if (r <= (uint32_t)(K + 1)) {
                            lvl = 1;
                            nidx = 0;
                            if (use_table) {
                                uint32_t cell = ((uint32_t)(ix >> (D - K)) << (2 * K)) |
                                                ((uint32_t)(iy >> (D - K)) << K) | (uint32_t)(iz >> (D - K));
                                uint32_t e = tbl[cell];
                                if (e & kTopLeaf) {
                                    desc = false;
                                    leaf_p = e & 0x07FFFFFFu;
                                    L = (e >> 27) & 7u;
                                    solid = (e & kTopSolid) != 0u;
                                } else {
                                    lvl = K + 1;
                                    nidx = e;
                                }
                            }
                        } else {
                            lvl = r;
                            nidx = stk[(r - SBASE) * BLOCK + tid];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Counter scan, compute.wgsl:26-47.  One word per lane, wave-aggregated append (ballot + mbcnt,
// one atomic per wave and list instead of one per node).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_kernel(const uint32_t *nodes, uint32_t n_words, uint32_t node_length,
                                                   uint32_t *sub, uint32_t *unsub, uint32_t capacity) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = (blockIdx.x * 256u + threadIdx.x) & ~63u; base < n_words; base += gridDim.x * 256u) {
        uint32_t id = base + lane;
        uint32_t node = (id < n_words) ? nodes[id] : 0u;
        uint32_t counter = node & 15u;
        bool live = (node != 0u) && (id < node_length);
        bool is_unsub = live && counter == 0u && (node >> 4) < kVoxelOffset;
        bool is_sub = live && !is_unsub && counter >= 4u && (node >> 4) > kVoxelOffset;
        uint64_t mu = __ballot(is_unsub), ms = __ballot(is_sub);
        if (mu) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&unsub[0], (uint32_t)__popcll(mu));
            b = __builtin_amdgcn_readfirstlane(b);
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mu, 0u));
            if (is_unsub && (uint64_t)b + rank + 1u < capacity) unsub[1u + b + rank] = id;
        }
        if (ms) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&sub[0], (uint32_t)__popcll(ms));
            b = __builtin_amdgcn_readfirstlane(b);
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
            if (is_sub && (uint64_t)b + rank + 1u < capacity) sub[1u + b + rank] = id;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_build_top_table(const uint32_t *nodes, uint32_t n_words, uint32_t *top_table, hipStream_t stream) {
    hipLaunchKernelGGL(build_top_table_kernel, dim3(kTopEntries / 256), dim3(256), 0, stream, nodes, n_words, top_table);
    return hipGetLastError();
}

constexpr int kStackBlock = 256;
constexpr int kStackLevels = 16;

hipError_t launch_trace(const TraceArgs &args, const LaunchInfo &li, hipStream_t stream) {
    const uint32_t strip_items = li.strip_items ? li.strip_items : 64u;
    uint32_t *work_counter = li.work_counter;
    if (args.work.n_items == 0) return hipSuccess;
    if (li.variant == SVO_VARIANT_RESTART) {
        uint32_t blocks = (args.work.n_items + 255u) / 256u;
        uint32_t cap = (uint32_t)li.num_cus * 8u;
        if (li.grid_blocks > 0) cap = (uint32_t)li.grid_blocks;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(trace_restart_kernel, dim3(blocks), dim3(256), 0, stream, args);
        return hipGetLastError();
    }
    auto kern = trace_stack_kernel<kStackBlock, kStackLevels>;
    size_t lds_bytes = (size_t)(kTopEntries + kStackLevels * kStackBlock) * sizeof(uint32_t);
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kStackBlock, lds_bytes);
        if (e != hipSuccess) return e;
        blocks_per_cu = n > 0 ? n : 1;
    }
    uint32_t blocks = (uint32_t)li.num_cus * (uint32_t)blocks_per_cu;
    if (li.grid_blocks > 0) blocks = (uint32_t)li.grid_blocks;
    uint32_t n_strips = (args.work.n_items + strip_items - 1) / strip_items;
    uint32_t need = (n_strips + (kStackBlock / 64) - 1) / (kStackBlock / 64);
    if (blocks > need) blocks = need;
    if (work_counter) {
        hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kStackBlock), lds_bytes, stream, args, strip_items, work_counter);
    return hipGetLastError();
}

hipError_t launch_scan(const uint32_t *nodes, uint32_t n_words, uint32_t node_length, uint32_t *sub, uint32_t *unsub,
                       uint32_t capacity, hipStream_t stream) {
    if (n_words == 0) return hipSuccess;
    uint32_t blocks = (n_words + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(scan_kernel, dim3(blocks), dim3(256), 0, stream, nodes, n_words, node_length, sub, unsub, capacity);
    return hipGetLastError();
}

}  // namespace svo
