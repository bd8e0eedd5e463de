// svo_trace_fn.h -- device helpers shared by the trace kernels (svo_kernels.hip, svo_dual.hip): the strict-f32
// restatement of shader.wgsl:54-80,177-212 (ray generation, cube entry), work-item decoding, the exact integer path codes
// and the fast division of DESIGN.md 4.2-4.3, the per-lane state word and the ray-pool record.  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include "svo_device.h"

#ifndef SVO_STREAM_RECORDS
#define SVO_STREAM_RECORDS 1
#endif

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

// Ray generation, shader.wgsl:54-59,253-259 (inv: camera_inverse, column-major; dim: the frame's width and height)
__device__ __forceinline__ RayIn gen_ray_from(const float *inv, float dim_x, float dim_y, uint32_t px, uint32_t py) {
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float cx = fx / dim_x * 2.0f;
    float cy = fy / dim_y * 2.0f;
    cx = cx - 1.0f;
    cy = cy - 1.0f;
    cy = cy * -1.0f;
    float p4[4], d4[4];
    mat_vec(inv, 0.0f, 0.0f, 0.0f, 1.0f, p4);
    mat_vec(inv, cx, cy, 1.0f, 1.0f, d4);
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
__device__ __forceinline__ RayIn gen_ray(const svo_uniforms &u, uint32_t px, uint32_t py) {
    return gen_ray_from(u.camera_inverse, u.dimensions[0], u.dimensions[1], px, py);
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
    // (round 5: a streaming store -- 33 MB of records per 1080p frame pass through the eight 4 MB L2s, as much as they hold, and the
    // node lines they push out are fetched again; nobody reads a record before the frame is over)
    typedef uint32_t rec_t __attribute__((ext_vector_type(4)));
    const rec_t rec = {value, __float_as_uint(t), (steps & 0xFFu) | ((depth & 0xFFu) << 8) | (hit << 16) | (ncode << 17), ncode};
#if SVO_STREAM_RECORDS
    __builtin_nontemporal_store(rec, reinterpret_cast<rec_t *>(hits) + out);
#else
    reinterpret_cast<rec_t *>(hits)[out] = rec;
#endif
}

__device__ __forceinline__ float code_to_normal(uint32_t c) { return c == 1u ? 1.0f : (c == 2u ? -1.0f : 0.0f); }

// normalize(u.sun_dir.xyz), shader.wgsl:275
__device__ __forceinline__ void sun_direction(const svo_uniforms &u, float s[3]) {
    const float sl = sqrtf((u.sun_dir[0] * u.sun_dir[0] + u.sun_dir[1] * u.sun_dir[1]) + u.sun_dir[2] * u.sun_dir[2]);
    s[0] = u.sun_dir[0] / sl; s[1] = u.sun_dir[1] / sl; s[2] = u.sun_dir[2] / sl;
}

// Where the secondary rays of a hit start (shader.wgsl:276): HitInfo.pos + normal * 2.5e-6, with HitInfo.pos the
// voxel_pos of the last step (the entry point when no step was taken) rebuilt from the primary ray (pos, dir: after
// ray_enter), the record's step count and normal codes, and t_current of the last step.
__device__ __forceinline__ void secondary_origin(const float pos[3], const float dir[3], uint32_t steps, uint32_t ncode, float t,
                                                 float org[3], float n[3]) {
    n[0] = code_to_normal(ncode & 3u); n[1] = code_to_normal((ncode >> 2) & 3u); n[2] = code_to_normal((ncode >> 4) & 3u);
    float h0 = pos[0], h1 = pos[1], h2 = pos[2];
    if (steps != 0u) {
        h0 = pos[0] + dir[0] * t - n[0] * 0.000002f;
        h1 = pos[1] + dir[1] * t - n[1] * 0.000002f;
        h2 = pos[2] + dir[2] * t - n[2] * 0.000002f;
    }
    org[0] = h0 + n[0] * 0.0000025f; org[1] = h1 + n[1] * 0.0000025f; org[2] = h2 + n[2] * 0.0000025f;
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
    uint32_t x = (bx << w.bw_log2) + (lane & ((1u << w.bw_log2) - 1u)), y = (by << (6u - w.bw_log2)) + (lane >> w.bw_log2);
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

// Path code of a position: bit (D - d) of the code is the child choice `pos > centre` (or `>=`)
// at level d.  General form (any float, NaN included): used at ray entry.
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

// n / d for n*d_err < 2^32 with one fix-up step; magic = floor(2^32 / d) + 1 (d >= 2), d == 1 handled by magic 0
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t d, uint32_t magic) {
    if (magic == 0u) return n;
    uint32_t q = __umulhi(n, magic);
    uint32_t r = n - q * d;
    return (r >= d) ? q - 1u : q;  // r wrapped below zero when q overshot by one
}

struct ItemFast {
    bool valid;
    uint32_t out, px, py;
};

__device__ __forceinline__ ItemFast decode_item_fast(const WorkDesc &w, uint32_t q) {
    ItemFast it;
    if (w.mode == 2) {
        it.valid = q < w.n_items;
        it.out = q;
        it.px = it.py = 0;
        return it;
    }
    uint32_t blk = q >> 6, lane = q & 63u;
    uint32_t rect = 0, b = blk;
    if (w.n_rects > 1u) {
        rect = fast_div(blk, w.bprect, w.magic_bprect);
        b = blk - rect * w.bprect;
    }
    uint32_t by = fast_div(b, w.bpr, w.magic_bpr), bx = b - by * w.bpr;
    uint32_t x = (bx << w.bw_log2) + (lane & ((1u << w.bw_log2) - 1u)), y = (by << (6u - w.bw_log2)) + (lane >> w.bw_log2);
    it.valid = (q < w.n_items) && (x < w.w) && (y < w.h);
    it.out = rect * (w.w * w.h) + y * w.w + x;
    uint32_t ox = w.x0, oy = w.y0;
    if (w.mode == 1) {
        uint32_t t = w.first_tile + rect * w.tile_stride;
        uint32_t ty = fast_div(t, w.tiles_x, w.magic_tiles_x);
        ox = (t - ty * w.tiles_x) * w.w;
        oy = ty * w.h;
    }
    it.px = ox + x;
    it.py = oy + y;
    return it;
}

// The same for the 64 items of one block, lane by lane: `first` is the item of lane 0 -- a multiple of 64 and the same on every
// lane, so the block's row, column and rectangle are found once per wave on the scalar unit (three divisions by
// multiplication: ~80 vector instructions per generated strip when done per lane) and only the position inside the block
// is per-lane work.
__device__ __forceinline__ ItemFast decode_item_wave(const WorkDesc &w, uint32_t first, uint32_t lane) {
    ItemFast it;
    const uint32_t q = first + lane;
    if (w.mode == 2) {
        it.valid = q < w.n_items;
        it.out = q;
        it.px = it.py = 0;
        return it;
    }
    const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(first >> 6));
    uint32_t rect = 0, b = blk;
    if (w.n_rects > 1u) {
        rect = fast_div(blk, w.bprect, w.magic_bprect);
        b = blk - rect * w.bprect;
    }
    const uint32_t by = fast_div(b, w.bpr, w.magic_bpr), bx = b - by * w.bpr;
    uint32_t ox = w.x0, oy = w.y0;
    if (w.mode == 1) {
        const uint32_t t = w.first_tile + rect * w.tile_stride;
        const uint32_t ty = fast_div(t, w.tiles_x, w.magic_tiles_x);
        ox = (t - ty * w.tiles_x) * w.w;
        oy = ty * w.h;
    }
    const uint32_t x0 = bx << w.bw_log2, y0 = by << (6u - w.bw_log2), out0 = rect * (w.w * w.h) + y0 * w.w + x0;  // (scalar)
    const uint32_t lx = lane & ((1u << w.bw_log2) - 1u), ly = lane >> w.bw_log2;
    const uint32_t x = x0 + lx, y = y0 + ly;
    it.valid = (q < w.n_items) && (x < w.w) && (y < w.h);
    it.out = out0 + ly * w.w + lx;
    it.px = ox + x;
    it.py = oy + y;
    return it;
}

// a / d given y = RN(1 / d):  q0 = a*y,  r = a - d*q0 (exact, one fma),  q0 + r*y rounded once -- the same bits as
// IEEE a / d.  Markstein's theorem gives this for a faithful q0; that q0 = RN(a*y) is always good enough is
// established exhaustively: tools/divtest_gpu.hip compares the sequence with a / d for all 2^23 x 2^23 pairs
// of significands (0 mismatches, profiles/r01_divtest_gpu.log; the uncorrected product fails on 27 %).  The
// sequence is invariant under power-of-two scaling and sign changes while nothing over/underflows: the caller
// guarantees 2^-17 <= |d| <= 2^63 and a == 0 or 2^-26 <= |a| <= 2^26 (grid units), so q0 and r stay normal.
__device__ __forceinline__ float div_by_recip(float a, float d, float y) {
    const float q = a * y;
    const float r = __builtin_fmaf(-d, q, a);
    return __builtin_fmaf(r, y, q);
}

// RN(1 / d) for 2^-20 <= |d| <= 2^64: the hardware reciprocal (1 ulp) and one Newton step; equal to the IEEE division for every
// such d ON gfx950 (tools/rcptest_gpu.hip compares all of them -- tests/test_lds_oob_gpu.py runs it on the GPU under test --; svo_kernels.hip
// refuses to build for another target; SVO_RECIP_IEEE=1 builds the division instead)
#ifndef SVO_RECIP_IEEE
#define SVO_RECIP_IEEE 0
#endif
__device__ __forceinline__ float recip_rn(float d) {
#if SVO_RECIP_IEEE
    return 1.0f / d;
#else
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
#endif
}

// floor(x) and floor(-x) = -ceil(x) as integers in one instruction (|x| < 2^24 here, no saturation involved)
__device__ __forceinline__ int32_t cvt_floor_i32(float x) {
    int32_t r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int32_t cvt_floor_neg_i32(float x) {
    int32_t r;
    asm("v_cvt_flr_i32_f32_e64 %0, -%1" : "=v"(r) : "v"(x));
    return r;
}

// Path code of a finite position given in grid units (|g| <= 2^(D+1)), clamped into the cube like path_code(), in the form
// the STACK kernel keeps it: the bits of the f32 number 2^D + code (0x4B000000 | code for D = 23).  Bit D - l of that word is
// the child choice at level l, and the same word read as a float takes part in the step's arithmetic without a conversion.
template <bool GE>
__device__ __forceinline__ uint32_t entry_magic(float g) {
    constexpr int32_t kHalf = 1 << (kPathBits - 1), kMax = (1 << kPathBits) - 1;
    const int32_t i = GE ? cvt_floor_i32(g) + kHalf : (kHalf - 1) - cvt_floor_neg_i32(g);
    return __float_as_uint((float)(1 << kPathBits)) | (uint32_t)min(max(i, 0), kMax);
}

__device__ __forceinline__ float copysign_bits(float mag, float sgn) {
    return __uint_as_float((__float_as_uint(mag) & 0x7FFFFFFFu) | (__float_as_uint(sgn) & 0x80000000u));
}

// "clean" ray: every quantity of the stepping arithmetic stays finite and inside the ranges the fast
// forms above are proven for.  pos is the entry point (|pos| <= 2 always holds for rays that enter).
__device__ __forceinline__ bool clean_component(float p, float d) {
    float ap = fabsf(p), ad = fabsf(d);
    bool p_ok = ap <= 2.0f;  // any magnitude below: scaling by 2^22 is exact, and A = (C - P) + H is 0 or >= 2^-27
    bool d_ok = (ad >= 9.094947017729282e-13f) && (ad <= 1099511627776.0f);     // 2^-40 .. 2^40
    return p_ok && d_ok;  // NaN fails both
}

// Per-lane state of the STACK kernel, kept as a FLOAT (round 4: on gfx950 the plain f32 instructions -- add, mul, fma, also with
// the clamp modifier -- issue beside the integer / compare / conversion instructions, which is where this kernel is bound; an
// integer state word cost a handful of those per round): 0 no ray, 1 a finished ray whose record is not written yet, 2 at its
// leaf, 4 needs a descent; negative: the lane traces the shadow ray of the pixel in `out` (SHD instantiation).  Transitions are
// multiplications (x 2, x 0.5), which keep the sign.  Steps, the last step's mask and the leaf's level live in registers of
// their own (stepsf, nmf, sh).
constexpr float ST_IDLE = 0.0f, ST_PENDING = 1.0f, ST_LEAF = 2.0f, ST_DESC = 4.0f;

// Ray pool: a wave generates the rays of up to 64 work items at once, with every lane busy (lanes that
// are still traversing compute a ray for somebody else), compacts the ones that enter the cube into LDS,
// and idle lanes later pick them up.  Ray generation and set-up (2 mat-vecs, 14 IEEE divisions, a square
// root) are thereby paid once per 64 rays at full lane utilisation instead of on
// every refill.  Pool record: P.xyz, Dr.xyz (position and biased direction in grid units), dist, out | entry normal code << 26.
// (The entry path codes and the reciprocals of the direction are recomputed at pick-up: 8 words per ray keep a workgroup at
// 22 KiB of LDS, i.e. 7 workgroups per CU.)
constexpr int kPoolWords = 8;
constexpr int kCountQueue = 128;  // CNT: queued (word, visits) pairs per wave
#ifndef SVO_SAT_TAGS
#define SVO_SAT_TAGS 256
#endif
constexpr int kSatTags = SVO_SAT_TAGS;  // CNT: words known to be saturated, direct-mapped, per workgroup (256: the counting workgroup stays at
                                        // 26 KiB of LDS, six per CU; 512 entries measured the same time at five)

}  // namespace svo
