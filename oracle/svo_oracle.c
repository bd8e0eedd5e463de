/*
 * svo_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see svo_oracle.h).
 *
 * Plain-C restatement of ria8651/octree-tracer's SVO path.  Every function cites the
 * reference file:line it follows.  Compile with strict IEEE f32 semantics:
 *     gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
 * PARITY STATUS: parity unpinned by the reference (no tests, not buildable here); pinned by
 * the hand-derived known-answer tests of SURVEY.md 8c only.
 *
 * Choices the reference leaves to the WGSL compiler are fixed here and the HIP kernel follows
 * THIS file: mat*vec is ((c0*x + c1*y) + c2*z) + c3*w; dot is (x*x' + y*y') + z*z';
 * normalize(v) is v / sqrt(dot(v,v)); min(a,b) is (b < a) ? b : a; max(a,b) is (a < b) ? b : a;
 * division and sqrt are correctly rounded; no FMA contraction; subnormals kept.
 */
#include "svo_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define VOXEL_OFFSET ORACLE_VOXEL_OFFSET
#define CHUNK_OFFSET ORACLE_CHUNK_OFFSET

/* Descent guard.  The reference loop (shader.wgsl:134-171) is unbounded and `1u << depth`
 * is undefined from depth 32; a malformed array (e.g. a zero word) would never terminate.
 * Both oracle and kernel stop at this depth and report the step-limit sentinel. */
#define MAX_DESCENT 31u

/* ------------------------------------------------------------------------------------ */
/* CpuOctree (src/cpu_octree.rs:5-45)                                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t pointer;
    uint8_t r, g, b;
} cnode;

struct oracle_tree {
    cnode *nodes;
    size_t len, cap;
    uint8_t top_mip[3];
};

static void tree_push(oracle_tree *t, uint32_t pointer, uint8_t r, uint8_t g, uint8_t b) {
    if (t->len == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (cnode *)realloc(t->nodes, t->cap * sizeof(cnode));
        if (!t->nodes) abort();
    }
    cnode n = {pointer, r, g, b};
    t->nodes[t->len++] = n;
}

/* CpuOctree::add_voxels, cpu_octree.rs:32-45: child-mask bit test. */
static void tree_add_voxels(oracle_tree *t, uint8_t mask) {
    for (int i = 0; i < 8; i++) {
        if ((mask >> i) & 1)
            tree_push(t, CHUNK_OFFSET + ((uint32_t)t->len % 8u) + 1u, 255, 0, 0);
        else
            tree_push(t, CHUNK_OFFSET, 0, 0, 0);
    }
}

oracle_tree *oracle_tree_new(uint8_t mask) {
    oracle_tree *t = (oracle_tree *)calloc(1, sizeof(*t));
    t->top_mip[0] = 50; t->top_mip[1] = 255; t->top_mip[2] = 50; /* cpu_octree.rs:25 */
    tree_add_voxels(t, mask);
    return t;
}

void oracle_tree_free(oracle_tree *t) {
    if (!t) return;
    free(t->nodes);
    free(t);
}

size_t oracle_tree_len(const oracle_tree *t) { return t->len; }

/* Octree::pos_offset, octree.rs:154-161 */
static void pos_offset(unsigned child, uint32_t depth, float out[3]) {
    float d = (float)(1u << depth);
    out[0] = ((float)((child >> 2) & 1) * 2.0f - 1.0f) / d;
    out[1] = ((float)((child >> 1) & 1) * 2.0f - 1.0f) / d;
    out[2] = ((float)(child & 1) * 2.0f - 1.0f) / d;
}

/* CpuOctree::find_voxel, cpu_octree.rs:48-76 (">=" tie-break). max_depth < 0 means None. */
void oracle_tree_find_voxel(const oracle_tree *t, float x, float y, float z, int64_t max_depth,
                            uint64_t *index, uint32_t *depth_out, float pos_out[3]) {
    size_t node_index = 0;
    float np[3] = {0.0f, 0.0f, 0.0f};
    uint32_t depth = 0;
    for (;;) {
        depth += 1;
        unsigned px = x >= np[0], py = y >= np[1], pz = z >= np[2];
        unsigned child = px * 4 + py * 2 + pz;
        float off[3];
        pos_offset(child, depth, off);
        np[0] += off[0]; np[1] += off[1]; np[2] += off[2];
        uint32_t ptr = t->nodes[node_index + child].pointer;
        if (ptr >= CHUNK_OFFSET || (max_depth >= 0 && (int64_t)depth == max_depth)) {
            *index = node_index + child;
            *depth_out = depth;
            if (pos_out) { pos_out[0] = np[0]; pos_out[1] = np[1]; pos_out[2] = np[2]; }
            return;
        }
        node_index = ptr;
    }
}

/* CpuOctree::put_in_voxel, cpu_octree.rs:100-111 */
void oracle_tree_put_in_voxel(oracle_tree *t, float x, float y, float z, uint8_t r, uint8_t g,
                              uint8_t b, uint32_t depth) {
    for (;;) {
        uint64_t node; uint32_t node_depth;
        oracle_tree_find_voxel(t, x, y, z, -1, &node, &node_depth, NULL);
        if (depth == node_depth) {
            cnode n = {CHUNK_OFFSET, r, g, b};
            t->nodes[node] = n;
            return;
        }
        t->nodes[node].pointer = (uint32_t)t->len;
        tree_add_voxels(t, 0);
    }
}

/* CpuOctree::to_octree, cpu_octree.rs:233-252; Voxel::to_value octree.rs:28-34;
 * create_node octree.rs:164-166 */
void oracle_tree_to_octree(const oracle_tree *t, uint32_t *out) {
    for (size_t i = 0; i < t->len; i++) {
        cnode n = t->nodes[i];
        if (n.pointer < CHUNK_OFFSET)
            out[i] = n.pointer << 4;
        else
            out[i] = (VOXEL_OFFSET + (((uint32_t)n.r << 16) | ((uint32_t)n.g << 8) | n.b)) << 4;
    }
}

void oracle_tree_raw(const oracle_tree *t, uint32_t *pointers, uint8_t *rgb) {
    for (size_t i = 0; i < t->len; i++) {
        if (pointers) pointers[i] = t->nodes[i].pointer;
        if (rgb) { rgb[3*i] = t->nodes[i].r; rgb[3*i+1] = t->nodes[i].g; rgb[3*i+2] = t->nodes[i].b; }
    }
}

/* ------------------------------------------------------------------------------------ */
/* .vox reader: restates what cpu_octree.rs:177-193 consumes from dot_vox 4.1.0          */
/* (Cargo.lock:439-440; crate source not vendored).  Published behaviour relied on:      */
/* model 0 only; Voxel{x,y,z,i} with i = file colour index - 1; palette = the 256 LE u32 */
/* of the RGBA chunk, so palette[i].to_le_bytes()[0..3] = R,G,B of file index i+1.       */
/* ------------------------------------------------------------------------------------ */
static uint32_t rd_u32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static void set_err(char *err, size_t errlen, const char *msg) {
    if (err && errlen) { strncpy(err, msg, errlen - 1); err[errlen - 1] = 0; }
}

int64_t oracle_vox_parse(const uint8_t *data, size_t len, uint32_t size[3], uint8_t *xyzi,
                         size_t xyzi_cap, uint32_t *palette256, char *err, size_t errlen) {
    if (len < 20 || memcmp(data, "VOX ", 4) != 0) { set_err(err, errlen, "not a .vox file"); return -1; }
    if (memcmp(data + 8, "MAIN", 4) != 0) { set_err(err, errlen, "missing MAIN chunk"); return -1; }
    size_t off = 8 + 12 + rd_u32(data + 12);
    int have_size = 0, have_xyzi = 0, have_rgba = 0;
    int64_t n_vox = -1;
    while (off + 12 <= len) {
        const uint8_t *c = data + off;
        uint32_t csize = rd_u32(c + 4), chsize = rd_u32(c + 8);
        const uint8_t *body = c + 12;
        if (off + 12 + (size_t)csize > len) { set_err(err, errlen, "truncated chunk"); return -1; }
        if (!memcmp(c, "SIZE", 4) && !have_size) {
            size[0] = rd_u32(body); size[1] = rd_u32(body + 4); size[2] = rd_u32(body + 8);
            have_size = 1;
        } else if (!memcmp(c, "XYZI", 4) && !have_xyzi) {
            uint32_t n = rd_u32(body);
            n_vox = n;
            if (xyzi) {
                if ((size_t)n * 4 > xyzi_cap) { set_err(err, errlen, "xyzi buffer too small"); return -1; }
                memcpy(xyzi, body + 4, (size_t)n * 4);
            }
            have_xyzi = 1;
        } else if (!memcmp(c, "RGBA", 4) && !have_rgba) {
            if (palette256)
                for (int i = 0; i < 256; i++) palette256[i] = rd_u32(body + 4 * i);
            have_rgba = 1;
        }
        off += 12 + (size_t)csize + (size_t)chsize;
    }
    if (!have_size || !have_xyzi) { set_err(err, errlen, "no model in .vox"); return -1; }
    if (!have_rgba && palette256) {
        /* dot_vox falls back to MagicaVoxel's default palette (cpu_octree.rs:177-193 loads such files).  The table of the
         * .vox format description, entry k = colour of file index k + 1: indices 1..215 a 6x6x6 cube (0xAABBGGRR, blue
         * fastest, steps of 0x33, without black), then four ramps of ten.  dot_vox's copy is unavailable: unpinned. */
        static const unsigned ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
        int idx = 1;
        for (int r = 5; r >= 0; r--)
            for (int g = 5; g >= 0; g--)
                for (int b = 5; b >= 0; b--)
                    if (r || g || b) palette256[idx++ - 1] = 0xff000000u | ((uint32_t)(b * 0x33) << 16) | ((uint32_t)(g * 0x33) << 8) | (uint32_t)(r * 0x33);
        for (int c = 0; c < 4; c++)
            for (int k = 0; k < 10; k++, idx++)
                palette256[idx - 1] = 0xff000000u | (c == 3 ? ramp[k] * 0x010101u : (uint32_t)ramp[k] << (8 * c));
        palette256[255] = 0;
    }
    return n_vox;
}

/* load_vox body, cpu_octree.rs:179-209 */
oracle_tree *oracle_tree_from_voxels(uint32_t size_u, const uint8_t *xyzi, size_t n_voxels,
                                     const uint32_t *palette256) {
    int size = (int)size_u;
    float depthf = log2f((float)size);         /* :186 */
    if (depthf != floorf(depthf)) return NULL; /* :187-189 */
    uint32_t depth = (uint32_t)depthf;
    oracle_tree *t = oracle_tree_new(0);       /* :191 */
    for (size_t k = 0; k < n_voxels; k++) {
        uint8_t vx = xyzi[4*k], vy = xyzi[4*k+1], vz = xyzi[4*k+2], fi = xyzi[4*k+3];
        uint8_t i = fi ? (uint8_t)(fi - 1) : 0; /* dot_vox: i = file index - 1 (saturating) */
        uint32_t colour = palette256[i];        /* :193, to_le_bytes()[0..3] = R,G,B */
        float px = (float)size - (float)vx - 1.0f; /* :194-198 axis swap (size-x-1, z, y) */
        float py = (float)vz;
        float pz = (float)vy;
        px /= (float)size; py /= (float)size; pz /= (float)size;   /* :199 */
        px = px * 2.0f - 1.0f; py = py * 2.0f - 1.0f; pz = pz * 2.0f - 1.0f; /* :200 */
        oracle_tree_put_in_voxel(t, px, py, pz, (uint8_t)(colour & 0xFF), (uint8_t)((colour >> 8) & 0xFF),
                                 (uint8_t)((colour >> 16) & 0xFF), depth);
    }
    return t;
}

oracle_tree *oracle_tree_from_vox(const uint8_t *data, size_t len, char *err, size_t errlen) {
    uint32_t size[3], pal[256];
    int64_t n = oracle_vox_parse(data, len, size, NULL, 0, pal, err, errlen);
    if (n < 0) return NULL;
    if (size[0] != size[1] || size[0] != size[2]) { set_err(err, errlen, "Voxel model is not a cube!"); return NULL; } /* :180-182 */
    uint8_t *xyzi = (uint8_t *)malloc((size_t)n * 4 + 4);
    oracle_vox_parse(data, len, size, xyzi, (size_t)n * 4, pal, err, errlen);
    oracle_tree *t = oracle_tree_from_voxels(size[0], xyzi, (size_t)n, pal);
    free(xyzi);
    if (!t) set_err(err, errlen, "Voxel model size is not a power of 2!"); /* :188 */
    return t;
}

/* CpuOctree::load_octree (.rsvo), cpu_octree.rs:128-175 */
oracle_tree *oracle_tree_from_rsvo(const uint8_t *data, size_t len, uint32_t octree_depth,
                                   char *err, size_t errlen) {
    const size_t top_level_start = 16, node_count_start = 20;
    if (len < 24) { set_err(err, errlen, "rsvo too short"); return NULL; }
    size_t top_level = data[top_level_start];
    size_t data_start = node_count_start + 4 * (top_level + 1);
    if (len <= data_start) { set_err(err, errlen, "rsvo too short"); return NULL; }
    if (octree_depth > top_level) {
        set_err(err, errlen, "Octree depth is greater than top level"); /* :148-153 */
        return NULL;
    }
    size_t node_end = 0;
    for (size_t i = 0; i < octree_depth; i++) node_end += rd_u32(data + node_count_start + 4 * i); /* :138-155 */
    oracle_tree *t = oracle_tree_new(data[data_start]); /* :157 */
    size_t data_index = 1, node_index = 0;
    while (node_index < t->len) {                       /* :160-172 */
        if (t->nodes[node_index].pointer > CHUNK_OFFSET) {
            if (data_index < node_end) {
                if (data_start + data_index >= len) { set_err(err, errlen, "rsvo truncated"); oracle_tree_free(t); return NULL; }
                uint8_t child_mask = data[data_start + data_index];
                t->nodes[node_index].pointer = (uint32_t)t->len;
                tree_add_voxels(t, child_mask);
            }
            data_index += 1;
        }
        node_index += 1;
    }
    return t;
}

/* World::generate_mip_tree restricted to one chunk with no block references
 * (world.rs:234-336).  BFS catalogue per level, then bottom-up average of the non-(0,0,0)
 * children, truncating cast to u8, clamped to >= 1 per channel.  Leaves that still carry a
 * block id (pointer > CHUNK_OFFSET) keep their value (no block table here). */
void oracle_tree_generate_mips(oracle_tree *t, uint8_t top_mip[3]) {
    size_t n = t->len;
    size_t *queue = (size_t *)malloc((n + 8) * sizeof(size_t));
    uint32_t *qdepth = (uint32_t *)malloc((n + 8) * sizeof(uint32_t));
    size_t qh = 0, qt = 0;
    for (size_t c = 0; c < 8; c++)
        if (t->nodes[c].pointer < CHUNK_OFFSET) { queue[qt] = c; qdepth[qt++] = 1; }
    while (qh < qt) { /* world.rs:260-293: the queue order IS the per-level order */
        size_t ni = queue[qh++];
        uint32_t d = qdepth[qh - 1];
        uint32_t ptr = t->nodes[ni].pointer;
        for (size_t c = 0; c < 8; c++)
            if (t->nodes[ptr + c].pointer < CHUNK_OFFSET) { queue[qt] = ptr + c; qdepth[qt++] = d + 1; }
    }
    /* bottom-up: levels in reverse, nodes of a level in catalogue order (:303-333).  Within a
     * level the order does not matter (children are on deeper levels), so walk the queue
     * backwards by level. */
    size_t end = qt;
    while (end > 0) {
        uint32_t d = qdepth[end - 1];
        size_t start = end;
        while (start > 0 && qdepth[start - 1] == d) start--;
        for (size_t k = start; k < end; k++) {
            size_t ni = queue[k];
            uint32_t ptr = t->nodes[ni].pointer;
            float cr = 0.0f, cg = 0.0f, cb = 0.0f, div = 0.0f;
            for (size_t c = 0; c < 8; c++) {
                cnode ch = t->nodes[ptr + c];
                if (ch.r || ch.g || ch.b) { cr += (float)ch.r; cg += (float)ch.g; cb += (float)ch.b; div += 1.0f; }
            }
            cr /= div; cg /= div; cb /= div; /* NaN when div == 0; `as u8` of NaN is 0 in Rust */
            uint8_t r = (cr == cr) ? (uint8_t)(cr < 0 ? 0 : (cr > 255 ? 255 : cr)) : 0;
            uint8_t g = (cg == cg) ? (uint8_t)(cg < 0 ? 0 : (cg > 255 ? 255 : cg)) : 0;
            uint8_t b = (cb == cb) ? (uint8_t)(cb < 0 ? 0 : (cb > 255 ? 255 : cb)) : 0;
            t->nodes[ni].r = r < 1 ? 1 : r; t->nodes[ni].g = g < 1 ? 1 : g; t->nodes[ni].b = b < 1 ? 1 : b;
        }
        end = start;
    }
    { /* level 0: the root group (node.pointer = 0), :306-310,:330-332 */
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, div = 0.0f;
        for (size_t c = 0; c < 8; c++) {
            cnode ch = t->nodes[c];
            if (ch.r || ch.g || ch.b) { cr += (float)ch.r; cg += (float)ch.g; cb += (float)ch.b; div += 1.0f; }
        }
        cr /= div; cg /= div; cb /= div;
        uint8_t r = (cr == cr) ? (uint8_t)cr : 0, g = (cg == cg) ? (uint8_t)cg : 0, b = (cb == cb) ? (uint8_t)cb : 0;
        t->top_mip[0] = r < 1 ? 1 : r; t->top_mip[1] = g < 1 ? 1 : g; t->top_mip[2] = b < 1 ? 1 : b;
    }
    if (top_mip) memcpy(top_mip, t->top_mip, 3);
    free(queue); free(qdepth);
}

/* ------------------------------------------------------------------------------------ */
/* Camera (render.rs:191-206, main.rs:139-162; cgmath 0.18 look_at_rh / invert).         */
/* cgmath's exact operation order is not reproducible here (crate source absent): the    */
/* matrices are an INPUT of the path; parity is defined on identical uniforms.           */
/* ------------------------------------------------------------------------------------ */
static void mat4_mul(const float a[16], const float b[16], float out[16]) {
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = 0.0f;
            for (int k = 0; k < 4; k++) s += a[k * 4 + r] * b[c * 4 + k];
            out[c * 4 + r] = s;
        }
}

static int mat4_invert(const float m[16], float inv[16]) {
    float t[16];
    t[0] = m[5]*m[10]*m[15] - m[5]*m[11]*m[14] - m[9]*m[6]*m[15] + m[9]*m[7]*m[14] + m[13]*m[6]*m[11] - m[13]*m[7]*m[10];
    t[4] = -m[4]*m[10]*m[15] + m[4]*m[11]*m[14] + m[8]*m[6]*m[15] - m[8]*m[7]*m[14] - m[12]*m[6]*m[11] + m[12]*m[7]*m[10];
    t[8] = m[4]*m[9]*m[15] - m[4]*m[11]*m[13] - m[8]*m[5]*m[15] + m[8]*m[7]*m[13] + m[12]*m[5]*m[11] - m[12]*m[7]*m[9];
    t[12] = -m[4]*m[9]*m[14] + m[4]*m[10]*m[13] + m[8]*m[5]*m[14] - m[8]*m[6]*m[13] - m[12]*m[5]*m[10] + m[12]*m[6]*m[9];
    t[1] = -m[1]*m[10]*m[15] + m[1]*m[11]*m[14] + m[9]*m[2]*m[15] - m[9]*m[3]*m[14] - m[13]*m[2]*m[11] + m[13]*m[3]*m[10];
    t[5] = m[0]*m[10]*m[15] - m[0]*m[11]*m[14] - m[8]*m[2]*m[15] + m[8]*m[3]*m[14] + m[12]*m[2]*m[11] - m[12]*m[3]*m[10];
    t[9] = -m[0]*m[9]*m[15] + m[0]*m[11]*m[13] + m[8]*m[1]*m[15] - m[8]*m[3]*m[13] - m[12]*m[1]*m[11] + m[12]*m[3]*m[9];
    t[13] = m[0]*m[9]*m[14] - m[0]*m[10]*m[13] - m[8]*m[1]*m[14] + m[8]*m[2]*m[13] + m[12]*m[1]*m[10] - m[12]*m[2]*m[9];
    t[2] = m[1]*m[6]*m[15] - m[1]*m[7]*m[14] - m[5]*m[2]*m[15] + m[5]*m[3]*m[14] + m[13]*m[2]*m[7] - m[13]*m[3]*m[6];
    t[6] = -m[0]*m[6]*m[15] + m[0]*m[7]*m[14] + m[4]*m[2]*m[15] - m[4]*m[3]*m[14] - m[12]*m[2]*m[7] + m[12]*m[3]*m[6];
    t[10] = m[0]*m[5]*m[15] - m[0]*m[7]*m[13] - m[4]*m[1]*m[15] + m[4]*m[3]*m[13] + m[12]*m[1]*m[7] - m[12]*m[3]*m[5];
    t[14] = -m[0]*m[5]*m[14] + m[0]*m[6]*m[13] + m[4]*m[1]*m[14] - m[4]*m[2]*m[13] - m[12]*m[1]*m[6] + m[12]*m[2]*m[5];
    t[3] = -m[1]*m[6]*m[11] + m[1]*m[7]*m[10] + m[5]*m[2]*m[11] - m[5]*m[3]*m[10] - m[9]*m[2]*m[7] + m[9]*m[3]*m[6];
    t[7] = m[0]*m[6]*m[11] - m[0]*m[7]*m[10] - m[4]*m[2]*m[11] + m[4]*m[3]*m[10] + m[8]*m[2]*m[7] - m[8]*m[3]*m[6];
    t[11] = -m[0]*m[5]*m[11] + m[0]*m[7]*m[9] + m[4]*m[1]*m[11] - m[4]*m[3]*m[9] - m[8]*m[1]*m[7] + m[8]*m[3]*m[5];
    t[15] = m[0]*m[5]*m[10] - m[0]*m[6]*m[9] - m[4]*m[1]*m[10] + m[4]*m[2]*m[9] + m[8]*m[1]*m[6] - m[8]*m[2]*m[5];
    float det = m[0]*t[0] + m[1]*t[4] + m[2]*t[8] + m[3]*t[12];
    if (det == 0.0f) return 0;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) inv[i] = t[i] * inv_det;
    return 1;
}

void oracle_camera(const float pos[3], const float look[3], float fov_deg, float width,
                   float height, float camera[16], float camera_inverse[16]) {
    /* look_at_rh(pos, pos + look, +Y) = look_to_rh(pos, look, +Y), render.rs:194-198 */
    float dir[3] = {(pos[0] + look[0]) - pos[0], (pos[1] + look[1]) - pos[1], (pos[2] + look[2]) - pos[2]};
    float fl = 1.0f / sqrtf(dir[0]*dir[0] + dir[1]*dir[1] + dir[2]*dir[2]);
    float f[3] = {dir[0]*fl, dir[1]*fl, dir[2]*fl};
    float up[3] = {0.0f, 1.0f, 0.0f};
    float s[3] = {f[1]*up[2] - f[2]*up[1], f[2]*up[0] - f[0]*up[2], f[0]*up[1] - f[1]*up[0]};
    float sl = 1.0f / sqrtf(s[0]*s[0] + s[1]*s[1] + s[2]*s[2]);
    s[0] *= sl; s[1] *= sl; s[2] *= sl;
    float u[3] = {s[1]*f[2] - s[2]*f[1], s[2]*f[0] - s[0]*f[2], s[0]*f[1] - s[1]*f[0]};
    float view[16] = {s[0], u[0], -f[0], 0.0f, s[1], u[1], -f[1], 0.0f, s[2], u[2], -f[2], 0.0f,
                      -(pos[0]*s[0] + pos[1]*s[1] + pos[2]*s[2]), -(pos[0]*u[0] + pos[1]*u[1] + pos[2]*u[2]),
                      pos[0]*f[0] + pos[1]*f[1] + pos[2]*f[2], 1.0f};
    /* create_proj_matrix(fov, H/W), main.rs:139-162, render.rs:200 */
    float aspect = height / width;
    float sc = 1.0f / tanf((fov_deg / 2.0f) * (3.14159265358979323846f / 180.0f));
    float proj[16] = {aspect * sc, 0, 0, 0, 0, sc, 0, 0, 0, 0, -1.0f, 0, 0, 0, 0, 1.0f};
    mat4_mul(proj, view, camera);          /* render.rs:201 */
    if (!mat4_invert(camera, camera_inverse)) memset(camera_inverse, 0, 64); /* :202 unwrap */
}

/* ------------------------------------------------------------------------------------ */
/* Traversal (src/shader.wgsl)                                                            */
/* ------------------------------------------------------------------------------------ */
static inline float fmin_w(float a, float b) { return (b < a) ? b : a; }
static inline float fmax_w(float a, float b) { return (a < b) ? b : a; }
static inline float sign_w(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

typedef struct { float pos[3], dir[3]; } ray_t;

typedef struct {
    uint32_t value;
    float pos[3];
    uint32_t depth;
} voxel_t;

typedef struct {
    int hit;
    uint32_t value;
    float pos[3], normal[3];
    uint32_t steps, depth;
    float t; /* dist + t_current of the last step (not a HitInfo field) */
} hitinfo_t;

typedef struct {
    const uint32_t *nodes;
    size_t n_nodes;
    uint32_t flags;
    uint32_t *visits; /* optional per-word visit counts (counter side effect) */
    /* perfect-reuse bookkeeping */
    uint32_t prev_path[MAX_DESCENT + 1], prev_len;
    uint32_t w_restart, w_reuse;
    int overflow;
} trace_ctx;

static inline uint32_t load_word(const trace_ctx *c, uint32_t i) {
    return (i < c->n_nodes) ? c->nodes[i] : 0u; /* out-of-range reads are defined as 0 */
}

/* ray_box_dist, shader.wgsl:66-80 */
static float ray_box_dist(const ray_t *r, float vmin, float vmax) {
    float v1 = (vmin - r->pos[0]) / r->dir[0];
    float v2 = (vmax - r->pos[0]) / r->dir[0];
    float v3 = (vmin - r->pos[1]) / r->dir[1];
    float v4 = (vmax - r->pos[1]) / r->dir[1];
    float v5 = (vmin - r->pos[2]) / r->dir[2];
    float v6 = (vmax - r->pos[2]) / r->dir[2];
    float v7 = fmax_w(fmax_w(fmin_w(v1, v2), fmin_w(v3, v4)), fmin_w(v5, v6));
    float v8 = fmin_w(fmin_w(fmax_w(v1, v2), fmax_w(v3, v4)), fmax_w(v5, v6));
    if (v8 < 0.0f || v7 > v8) return 0.0f;
    return v7;
}

/* in_bounds, shader.wgsl:177-180: step(-1,v) - step(1,v) per component, product > 0.5 */
static int in_bounds(const float v[3]) {
    float s0 = ((-1.0f <= v[0]) ? 1.0f : 0.0f) - ((1.0f <= v[0]) ? 1.0f : 0.0f);
    float s1 = ((-1.0f <= v[1]) ? 1.0f : 0.0f) - ((1.0f <= v[1]) ? 1.0f : 0.0f);
    float s2 = ((-1.0f <= v[2]) ? 1.0f : 0.0f) - ((1.0f <= v[2]) ? 1.0f : 0.0f);
    return (s0 * s1 * s2) > 0.5f;
}

/* find_voxel, shader.wgsl:130-175 */
static voxel_t find_voxel(trace_ctx *c, const float pos[3], int primary) {
    uint32_t node_index = 0;
    float np[3] = {0.0f, 0.0f, 0.0f};
    uint32_t depth = 0;
    uint32_t path[MAX_DESCENT + 1];
    voxel_t v;
    int misc_bool = (c->flags & ORACLE_F_MISC_BOOL) != 0;
    for (;;) {
        depth += 1;
        unsigned px, py, pz;
        if (misc_bool) { px = pos[0] >= np[0]; py = pos[1] >= np[1]; pz = pos[2] >= np[2]; }
        else           { px = pos[0] >  np[0]; py = pos[1] >  np[1]; pz = pos[2] >  np[2]; }
        unsigned child = px * 4u + py * 2u + pz;
        float d = (float)(1u << depth);
        np[0] = np[0] + ((float)px * 2.0f - 1.0f) / d;
        np[1] = np[1] + ((float)py * 2.0f - 1.0f) / d;
        np[2] = np[2] + ((float)pz * 2.0f - 1.0f) / d;
        uint32_t p = node_index + child;
        uint32_t value = load_word(c, p);
        path[depth - 1] = p;
        if (primary && c->visits && !(c->flags & ORACLE_F_PAUSE_ADAPTIVE) && p < c->n_nodes)
            c->visits[p] += 1; /* :157-161; saturation applied by the caller */
        uint32_t tnipt = value >> 4;
        if (tnipt >= VOXEL_OFFSET || depth >= MAX_DESCENT) {
            if (tnipt < VOXEL_OFFSET) c->overflow = 1;
            v.value = p; v.pos[0] = np[0]; v.pos[1] = np[1]; v.pos[2] = np[2]; v.depth = depth;
            break;
        }
        node_index = tnipt;
    }
    /* bookkeeping: words read by the reference (restart) and with perfect per-ray reuse */
    c->w_restart += depth;
    uint32_t common = 0;
    while (common < depth && common < c->prev_len && c->prev_path[common] == path[common]) common++;
    c->w_reuse += depth - common;
    memcpy(c->prev_path, path, depth * sizeof(uint32_t));
    c->prev_len = depth;
    return v;
}

/* octree_ray, shader.wgsl:191-248 */
static hitinfo_t octree_ray(trace_ctx *c, const ray_t *r, int primary) {
    hitinfo_t h;
    memset(&h, 0, sizeof(h));
    c->prev_len = 0; c->overflow = 0;
    float pos[3] = {r->pos[0], r->pos[1], r->pos[2]};
    float dir[3];
    for (int i = 0; i < 3; i++) {
        float dir_mask = (r->dir[i] == 0.0f) ? 1.0f : 0.0f;
        dir[i] = r->dir[i] + dir_mask * 0.000001f;       /* :193-194 */
    }
    float dist = 0.0f;
    if (!in_bounds(r->pos)) {
        dist = ray_box_dist(r, -1.0f, 1.0f);              /* :199 */
        if (dist == 0.0f) return h;                       /* :200-202: miss, value 0 */
        for (int i = 0; i < 3; i++) pos[i] = r->pos[i] + dir[i] * dist; /* :204 */
    }
    float r_sign[3] = {sign_w(dir[0]), sign_w(dir[1]), sign_w(dir[2])};
    voxel_t voxel;
    float voxel_pos[3] = {pos[0], pos[1], pos[2]};
    uint32_t steps = 0;
    float normal[3] = {truncf(pos[0] * 1.000001f), truncf(pos[1] * 1.000001f), truncf(pos[2] * 1.000001f)}; /* :212 */
    float t_current = 0.0f;
    for (;;) {
        voxel = find_voxel(c, voxel_pos, primary);
        if (c->overflow) { /* malformed array: report the step-limit sentinel */
            h.hit = 1; h.value = 0xFF000000u; memcpy(h.pos, voxel_pos, 12); memcpy(h.normal, normal, 12);
            h.steps = steps; h.depth = 100; h.t = dist + t_current; return h;
        }
        if (!(c->flags & ORACLE_F_PAUSE_ADAPTIVE) || !(c->flags & ORACLE_F_SHOW_HITS)) { /* :215-219 */
            uint32_t tnipt = (load_word(c, voxel.value) >> 4) - VOXEL_OFFSET;
            if (tnipt > 0u) break;
        } else {                                                                          /* :220-224 */
            uint32_t value = load_word(c, voxel.value);
            if ((value & 15u) > 0u) break;
        }
        float voxel_size = 2.0f / (float)(1u << voxel.depth);   /* :227 */
        float t_max[3];
        for (int i = 0; i < 3; i++)
            t_max[i] = (voxel.pos[i] - pos[i] + r_sign[i] * voxel_size / 2.0f) / dir[i]; /* :228 */
        /* mask = t_max.xyz <= min(t_max.yzx, t_max.zxy), :231 */
        float mask[3];
        mask[0] = (t_max[0] <= fmin_w(t_max[1], t_max[2])) ? 1.0f : 0.0f;
        mask[1] = (t_max[1] <= fmin_w(t_max[2], t_max[0])) ? 1.0f : 0.0f;
        mask[2] = (t_max[2] <= fmin_w(t_max[0], t_max[1])) ? 1.0f : 0.0f;
        for (int i = 0; i < 3; i++) normal[i] = mask[i] * -r_sign[i];                    /* :232 */
        t_current = fmin_w(fmin_w(t_max[0], t_max[1]), t_max[2]);                        /* :234 */
        for (int i = 0; i < 3; i++)
            voxel_pos[i] = pos[i] + dir[i] * t_current - normal[i] * 0.000002f;          /* :235 */
        if (!in_bounds(voxel_pos)) {                                                     /* :237-239 */
            h.hit = 0; h.value = 0x20202000u; h.steps = steps; h.depth = voxel.depth; h.t = dist + t_current;
            return h;
        }
        steps += 1;
        if (steps > 100u) {                                                              /* :241-244 */
            h.hit = 1; h.value = 0xFF000000u; memcpy(h.pos, voxel_pos, 12); memcpy(h.normal, normal, 12);
            h.steps = steps; h.depth = 100; h.t = dist + t_current; return h;
        }
    }
    h.hit = 1; h.value = voxel.value; memcpy(h.pos, voxel_pos, 12); memcpy(h.normal, normal, 12); /* :247 */
    h.steps = steps; h.depth = voxel.depth; h.t = dist + t_current;
    return h;
}

static uint32_t normal_code(float n) { return (n == 0.0f) ? 0u : ((n == 1.0f) ? 1u : ((n == -1.0f) ? 2u : 3u)); }

static oracle_hit pack_hit(const hitinfo_t *h) {
    oracle_hit o;
    uint32_t nb = normal_code(h->normal[0]) | (normal_code(h->normal[1]) << 2) | (normal_code(h->normal[2]) << 4);
    o.value = h->value;
    o.t = h->t;
    o.info = (h->steps & 0xFFu) | ((h->depth & 0xFFu) << 8) | ((uint32_t)(h->hit != 0) << 16) | (nb << 17);
    o.normal_bits = nb;
    return o;
}

/* mat4 * vec4, column-major, ((c0*x + c1*y) + c2*z) + c3*w */
static void mat_vec(const float m[16], float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

/* Ray generation, shader.wgsl:54-59,253-259 */
static ray_t gen_ray(const oracle_uniforms *u, int px, int py) {
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f; /* frag_pos at pixel centres */
    float cx = fx / u->dimensions[0] * 2.0f;            /* :55 */
    float cy = fy / u->dimensions[1] * 2.0f;
    cx = cx - 1.0f; cy = cy - 1.0f;                     /* :56 */
    cx = cx * 1.0f; cy = cy * -1.0f;                    /* :57 */
    float p4[4], d4[4];
    mat_vec(u->camera_inverse, 0.0f, 0.0f, 0.0f, 1.0f, p4); /* :255 */
    mat_vec(u->camera_inverse, cx, cy, 1.0f, 1.0f, d4);     /* :256 */
    ray_t r;
    for (int i = 0; i < 3; i++) r.pos[i] = p4[i] / p4[3];   /* :257 */
    float d[3];
    for (int i = 0; i < 3; i++) d[i] = d4[i] / d4[3] - r.pos[i]; /* :258 */
    float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (int i = 0; i < 3; i++) r.dir[i] = d[i] / len;
    return r;
}

void oracle_find_voxel(const uint32_t *nodes, size_t n_nodes, const float pos[3], int misc_bool,
                       uint32_t *value, float vpos[3], uint32_t *depth) {
    trace_ctx c;
    memset(&c, 0, sizeof(c));
    c.nodes = nodes; c.n_nodes = n_nodes; c.flags = ORACLE_F_PAUSE_ADAPTIVE | (misc_bool ? ORACLE_F_MISC_BOOL : 0);
    voxel_t v = find_voxel(&c, pos, 0);
    *value = v.value; *depth = v.depth;
    if (vpos) memcpy(vpos, v.pos, 12);
}

/* ---- threading: a persistent pool of pthreads, work handed out in small dynamic chunks from an atomic cursor ----
 * (Round 4: the pool used to create and join its threads on every call and dealt a frame in chunks of 4 rows behind a
 * mutex -- 270 chunks for 256 threads, sky rows cheap and terrain rows dear -- so the CPU baseline got SLOWER with more
 * cores, VERDICT r3.  Workers now live across calls and sleep on a condition variable between jobs; a frame is dealt in
 * spans of 256 pixels.)  The calling thread works too: n_threads = callers + workers. */
typedef struct {
    void (*fn)(void *arg, size_t begin, size_t end);
    void *arg;
    size_t n, chunk;
    atomic_size_t next;
} par_job;

static void par_run(par_job *j) {
    for (;;) {
        size_t b = atomic_fetch_add_explicit(&j->next, j->chunk, memory_order_relaxed);
        if (b >= j->n) break;
        size_t e = b + j->chunk < j->n ? b + j->chunk : j->n;
        j->fn(j->arg, b, e);
    }
}

#define PAR_MAX_THREADS 256
#define PAR_SPAN 256     /* pixels per chunk of a frame */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t work, done;
    int n_workers;        /* threads created so far (they never exit) */
    unsigned long gen;    /* bumped for every job */
    par_job *job;
    int want;             /* workers 0 .. want-1 take part in the current job */
    int busy;             /* of those, how many have not finished it yet */
    int atfork;
} par_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, NULL, 0, 0, 0};

static void *par_worker(void *p) {
    const int id = (int)(size_t)p;
    unsigned long seen = 0;
    pthread_mutex_lock(&par_pool.mu);
    for (;;) {
        while (par_pool.gen == seen) pthread_cond_wait(&par_pool.work, &par_pool.mu);
        seen = par_pool.gen;
        if (id >= par_pool.want) continue;
        par_job *j = par_pool.job;
        pthread_mutex_unlock(&par_pool.mu);
        par_run(j);
        pthread_mutex_lock(&par_pool.mu);
        if (--par_pool.busy == 0) pthread_cond_signal(&par_pool.done);
    }
    return NULL;
}

/* a forked child has none of the parent's threads: start over with an empty pool */
static void par_after_fork_child(void) {
    pthread_mutex_init(&par_pool.mu, NULL);
    pthread_cond_init(&par_pool.work, NULL);
    pthread_cond_init(&par_pool.done, NULL);
    par_pool.n_workers = 0; par_pool.gen = 0; par_pool.job = NULL; par_pool.want = 0; par_pool.busy = 0;
}

static void par_for(void (*fn)(void *, size_t, size_t), void *arg, size_t n, size_t chunk, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > PAR_MAX_THREADS) n_threads = PAR_MAX_THREADS;
    par_job j;
    j.fn = fn; j.arg = arg; j.n = n; j.chunk = chunk ? chunk : 1;
    atomic_init(&j.next, 0);
    size_t n_chunks = (n + j.chunk - 1) / j.chunk;
    if ((size_t)n_threads > n_chunks) n_threads = n_chunks ? (int)n_chunks : 1;
    if (n_threads == 1) { par_run(&j); return; }
    /* the pool has ONE job slot: concurrent callers -- two Python threads are enough, ctypes releases the GIL around a call -- take
     * turns for the whole of their job */
    static pthread_mutex_t submit = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&submit);
    pthread_mutex_lock(&par_pool.mu);
    if (!par_pool.atfork) { pthread_atfork(NULL, NULL, par_after_fork_child); par_pool.atfork = 1; }
    while (par_pool.n_workers < n_threads - 1) {
        pthread_t th;
        if (pthread_create(&th, NULL, par_worker, (void *)(size_t)par_pool.n_workers) != 0) break;
        pthread_detach(th);
        par_pool.n_workers++;
    }
    int want = n_threads - 1 < par_pool.n_workers ? n_threads - 1 : par_pool.n_workers;
    par_pool.job = &j; par_pool.want = want; par_pool.busy = want;
    par_pool.gen++;
    pthread_cond_broadcast(&par_pool.work);
    pthread_mutex_unlock(&par_pool.mu);
    par_run(&j);
    pthread_mutex_lock(&par_pool.mu);
    while (par_pool.busy != 0) pthread_cond_wait(&par_pool.done, &par_pool.mu);
    par_pool.job = NULL;
    pthread_mutex_unlock(&par_pool.mu);
    pthread_mutex_unlock(&submit);
}

typedef struct {
    const uint32_t *nodes; size_t n_nodes; uint32_t flags;
    const float *rays; oracle_hit *out; uint32_t *stats;
} rays_job;

static void rays_fn(void *arg, size_t b, size_t e) {
    rays_job *j = (rays_job *)arg;
    trace_ctx c;
    memset(&c, 0, sizeof(c));
    c.nodes = j->nodes; c.n_nodes = j->n_nodes; c.flags = j->flags;
    for (size_t i = b; i < e; i++) {
        ray_t r;
        memcpy(r.pos, j->rays + 6 * i, 12);
        memcpy(r.dir, j->rays + 6 * i + 3, 12);
        c.w_restart = c.w_reuse = 0;
        hitinfo_t h = octree_ray(&c, &r, 1);
        j->out[i] = pack_hit(&h);
        if (j->stats) { j->stats[2 * i] = c.w_restart; j->stats[2 * i + 1] = c.w_reuse; }
    }
}

void oracle_trace_rays(const uint32_t *nodes, size_t n_nodes, uint32_t flags, const float *rays,
                       size_t n_rays, oracle_hit *out, uint32_t *stats, int n_threads) {
    rays_job j = {nodes, n_nodes, flags, rays, out, stats};
    par_for(rays_fn, &j, n_rays, 1024, n_threads);
}

typedef struct {
    const uint32_t *nodes; size_t n_nodes; const oracle_uniforms *u;
    int x0, y0, w, h; oracle_hit *out; uint32_t *stats; float *rgba; uint32_t *visits;
} frame_job;

static void frame_fn(void *arg, size_t b, size_t e) {
    frame_job *j = (frame_job *)arg;
    trace_ctx c;
    memset(&c, 0, sizeof(c));
    c.nodes = j->nodes; c.n_nodes = j->n_nodes; c.flags = j->u->flags; c.visits = j->visits; /* visits==NULL: read-only */
    for (size_t i = b; i < e; i++) {  /* pixel i of the rectangle, row-major */
        const int row = (int)(i / (size_t)j->w), col = (int)(i % (size_t)j->w);
        ray_t r = gen_ray(j->u, j->x0 + col, j->y0 + row);
        c.w_restart = c.w_reuse = 0;
        hitinfo_t h = octree_ray(&c, &r, 1);
        if (j->out) j->out[i] = pack_hit(&h);
        if (j->stats) { j->stats[2 * i] = c.w_restart; j->stats[2 * i + 1] = c.w_reuse; }
    }
}

void oracle_trace_frame(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0,
                        int y0, int w, int h, oracle_hit *out, uint32_t *stats, int n_threads) {
    frame_job j = {nodes, n_nodes, u, x0, y0, w, h, out, stats, NULL, NULL};
    par_for(frame_fn, &j, (size_t)w * (size_t)h, PAR_SPAN, n_threads);
}

/* fs_main shading, shader.wgsl:261-304 */
static void shade_pixel(trace_ctx *c, const oracle_uniforms *u, int px, int py, float rgba[4]) {
    float col[3] = {0.0f, 0.0f, 0.0f};
    ray_t ray = gen_ray(u, px, py);
    hitinfo_t hit = octree_ray(c, &ray, 1);
    if (u->flags & ORACLE_F_SHOW_STEPS) {
        col[0] = col[1] = col[2] = (float)hit.steps / 64.0f;                       /* :264 */
    } else if (hit.hit) {
        if (u->flags & ORACLE_F_SHOW_HITS) {
            col[0] = col[1] = col[2] = (float)(load_word(c, hit.value) & 15u) / 15.0f; /* :268 */
        } else {
            float sl = sqrtf((u->sun_dir[0]*u->sun_dir[0] + u->sun_dir[1]*u->sun_dir[1]) + u->sun_dir[2]*u->sun_dir[2]);
            float sun[3] = {u->sun_dir[0] / sl, u->sun_dir[1] / sl, u->sun_dir[2] / sl}; /* :270 */
            float ambient = 0.3f;
            float diffuse = fmax_w((hit.normal[0] * -sun[0] + hit.normal[1] * -sun[1]) + hit.normal[2] * -sun[2], 0.0f); /* :273 */
            if (u->flags & ORACLE_F_SHADOWS) {                                       /* :275-280 */
                ray_t sr;
                for (int i = 0; i < 3; i++) { sr.pos[i] = hit.pos[i] + hit.normal[i] * 0.0000025f; sr.dir[i] = -sun[i]; }
                hitinfo_t sh = octree_ray(c, &sr, 1);
                if (sh.hit) diffuse = 0.0f;
            }
            uint32_t value = (load_word(c, hit.value) >> 4) - VOXEL_OFFSET;          /* :282 */
            float colour[3] = {(float)((value >> 16) & 0xFFu) / 255.0f, (float)((value >> 8) & 0xFFu) / 255.0f,
                               (float)(value & 0xFFu) / 255.0f};                     /* :283 unpack_u8(value).yzw */
            for (int i = 0; i < 3; i++) col[i] = (ambient + diffuse) * colour[i];    /* :284 */
        }
    } else {
        col[0] = col[1] = col[2] = 0.2f;                                             /* :287 */
    }
    float gamma = ((u->flags & ORACLE_F_MISC_BOOL) ? 1.0f : 0.0f) * -1.2f + 2.2f;    /* :304 */
    for (int i = 0; i < 3; i++) {
        float cl = fmin_w(fmax_w(col[i], 0.0f), 1.0f);
        rgba[i] = powf(cl, gamma);
    }
    rgba[3] = 0.5f;
}

static void shade_fn(void *arg, size_t b, size_t e) {
    frame_job *j = (frame_job *)arg;
    trace_ctx c;
    memset(&c, 0, sizeof(c));
    c.nodes = j->nodes; c.n_nodes = j->n_nodes; c.flags = j->u->flags; c.visits = j->visits;
    float dummy[4];
    for (size_t i = b; i < e; i++)
        shade_pixel(&c, j->u, j->x0 + (int)(i % (size_t)j->w), j->y0 + (int)(i / (size_t)j->w), j->rgba ? j->rgba + 4 * i : dummy);
}

void oracle_shade_frame(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0,
                        int y0, int w, int h, float *rgba, int n_threads) {
    frame_job j = {nodes, n_nodes, u, x0, y0, w, h, NULL, NULL, rgba, NULL};
    par_for(shade_fn, &j, (size_t)w * (size_t)h, PAR_SPAN, n_threads);
}

/* Benchmark config 5 (no reference counterpart beyond the shadow ray): per hit pixel, ray 0 = the shadow ray
 * of shader.wgsl:275-280 and rays 1.. from the same origin along hashed directions (include/svo_hip.h,
 * svo_render_secondary).  Pixels without a hit trace a ray that never enters the cube. */
static uint32_t mix32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

typedef struct {
    frame_job f;
    uint32_t n_secondary;
    oracle_hit *secondary;
} secondary_job;

static void secondary_fn(void *arg, size_t b, size_t e) {
    secondary_job *sj = (secondary_job *)arg;
    frame_job *j = &sj->f;
    const oracle_uniforms *u = j->u;
    trace_ctx c;
    memset(&c, 0, sizeof(c));
    c.nodes = j->nodes; c.n_nodes = j->n_nodes; c.flags = u->flags; c.visits = j->visits;
    float sl = sqrtf((u->sun_dir[0]*u->sun_dir[0] + u->sun_dir[1]*u->sun_dir[1]) + u->sun_dir[2]*u->sun_dir[2]);
    float sun[3] = {u->sun_dir[0] / sl, u->sun_dir[1] / sl, u->sun_dir[2] / sl};
    uint32_t width = (uint32_t)u->dimensions[0];
    size_t n = (size_t)j->w * (size_t)j->h;
    for (size_t i = b; i < e; i++) {
            int px = j->x0 + (int)(i % (size_t)j->w), py = j->y0 + (int)(i / (size_t)j->w);
            ray_t r = gen_ray(u, px, py);
            hitinfo_t h = octree_ray(&c, &r, 1);
            if (j->out) j->out[i] = pack_hit(&h);
            for (uint32_t k = 0; k < sj->n_secondary; k++) {
                ray_t sr = {{5.0f, 5.0f, 5.0f}, {1.0f, 1.0f, 1.0f}};
                if (h.hit) {
                    for (int a = 0; a < 3; a++) sr.pos[a] = h.pos[a] + h.normal[a] * 0.0000025f;
                    if (k == 0) {
                        for (int a = 0; a < 3; a++) sr.dir[a] = -sun[a];
                    } else {
                        uint32_t hs = mix32(((uint32_t)py * width + (uint32_t)px) * 4u + k + 0x9E3779B9u);
                        float d[3] = {(float)(int)(hs & 1023u) - 511.5f, (float)(int)((hs >> 10) & 1023u) - 511.5f,
                                      (float)(int)((hs >> 20) & 1023u) - 511.5f};
                        float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
                        for (int a = 0; a < 3; a++) d[a] = d[a] / len;
                        int flip = (h.normal[0] * d[0] + h.normal[1] * d[1]) + h.normal[2] * d[2] < 0.0f;
                        for (int a = 0; a < 3; a++) sr.dir[a] = flip ? -d[a] : d[a];
                    }
                }
                hitinfo_t sh = octree_ray(&c, &sr, 1);
                sj->secondary[(size_t)k * n + i] = pack_hit(&sh);
            }
        }
}

void oracle_secondary_frame(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0, int y0, int w,
                            int h, uint32_t n_secondary, oracle_hit *primary, oracle_hit *secondary, int n_threads) {
    secondary_job sj = {{nodes, n_nodes, u, x0, y0, w, h, primary, NULL, NULL, NULL}, n_secondary, secondary};
    par_for(secondary_fn, &sj, (size_t)w * (size_t)h, PAR_SPAN, n_threads);
}

void oracle_count_frame(uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, int x0, int y0,
                        int w, int h) {
    uint32_t *visits = (uint32_t *)calloc(n_nodes ? n_nodes : 1, sizeof(uint32_t));
    frame_job j = {nodes, n_nodes, u, x0, y0, w, h, NULL, NULL, NULL, visits};
    /* single thread: the traversal only reads pointers, so the final counters
     * (min(15, old + visits)) do not depend on ray order */
    if (u->flags & (ORACLE_F_SHOW_STEPS | ORACLE_F_SHOW_HITS)) frame_fn(&j, 0, (size_t)w * (size_t)h);
    else shade_fn(&j, 0, (size_t)w * (size_t)h); /* the shadow ray also counts (:276 passes primary=true) */
    for (size_t i = 0; i < n_nodes; i++) {
        uint32_t cnt = (nodes[i] & 15u) + visits[i];
        if (visits[i] > 15u || cnt > 15u) cnt = 15u;
        nodes[i] = (nodes[i] & ~15u) | cnt;
    }
    free(visits);
}

/* compute.wgsl:26-47 + adaptive.rs:22 */
void oracle_scan(const uint32_t *nodes, size_t n_nodes, uint32_t node_length, uint32_t *sub,
                 uint32_t *unsub, size_t capacity) {
    uint32_t ns = 0, nu = 0;
    for (size_t id = 0; id < n_nodes; id++) {
        uint32_t node = nodes[id];
        if (node == 0u) continue;                                                   /* :35-37 */
        uint32_t counter = node & 15u;
        if (counter == 0u && (node >> 4) < VOXEL_OFFSET && id < node_length) {     /* :40-42 */
            if ((size_t)nu + 1 < capacity) unsub[1 + nu] = (uint32_t)id;
            nu++;
        } else if (counter >= 4u && (node >> 4) > VOXEL_OFFSET && id < node_length) { /* :43-46 */
            if ((size_t)ns + 1 < capacity) sub[1 + ns] = (uint32_t)id;
            ns++;
        }
    }
    sub[0] = ns; unsub[0] = nu;
}
