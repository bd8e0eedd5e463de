// svo_host.cpp -- host-side data model behind include/svo_host.h: CpuOctree and Octree containers
// with the reference's method set (cpu_octree.rs, octree.rs), the .vox / .rsvo formats, mip
// colours (world.rs:234-336), camera matrices (render.rs:191-206) and deterministic scene
// generators for the benchmark configs.  Pure host C++ (no HIP calls).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "svo_hip.h"
#include "svo_host.h"

namespace {

constexpr uint32_t kVoxelOffset = SVO_VOXEL_OFFSET;
constexpr uint32_t kChunkOffset = SVO_CHUNK_OFFSET;

struct Rgb {
    uint8_t r = 0, g = 0, b = 0;
    bool is_zero() const { return !(r | g | b); }
    uint32_t cpu_value() const { return (uint32_t(r) << 16) | (uint32_t(g) << 8) | b; }  // octree.rs:32-34
    uint32_t gpu_word() const { return (kVoxelOffset + cpu_value()) << 4; }              // octree.rs:28-30
};

struct Vec3 {
    float x = 0, y = 0, z = 0;
};

// Octree::pos_offset (octree.rs:154-161)
Vec3 pos_offset(uint32_t child, uint32_t depth) {
    const float d = float(1u << depth);
    return {(float((child >> 2) & 1u) * 2.0f - 1.0f) / d, (float((child >> 1) & 1u) * 2.0f - 1.0f) / d,
            (float(child & 1u) * 2.0f - 1.0f) / d};
}

struct Located {
    size_t index;
    uint32_t depth;
    Vec3 pos;
};

// The `>=` point location shared by CpuOctree::find_voxel (cpu_octree.rs:48-76) and
// Octree::find_voxel (octree.rs:113-141); is_leaf tells when the walk ends.
template <class IsLeaf, class Next>
Located locate(Vec3 p, int64_t max_depth, IsLeaf is_leaf, Next next) {
    size_t base = 0;
    Vec3 c;
    for (uint32_t depth = 1;; ++depth) {
        const uint32_t child = (p.x >= c.x ? 4u : 0u) | (p.y >= c.y ? 2u : 0u) | (p.z >= c.z ? 1u : 0u);
        const Vec3 o = pos_offset(child, depth);
        c.x += o.x; c.y += o.y; c.z += o.z;
        const size_t at = base + child;
        if (is_leaf(at) || (max_depth >= 0 && int64_t(depth) == max_depth)) return {at, depth, c};
        base = next(at);
    }
}

void put_err(char *err, size_t n, const std::string &msg) {
    if (err && n) snprintf(err, n, "%s", msg.c_str());
}

uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (uint32_t(p[3]) << 24); }
void put32(uint8_t *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }

uint32_t mix32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}
uint32_t hash4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return mix32(a * 0x9E3779B1u ^ mix32(b * 0x85EBCA77u ^ mix32(c * 0xC2B2AE3Du ^ mix32(d + 0x27D4EB2Fu))));
}

}  // namespace

// ------------------------------------------------------------------------------------------
// CpuOctree
// ------------------------------------------------------------------------------------------
struct svo_cpu_octree {
    struct Node {
        uint32_t pointer;
        Rgb value;
    };
    std::vector<Node> nodes;
    Rgb top_mip{50, 255, 50};  // cpu_octree.rs:25

    // add_voxels (cpu_octree.rs:32-45): child-mask bit i -> a block reference, else empty
    void add_voxels(uint8_t mask) {
        for (int i = 0; i < 8; i++) {
            if ((mask >> i) & 1)
                nodes.push_back({kChunkOffset + uint32_t(nodes.size() % 8) + 1u, Rgb{255, 0, 0}});
            else
                nodes.push_back({kChunkOffset, Rgb{}});
        }
    }

    Located find_voxel(Vec3 p, int64_t max_depth) const {
        return locate(
            p, max_depth, [&](size_t i) { return nodes[i].pointer >= kChunkOffset; },
            [&](size_t i) { return size_t(nodes[i].pointer); });
    }

    // put_in_voxel / put_in_block (cpu_octree.rs:87-111): split the containing leaf until `depth`
    void put(Vec3 p, Node leaf, uint32_t depth) {
        for (;;) {
            const Located l = find_voxel(p, -1);
            if (l.depth == depth) {
                nodes[l.index] = leaf;
                return;
            }
            nodes[l.index].pointer = uint32_t(nodes.size());
            add_voxels(0);
        }
    }
};

namespace {

struct VoxFile {
    uint32_t size[3] = {0, 0, 0};
    std::vector<uint8_t> xyzi;
    uint32_t palette[256];
    bool ok = false;
};

// MagicaVoxel's default palette (the table of the .vox format description), for files saved without an RGBA chunk --
// dot_vox::load_bytes falls back to it and the reference loads such files (cpu_octree.rs:177-193).  Entry k of the
// array is the colour of FILE index k + 1, the position an RGBA chunk would give it: a 6x6x6 colour cube (blue fastest,
// steps of 0x33, black left out) followed by red, green, blue and grey ramps.  dot_vox's own copy of the table cannot be
// compared here (crate source absent): the index convention is the file format's.
void default_palette(uint32_t pal[256]) {
    static const uint8_t ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    for (uint32_t j = 0; j < 215; j++) {
        const uint32_t r = 0xff - 0x33 * (j / 36), g = 0xff - 0x33 * ((j / 6) % 6), b = 0xff - 0x33 * (j % 6);
        pal[j] = 0xff000000u | (b << 16) | (g << 8) | r;
    }
    for (uint32_t k = 0; k < 10; k++) {
        pal[215 + k] = 0xff000000u | ramp[k];
        pal[225 + k] = 0xff000000u | (uint32_t(ramp[k]) << 8);
        pal[235 + k] = 0xff000000u | (uint32_t(ramp[k]) << 16);
        pal[245 + k] = 0xff000000u | (uint32_t(ramp[k]) << 16) | (uint32_t(ramp[k]) << 8) | ramp[k];
    }
    pal[255] = 0;
}

// Minimal MagicaVoxel reader: first SIZE/XYZI pair (model 0) and the RGBA chunk.  This is what
// cpu_octree.rs:178-193 consumes from dot_vox 4.1.0 (i = file colour index - 1; palette words LE).
bool parse_vox(const uint8_t *d, size_t len, VoxFile &v, std::string &why) {
    if (len < 20 || memcmp(d, "VOX ", 4) || memcmp(d + 8, "MAIN", 4)) { why = "not a .vox file"; return false; }
    bool got_size = false, got_xyzi = false, got_rgba = false;
    size_t at = 20 + le32(d + 12);
    while (at + 12 <= len) {
        const uint32_t n = le32(d + at + 4), kids = le32(d + at + 8);
        const uint8_t *body = d + at + 12;
        if (at + 12 + size_t(n) > len) { why = "truncated chunk"; return false; }
        if (!got_size && !memcmp(d + at, "SIZE", 4) && n >= 12) {
            for (int i = 0; i < 3; i++) v.size[i] = le32(body + 4 * i);
            got_size = true;
        } else if (!got_xyzi && !memcmp(d + at, "XYZI", 4) && n >= 4) {
            const uint32_t count = le32(body);
            if (4 + size_t(count) * 4 > n) { why = "truncated XYZI"; return false; }
            v.xyzi.assign(body + 4, body + 4 + size_t(count) * 4);
            got_xyzi = true;
        } else if (!got_rgba && !memcmp(d + at, "RGBA", 4) && n >= 1024) {
            for (int i = 0; i < 256; i++) v.palette[i] = le32(body + 4 * i);
            got_rgba = true;
        }
        at += 12 + size_t(n) + kids;
    }
    if (!got_size || !got_xyzi) { why = "no model in .vox"; return false; }
    if (!got_rgba) default_palette(v.palette);
    return v.ok = true;
}

svo_cpu_octree *tree_from_voxels(uint32_t size, const uint8_t *xyzi, size_t n, const uint32_t *pal, std::string &why) {
    // cpu_octree.rs:184-189
    const float depthf = std::log2(float(int(size)));
    if (size == 0 || depthf != std::floor(depthf)) { why = "Voxel model size is not a power of 2!"; return nullptr; }
    const uint32_t depth = uint32_t(depthf);
    const float fs = float(int(size));
    auto *t = new svo_cpu_octree();
    t->add_voxels(0);  // CpuOctree::new(0), :191
    for (size_t k = 0; k < n; k++) {
        const uint8_t *v = xyzi + 4 * k;
        const uint32_t rgba = pal[v[3] ? v[3] - 1 : 0];  // dot_vox index convention, :193
        // axis swap (size - x - 1, z, y), scale to [-1, 1) (:194-200)
        Vec3 p{fs - float(v[0]) - 1.0f, float(v[2]), float(v[1])};
        p.x /= fs; p.y /= fs; p.z /= fs;
        p.x = p.x * 2.0f - 1.0f; p.y = p.y * 2.0f - 1.0f; p.z = p.z * 2.0f - 1.0f;
        t->put(p, {kChunkOffset, Rgb{uint8_t(rgba), uint8_t(rgba >> 8), uint8_t(rgba >> 16)}}, depth);
    }
    return t;
}

void average_children(const svo_cpu_octree &t, size_t first_child, Rgb &dst) {
    // world.rs:311-328: mean of the non-(0,0,0) children, `as u8` (saturating, NaN -> 0), max 1
    float sum[3] = {0, 0, 0}, div = 0;
    for (int i = 0; i < 8; i++) {
        const Rgb c = t.nodes[first_child + i].value;
        if (!c.is_zero()) { sum[0] += c.r; sum[1] += c.g; sum[2] += c.b; div += 1.0f; }
    }
    uint8_t out[3];
    for (int i = 0; i < 3; i++) {
        const float m = sum[i] / div;
        const uint8_t q = (m != m) ? 0 : (m <= 0.0f ? 0 : (m >= 255.0f ? 255 : uint8_t(m)));
        out[i] = std::max<uint8_t>(q, 1);
    }
    dst = Rgb{out[0], out[1], out[2]};
}

}  // namespace

extern "C" {

svo_cpu_octree *svo_cpu_octree_new(uint8_t mask) {
    auto *t = new svo_cpu_octree();
    t->add_voxels(mask);
    return t;
}

void svo_cpu_octree_free(svo_cpu_octree *t) { delete t; }
size_t svo_cpu_octree_len(const svo_cpu_octree *t) { return t->nodes.size(); }

svo_cpu_octree *svo_cpu_octree_load_vox(const uint8_t *data, size_t len, char *err, size_t errlen) {
    VoxFile v;
    std::string why;
    if (!parse_vox(data, len, v, why)) { put_err(err, errlen, why); return nullptr; }
    if (v.size[0] != v.size[1] || v.size[0] != v.size[2]) {  // cpu_octree.rs:180-182
        put_err(err, errlen, "Voxel model is not a cube!");
        return nullptr;
    }
    auto *t = tree_from_voxels(v.size[0], v.xyzi.data(), v.xyzi.size() / 4, v.palette, why);
    if (!t) put_err(err, errlen, why);
    return t;
}

svo_cpu_octree *svo_cpu_octree_from_voxels(uint32_t size, const uint8_t *xyzi, size_t n_voxels,
                                           const uint32_t *palette256, char *err, size_t errlen) {
    std::string why;
    auto *t = tree_from_voxels(size, xyzi, n_voxels, palette256, why);
    if (!t) put_err(err, errlen, why);
    return t;
}

// .rsvo: byte 16 = top level, LE u32 node count per level from byte 20, then one child-mask byte
// per node in BFS order (cpu_octree.rs:128-175)
svo_cpu_octree *svo_cpu_octree_load_rsvo(const uint8_t *data, size_t len, uint32_t octree_depth, char *err,
                                         size_t errlen) {
    if (len < 24) { put_err(err, errlen, "rsvo too short"); return nullptr; }
    const size_t top_level = data[16];
    const size_t masks = 20 + 4 * (top_level + 1);
    if (len <= masks) { put_err(err, errlen, "rsvo too short"); return nullptr; }
    if (octree_depth > top_level) {
        put_err(err, errlen, "Octree depth (" + std::to_string(octree_depth) + ") is greater than top level (" +
                                 std::to_string(top_level) + ")");
        return nullptr;
    }
    size_t usable = 0;  // masks consumed = nodes of the first octree_depth levels
    for (size_t i = 0; i < octree_depth; i++) usable += le32(data + 20 + 4 * i);
    auto *t = new svo_cpu_octree();
    t->add_voxels(data[masks]);
    size_t cursor = 1;
    for (size_t i = 0; i < t->nodes.size(); i++) {
        if (t->nodes[i].pointer <= kChunkOffset) continue;  // empty child: no mask byte
        if (cursor < usable) {
            if (masks + cursor >= len) { put_err(err, errlen, "rsvo truncated"); delete t; return nullptr; }
            t->nodes[i].pointer = uint32_t(t->nodes.size());
            t->add_voxels(data[masks + cursor]);
        }
        cursor++;
    }
    return t;
}

svo_cpu_octree *svo_cpu_octree_load_file(const char *path, uint32_t octree_depth, char *err, size_t errlen) {
    std::string p(path ? path : "");
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { put_err(err, errlen, "cannot open " + p); return nullptr; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    const size_t dot = p.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : p.substr(dot + 1);
    if (ext == "rsvo") return svo_cpu_octree_load_rsvo(buf.data(), buf.size(), octree_depth, err, errlen);
    if (ext == "vox") return svo_cpu_octree_load_vox(buf.data(), buf.size(), err, errlen);
    put_err(err, errlen, "Unknown file type");  // cpu_octree.rs:120
    return nullptr;
}

void svo_cpu_octree_put_in_voxel(svo_cpu_octree *t, const float pos[3], const uint8_t rgb[3], uint32_t depth) {
    t->put({pos[0], pos[1], pos[2]}, {kChunkOffset, Rgb{rgb[0], rgb[1], rgb[2]}}, depth);
}

void svo_cpu_octree_put_in_block(svo_cpu_octree *t, const float pos[3], uint32_t block_id, uint32_t depth) {
    t->put({pos[0], pos[1], pos[2]}, {kChunkOffset + block_id, Rgb{}}, depth);
}

void svo_cpu_octree_find_voxel(const svo_cpu_octree *t, const float pos[3], int64_t max_depth, uint64_t *index,
                               uint32_t *depth, float node_pos[3]) {
    const Located l = t->find_voxel({pos[0], pos[1], pos[2]}, max_depth);
    if (index) *index = l.index;
    if (depth) *depth = l.depth;
    if (node_pos) { node_pos[0] = l.pos.x; node_pos[1] = l.pos.y; node_pos[2] = l.pos.z; }
}

void svo_cpu_octree_get_node_mask(const svo_cpu_octree *t, size_t node, uint8_t rgb_out[24]) {
    for (int i = 0; i < 8; i++) {
        const Rgb v = t->nodes[node + i].value;
        rgb_out[3 * i] = v.r; rgb_out[3 * i + 1] = v.g; rgb_out[3 * i + 2] = v.b;
    }
}

void svo_cpu_octree_to_octree(const svo_cpu_octree *t, uint32_t *out) {
    size_t i = 0;
    for (const auto &n : t->nodes) out[i++] = n.pointer < kChunkOffset ? n.pointer << 4 : n.value.gpu_word();
}

void svo_cpu_octree_raw(const svo_cpu_octree *t, uint32_t *pointers, uint8_t *rgb) {
    size_t i = 0;
    for (const auto &n : t->nodes) {
        if (pointers) pointers[i] = n.pointer;
        if (rgb) { rgb[3 * i] = n.value.r; rgb[3 * i + 1] = n.value.g; rgb[3 * i + 2] = n.value.b; }
        i++;
    }
}

// generate_mip_tree for one chunk without block tables (world.rs:234-336): catalogue the interior
// nodes breadth-first, then average bottom-up.
void svo_cpu_octree_generate_mips(svo_cpu_octree *t, uint8_t top_mip[3]) {
    std::vector<uint32_t> order;  // interior nodes, BFS (parents before children)
    for (uint32_t c = 0; c < 8 && c < t->nodes.size(); c++)
        if (t->nodes[c].pointer < kChunkOffset) order.push_back(c);
    for (size_t head = 0; head < order.size(); head++) {
        const uint32_t kids = t->nodes[order[head]].pointer;
        for (uint32_t c = 0; c < 8; c++)
            if (t->nodes[kids + c].pointer < kChunkOffset) order.push_back(kids + c);
    }
    for (size_t k = order.size(); k-- > 0;) average_children(*t, t->nodes[order[k]].pointer, t->nodes[order[k]].value);
    average_children(*t, 0, t->top_mip);
    if (top_mip) { top_mip[0] = t->top_mip.r; top_mip[1] = t->top_mip.g; top_mip[2] = t->top_mip.b; }
}

int64_t svo_vox_parse(const uint8_t *data, size_t len, uint32_t size[3], uint8_t *xyzi, size_t xyzi_cap,
                      uint32_t *palette256, char *err, size_t errlen) {
    VoxFile v;
    std::string why;
    if (!parse_vox(data, len, v, why)) { put_err(err, errlen, why); return -1; }
    if (size) memcpy(size, v.size, sizeof v.size);
    if (xyzi) {
        if (xyzi_cap < v.xyzi.size()) { put_err(err, errlen, "xyzi buffer too small"); return -1; }
        memcpy(xyzi, v.xyzi.data(), v.xyzi.size());
    }
    if (palette256) memcpy(palette256, v.palette, sizeof v.palette);
    return int64_t(v.xyzi.size() / 4);
}

size_t svo_vox_write(uint32_t size, const uint8_t *xyzi, size_t n_voxels, const uint32_t *palette256, uint8_t *out,
                     size_t cap) {
    const size_t size_chunk = 12 + 12, xyzi_chunk = 12 + 4 + 4 * n_voxels, rgba_chunk = 12 + 1024;
    const size_t total = 8 + 12 + size_chunk + xyzi_chunk + rgba_chunk;
    if (!out || cap < total) return total;
    uint8_t *p = out;
    memcpy(p, "VOX ", 4); put32(p + 4, 150); p += 8;
    memcpy(p, "MAIN", 4); put32(p + 4, 0); put32(p + 8, uint32_t(size_chunk + xyzi_chunk + rgba_chunk)); p += 12;
    memcpy(p, "SIZE", 4); put32(p + 4, 12); put32(p + 8, 0);
    put32(p + 12, size); put32(p + 16, size); put32(p + 20, size); p += size_chunk;
    memcpy(p, "XYZI", 4); put32(p + 4, uint32_t(4 + 4 * n_voxels)); put32(p + 8, 0); put32(p + 12, uint32_t(n_voxels));
    memcpy(p + 16, xyzi, 4 * n_voxels); p += xyzi_chunk;
    memcpy(p, "RGBA", 4); put32(p + 4, 1024); put32(p + 8, 0);
    for (int i = 0; i < 256; i++) put32(p + 12 + 4 * i, palette256[i]);
    return total;
}

size_t svo_rsvo_write(const svo_cpu_octree *t, uint8_t *out, size_t cap) {
    // BFS over the tree in the loader's order: one mask byte per non-empty node
    std::vector<uint8_t> masks;
    std::vector<uint32_t> level_counts;
    std::vector<size_t> frontier{0}, nextf;  // child-group starts of the current level
    while (!frontier.empty()) {
        level_counts.push_back(uint32_t(frontier.size()));
        nextf.clear();
        for (size_t g : frontier) {
            uint8_t m = 0;
            for (int i = 0; i < 8; i++) {
                const auto &n = t->nodes[g + i];
                if (n.pointer != kChunkOffset || !n.value.is_zero()) m |= uint8_t(1u << i);  // interior, block or coloured leaf
                if (n.pointer < kChunkOffset) nextf.push_back(n.pointer);
            }
            masks.push_back(m);
        }
        // a non-empty child that is not interior is only representable on the last level
        size_t non_empty = 0;
        for (size_t g : frontier)
            for (int i = 0; i < 8; i++) non_empty += t->nodes[g + i].pointer != kChunkOffset || !t->nodes[g + i].value.is_zero();
        if (!nextf.empty() && nextf.size() != non_empty) return 0;
        frontier.swap(nextf);
    }
    // leaves of the last level carry no masks: the loader needs counts for levels 0..top_level where
    // octree_depth <= top_level selects how many mask levels are consumed
    const size_t top_level = level_counts.size();
    const size_t header = 20 + 4 * (top_level + 1);
    const size_t total = header + masks.size();
    if (!out || cap < total) return total;
    memset(out, 0, header);
    memcpy(out, "RSVO", 4);
    out[16] = uint8_t(top_level);
    for (size_t i = 0; i < top_level; i++) put32(out + 20 + 4 * i, level_counts[i]);
    put32(out + 20 + 4 * top_level, 0);
    memcpy(out + header, masks.data(), masks.size());
    return total;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Octree (host mirror of the device array)
// ------------------------------------------------------------------------------------------
struct svo_octree {
    std::vector<uint32_t> nodes;
    std::vector<Vec3> positions;
    std::vector<size_t> hole_stack;
    // words written since the last svo_octree_take_dirty (each index once): what an incremental upload has to send
    std::vector<uint32_t> dirty;
    std::vector<uint8_t> dirty_flag;
    void touch(size_t i) {
        if (dirty_flag.size() < nodes.size()) dirty_flag.resize(nodes.size(), 0);
        if (!dirty_flag[i]) { dirty_flag[i] = 1; dirty.push_back(uint32_t(i)); }
    }
};

extern "C" {

svo_octree *svo_octree_new(const uint8_t mask_rgb[24]) {
    auto *o = new svo_octree();
    for (uint32_t i = 0; i < 8; i++) {
        o->nodes.push_back(Rgb{mask_rgb[3 * i], mask_rgb[3 * i + 1], mask_rgb[3 * i + 2]}.gpu_word());
        o->positions.push_back(pos_offset(i, 1));
    }
    return o;
}

svo_octree *svo_octree_from_words(const uint32_t *words, size_t n) {
    auto *o = new svo_octree();
    o->nodes.assign(words, words + n);
    o->positions.assign(n, Vec3{});
    // recover node centres top-down so unsubdivide's position check works
    struct Item { size_t base; Vec3 c; uint32_t depth; };
    std::vector<Item> todo{{0, Vec3{}, 1}};
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        for (uint32_t i = 0; i < 8 && it.base + i < n; i++) {
            const Vec3 off = pos_offset(i, it.depth);
            const Vec3 c{it.c.x + off.x, it.c.y + off.y, it.c.z + off.z};
            o->positions[it.base + i] = c;
            const uint32_t ptr = o->nodes[it.base + i] >> 4;
            if (ptr < kVoxelOffset && ptr != 0 && ptr + 8 <= n && it.depth < 31) todo.push_back({ptr, c, it.depth + 1});
        }
    }
    return o;
}

void svo_octree_free(svo_octree *o) { delete o; }
size_t svo_octree_len(const svo_octree *o) { return o->nodes.size(); }
const uint32_t *svo_octree_raw_data(const svo_octree *o) { return o->nodes.data(); }
uint32_t svo_octree_get_node(const svo_octree *o, size_t index) { return o->nodes[index] >> 4; }
size_t svo_octree_holes(const svo_octree *o) { return o->hole_stack.size(); }
void svo_octree_set_node(svo_octree *o, size_t index, uint32_t word) { o->nodes[index] = word; o->touch(index); }
void svo_octree_position(const svo_octree *o, size_t index, float out[3]) {
    out[0] = o->positions[index].x; out[1] = o->positions[index].y; out[2] = o->positions[index].z;
}

int svo_octree_subdivide(svo_octree *o, size_t node, const uint8_t mask_rgb[24], uint32_t depth) {
    if ((o->nodes[node] >> 4) < kVoxelOffset) return -1;  // "Node already subdivided!" octree.rs:73-75
    const Vec3 c = o->positions[node];
    size_t first;
    if (!o->hole_stack.empty()) {  // reuse a freed group, octree.rs:78-84
        first = o->hole_stack.back();
        o->hole_stack.pop_back();
    } else {
        first = o->nodes.size();
        o->nodes.resize(first + 8);
        o->positions.resize(first + 8);
    }
    o->nodes[node] = uint32_t(first) << 4;
    o->touch(node);
    for (uint32_t i = 0; i < 8; i++) {
        const Vec3 off = pos_offset(i, depth);
        o->nodes[first + i] = Rgb{mask_rgb[3 * i], mask_rgb[3 * i + 1], mask_rgb[3 * i + 2]}.gpu_word();
        o->positions[first + i] = Vec3{c.x + off.x, c.y + off.y, c.z + off.z};
        o->touch(first + i);
    }
    return 0;
}

int svo_octree_unsubdivide(svo_octree *o, size_t node) {
    const uint32_t ptr = o->nodes[node] >> 4;
    if (ptr >= kVoxelOffset) return 1;  // "not subdivided", octree.rs:97-100
    const Vec3 c = o->positions[node];
    if (c.x == 0.0f && c.y == 0.0f && c.z == 0.0f) return -1;  // octree.rs:104-107
    o->hole_stack.push_back(ptr);
    o->nodes[node] = Rgb{255, 0, 0}.gpu_word();  // octree.rs:109
    o->touch(node);
    return 0;
}

void svo_octree_find_voxel(const svo_octree *o, const float pos[3], int64_t max_depth, uint64_t *index,
                           uint32_t *depth, float node_pos[3]) {
    const Located l = locate(
        Vec3{pos[0], pos[1], pos[2]}, max_depth, [&](size_t i) { return (o->nodes[i] >> 4) >= kVoxelOffset; },
        [&](size_t i) { return size_t(o->nodes[i] >> 4); });
    if (index) *index = l.index;
    if (depth) *depth = l.depth;
    if (node_pos) { node_pos[0] = l.pos.x; node_pos[1] = l.pos.y; node_pos[2] = l.pos.z; }
}

size_t svo_octree_take_dirty(svo_octree *o, uint32_t *indices, uint32_t *words, size_t cap) {
    const size_t n = o->dirty.size();
    if (!indices || !words || cap < n) return n;
    for (size_t k = 0; k < n; k++) {
        indices[k] = o->dirty[k];
        words[k] = o->nodes[o->dirty[k]];
        o->dirty_flag[o->dirty[k]] = 0;
    }
    o->dirty.clear();
    return n;
}

int svo_octree_expanded(const svo_octree *o, size_t size, uint32_t *out) {
    if (size < o->nodes.size()) return -1;
    std::copy(o->nodes.begin(), o->nodes.end(), out);
    std::fill(out + o->nodes.size(), out + size, 0u);
    return 0;
}

void svo_octree_pos_offset(uint32_t child_index, uint32_t depth, float out[3]) {
    const Vec3 v = pos_offset(child_index, depth);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// World: chunk table with block instancing (world.rs), chunk dumps, and the CPU half of the
// streaming loop (adaptive.rs)
// ------------------------------------------------------------------------------------------
struct svo_world {
    std::string path;
    std::map<uint32_t, svo_cpu_octree *> chunks;
    mutable std::string error;

    ~svo_world() {
        for (auto &kv : chunks) delete kv.second;
    }
    svo_cpu_octree *get(uint32_t id) const {
        auto it = chunks.find(id);
        return it == chunks.end() ? nullptr : it->second;
    }
};

namespace {

struct WorldLocated {
    uint32_t chunk;
    size_t index;
    uint32_t depth;
    Vec3 pos;
    bool ok;
};

// World::find_voxel (world.rs:201-232): the `>=` walk, hopping into the referenced chunk's root group at a
// block leaf.  The reference unwraps a missing chunk (panic); here ok = false.
WorldLocated world_locate(const svo_world &w, Vec3 p, int64_t max_depth) {
    uint32_t chunk = 0;
    size_t base = 0;
    Vec3 c;
    const svo_cpu_octree *cur = w.get(0);
    for (uint32_t depth = 1;; ++depth) {
        if (!cur) return {chunk, 0, depth, c, false};
        const uint32_t child = (p.x >= c.x ? 4u : 0u) | (p.y >= c.y ? 2u : 0u) | (p.z >= c.z ? 1u : 0u);
        const Vec3 o = pos_offset(child, depth);
        c.x += o.x; c.y += o.y; c.z += o.z;
        if (base + child >= cur->nodes.size()) return {chunk, base + child, depth, c, false};
        const uint32_t ptr = cur->nodes[base + child].pointer;
        if (ptr == kChunkOffset || (max_depth >= 0 && int64_t(depth) == max_depth)) return {chunk, base + child, depth, c, true};
        if (ptr > kChunkOffset) {
            chunk = ptr - kChunkOffset;
            cur = w.get(chunk);
            base = 0;
        } else {
            base = ptr;
        }
        if (depth >= 64) return {chunk, base, depth, c, false};  // cyclic chunk references
    }
}

// <id>.bin (world.rs:176-184, cpu_octree.rs:262-272): the reference dumps its Vec<Node> as raw memory.  Node is
// {pointer: u32, value: Voxel{r, g, b: u8}}: 8 bytes with rustc's layout (pointer at 0, r g b at 4..6, one
// padding byte); little-endian.  The layout is what rustc produces in practice, not a language guarantee.
constexpr size_t kBinNodeBytes = 8;

std::string chunk_file(const svo_world &w, uint32_t id) { return w.path + "/" + std::to_string(id) + ".bin"; }

svo_cpu_octree *tree_from_bin(const uint8_t *d, size_t len) {
    auto *t = new svo_cpu_octree();
    t->top_mip = Rgb{};  // from_bin, cpu_octree.rs:270
    t->nodes.resize(len / kBinNodeBytes);
    for (size_t i = 0; i < t->nodes.size(); i++) {
        const uint8_t *n = d + i * kBinNodeBytes;
        t->nodes[i] = {le32(n), Rgb{n[4], n[5], n[6]}};
    }
    return t;
}

bool read_file(const std::string &p, std::vector<uint8_t> &buf) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    uint8_t tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    return true;
}

int world_fail(svo_world *w, const std::string &msg) {
    w->error = msg;
    return -1;
}

// what process_subdivision does for one listed node (adaptive.rs:29-61); `loaded` counts chunk loads
// returns 1 subdivided, 0 skipped, -1 error
int subdivide_one(svo_world *w, svo_octree *o, size_t node_index, uint64_t *loaded) {
    if (node_index >= o->nodes.size()) return world_fail(w, "node index past the octree");
    if ((o->nodes[node_index] >> 4) < kVoxelOffset) return 0;  // "Doubleup!" :32-35
    const Vec3 pos = o->positions[node_index];
    uint32_t voxel_depth = 0;
    {
        const float p[3] = {pos.x, pos.y, pos.z};
        svo_octree_find_voxel(o, p, -1, nullptr, &voxel_depth, nullptr);
    }
    const WorldLocated l = world_locate(*w, pos, voxel_depth);
    if (!l.ok) return world_fail(w, "world walk left the loaded chunks (chunk " + std::to_string(l.chunk) + ")");
    const svo_cpu_octree *chunk = w->get(l.chunk);
    const uint32_t ptr = chunk->nodes[l.index].pointer;
    uint8_t mask[24];
    if (ptr < kChunkOffset) {  // :42-48
        if (size_t(ptr) + 8 > chunk->nodes.size()) return world_fail(w, "child pointer past the chunk");
        svo_cpu_octree_get_node_mask(chunk, ptr, mask);
    } else if (ptr > kChunkOffset) {  // :49-58
        const uint32_t id = ptr - kChunkOffset;
        const svo_cpu_octree *block = w->get(id);
        if (!block) {
            // the reference starts an asynchronous load and retries when the node is listed again
            if (svo_world_load_chunk(w, id) == 0 && loaded) ++*loaded;
            return 0;
        }
        if (block->nodes.size() < 8) return 0;  // a chunk whose nodes were dropped to save memory (world.rs:134)
        svo_cpu_octree_get_node_mask(block, 0, mask);
    } else {
        return 0;
    }
    return svo_octree_subdivide(o, node_index, mask, voxel_depth + 1) == 0 ? 1 : 0;
}

}  // namespace

extern "C" {

svo_world *svo_world_new(const char *path) {
    auto *w = new svo_world();
    w->path = path ? path : "";
    return w;
}
void svo_world_free(svo_world *w) { delete w; }
const char *svo_world_last_error(const svo_world *w) { return w->error.c_str(); }

int svo_world_insert(svo_world *w, uint32_t id, svo_cpu_octree *chunk) {
    if (!chunk) return world_fail(w, "null chunk");
    auto it = w->chunks.find(id);
    if (it != w->chunks.end()) {
        if (it->second != chunk) delete it->second;
        it->second = chunk;
    } else {
        w->chunks[id] = chunk;
    }
    return 0;
}
int svo_world_remove(svo_world *w, uint32_t id) {
    auto it = w->chunks.find(id);
    if (it == w->chunks.end()) return 1;
    delete it->second;
    w->chunks.erase(it);
    return 0;
}
svo_cpu_octree *svo_world_chunk(const svo_world *w, uint32_t id) { return w->get(id); }
size_t svo_world_chunk_ids(const svo_world *w, uint32_t *ids, size_t cap) {
    size_t n = 0;
    for (auto &kv : w->chunks) {
        if (ids && n < cap) ids[n] = kv.first;
        n++;
    }
    return n;
}

int svo_world_find_voxel(const svo_world *w, const float pos[3], int64_t max_depth, uint32_t *chunk, uint64_t *index,
                         uint32_t *depth, float node_pos[3]) {
    const WorldLocated l = world_locate(*w, {pos[0], pos[1], pos[2]}, max_depth);
    if (chunk) *chunk = l.chunk;
    if (index) *index = l.index;
    if (depth) *depth = l.depth;
    if (node_pos) { node_pos[0] = l.pos.x; node_pos[1] = l.pos.y; node_pos[2] = l.pos.z; }
    if (!l.ok) w->error = "world walk left the loaded chunks (chunk " + std::to_string(l.chunk) + ")";
    return l.ok ? 0 : -1;
}

// World::generate_mip_tree (world.rs:234-336): block leaves take the referenced chunk's top_mip first
// (:249-253, :273-279), then interior nodes are averaged bottom-up and the root average becomes top_mip.
int svo_world_generate_mip_tree(svo_world *w, uint32_t id, uint8_t top_mip[3]) {
    svo_cpu_octree *t = w->get(id);
    if (!t) return world_fail(w, "no chunk " + std::to_string(id));
    if (t->nodes.size() < 8) return world_fail(w, "chunk " + std::to_string(id) + " has no nodes");
    for (auto &n : t->nodes) {
        // every node of the array is reachable in a tree built by the loaders, so one linear pass equals
        // the reference's walk over reachable block leaves
        if (n.pointer > kChunkOffset) {
            const svo_cpu_octree *b = w->get(n.pointer - kChunkOffset);
            if (!b) return world_fail(w, "block " + std::to_string(n.pointer - kChunkOffset) + " is not loaded");
            n.value = b->top_mip;
        }
    }
    svo_cpu_octree_generate_mips(t, top_mip);
    return 0;
}

size_t svo_cpu_octree_bin(const svo_cpu_octree *t, uint8_t *out, size_t cap) {
    const size_t total = t->nodes.size() * kBinNodeBytes;
    if (!out || cap < total) return total;
    for (size_t i = 0; i < t->nodes.size(); i++) {
        uint8_t *n = out + i * kBinNodeBytes;
        put32(n, t->nodes[i].pointer);
        n[4] = t->nodes[i].value.r; n[5] = t->nodes[i].value.g; n[6] = t->nodes[i].value.b; n[7] = 0;
    }
    return total;
}

svo_cpu_octree *svo_cpu_octree_from_bin(const uint8_t *data, size_t len, char *err, size_t errlen) {
    if (len == 0 || len % kBinNodeBytes || (len / kBinNodeBytes) % 8) {
        put_err(err, errlen, "chunk dump is not a whole number of 8-node groups");
        return nullptr;
    }
    return tree_from_bin(data, len);
}

int svo_world_save_chunk(svo_world *w, uint32_t id) {
    const svo_cpu_octree *t = w->get(id);
    if (!t) return world_fail(w, "no chunk " + std::to_string(id));
    std::vector<uint8_t> buf(svo_cpu_octree_bin(t, nullptr, 0));
    svo_cpu_octree_bin(t, buf.data(), buf.size());
    FILE *f = fopen(chunk_file(*w, id).c_str(), "wb");
    if (!f) return world_fail(w, "cannot create " + chunk_file(*w, id));
    const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    return ok ? 0 : world_fail(w, "short write to " + chunk_file(*w, id));
}

// World::load_chunk (world.rs:186-198) without the tokio task: the chunk is there when the call returns.
int svo_world_load_chunk(svo_world *w, uint32_t id) {
    std::vector<uint8_t> buf;
    if (!read_file(chunk_file(*w, id), buf)) return world_fail(w, "cannot read " + chunk_file(*w, id));
    char why[128] = "";
    svo_cpu_octree *t = svo_cpu_octree_from_bin(buf.data(), buf.size(), why, sizeof why);
    if (!t) return world_fail(w, chunk_file(*w, id) + ": " + why);
    return svo_world_insert(w, id, t);
}

// World::load_world (world.rs:159-174): a world directory is opened by reading its root chunk 0.bin.
svo_world *svo_world_load(const char *path, char *err, size_t errlen) {
    svo_world *w = svo_world_new(path);
    if (svo_world_load_chunk(w, 0) != 0) {
        put_err(err, errlen, w->error);
        delete w;
        return nullptr;
    }
    return w;
}

int64_t svo_adaptive_subdivide(svo_world *w, svo_octree *o, const uint32_t *list, size_t n, uint64_t *chunks_loaded) {
    int64_t done = 0;
    if (chunks_loaded) *chunks_loaded = 0;
    for (size_t i = 0; i < n; i++) {
        const int r = subdivide_one(w, o, list[i], chunks_loaded);
        if (r < 0) return -1;
        done += r;
    }
    return done;
}

// process_unsubdivision (adaptive.rs:93-121)
int64_t svo_adaptive_unsubdivide(svo_world *w, svo_octree *o, const uint32_t *list, size_t n) {
    int64_t done = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t node_index = list[i];
        if (node_index >= o->nodes.size()) return world_fail(w, "node index past the octree");
        const int r = svo_octree_unsubdivide(o, node_index);  // :95
        if (r < 0) return world_fail(w, "Tried to unsubdivide a node without position!");
        const Vec3 pos = o->positions[node_index];
        const float p[3] = {pos.x, pos.y, pos.z};
        uint32_t voxel_depth = 0;
        svo_octree_find_voxel(o, p, -1, nullptr, &voxel_depth, nullptr);
        const WorldLocated l = world_locate(*w, pos, voxel_depth);
        if (!l.ok) return world_fail(w, "world walk left the loaded chunks (chunk " + std::to_string(l.chunk) + ")");
        const auto node = w->get(l.chunk)->nodes[l.index];
        if (node.pointer > kChunkOffset) {  // :104-110: streamed chunks are dropped, blocks stay
            const uint32_t id = node.pointer - kChunkOffset;
            if (id >= kChunkOffset / 2) svo_world_remove(w, id);
        }
        o->nodes[node_index] = node.value.gpu_word();  // :117
        o->touch(node_index);
        done += r == 0;
    }
    return done;
}

// The streaming loop's fixed point for a given view rule, without a device: pass after pass, every leaf whose
// world node has children is subdivided in index order (= process_subdivision fed all such leaves, sorted), so
// the array comes out breadth-first.  A leaf at depth d is refined while d < max_depth and, if lod_c > 0,
// 2^d < lod_c / distance(cam, its cube) -- the rule of the terrain generator below.  Stops at max_words.
uint64_t svo_world_expand(svo_world *w, svo_octree *o, uint32_t max_depth, const float cam[3], float lod_c,
                          uint64_t max_words) {
    struct Leaf { size_t index; uint32_t depth; };
    std::vector<Leaf> frontier, next;
    for (size_t i = 0; i < o->nodes.size(); i++) {
        if ((o->nodes[i] >> 4) < kVoxelOffset) continue;
        const Vec3 pos = o->positions[i];
        const float p[3] = {pos.x, pos.y, pos.z};
        uint32_t d = 0;
        svo_octree_find_voxel(o, p, -1, nullptr, &d, nullptr);
        frontier.push_back({i, d});
    }
    uint64_t done = 0;
    while (!frontier.empty()) {
        next.clear();
        for (const Leaf &lf : frontier) {
            if (lf.depth >= max_depth) continue;
            if (o->nodes.size() + 8 > max_words && o->hole_stack.empty()) return done;
            if (lod_c > 0.0f && cam) {
                const Vec3 c = o->positions[lf.index];
                const float h = 1.0f / float(1u << lf.depth);  // half edge of a depth-d cube
                float d2 = 0.0f;
                const float cc[3] = {c.x, c.y, c.z};
                for (int k = 0; k < 3; k++) {
                    const float lo = cc[k] - h, hi = cc[k] + h;
                    const float d = cam[k] < lo ? lo - cam[k] : (cam[k] > hi ? cam[k] - hi : 0.0f);
                    d2 += d * d;
                }
                const float r = std::sqrt(d2);
                if (!(float(1u << lf.depth) * r < lod_c)) continue;
            }
            const size_t before_holes = o->hole_stack.size();
            const size_t before = o->nodes.size();
            const int r = subdivide_one(w, o, lf.index, nullptr);
            if (r < 0) return done;
            if (r == 1) {
                const size_t first = before_holes ? size_t(o->nodes[lf.index] >> 4) : before;
                for (size_t k = 0; k < 8; k++) next.push_back({first + k, lf.depth + 1});
                done++;
            }
        }
        frontier.swap(next);
    }
    return done;
}

// ------------------------------------------------------------------------------------------
// camera (render.rs:191-206, main.rs:139-162).  cgmath's operation order is not available, so
// these are the textbook forms; the matrices are inputs of the device path.
// ------------------------------------------------------------------------------------------
void svo_camera_matrices(const float pos[3], const float look[3], float fov_deg, float width, float height,
                         float camera[16], float camera_inverse[16]) {
    auto norm = [](float v[3]) {
        const float s = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] *= s; v[1] *= s; v[2] *= s;
    };
    float f[3] = {(pos[0] + look[0]) - pos[0], (pos[1] + look[1]) - pos[1], (pos[2] + look[2]) - pos[2]};
    norm(f);
    float s[3] = {f[1] * 0.0f - f[2] * 1.0f, f[2] * 0.0f - f[0] * 0.0f, f[0] * 1.0f - f[1] * 0.0f};  // f x (0,1,0)
    norm(s);
    const float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    const float view[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
                            -(pos[0] * s[0] + pos[1] * s[1] + pos[2] * s[2]),
                            -(pos[0] * u[0] + pos[1] * u[1] + pos[2] * u[2]),
                            pos[0] * f[0] + pos[1] * f[1] + pos[2] * f[2], 1};
    const float aspect = height / width;  // render.rs:200 passes H/W
    const float sc = 1.0f / std::tan((fov_deg / 2.0f) * (3.14159265358979323846f / 180.0f));
    const float proj[16] = {aspect * sc, 0, 0, 0, 0, sc, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float acc = 0;
            for (int k = 0; k < 4; k++) acc += proj[k * 4 + r] * view[c * 4 + k];
            camera[c * 4 + r] = acc;
        }
    // inverse by cofactors
    const float *m = camera;
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
    const float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
    if (det == 0.0f) { memset(camera_inverse, 0, 64); return; }
    const float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) camera_inverse[i] = t[i] * inv_det;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// deterministic scene generators (BFS layout: a node's 8 children are appended when it is
// dequeued, so the array is level-ordered with the root group at words 0..7)
// ------------------------------------------------------------------------------------------
namespace {

struct Builder {
    uint32_t *out;
    uint64_t cap, n = 0;
    bool full() const { return n + 8 > cap; }
    uint64_t alloc8() { const uint64_t at = n; n += 8; return at; }
};

uint32_t leaf_colour(uint32_t h) {
    const uint32_t r = 1 + (h & 0xFF) % 255, g = 1 + ((h >> 8) & 0xFF) % 255, b = 1 + ((h >> 16) & 0xFF) % 255;
    return (kVoxelOffset + ((r << 16) | (g << 8) | b)) << 4;
}
constexpr uint32_t kEmptyWord = kVoxelOffset << 4;

// distance from the LOD centre to the cube (x,y,z) of level `level` (cell size 2 / 2^level)
float cube_distance(const float cam[3], uint32_t level, uint32_t x, uint32_t y, uint32_t z) {
    const float s = 2.0f / float(1u << level);
    const uint32_t c[3] = {x, y, z};
    float d2 = 0;
    for (int i = 0; i < 3; i++) {
        const float lo = -1.0f + s * float(c[i]), hi = lo + s;
        const float d = cam[i] < lo ? lo - cam[i] : (cam[i] > hi ? cam[i] - hi : 0.0f);
        d2 += d * d;
    }
    return std::sqrt(d2);
}

bool lod_refine(const svo_terrain_params &p, uint32_t level, uint32_t x, uint32_t y, uint32_t z) {
    if (level >= p.max_depth) return false;
    if (level < p.min_depth) return true;
    const float r = cube_distance(p.cam, level, x, y, z);
    return float(1u << level) * r < p.lod_c;
}

struct Terrain {
    uint32_t seed, max_depth;
    int32_t amp[32], rem[32];  // per-level offset amplitude and the bound of all deeper offsets
    Terrain(uint32_t seed_, uint32_t max_depth_) : seed(seed_), max_depth(max_depth_) {
        const int64_t R = int64_t(1) << max_depth;
        for (uint32_t d = 0; d < 32; d++) amp[d] = d <= max_depth ? int32_t((R * 2 / 5) >> d) : 0;
        for (uint32_t d = 0; d < 32; d++) {
            int64_t s = 0;
            for (uint32_t k = d + 1; k <= max_depth; k++) s += amp[k] / 2 + 1;
            rem[d] = int32_t(s);
        }
    }
    int32_t root_height() const { return int32_t((int64_t(1) << max_depth) / 2); }
    // height of cell (i, j) of level d given its parent's height
    int32_t child_height(int32_t parent_h, uint32_t d, uint32_t i, uint32_t j) const {
        const int32_t u = int32_t(hash4(i, j, d, seed) & 0xFFFFu) - 32768;
        return parent_h + int32_t((int64_t(u) * amp[d]) >> 16);
    }
};

}  // namespace

extern "C" {

int32_t svo_gen_terrain_height(uint32_t seed, uint32_t max_depth, uint32_t X, uint32_t Z) {
    Terrain t(seed, max_depth);
    int32_t h = t.root_height();
    for (uint32_t d = 1; d <= max_depth; d++) h = t.child_height(h, d, X >> (max_depth - d), Z >> (max_depth - d));
    return h;
}

uint64_t svo_gen_terrain(const svo_terrain_params *p, uint32_t *out, uint64_t cap) {
    if (!p || !out || p->max_depth < 1 || p->max_depth > 20) return 0;
    const uint64_t limit = std::min<uint64_t>({cap, p->max_words ? p->max_words : cap, uint64_t(kVoxelOffset)});
    if (limit < 8) return 0;
    Terrain t(p->seed, p->max_depth);
    Builder b{out, limit};
    struct Job {
        uint32_t word;   // index of the parent word to patch (0xFFFFFFFF for the root)
        uint32_t x, y, z;
        int32_t h;       // column height estimate of this cell
        uint8_t level;
    };
    std::deque<Job> queue;
    queue.push_back({0xFFFFFFFFu, 0, 0, 0, t.root_height(), 0});
    while (!queue.empty()) {
        const Job j = queue.front();
        queue.pop_front();
        if (b.full()) {  // cap reached: the node stays a coarse leaf
            if (j.word != 0xFFFFFFFFu) {
                const uint32_t s = 1u << (p->max_depth - j.level);
                const bool solid = int64_t(2 * j.y + 1) * s < 2 * int64_t(j.h);
                out[j.word] = solid ? leaf_colour(hash4(j.x, j.y, j.z, p->seed ^ j.level)) : kEmptyWord;
            }
            continue;
        }
        const uint64_t base = b.alloc8();
        if (j.word != 0xFFFFFFFFu) out[j.word] = uint32_t(base) << 4;
        const uint32_t cl = j.level + 1u;
        const int64_t s = int64_t(1) << (p->max_depth - cl);  // child cell size in finest voxels
        for (uint32_t c = 0; c < 8; c++) {
            const uint32_t cx = 2 * j.x + ((c >> 2) & 1u), cy = 2 * j.y + ((c >> 1) & 1u), cz = 2 * j.z + (c & 1u);
            const int32_t h = t.child_height(j.h, cl, cx, cz);
            const int64_t lo = int64_t(cy) * s, hi = lo + s;
            uint32_t word;
            if (hi <= int64_t(h) - t.rem[cl]) {
                word = leaf_colour(hash4(cx, cy, cz, p->seed ^ cl));          // whole cube under the surface
            } else if (lo >= int64_t(h) + t.rem[cl]) {
                word = kEmptyWord;                                             // whole cube above it
            } else if (lod_refine(*p, cl, cx, cy, cz)) {
                word = kEmptyWord;  // patched when dequeued
                queue.push_back({uint32_t(base + c), cx, cy, cz, h, uint8_t(cl)});
            } else {
                const bool solid = (2 * lo + s) < 2 * int64_t(h);             // cube centre under the surface
                word = solid ? leaf_colour(hash4(cx, cy, cz, p->seed ^ cl)) : kEmptyWord;
            }
            out[base + c] = word;
        }
    }
    return b.n;
}

uint64_t svo_gen_fractal(const svo_terrain_params *p, uint32_t *out, uint64_t cap) {
    // Sierpinski-tetrahedron octree (children with even coordinate parity survive), refined by the
    // same distance rule as the terrain so a depth-20 tree stays under the 2^27-word layout cap.
    if (!p || !out || p->max_depth < 1 || p->max_depth > 24) return 0;
    const uint64_t limit = std::min<uint64_t>({cap, p->max_words ? p->max_words : cap, uint64_t(kVoxelOffset)});
    if (limit < 8) return 0;
    Builder b{out, limit};
    struct Job { uint32_t word, x, y, z; uint8_t level; };
    std::deque<Job> queue;
    queue.push_back({0xFFFFFFFFu, 0, 0, 0, 0});
    while (!queue.empty()) {
        const Job j = queue.front();
        queue.pop_front();
        if (b.full()) {
            if (j.word != 0xFFFFFFFFu) out[j.word] = leaf_colour(hash4(j.x, j.y, j.z, p->seed ^ j.level));
            continue;
        }
        const uint64_t base = b.alloc8();
        if (j.word != 0xFFFFFFFFu) out[j.word] = uint32_t(base) << 4;
        const uint32_t cl = j.level + 1u;
        for (uint32_t c = 0; c < 8; c++) {
            const uint32_t bx = (c >> 2) & 1u, by = (c >> 1) & 1u, bz = c & 1u;
            const uint32_t cx = 2 * j.x + bx, cy = 2 * j.y + by, cz = 2 * j.z + bz;
            uint32_t word = kEmptyWord;
            if (((bx ^ by ^ bz) & 1u) == 0u) {
                if (lod_refine(*p, cl, cx, cy, cz)) queue.push_back({uint32_t(base + c), cx, cy, cz, uint8_t(cl)});
                else word = leaf_colour(hash4(cx, cy, cz, p->seed ^ cl));
            }
            out[base + c] = word;
        }
    }
    return b.n;
}

uint64_t svo_gen_random(uint32_t seed, uint32_t max_depth, float p_split, float p_solid, uint64_t max_words,
                        uint32_t *out, uint64_t cap) {
    const uint64_t limit = std::min<uint64_t>({cap, max_words ? max_words : cap, uint64_t(kVoxelOffset)});
    if (!out || limit < 8 || max_depth < 1) return 0;
    Builder b{out, limit};
    struct Job { uint32_t word; uint8_t level; };
    std::deque<Job> queue;
    queue.push_back({0xFFFFFFFFu, 0});
    uint32_t counter = 0;
    const uint32_t t_split = uint32_t(double(p_split) * 4294967295.0), t_solid = uint32_t(double(p_solid) * 4294967295.0);
    while (!queue.empty()) {
        const Job j = queue.front();
        queue.pop_front();
        if (b.full()) {
            if (j.word != 0xFFFFFFFFu) out[j.word] = leaf_colour(mix32(seed ^ j.word));
            continue;
        }
        const uint64_t base = b.alloc8();
        if (j.word != 0xFFFFFFFFu) out[j.word] = uint32_t(base) << 4;
        for (uint32_t c = 0; c < 8; c++) {
            const uint32_t h1 = hash4(counter++, c, j.level, seed), h2 = mix32(h1 ^ 0xA511E9B3u);
            uint32_t word = kEmptyWord;
            if (uint32_t(j.level + 1) < max_depth && h1 < t_split) queue.push_back({uint32_t(base + c), uint8_t(j.level + 1)});
            else if (h2 < t_solid) word = leaf_colour(mix32(h2));
            out[base + c] = word;
        }
    }
    return b.n;
}

// Re-linearisation of a node array: the same tree, its child groups in another order.  Groups of the levels
// 1 .. block_level in breadth-first order first (what the reference's .rsvo loader and this repository's generators emit for the whole
// tree), then every subtree below a level-`block_level` interior word as ONE contiguous block, breadth-first inside: the levels a ray walks
// below that word lie within one block instead of one region of the array per level.  out: the new array (at most n words: words that are
// not reachable from the root are dropped); perm (optional): perm[new word] = old word, so that a hit's voxel index can be translated
// back.  Returns the number of words written, 0 for a malformed tree (pointer out of range, deeper than 31 levels, a cycle).
uint64_t svo_nodes_relayout(const uint32_t *words, uint64_t n, uint32_t block_level, uint32_t *out, uint32_t *perm) {
    if (!words || !out || n < 8) return 0;
    std::vector<uint32_t> order;           // old group start of the k-th group of the new array
    order.reserve(n / 8);
    std::vector<uint32_t> cur{0u}, next, roots;  // BFS frontier (old group starts); subtree roots found at block_level
    for (uint32_t level = 1; !cur.empty(); level++) {
        if (level > 31) return 0;
        next.clear();
        for (uint32_t g : cur) {
            if (uint64_t(g) + 8 > n || order.size() >= n / 8) return 0;
            order.push_back(g);
            for (uint32_t c = 0; c < 8; c++) {
                const uint32_t ptr = words[g + c] >> 4;
                if (ptr < kVoxelOffset) (level == block_level ? roots : next).push_back(ptr);
            }
        }
        cur.swap(next);
    }
    for (uint32_t r : roots) {  // one block per subtree, breadth-first inside
        cur.assign(1, r);
        for (uint32_t level = block_level + 1; !cur.empty(); level++) {
            if (level > 31) return 0;
            next.clear();
            for (uint32_t g : cur) {
                if (uint64_t(g) + 8 > n || order.size() >= n / 8) return 0;
                order.push_back(g);
                for (uint32_t c = 0; c < 8; c++) {
                    const uint32_t ptr = words[g + c] >> 4;
                    if (ptr < kVoxelOffset) next.push_back(ptr);
                }
            }
            cur.swap(next);
        }
    }
    std::vector<uint32_t> new_of(n / 8 + 1, 0xFFFFFFFFu);  // old group number -> new group start
    for (size_t k = 0; k < order.size(); k++) {
        if (order[k] % 8u != 0u || new_of[order[k] / 8u] != 0xFFFFFFFFu) return 0;  // (groups are 8-aligned in every array this library builds; a group reached twice: not a tree)
        new_of[order[k] / 8u] = uint32_t(k * 8);
    }
    for (size_t k = 0; k < order.size(); k++)
        for (uint32_t c = 0; c < 8; c++) {
            const uint32_t w = words[order[k] + c], ptr = w >> 4;
            out[k * 8 + c] = ptr < kVoxelOffset ? ((new_of[ptr / 8u] << 4) | (w & 15u)) : w;
            if (perm) perm[k * 8 + c] = order[k] + c;
        }
    return uint64_t(order.size()) * 8;
}

uint32_t svo_nodes_max_depth(const uint32_t *words, uint64_t n) {
    if (!words || n < 8) return 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};
    uint32_t deepest = 0;
    uint64_t visited = 0;
    while (!todo.empty()) {
        const auto [base, depth] = todo.back();
        todo.pop_back();
        if (++visited > n) return 0;  // cycle
        deepest = std::max(deepest, depth);
        for (uint32_t c = 0; c < 8; c++) {
            if (base + c >= n) return 0;
            const uint32_t ptr = words[base + c] >> 4;
            if (ptr < kVoxelOffset) {
                if (uint64_t(ptr) + 8 > n || depth >= 31) return 0;
                todo.push_back({ptr, depth + 1});
            }
        }
    }
    return deepest;
}

}  // extern "C"
