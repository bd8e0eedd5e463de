// svo_render.hpp -- C++ host-side mirror of the reference's dispatch API over the C ABI (include/svo_hip.h,
// include/svo_host.h): the same type and method names, argument meaning and error behaviour as
//   src/gpu.rs      Gpu::new
//   src/render.rs   Render::{new, update, render, resize}, Uniforms
//   src/compute.rs  Compute::{new, update}
//   src/octree.rs   Octree        src/cpu_octree.rs  CpuOctree      src/main.rs  Settings, Character
// Where the reference unwraps/panics (gpu.rs:24,39; adaptive.rs:66,124; octree.rs:73-75) these classes throw
// svo::Error.  Header-only; link against libsvo_hip.so.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "svo_hip.h"
#include "svo_host.h"

namespace svo {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

struct Voxel {  // octree.rs:7-35
    uint8_t r = 0, g = 0, b = 0;
    uint32_t to_cpu_value() const { return (uint32_t(r) << 16) | (uint32_t(g) << 8) | b; }
    uint32_t to_value() const { return (SVO_VOXEL_OFFSET + to_cpu_value()) << 4; }
};

struct Settings {  // main.rs:116-120, defaults app.rs:23-27
    uint32_t octree_depth = 12;
    float fov = 90.0f;
    float sensitivity = 0.00005f;
};

struct Character {  // main.rs:122-137
    float pos[3] = {0.1f, 0.2f, -1.5f};
    float look[3] = {0.0f, 0.0f, 1.5f};
};

class Gpu {  // gpu.rs:3-50 (wgpu instance/adapter/device/queue -> one HIP context + stream)
  public:
    explicit Gpu(int device = 0) {
        int rc = svo_ctx_create(device, &ctx_);
        if (rc != SVO_OK) throw Error(rc, "svo_ctx_create failed (no usable HIP device?)");
    }
    ~Gpu() { svo_ctx_destroy(ctx_); }
    Gpu(const Gpu &) = delete;
    Gpu &operator=(const Gpu &) = delete;
    svo_ctx *ctx() const { return ctx_; }
    void check(int rc) const {
        if (rc != SVO_OK) throw Error(rc, svo_last_error(ctx_));
    }
    void poll_wait() const { check(svo_sync(ctx_)); }  // device.poll(Maintain::Wait)
    void set_option(int option, int64_t value) const { check(svo_set_option(ctx_, option, value)); }

  private:
    svo_ctx *ctx_ = nullptr;
};

class CpuOctree {  // cpu_octree.rs:17-273
  public:
    explicit CpuOctree(uint8_t mask = 0) : t_(svo_cpu_octree_new(mask)) {}
    static CpuOctree load_file(const std::string &file, uint32_t octree_depth) {  // :113-125, Err(String) -> throw
        char err[256] = {0};
        svo_cpu_octree *t = svo_cpu_octree_load_file(file.c_str(), octree_depth, err, sizeof err);
        if (!t) throw Error(SVO_ERR_ARG, err);
        return CpuOctree(t);
    }
    ~CpuOctree() { svo_cpu_octree_free(t_); }
    CpuOctree(CpuOctree &&o) noexcept : t_(std::exchange(o.t_, nullptr)) {}
    CpuOctree(const CpuOctree &) = delete;
    size_t len() const { return svo_cpu_octree_len(t_); }
    void put_in_voxel(const float pos[3], Voxel v, uint32_t depth) {
        const uint8_t rgb[3] = {v.r, v.g, v.b};
        svo_cpu_octree_put_in_voxel(t_, pos, rgb, depth);
    }
    std::vector<uint32_t> to_octree_words() const {  // to_octree(), :233-252
        std::vector<uint32_t> w(len());
        svo_cpu_octree_to_octree(t_, w.data());
        return w;
    }
    Voxel generate_mip_tree() {  // world.rs:234-336
        uint8_t top[3];
        svo_cpu_octree_generate_mips(t_, top);
        return Voxel{top[0], top[1], top[2]};
    }
    void put_in_block(const float pos[3], uint32_t block_id, uint32_t depth) { svo_cpu_octree_put_in_block(t_, pos, block_id, depth); }
    std::vector<uint8_t> bin() const {  // CpuOctree::bin, :262-264
        std::vector<uint8_t> b(svo_cpu_octree_bin(t_, nullptr, 0));
        svo_cpu_octree_bin(t_, b.data(), b.size());
        return b;
    }
    static CpuOctree from_bin(const std::vector<uint8_t> &bin) {  // :266-272
        char err[256] = {0};
        svo_cpu_octree *t = svo_cpu_octree_from_bin(bin.data(), bin.size(), err, sizeof err);
        if (!t) throw Error(SVO_ERR_ARG, err);
        return CpuOctree(t);
    }
    svo_cpu_octree *raw() const { return t_; }
    svo_cpu_octree *release() { return std::exchange(t_, nullptr); }  // hand the tree to a World

  private:
    explicit CpuOctree(svo_cpu_octree *t) : t_(t) {}
    svo_cpu_octree *t_;
};

class Octree {  // octree.rs:43-162
  public:
    explicit Octree(const std::vector<uint32_t> &words) : o_(svo_octree_from_words(words.data(), words.size())) {}
    explicit Octree(const Voxel (&mask)[8]) {
        uint8_t rgb[24];
        for (int i = 0; i < 8; i++) { rgb[3 * i] = mask[i].r; rgb[3 * i + 1] = mask[i].g; rgb[3 * i + 2] = mask[i].b; }
        o_ = svo_octree_new(rgb);
    }
    ~Octree() { svo_octree_free(o_); }
    Octree(const Octree &) = delete;
    size_t len() const { return svo_octree_len(o_); }
    const uint32_t *raw_data() const { return svo_octree_raw_data(o_); }
    uint32_t get_node(size_t i) const { return svo_octree_get_node(o_, i); }
    void subdivide(size_t node, const Voxel (&mask)[8], uint32_t depth) {
        uint8_t rgb[24];
        for (int i = 0; i < 8; i++) { rgb[3 * i] = mask[i].r; rgb[3 * i + 1] = mask[i].g; rgb[3 * i + 2] = mask[i].b; }
        if (svo_octree_subdivide(o_, node, rgb, depth) != 0) throw Error(SVO_ERR_STATE, "Node already subdivided!");
    }
    bool unsubdivide(size_t node) {
        int rc = svo_octree_unsubdivide(o_, node);
        if (rc < 0) throw Error(SVO_ERR_STATE, "Tried to unsubdivide a node without position!");
        return rc == 0;
    }
    // words written since the previous call (index, value), each index once: the input of Render::scatter_nodes
    std::pair<std::vector<uint32_t>, std::vector<uint32_t>> take_dirty() {
        const size_t n = svo_octree_take_dirty(o_, nullptr, nullptr, 0);
        std::vector<uint32_t> idx(n), val(n);
        if (n) svo_octree_take_dirty(o_, idx.data(), val.data(), n);
        return {std::move(idx), std::move(val)};
    }
    svo_octree *raw() const { return o_; }

  private:
    svo_octree *o_;
};

class World {  // world.rs:5-336: chunk table with block instancing
  public:
    struct Found { uint32_t chunk; size_t index; uint32_t depth; float pos[3]; };
    explicit World(const std::string &path = "") : w_(svo_world_new(path.c_str())) {}
    static World load_world(const std::string &path) {  // :159-174
        char err[256] = {0};
        svo_world *w = svo_world_load(path.c_str(), err, sizeof err);
        if (!w) throw Error(SVO_ERR_ARG, err);
        return World(w);
    }
    ~World() { svo_world_free(w_); }
    World(World &&o) noexcept : w_(std::exchange(o.w_, nullptr)) {}
    World(const World &) = delete;
    void insert(uint32_t id, CpuOctree &&chunk) { check(svo_world_insert(w_, id, chunk.release())); }  // chunks.insert
    bool remove(uint32_t id) { return svo_world_remove(w_, id) == 0; }
    bool contains(uint32_t id) const { return svo_world_chunk(w_, id) != nullptr; }
    Found find_voxel(const float pos[3], int64_t max_depth = -1) const {  // :201-232; the reference panics on a missing chunk
        Found f{};
        uint64_t index = 0;
        check(svo_world_find_voxel(w_, pos, max_depth, &f.chunk, &index, &f.depth, f.pos));
        f.index = size_t(index);
        return f;
    }
    Voxel generate_mip_tree(uint32_t id) {  // :234-336
        uint8_t top[3];
        check(svo_world_generate_mip_tree(w_, id, top));
        return Voxel{top[0], top[1], top[2]};
    }
    void save_chunk(uint32_t id) { check(svo_world_save_chunk(w_, id)); }  // :176-184
    void load_chunk(uint32_t id) { check(svo_world_load_chunk(w_, id)); }  // :186-198, synchronous
    Octree root_octree() const {  // App::new, app.rs:47-48
        uint8_t rgb[24];
        const svo_cpu_octree *root = svo_world_chunk(w_, 0);
        if (!root) throw Error(SVO_ERR_STATE, "world has no root chunk");
        svo_cpu_octree_get_node_mask(root, 0, rgb);
        Voxel mask[8];
        for (int i = 0; i < 8; i++) mask[i] = Voxel{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        return Octree(mask);
    }
    // process_subdivision / process_unsubdivision over the lists Compute::read_lists returns (adaptive.rs:29-61, 93-121)
    size_t process_subdivision(const std::vector<uint32_t> &list, Octree &octree) {
        const int64_t n = svo_adaptive_subdivide(w_, octree.raw(), list.data(), list.size(), nullptr);
        check(n < 0 ? -1 : 0);
        return size_t(n);
    }
    size_t process_unsubdivision(const std::vector<uint32_t> &list, Octree &octree) {
        const int64_t n = svo_adaptive_unsubdivide(w_, octree.raw(), list.data(), list.size());
        check(n < 0 ? -1 : 0);
        return size_t(n);
    }
    uint64_t expand(Octree &octree, uint32_t max_depth, const float *cam = nullptr, float lod_c = 0.0f,
                    uint64_t max_words = 1ull << 27) {
        return svo_world_expand(w_, octree.raw(), max_depth, cam, cam ? lod_c : 0.0f, max_words);
    }
    svo_world *raw() const { return w_; }

  private:
    explicit World(svo_world *w) : w_(w) {}
    void check(int rc) const {
        if (rc < 0) throw Error(SVO_ERR_STATE, svo_world_last_error(w_));
    }
    svo_world *w_;
};

class Render {  // render.rs:3-285
  public:
    static constexpr size_t kDefaultCapacity = 10000000;  // render.rs:53
    svo_uniforms uniforms{};                              // render.rs:287-322 (bools as flag bits)
    uint32_t width, height;

    // Render::new: node buffer = octree.expanded(capacity), uniform defaults (render.rs:42-61, 306-321)
    Render(const Gpu &gpu, uint32_t w, uint32_t h, const uint32_t *words, size_t n_words, size_t capacity = kDefaultCapacity)
        : width(w), height(h), gpu_(gpu) {
        gpu_.check(svo_nodes_alloc(gpu_.ctx(), capacity > n_words ? capacity : n_words));
        write_nodes(words, n_words);
        const float sun[4] = {-1.7f, -1.0f, 0.8f, 0.0f};
        for (int i = 0; i < 4; i++) uniforms.sun_dir[i] = sun[i];
        uniforms.flags = SVO_F_SHADOWS;
    }
    Render(const Gpu &gpu, uint32_t w, uint32_t h, const Octree &octree, size_t capacity = kDefaultCapacity)
        : Render(gpu, w, h, octree.raw_data(), octree.len(), capacity) {}
    // queue.write_buffer(&node_buffer, 0, nodes) (app.rs:113-118)
    void write_nodes(const uint32_t *words, size_t n) { gpu_.check(svo_nodes_write(gpu_.ctx(), 0, words, n)); }
    // incremental form of the same upload: only the words that changed (svo_nodes_scatter; pair it with
    // gpu.set_option(SVO_OPT_SCAN_CLEARS_COUNTERS, 1), which takes over the counter reset of the full upload)
    void scatter_nodes(const std::vector<uint32_t> &indices, const std::vector<uint32_t> &words) {
        gpu_.check(svo_nodes_scatter(gpu_.ctx(), indices.data(), words.data(), indices.size()));
        gpu_.poll_wait();  // the vectors may go away
    }
    // Render::resize ignores zero sizes (render.rs:182-189)
    void resize(uint32_t w, uint32_t h) {
        if (w > 0 && h > 0) { width = w; height = h; }
    }
    // Render::update: camera = proj * look_at_rh, inverse, dimensions; upload (render.rs:191-212)
    void update(const Settings &settings, const Character &character) {
        svo_camera_matrices(character.pos, character.look, settings.fov, float(width), float(height), uniforms.camera,
                            uniforms.camera_inverse);
        uniforms.dimensions[0] = float(width);
        uniforms.dimensions[1] = float(height);
        uniforms.dimensions[2] = uniforms.dimensions[3] = 0.0f;
        gpu_.check(svo_set_uniforms(gpu_.ctx(), &uniforms));
    }
    // Render::render: one pass over every pixel (render.rs:217-284).  Device pointers; asynchronous.
    void render(svo_hit *hits_dev, uint32_t *rgba_dev = nullptr) {
        gpu_.check(svo_render(gpu_.ctx(), width, height, 0, 0, width, height, hits_dev, rgba_dev));
    }
    // primary rays + n_secondary rays per hit pixel (ray 0: the shadow ray of shader.wgsl:275-280); device pointers
    void render_secondary(uint32_t n_secondary, svo_hit *primary_dev, svo_hit *secondary_dev) {
        gpu_.check(svo_render_secondary(gpu_.ctx(), width, height, 0, 0, width, height, n_secondary, primary_dev, secondary_dev));
    }
    // same, results copied to host memory (blocking)
    void render_host(svo_hit *hits, uint32_t *rgba = nullptr) {
        gpu_.check(svo_render_host(gpu_.ctx(), width, height, 0, 0, width, height, hits, rgba));
    }
    // A second Render on another context of the same device over the SAME node buffer (svo_nodes_share): frames in
    // flight on several streams.  Writes through either one are seen by both (shared generation counter).
    struct Shared {};
    Render(const Gpu &gpu, const Render &owner, Shared) : uniforms(owner.uniforms), width(owner.width), height(owner.height), gpu_(gpu) {
        gpu_.check(svo_nodes_share(gpu_.ctx(), owner.gpu_.ctx()));
        gpu_.check(svo_set_uniforms(gpu_.ctx(), &uniforms));
    }
    // this rank's tiles of a frame sharded over `world` GPUs (tile t belongs to rank t % world), contiguous in hits_dev
    void render_tiles(uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world, svo_hit *hits_dev, uint32_t *rgba_dev = nullptr) {
        gpu_.check(svo_render_tiles(gpu_.ctx(), width, height, tile_w, tile_h, rank, world, hits_dev, rgba_dev));
    }
    const Gpu &gpu() const { return gpu_; }

  private:
    const Gpu &gpu_;
};

// One process, one thread, N GPUs -- the reference's threading model (main.rs:40-88) extended to a tile-sharded frame
// (SURVEY.md 8e): every device holds a replica of the node array and traces the tiles t = rank, rank + N, ...; the frame
// ends with ONE gather of 12-byte wire records to device 0 over RCCL (svo_gather_frame_all) and the un-permute there.
// Buffers live as long as the object; frame() returns a device pointer on device 0 that stays valid until the next frame().
class MultiGpuFrame {
  public:
    MultiGpuFrame(const std::vector<int> &devices, uint32_t w, uint32_t h, const uint32_t *words, size_t n_words, size_t capacity,
                  uint32_t tile_w = 64, uint32_t tile_h = 8)
        : w_(w), h_(h), tw_(tile_w), th_(tile_h), world_(uint32_t(devices.size())) {
        if (devices.empty() || w % tile_w || h % tile_h) throw Error(SVO_ERR_ARG, "frame must be a whole number of tiles");
        const uint32_t tiles = (w / tile_w) * (h / tile_h);
        n_pad_ = (tiles + world_ - 1) / world_;
        for (int d : devices) {
            gpus_.emplace_back(new Gpu(d));
            renders_.emplace_back(new Render(*gpus_.back(), w, h, words, n_words, capacity));
            ctxs_.push_back(gpus_.back()->ctx());
        }
        check(svo_comm_init_all(int(world_), ctxs_.data()));
        const size_t rec = size_t(n_pad_) * tw_ * th_;
        for (uint32_t r = 0; r < world_; r++) {
            hits_.push_back(static_cast<svo_hit *>(dev_alloc(r, rec * sizeof(svo_hit))));
            wire_.push_back(static_cast<uint32_t *>(dev_alloc(r, rec * 12)));
        }
        gathered_ = static_cast<uint32_t *>(dev_alloc(0, rec * 12 * world_));
        frame_ = static_cast<svo_hit *>(dev_alloc(0, size_t(w) * h * sizeof(svo_hit)));
    }
    ~MultiGpuFrame() {
        for (uint32_t r = 0; r < world_ && r < hits_.size(); r++) {
            (void)svo_buffer_free(ctxs_[r], hits_[r]);
            if (r < wire_.size()) (void)svo_buffer_free(ctxs_[r], wire_[r]);
        }
        if (!ctxs_.empty()) {
            (void)svo_buffer_free(ctxs_[0], gathered_);
            (void)svo_buffer_free(ctxs_[0], frame_);
        }
    }
    MultiGpuFrame(const MultiGpuFrame &) = delete;
    MultiGpuFrame &operator=(const MultiGpuFrame &) = delete;
    // the assembled frame of the last frame() call, copied to host memory (blocking)
    void read_frame(svo_hit *host) { check(svo_buffer_read(ctxs_[0], frame_, host, size_t(w_) * h_ * sizeof(svo_hit))); }
    Render &render(uint32_t rank) { return *renders_[rank]; }
    // trace every rank's tiles, gather, assemble: all asynchronous; sync() (or the next blocking call) completes it
    svo_hit *frame() {
        const size_t rec = size_t(n_pad_) * tw_ * th_;
        std::vector<const void *> send(world_);
        for (uint32_t r = 0; r < world_; r++) {
            // the previous frame's gather may still be reading wire_[r] on this rank's communication stream: this frame's
            // trace and pack, which overwrite hits_[r] / wire_[r], are ordered behind it (svo_hip.h: svo_gather_wait before
            // `send` is overwritten)
            check(svo_gather_wait(ctxs_[r]));
            renders_[r]->render_tiles(tw_, th_, r, world_, hits_[r]);
            check(svo_pack_records(ctxs_[r], hits_[r], rec, wire_[r]));
            send[r] = wire_[r];
        }
        check(svo_gather_frame_all(int(world_), ctxs_.data(), send.data(), rec * 12, gathered_, 0));
        check(svo_gather_wait(ctxs_[0]));
        check(svo_assemble_tiles_packed(ctxs_[0], gathered_, world_, n_pad_, w_, h_, tw_, th_, frame_));
        return frame_;
    }
    void sync() {
        for (auto &g : gpus_) g->poll_wait();
    }

  private:
    void check(int rc) const {
        if (rc != SVO_OK) throw Error(rc, svo_last_error(ctxs_.empty() ? nullptr : ctxs_[0]));
    }
    void *dev_alloc(uint32_t rank, size_t bytes) {
        void *p = nullptr;
        check(svo_buffer_alloc(ctxs_[rank], bytes, &p));
        return p;
    }
    uint32_t w_, h_, tw_, th_, world_, n_pad_ = 0;
    std::vector<std::unique_ptr<Gpu>> gpus_;
    std::vector<std::unique_ptr<Render>> renders_;
    std::vector<svo_ctx *> ctxs_;
    std::vector<svo_hit *> hits_;
    std::vector<uint32_t *> wire_;
    uint32_t *gathered_ = nullptr;
    svo_hit *frame_ = nullptr;
};

class Compute {  // compute.rs:6-127 + the read-back of adaptive.rs:12-23, 76-87
  public:
    static constexpr size_t kMaxPerFrame = 1024000;  // adaptive.rs:3-4
    Compute(const Gpu &gpu, const Render &) : gpu_(gpu) {}
    void update(size_t node_length) { gpu_.check(svo_scan_dispatch(gpu_.ctx(), uint32_t(node_length))); }
    void update(const Octree &octree) { update(octree.len()); }
    // (subdivide list, unsubdivide list); counts clamped and the device counters reset like adaptive.rs:22-23
    std::pair<std::vector<uint32_t>, std::vector<uint32_t>> read_lists() {
        std::vector<uint32_t> sub(kMaxPerFrame), unsub(kMaxPerFrame);
        uint32_t ns = 0, nu = 0;
        gpu_.check(svo_scan_read(gpu_.ctx(), sub.data(), &ns, unsub.data(), &nu, kMaxPerFrame));
        return {std::vector<uint32_t>(sub.begin() + 1, sub.begin() + 1 + ns),
                std::vector<uint32_t>(unsub.begin() + 1, unsub.begin() + 1 + nu)};
    }

  private:
    const Gpu &gpu_;
};

}  // namespace svo
