// One process, one thread, N GPUs: the tile-sharded frame with its single RCCL gather, through the C ABI only
// (svo::MultiGpuFrame of include/svo_render.hpp: svo_comm_init_all, svo_render_tiles, svo_pack_records,
// svo_gather_frame_all, svo_gather_wait, svo_assemble_tiles_packed).  The assembled frame is compared with the same
// frame traced unsharded on device 0.  usage: multi_gpu_frame [n_gpus]   (default: every visible device)
// build: g++ -std=c++17 -Iinclude examples/multi_gpu_frame.cpp -Loctree-tracer_amd -lsvo_hip -Wl,-rpath,$PWD/octree-tracer_amd -o /tmp/multi_gpu_frame
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "svo_render.hpp"

int main(int argc, char **argv) {
    int n = svo_device_count();
    if (argc > 1) n = std::atoi(argv[1]) < n ? std::atoi(argv[1]) : n;
    if (n < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 1;
    }
    svo::CpuOctree cpu;
    for (int x = 0; x < 16; x++)
        for (int y = 0; y < 16; y++)
            for (int z = 0; z < 16; z++)
                if (((x * 7 + y * 3 + z * 5) % 11) < 3 && y < 12) {
                    const float p[3] = {(x + 0.5f) / 8 - 1, (y + 0.5f) / 8 - 1, (z + 0.5f) / 8 - 1};
                    cpu.put_in_voxel(p, svo::Voxel{uint8_t(20 + 14 * x), uint8_t(230 - 12 * y), uint8_t(15 + 13 * z)}, 4);
                }
    const std::vector<uint32_t> words = cpu.to_octree_words();
    const uint32_t W = 640, H = 384;
    try {
        std::vector<int> devices;
        for (int d = 0; d < n; d++) devices.push_back(d);
        svo::MultiGpuFrame mg(devices, W, H, words.data(), words.size(), words.size() + 64);
        for (int r = 0; r < n; r++) {
            mg.render(r).uniforms.flags = SVO_F_PAUSE_ADAPTIVE;
            mg.render(r).update(svo::Settings{}, svo::Character{});
        }
        std::vector<svo_hit> sharded(size_t(W) * H), whole(size_t(W) * H);
        // three frames queue up behind each other on the streams, each from a different camera: a frame whose records were
        // overwritten while the previous frame's gather still read them, or that kept stale ones, differs from the check below
        svo::Character cam;
        for (int frame = 0; frame < 3; frame++) {
            cam.pos[0] = 0.1f + 0.35f * float(frame);
            cam.look[0] = -0.2f * float(frame);
            for (int r = 0; r < n; r++) mg.render(r).update(svo::Settings{}, cam);
            mg.frame();
        }
        mg.read_frame(sharded.data());
        mg.sync();
        mg.render(0).render_host(whole.data());  // the last camera, unsharded
        const bool same = !std::memcmp(sharded.data(), whole.data(), whole.size() * sizeof(svo_hit));
        size_t hits = 0;
        for (const svo_hit &h : whole) hits += (h.steps_depth_hit >> 16) & 1;
        std::printf("%d GPU(s), %ux%u frame, %zu hits: sharded frame %s the unsharded one\n", n, W, H, hits, same ? "equals" : "DIFFERS FROM");
        return same && hits > 0 ? 0 : 2;
    } catch (const svo::Error &e) {
        std::fprintf(stderr, "svo error %d: %s\n", e.status, e.what());
        return 1;
    }
}
