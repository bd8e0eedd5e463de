// C++ host using the drop-in classes of include/svo_render.hpp exactly the way the reference's App drives
// Gpu / Render / Compute (app.rs:19-118): build a tree, upload, update uniforms, render, scan.
// build: hipcc -std=c++17 -Iinclude examples/render_frame.cpp -Loctree-tracer_amd -lsvo_hip -Wl,-rpath,$PWD/octree-tracer_amd -o /tmp/render_frame
#include <cstdio>
#include <cstring>

#include "svo_render.hpp"

int main(int argc, char **argv) {
    // an 8^3 checker scene through CpuOctree::put_in_voxel (cpu_octree.rs:100-111)
    svo::CpuOctree cpu;
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++)
            for (int z = 0; z < 8; z++)
                if (((x ^ y ^ z) & 1) && y < 5) {
                    const float p[3] = {(x + 0.5f) / 4 - 1, (y + 0.5f) / 4 - 1, (z + 0.5f) / 4 - 1};
                    cpu.put_in_voxel(p, svo::Voxel{uint8_t(40 + 25 * x), uint8_t(200 - 20 * y), uint8_t(30 + 25 * z)}, 3);
                }
    const std::vector<uint32_t> words = cpu.to_octree_words();
    std::printf("tree: %zu words\n", words.size());
    // a world whose root chunk instances that tree as block 1 in two octants (world.rs:5-9, cpu_octree.rs:78-90)
    svo::World world;
    world.insert(1, std::move(cpu));
    world.generate_mip_tree(1);
    {
        svo::CpuOctree root;
        const float a[3] = {-0.5f, -0.5f, -0.5f}, b[3] = {0.5f, 0.5f, 0.5f};
        root.put_in_block(a, 1, 1);
        root.put_in_block(b, 1, 1);
        world.insert(0, std::move(root));
    }
    const svo::Voxel top = world.generate_mip_tree(0);
    {
        svo::Octree full = world.root_octree();
        world.expand(full, 4);
        std::printf("world: %zu words fully expanded, top mip (%u, %u, %u)\n", full.len(), top.r, top.g, top.b);
    }
    if (argc > 1 && !std::strcmp(argv[1], "--host-only")) return 0;
    try {
        svo::Gpu gpu(0);
        svo::Octree octree(words);
        svo::Render render(gpu, 320, 200, octree, 4096);
        render.uniforms.flags = SVO_F_PAUSE_ADAPTIVE | SVO_F_SHADOWS;
        render.update(svo::Settings{}, svo::Character{});
        std::vector<svo_hit> hits(320 * 200);
        std::vector<uint32_t> rgba(320 * 200);
        render.render_host(hits.data(), rgba.data());
        size_t n_hit = 0, steps = 0;
        for (const svo_hit &h : hits) { n_hit += (h.steps_depth_hit >> 16) & 1; steps += h.steps_depth_hit & 0xFF; }
        std::printf("rays %zu hits %zu mean steps %.2f centre pixel rgba %08x\n", hits.size(), n_hit, double(steps) / hits.size(),
                    rgba[100 * 320 + 160]);
        render.uniforms.flags = 0;  // adaptive: counters live
        render.update(svo::Settings{}, svo::Character{});
        render.render_host(hits.data());
        svo::Compute compute(gpu, render);
        compute.update(octree);
        auto lists = compute.read_lists();
        std::printf("scan: %zu to subdivide, %zu to unsubdivide\n", lists.first.size(), lists.second.size());
        // the streaming loop of App::update (app.rs:94-118) over the world: the device tree grows from the root group
        svo::Octree streamed = world.root_octree();
        size_t subdivided = 0;
        render.write_nodes(streamed.raw_data(), streamed.len());
        streamed.take_dirty();
        gpu.set_option(SVO_OPT_SCAN_CLEARS_COUNTERS, 1);  // the scan resets the counters, so only changed words are re-sent
        for (int frame = 0; frame < 8; frame++) {
            render.render_host(hits.data());
            compute.update(streamed);
            auto l = compute.read_lists();
            subdivided += world.process_subdivision(l.first, streamed);
            world.process_unsubdivision(l.second, streamed);
            auto dirty = streamed.take_dirty();
            render.scatter_nodes(dirty.first, dirty.second);
        }
        std::printf("streaming: %zu subdivisions, device tree %zu words\n", subdivided, streamed.len());
        return n_hit > 0 && subdivided > 0 ? 0 : 2;
    } catch (const svo::Error &e) {
        std::fprintf(stderr, "svo error %d: %s\n", e.status, e.what());
        return 1;
    }
}
