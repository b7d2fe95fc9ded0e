// offline_nerf.cpp -- headless equivalent of the reference's OfflineNeRF executable (MON/main.cpp:287-343) without the
// Pangolin viewer: OfflineNeRF <config.json> <dataset_path> <UseGTdepth 0|1> [n_objects=4] [out_dir=./output] [gather]
// Links only against the C ABI (include/mon_core.h).  With "gather" the test images of all objects go through the in-process RCCL gather-to-root of
// libmon_core_rccl.so (include/mon_core_rccl.h; loaded on demand, so the plain job needs no RCCL): every object renders on its own device (k mod nGPU,
// nerf_manager.cu:96), the crops travel to device 0 over xGMI and one writer stores the same PNG bytes mon_offline_render_test writes.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dlfcn.h>
#include "../include/mon_core.h"
#include "../include/mon_core_rccl.h"

static int fail(const char* what) { std::fprintf(stderr, "%s: %s\n", what, mon_last_error()); return 1; }

int main(int argc, char** argv) {
    std::puts("......Multi-Object NeRF Offline (MI355X core)......");
    if (argc < 4) { std::fprintf(stderr, "param error...\n./offline_nerf ./configs/base.json dataset_path UseGTdepth [n_objects] [out_dir] [gather]\n"); return 0; }
    const std::string cfg = argv[1], dataset = argv[2]; const int use_depth = std::atoi(argv[3]);
    const int n_objects = argc > 4 ? std::atoi(argv[4]) : 4;                 // main.cpp:315-319 hard-codes 4
    const std::string out = argc > 5 ? argv[5] : "./output";
    const bool gather = argc > 6 && std::strcmp(argv[6], "gather") == 0;
    if (use_depth != 0 && use_depth != 1) { std::fprintf(stderr, "UseGTdepth param error...\n0 or 1\n"); return 0; }
    // harness convenience shared with the Python binding and tests/compat_driver.cpp (the library itself reads no environment variable):
    // MON_OPTIONS="name=value,..."
    if (const char* e = std::getenv("MON_OPTIONS")) {
        std::string kv, all = e; size_t p0 = 0;
        while (p0 <= all.size()) {
            const size_t p1 = all.find(',', p0); kv = all.substr(p0, p1 == std::string::npos ? std::string::npos : p1 - p0);
            p0 = p1 == std::string::npos ? all.size() + 1 : p1 + 1;
            const size_t q = kv.find('=');
            if (q == std::string::npos) continue;
            int orc;
            if (kv.substr(0, q) == "offline_schedule") { int o = 0, i = 0; orc = std::sscanf(kv.c_str() + q + 1, "%dx%d", &o, &i) == 2 ? mon_offline_set_schedule(o, i) : 1; }
            else orc = mon_set_option(kv.substr(0, q).c_str(), std::atol(kv.c_str() + q + 1));
            if (orc) return fail("MON_OPTIONS");
        }
    }
    // wall clock per phase (like the reference's steady_clock around Train_Step, nerf_model.cu:1632,1659)
    auto t_last = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) { const auto now = std::chrono::steady_clock::now();
        std::printf("phase %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count()); t_last = now; };
    mon_offline* mgr = nullptr;
    if (mon_offline_create(dataset.c_str(), cfg.c_str(), use_depth, &mgr)) return fail("create");
    if (mon_offline_init(mgr)) return fail("Init");
    phase("create + Init (device)");
    if (mon_offline_read_dataset(mgr)) return fail("ReadDataset");
    phase("ReadDataset (PNGs -> HBM)");
    mon_offline_set_output_dir(mgr, out.c_str());                          // <out>/<id>.ply, the reference writes ./output/<id>.ply
    for (int i = 0; i < n_objects; ++i) {
        const std::string obj = dataset + "/obj_offline/" + std::to_string(i) + ".txt";
        const auto tc = std::chrono::steady_clock::now();
        if (mon_offline_create_nerf(mgr, obj.c_str())) return fail("CreateNeRF");
        std::printf("CreateNeRF %d: %.1f ms\n", i, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count());
    }
    phase("CreateNeRF (threads started)");
    if (mon_offline_wait_threads_end(mgr)) return fail("WaitThreadsEnd");
    phase("training + meshes (threads)");
    for (int i = 0; i < n_objects; ++i) {
        float loss = 0.f; int dev = 0; mon_offline_object_loss(mgr, i, &loss, &dev);
        std::printf("object %d on device %d: final loss %f\n", i, dev, loss);
    }
    if (!gather) {          // every object's test images + obj.ply from a thread of its own: the renders take turns on the device, PNG / ply encoding runs side by side
        std::vector<std::thread> th; std::vector<int> rcs(n_objects, 0); std::vector<std::string> errs(n_objects);
        for (int i = 0; i < n_objects; ++i) th.emplace_back([&, i] { rcs[i] = mon_offline_render_test(mgr, i, out.c_str(), 4); if (rcs[i]) errs[i] = mon_last_error(); });
        for (auto& t : th) t.join();
        for (int i = 0; i < n_objects; ++i) if (rcs[i]) { std::fprintf(stderr, "OfflineNeRF: render of object %d failed: %s\n", i, errs[i].c_str()); return 1; }
    }
    if (gather) {
        void* lib = dlopen("libmon_core_rccl.so", RTLD_NOW);
        if (!lib) { std::fprintf(stderr, "gather: %s\n", dlerror()); return 1; }
        const auto create = reinterpret_cast<decltype(&mon_gather_create)>(dlsym(lib, "mon_gather_create"));
        const auto destroy = reinterpret_cast<decltype(&mon_gather_destroy)>(dlsym(lib, "mon_gather_destroy"));
        const auto stats = reinterpret_cast<decltype(&mon_gather_stats)>(dlsym(lib, "mon_gather_stats"));
        const auto render = reinterpret_cast<decltype(&mon_offline_render_test_gathered)>(dlsym(lib, "mon_offline_render_test_gathered"));
        if (!create || !destroy || !stats || !render) { std::fprintf(stderr, "gather: libmon_core_rccl.so lacks an entry point\n"); return 1; }
        mon_gather* g = nullptr;
        if (create(0, &g)) return fail("mon_gather_create");
        if (render(g, mgr, out.c_str(), 4)) return fail("mon_offline_render_test_gathered");
        uint64_t over_links = 0, on_root = 0; int senders = 0; double ms = 0.0;
        stats(g, &over_links, &on_root, &senders, &ms);
        std::printf("gathered to device 0: last view %llu B over device links from %d device(s), %llu B already on the root, %.3f ms\n",
                    (unsigned long long)over_links, senders, (unsigned long long)on_root, ms);
        destroy(g);
    }
    phase("test images + obj.ply");
    mon_offline_destroy(mgr);
    phase("destroy");
    std::puts("Training completed");
    return 0;
}
