// tsan_driver.cpp -- drives the C ABI of a ThreadSanitizer build of the host side (linked against hip_stub.cpp) through the concurrent situations the
// managers create on the GPU box: several objects trained from their own threads through the device's training lanes while the lane count flips, a viewer
// rendering from published snapshots, objects created and destroyed meanwhile, and the online manager's whole protocol (frames, boxes, empty updates, renders,
// finish).  Exit status 0 and no TSAN report = pass.  TEST INFRASTRUCTURE.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/mon_core.h"

#define OK(expr) do { const int rc_ = (expr); \
        if (rc_ != MON_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, mon_last_error()); std::exit(3); } } while (0)

int main(int argc, char** argv) {
    const char* cfg_json = argc > 1 ? argv[1] : "ro-map_amd/configs/base.json";
    const int H = 48, W = 64, n_frames = 24;
    int n_dev = 0; OK(mon_device_count(&n_dev));
    // every render / density query through the per-device tile workspace (its mutexes, the image key, the refcount at object destruction)
    OK(mon_set_option("tile_render", 2));
    // the level-tile chain's host side (second candidate set, position mode of the optimizer launch) also for the small objects below
    OK(mon_set_option("lds_encode", 2));
    mon_config cfg; OK(mon_config_default(&cfg)); cfg.rays_per_batch = 256;
    std::vector<unsigned char> rgb((size_t)H * W * 3, 128), inst((size_t)H * W, 7);
    float pose[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, -2, 1 }, Tow[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    const float amin[3] = { -0.3f, -0.3f, -0.3f }, amax[3] = { 0.3f, 0.3f, 0.3f };

    {   // ---- 1. objects on the C ABI directly: lanes, snapshots, churn
        mon_dataset* ds = nullptr; OK(mon_dataset_create(0, H, W, 60.f, 60.f, 32.f, 24.f, n_frames, 0, &ds));
        for (int v = 0; v < n_frames; ++v) OK(mon_dataset_add_frame(ds, v, rgb.data(), 3, 0, inst.data(), nullptr, pose));
        const int K = 5; std::vector<mon_object*> objs(K, nullptr);
        std::vector<mon_frame_bbox> boxes; for (int v = 0; v < n_frames; ++v) boxes.push_back(mon_frame_bbox{ (uint32_t)v, 8, 8, 24, 32 });
        for (auto& o : objs) { OK(mon_object_create(ds, &cfg, 7, Tow, amin, amax, &o)); OK(mon_object_add_boxes(o, boxes.data(), boxes.size())); }
        std::atomic<bool> stop{ false };
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) th.emplace_back([&,
                k] { float loss = 0.f; for (int r = 0; r < 40; ++r) OK(mon_object_train(objs[k], 3 + (r + k) % 19, &loss)); });
        th.emplace_back([&] { for (int i = 0; !stop.load(); ++i) { mon_set_option("train_lanes", (i & 1) ? 2
                : 0); std::this_thread::sleep_for(std::chrono::microseconds(300)); } mon_set_option("train_lanes", 2); });
        th.emplace_back([&] {          // a viewer
            std::vector<float> c(3 * 16 * 16), d(16 * 16), m(16 * 16); uint32_t step = 0;
            while (!stop.load()) { for (int k = 0; k < K; ++k) {
                    const int rc = mon_object_render_snapshot(objs[k], mon_frame_bbox{ 0, 8, 8, 16, 16 }, pose, 0, c.data(), d.data(), m.data(), &step);
                    if (rc != MON_OK && rc != MON_ERR_STATE) std::exit(4); } }
        });
        th.emplace_back([&] {          // objects come and go
            for (int i = 0; i < 12; ++i) { mon_object* o = nullptr; float loss; OK(mon_object_create(ds, &cfg, 7, Tow, amin, amax, &o));
                OK(mon_object_add_boxes(o, boxes.data(), 12)); OK(mon_object_train(o, 5, &loss)); OK(mon_object_destroy(o)); }
        });
        for (int k = 0; k < K; ++k) th[k].join();
        th.back().join(); th.pop_back();
        stop.store(true); for (size_t k = K; k < th.size(); ++k) th[k].join();
        for (auto& o : objs) OK(mon_object_destroy(o));
        OK(mon_dataset_destroy(ds));
    }
    {   // ---- 2. the online manager's protocol (NerfManagerOnline + NeRF::TrainOnline)
        mon_online* om = nullptr; OK(mon_online_create(cfg_json, 0, 20, &om)); OK(mon_online_init(om));
        OK(mon_online_dataset_init(om, 60.f, 60.f, 32.f, 24.f, H, W, n_frames));
        std::vector<size_t> ids; std::atomic<bool> stop{ false };
        std::thread viewer([&] {
            std::vector<float> c(3 * 16 * 16), d(16 * 16), m(16 * 16);
            while (!stop.load()) {
                for (size_t k = 0; k < 3; ++k) (void)mon_online_render(om, k, mon_frame_bbox{ 0, 8, 8, 16, 16 }, pose, c.data(), d.data(), m.data());
                std::this_thread::sleep_for(std::chrono::microseconds(200)); }
        });
        for (int v = 0; v < n_frames; ++v) {
            char stamp[32]; std::snprintf(stamp, sizeof stamp, "%.6f", v * 0.1);
            OK(mon_online_new_frame(om, (uint32_t)v, stamp, rgb.data(), 3, inst.data(), nullptr, pose));
            if (v < 3) { size_t idx = 0; const float bb[6] = { -0.3f, -0.3f, -0.3f, 0.3f, 0.3f, 0.3f };
                OK(mon_online_create_nerf(om, 7, Tow, bb, bb + 3, &idx)); ids.push_back(idx); }
            // (an empty update too)
            for (size_t idx : ids) { mon_frame_bbox b{ (uint32_t)v, 8, 8, 24, 32 }; OK(mon_online_update_nerf_bbox(om, idx, &b, 1, 1));
                OK(mon_online_update_nerf_bbox(om, idx, nullptr, 0, 1)); }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        OK(mon_online_wait_threads_end(om));
        stop.store(true); viewer.join();
        OK(mon_online_destroy(om));
    }
    std::printf("tsan driver finished\n");
    return 0;
}
