// Host cost of a draw through the C ABI without Python in the way: K frames of the 1080p bunny dealt to S frame
// slots (inv_render_in_flight), time per frame for the enqueue loop alone and until the GPU is done.
//   hipcc -O2 -o host_loop host_loop.cpp -I../../include ../../renderer_amd/libinvert_hip.so && ./host_loop bunny.obj
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "invert.h"

#define CHECK(x)                                                         \
    do {                                                                 \
        if ((x) != 0) {                                                  \
            std::fprintf(stderr, "%s failed: %s\n", #x, inv_last_error()); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return std::fprintf(stderr, "usage: host_loop bunny.obj\n"), 2;
    const uint32_t w = 1920, h = 1080;
    const int K = 20000;
    inv_ctx* ctx = nullptr;
    inv_mesh* mesh = nullptr;
    inv_scene* scene = nullptr;
    CHECK(inv_ctx_create(0, nullptr, &ctx));
    CHECK(inv_mesh_load_obj(argv[1], &mesh));
    CHECK(inv_scene_build(ctx, mesh, &scene));
    inv_camera cam;
    CHECK(inv_camera_reference(w, h, &cam));
    inv_params prm = {};
    prm.width = w, prm.height = h, prm.flags = INV_LAMBERT | INV_INVERT;
    inv_outputs out[INV_MAX_FRAMES_IN_FLIGHT];
    for (auto& o : out) {
        if (hipMalloc(&o.color, (size_t)w * h * 4) != hipSuccess || hipMalloc(&o.depth, (size_t)w * h * 4) != hipSuccess ||
            hipMalloc(&o.primid, (size_t)w * h * 4) != hipSuccess)
            return std::fprintf(stderr, "hipMalloc failed\n"), 1;
        o.on_device = 1;
    }
    for (int S = 1; S <= INV_MAX_FRAMES_IN_FLIGHT; S++) {
        for (int rep = 0; rep < 2; rep++) {
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < K; i++) CHECK(inv_render_in_flight(scene, (uint32_t)(i % S), &cam, &prm, &out[i % S]));
            const auto t1 = std::chrono::steady_clock::now();
            (void)hipDeviceSynchronize();
            const auto t2 = std::chrono::steady_clock::now();
            if (rep)
                std::printf("%d in flight: enqueue %.2f us per frame, done %.2f us per frame\n", S,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / K,
                            std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
        }
    }
    inv_scene_destroy(scene);
    inv_mesh_destroy(mesh);
    inv_ctx_destroy(ctx);
    return 0;
}
