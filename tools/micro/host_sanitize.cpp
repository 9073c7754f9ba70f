// The host side of the library (both OBJ readers, the camera, the image writers, the layout arithmetic and the
// argument checks of the C ABI) driven with well-formed, truncated and mutated input under AddressSanitizer and
// UndefinedBehaviorSanitizer.  No GPU is needed: entry points that need one must fail with a status code.
//   make -C tools/micro sanitize      (host code only: the sanitizer flags are given to the host compilation)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "invert.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd() {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 17, rng_state ^= rng_state << 5;
    return rng_state;
}

static int checks = 0;
#define EXPECT(cond)                                                                 \
    do {                                                                             \
        checks++;                                                                    \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond, inv_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    // a small well-formed mesh in the reference's fixed format
    std::string good;
    char line[128];
    for (int i = 0; i < 40; i++) {
        std::snprintf(line, sizeof line, "v %d.%07de-%03d %d.%07de-%03d -%d.%07de-%03d\n", (int)(rnd() % 10), (int)(rnd() % 10000000),
                      (int)(rnd() % 6), (int)(rnd() % 10), (int)(rnd() % 10000000), (int)(rnd() % 6), (int)(rnd() % 10),
                      (int)(rnd() % 10000000), (int)(rnd() % 6));
        good += line;
    }
    for (int i = 0; i < 60; i++) {
        std::snprintf(line, sizeof line, "f %d %d %d\n", 1 + (int)(rnd() % 40), 1 + (int)(rnd() % 40), 1 + (int)(rnd() % 40));
        good += line;
    }
    inv_mesh* mesh = nullptr;
    EXPECT(inv_mesh_parse_obj(good.data(), good.size(), &mesh) == INV_OK);
    uint32_t nv = 0, ni = 0;
    EXPECT(inv_mesh_counts(mesh, &nv, &ni) == INV_OK && nv == 40 && ni == 180);
    EXPECT(inv_mesh_pin(mesh) != INV_OK || inv_mesh_vertices(mesh) != nullptr); // needs a device; either way no fault
    EXPECT(inv_mesh_destroy(mesh) == INV_OK);

    // every prefix of the text, then thousands of single-byte and splice mutations: a status code, never a fault
    int ok = 0, refused = 0;
    for (size_t n = 0; n <= good.size(); n += 7) {
        mesh = nullptr;
        const int rc = inv_mesh_parse_obj(good.data(), n, &mesh);
        EXPECT(rc == INV_OK || rc == INV_ERR_FORMAT);
        (rc == INV_OK ? ok : refused)++;
        inv_mesh_destroy(mesh);
    }
    const char alphabet[] = "vf 0123456789.-+e\n\r/#x\0\xff";
    for (int round = 0; round < 6000; round++) {
        std::string t = good;
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits; e++) {
            const size_t at = rnd() % t.size();
            switch (rnd() % 3) {
                case 0: t[at] = alphabet[rnd() % (sizeof alphabet - 1)]; break;
                case 1: t.erase(at, 1 + rnd() % 5); break;
                default: t.insert(at, 1 + rnd() % 3, alphabet[rnd() % (sizeof alphabet - 1)]); break;
            }
            if (t.empty()) t = "v";
        }
        for (int general = 0; general < 2; general++) {
            mesh = nullptr;
            int rc;
            if (general) { // the tolerant reader reads a file
                const char* path = "/tmp/host_sanitize.obj";
                FILE* fp = std::fopen(path, "wb");
                EXPECT(fp != nullptr);
                std::fwrite(t.data(), 1, t.size(), fp);
                std::fclose(fp);
                rc = inv_mesh_load_obj_general(path, &mesh);
            } else {
                rc = inv_mesh_parse_obj(t.data(), t.size(), &mesh);
            }
            EXPECT(rc == INV_OK || rc == INV_ERR_FORMAT || rc == INV_ERR_ARG);
            if (rc == INV_OK) { // whatever was accepted is a consistent triangle list
                EXPECT(inv_mesh_counts(mesh, &nv, &ni) == INV_OK && ni % 3 == 0);
                const uint32_t* idx = inv_mesh_indices(mesh);
                for (uint32_t i = 0; i < ni; i++) EXPECT(idx[i] < nv);
                ok++;
            } else {
                EXPECT(mesh == nullptr);
                refused++;
            }
            inv_mesh_destroy(mesh);
        }
    }

    // cameras, layouts, writers, argument checks
    inv_camera cam;
    EXPECT(inv_camera_reference(1920, 1080, &cam) == INV_OK && inv_camera_reference(0, 4, &cam) != INV_OK);
    EXPECT(inv_camera_turntable(640, 480, 1.25f, &cam) == INV_OK);
    for (int i = 0; i < 2000; i++) {
        inv_params p = {};
        p.width = 1 + rnd() % 5000, p.height = 1 + rnd() % 3000, p.tile = 32u * (1 + rnd() % 8), p.nranks = 1 + rnd() % 9;
        p.rank = rnd() % p.nranks, p.flags = INV_SPARSE;
        uint64_t npix = 0;
        uint32_t ids[64], nt = 0;
        EXPECT(inv_tile_layout(&p, &npix, ids, 64, &nt) == INV_OK);
        uint32_t x0 = rnd() % p.width, y0 = rnd() % p.height;
        uint32_t rect[4] = {x0, y0, x0 + rnd() % (p.width - x0 + 1), y0 + rnd() % (p.height - y0 + 1)}, grid[4];
        const int rc = inv_sparse_layout(&p, rect, grid, &npix, ids, 64, &nt);
        EXPECT(p.nranks > 1 ? rc == INV_OK : rc == INV_ERR_ARG);
        // rank 0 without tiles: three ranks or more, the tiles dealt to the others
        p.flags |= INV_ROOT_IDLE;
        const int ri = inv_sparse_layout(&p, rect, grid, &npix, ids, 64, &nt);
        EXPECT(p.nranks >= 3 ? ri == INV_OK && (p.rank != 0 || nt == 0) : ri == INV_ERR_ARG);
        const int rd = inv_tile_layout(&p, &npix, ids, 64, &nt);
        EXPECT(p.nranks >= 3 ? rd == INV_OK && (p.rank != 0 || (nt == 0 && npix == 0)) : rd == INV_ERR_ARG);
    }
    {
        // argument caps: extents, samples, ranks (the layout arithmetic is 64-bit up to them and refused beyond)
        const uint32_t huge[] = {65537u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu};
        for (uint32_t v : huge) {
            inv_params p = {};
            uint64_t npix = 0;
            uint32_t nt = 0;
            p.width = v, p.height = 1;
            EXPECT(inv_tile_layout(&p, &npix, nullptr, 0, &nt) == INV_ERR_ARG);
            p.width = 1, p.height = v;
            EXPECT(inv_tile_layout(&p, &npix, nullptr, 0, &nt) == INV_ERR_ARG);
            p.width = 64, p.height = 64, p.spp = v;
            EXPECT(inv_tile_layout(&p, &npix, nullptr, 0, &nt) == INV_ERR_ARG);
            p.spp = 1, p.nranks = v, p.rank = 1;
            EXPECT(inv_tile_layout(&p, &npix, nullptr, 0, &nt) == INV_ERR_ARG);
        }
        inv_params p = {};
        uint64_t npix = 0;
        uint32_t nt = 0;
        p.width = 65536, p.height = 32767, p.spp = 4096, p.nranks = 65536, p.rank = 65535, p.tile = 1024;
        EXPECT(inv_tile_layout(&p, &npix, nullptr, 0, &nt) == INV_OK && npix == 1024ull * 1024ull && nt <= 1);
        p.width = 65536, p.height = 32768; // 2^31 pixels
        EXPECT(inv_tile_layout(&p, &npix, nullptr, 0, &nt) == INV_ERR_ARG);
    }
    std::vector<uint8_t> color(4 * 33 * 17, 128);
    std::vector<float> depth(33 * 17, 0.5f);
    std::vector<uint32_t> prim(33 * 17, 7);
    EXPECT(inv_write_ppm("/tmp/host_sanitize.ppm", color.data(), 33, 17, 0) == INV_OK);
    EXPECT(inv_write_png("/tmp/host_sanitize.png", color.data(), 33, 17, 0) == INV_OK);
    EXPECT(inv_write_png("/tmp/host_sanitize.png", color.data(), 33, 17, INV_RGBA) == INV_OK);
    EXPECT(inv_write_pfm("/tmp/host_sanitize.pfm", depth.data(), 33, 17) == INV_OK);
    EXPECT(inv_write_primid("/tmp/host_sanitize.pid", prim.data(), 33, 17) == INV_OK);
    EXPECT(inv_write_ppm("/nonexistent-dir/x.ppm", color.data(), 33, 17, 0) == INV_ERR_IO);
    EXPECT(inv_mesh_load_obj("/nonexistent-dir/x.obj", &mesh) == INV_ERR_IO);
    EXPECT(inv_render(nullptr, nullptr, nullptr, nullptr) == INV_ERR_ARG);
    EXPECT(inv_render_batch(nullptr, 0, 0, nullptr, nullptr, nullptr) == INV_ERR_ARG);
    EXPECT(inv_frame_wait(nullptr, 0) == INV_ERR_ARG && inv_frames_join(nullptr) == INV_ERR_ARG);
    EXPECT(inv_ctx_trim(nullptr) == INV_ERR_ARG);
    inv_ctx* ctx = nullptr;
    const int rc = inv_ctx_create(0, nullptr, &ctx); // INV_ERR_NO_DEVICE here, INV_OK on a GPU box
    EXPECT(rc == INV_OK || rc == INV_ERR_NO_DEVICE);
    if (rc == INV_OK) {
        EXPECT(inv_frame_wait(ctx, INV_MAX_FRAMES_IN_FLIGHT) == INV_ERR_ARG && inv_frame_wait(ctx, 1) == INV_OK);
        EXPECT(inv_ctx_destroy(ctx) == INV_OK);
    }
    std::printf("host_sanitize: %d checks, %d inputs accepted, %d refused, no sanitizer report\n", checks, ok, refused);
    (void)argc, (void)argv;
    return 0;
}
