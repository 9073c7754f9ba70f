// Headless host program on the C ABI: what main(), Window and the swapchain are to the reference
// (src/invert.cpp:1770-1897) when there is no window to present to.
//
//   load the OBJ (Obj stanfordBunny, src/invert.cpp:1763) -> context (initializeVulkan, :1483) -> scene
//   (vertex / index buffers, :1113-1164) -> frame loop (mainLoop / drawFrame, :477-487, :644-699) with frames in
//   flight -> the last frame's attachments written as PPM / PFM / primitive-id files instead of presented
//   (:674-683) -> one JSON line with the timings.
//
//   invert_headless --obj res/bunny.obj [--general] [--pin] [--width 1920 --height 1080] [--spp 1] [--ao 0]
//                   [--lambert] [--invert] [--rgba] [--turntable] [--frames 64] [--in-flight 3] [--per-launch 3]
//                   [--out PREFIX]
//                   [--ranks N --rank R --rendezvous DIR [--transport rccl|host] [--gpu G] [--tile 32]]
//
// With --ranks N (one process per rank, all given the same DIR, e.g. a fresh directory under /dev/shm) the frame is
// sharded: only the 32x32 tiles under the scene's projected bounds are dealt out, rank R draws the tiles with
// tile % N == R, the slot buffers go to rank 0 (inv_comm_gather: RCCL send / receive over xGMI, or staging files in
// DIR with --transport host, which also works with every rank on one GPU), and rank 0 puts them into its frame
// (inv_deinterleave_sparse) and writes the files.  Rank 0 hands the RCCL id to the others through DIR/rccl_id.
//
// The loop waits for a slot's fence before it submits to that slot again, as drawFrame does, so the host's wake-up
// after each wait is part of the frame time (1080p bunny: 4.9 us per frame with three slots of three frames; a loop
// that queues ahead without looking at the frames, like bench.py's, reaches 3.9 us).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "invert.h"

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int die(const char* what) {
    std::fprintf(stderr, "invert_headless: %s: %s\n", what, inv_last_error());
    return 1;
}

struct Attachments {
    void *color = nullptr, *depth = nullptr, *primid = nullptr;
};

// Rank 0 leaves the RCCL id in DIR/rccl_id (written under another name, then renamed: it appears whole); the other
// ranks wait for it.
bool share_rccl_id(const std::string& dir, uint32_t rank, unsigned char id[INV_COMM_ID_BYTES]) {
    const std::string fin = dir + "/rccl_id", tmp = fin + ".tmp";
    if (rank == 0) {
        if (inv_comm_unique_id(id) != INV_OK) return false;
        FILE* fp = std::fopen(tmp.c_str(), "wb");
        if (!fp) return false;
        const bool ok = std::fwrite(id, 1, INV_COMM_ID_BYTES, fp) == INV_COMM_ID_BYTES;
        return std::fclose(fp) == 0 && ok && std::rename(tmp.c_str(), fin.c_str()) == 0;
    }
    struct stat st;
    for (int tries = 0; ::stat(fin.c_str(), &st) != 0; tries++) {
        if (tries > 600000) return false; // two minutes
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    FILE* fp = std::fopen(fin.c_str(), "rb");
    if (!fp) return false;
    const bool ok = std::fread(id, 1, INV_COMM_ID_BYTES, fp) == INV_COMM_ID_BYTES;
    std::fclose(fp);
    return ok;
}

} // namespace

int main(int argc, char** argv) {
    std::string obj, out_prefix, rendezvous, transport = "rccl";
    bool general = false, pin = false, turntable = false;
    uint32_t width = 1920, height = 1080, spp = 1, ao = 0, flags = 0, frames = 64, in_flight = 3, per_launch = 3;
    uint32_t ranks = 1, rank = 0, gpu = 0xFFFFFFFFu, tile = 32;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        const auto value = [&](uint32_t* v) {
            if (i + 1 >= argc) return false;
            *v = (uint32_t)std::strtoul(argv[++i], nullptr, 10);
            return true;
        };
        bool ok = true;
        if (a == "--obj" && i + 1 < argc) obj = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_prefix = argv[++i];
        else if (a == "--general") general = true;
        else if (a == "--pin") pin = true;
        else if (a == "--turntable") turntable = true;
        else if (a == "--lambert") flags |= INV_LAMBERT;
        else if (a == "--invert") flags |= INV_INVERT;
        else if (a == "--rgba") flags |= INV_RGBA;
        else if (a == "--rank0-idle") flags |= INV_ROOT_IDLE; // several ranks: rank 0 draws no tiles
        else if (a == "--width") ok = value(&width);
        else if (a == "--height") ok = value(&height);
        else if (a == "--spp") ok = value(&spp);
        else if (a == "--ao") ok = value(&ao);
        else if (a == "--frames") ok = value(&frames);
        else if (a == "--in-flight") ok = value(&in_flight);
        else if (a == "--per-launch") ok = value(&per_launch);
        else if (a == "--ranks") ok = value(&ranks);
        else if (a == "--rank") ok = value(&rank);
        else if (a == "--gpu") ok = value(&gpu);
        else if (a == "--tile") ok = value(&tile);
        else if (a == "--rendezvous" && i + 1 < argc) rendezvous = argv[++i];
        else if (a == "--transport" && i + 1 < argc) transport = argv[++i];
        else ok = false;
        if (!ok) {
            std::fprintf(stderr, "invert_headless: bad argument %s (see the head of invert_headless.cpp)\n", a.c_str());
            return 2;
        }
    }
    if (obj.empty() || !frames || !in_flight || in_flight > INV_MAX_FRAMES_IN_FLIGHT || !per_launch || per_launch > 8) {
        std::fprintf(stderr, "invert_headless: --obj is required, --frames >= 1, --in-flight 1..%d, --per-launch 1..8\n",
                     INV_MAX_FRAMES_IN_FLIGHT);
        return 2;
    }

    if (!ranks || rank >= ranks || (ranks > 1 && rendezvous.empty()) || (transport != "rccl" && transport != "host")) {
        std::fprintf(stderr, "invert_headless: --rank must be below --ranks, several ranks need --rendezvous DIR, --transport is rccl or host\n");
        return 2;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) ndev = 1;
    const int device = gpu != 0xFFFFFFFFu ? (int)gpu : (int)(rank % (uint32_t)ndev);

    const double t_load = now_ms();
    inv_mesh* mesh = nullptr;
    if ((general ? inv_mesh_load_obj_general(obj.c_str(), &mesh) : inv_mesh_load_obj(obj.c_str(), &mesh)) != INV_OK)
        return die("loading the OBJ");
    uint32_t nv = 0, nidx = 0;
    inv_mesh_counts(mesh, &nv, &nidx);
    inv_ctx* ctx = nullptr;
    if (inv_ctx_create(device, nullptr, &ctx) != INV_OK) return die("inv_ctx_create");
    if (pin && inv_mesh_pin(mesh) != INV_OK) return die("inv_mesh_pin");
    const double t_build = now_ms();
    inv_scene* scene = nullptr;
    if (inv_scene_build(ctx, mesh, &scene) != INV_OK) return die("inv_scene_build");
    const double t_built = now_ms();

    const size_t npix = (size_t)width * height;
    if (ranks > 1) {
        // ---- several ranks: the sparse composite, frame by frame on the context's stream ----
        inv_comm* comm = nullptr;
        if (transport == "host") {
            if (inv_comm_create_host(ctx, rendezvous.c_str(), rank, ranks, &comm) != INV_OK) return die("inv_comm_create_host");
        } else {
            unsigned char id[INV_COMM_ID_BYTES];
            if (!share_rccl_id(rendezvous, rank, id)) return die("sharing the RCCL id");
            if (inv_comm_create_rccl(ctx, id, rank, ranks, &comm) != INV_OK) return die("inv_comm_create_rccl");
        }
        inv_params prm = {};
        prm.width = width, prm.height = height, prm.spp = spp, prm.ao_rays = ao, prm.flags = flags | INV_SPARSE;
        prm.tile = tile, prm.rank = rank, prm.nranks = ranks;
        inv_params prm_other = prm, prm0 = prm;
        prm_other.rank = 1, prm0.rank = 0;
        inv_camera cam;
        if (inv_camera_reference(width, height, &cam) != INV_OK) return die("camera");
        uint32_t rect[4], grid[4], ntiles = 0;
        uint64_t slot_elems = 0; // elements per attachment of a rank > 0 (the same for all of them)
        if (inv_scene_hit_rect(scene, &cam, width, height, rect) != INV_OK) return die("inv_scene_hit_rect");
        if (inv_sparse_layout(&prm_other, rect, grid, &slot_elems, nullptr, 0, &ntiles) != INV_OK) return die("inv_sparse_layout");
        uint32_t *planes = nullptr, *gathered = nullptr, *frame = nullptr;
        const size_t part = (size_t)std::max<uint64_t>(slot_elems, 1) * 3;
        if (rank == 0) {
            if (hipMalloc((void**)&frame, npix * 12) != hipSuccess || hipMalloc((void**)&gathered, part * ranks * 4) != hipSuccess)
                return die("hipMalloc");
        } else if (hipMalloc((void**)&planes, part * 4) != hipSuccess) {
            return die("hipMalloc");
        }
        const double t_loop = now_ms();
        for (uint32_t k = 0; k < frames; k++) {
            if (turntable && inv_camera_turntable(width, height, (float)k / 60.0f, &cam) != INV_OK) return die("camera");
            if (turntable) { // the rectangle moves with the model: every rank works it out the same way
                if (inv_scene_hit_rect(scene, &cam, width, height, rect) != INV_OK) return die("inv_scene_hit_rect");
                uint64_t now_elems = 0;
                if (inv_sparse_layout(&prm_other, rect, grid, &now_elems, nullptr, 0, &ntiles) != INV_OK) return die("inv_sparse_layout");
                if (now_elems > slot_elems) {
                    std::fprintf(stderr, "invert_headless: the turning model needs more tile slots than the first frame\n");
                    return 1;
                }
            }
            uint32_t* mine = rank == 0 ? frame : planes;
            const size_t stride = rank == 0 ? npix : (size_t)slot_elems;
            const inv_outputs out = {mine, mine + stride, mine + 2 * stride, 1};
            if (inv_render(scene, &cam, &prm, &out) != INV_OK) return die("inv_render");
            if (inv_comm_gather(comm, planes, slot_elems * 3, gathered, slot_elems * 3) != INV_OK) return die("inv_comm_gather");
            if (rank == 0 && inv_deinterleave_sparse(ctx, &prm0, rect, gathered, slot_elems * 3, slot_elems, 3, frame, npix) != INV_OK)
                return die("inv_deinterleave_sparse");
        }
        if (inv_ctx_synchronize(ctx) != INV_OK) return die("inv_ctx_synchronize");
        const double loop_ms = now_ms() - t_loop;
        uint64_t hits = 0;
        if (rank == 0) {
            std::vector<uint32_t> host(npix * 3);
            if (hipMemcpy(host.data(), frame, npix * 12, hipMemcpyDeviceToHost) != hipSuccess) return die("hipMemcpy");
            for (size_t i = 0; i < npix; i++) hits += host[2 * npix + i] != 0xFFFFFFFFu;
            if (!out_prefix.empty() &&
                (inv_write_ppm((out_prefix + ".ppm").c_str(), host.data(), width, height, flags) != INV_OK ||
                 inv_write_pfm((out_prefix + ".pfm").c_str(), (const float*)(host.data() + npix), width, height) != INV_OK ||
                 inv_write_primid((out_prefix + ".pid").c_str(), host.data() + 2 * npix, width, height) != INV_OK))
                return die("writing the attachments");
        }
        std::printf("{\"rank\": %u, \"ranks\": %u, \"transport\": \"%s\", \"device\": %d, \"triangles\": %u, \"width\": %u, "
                    "\"height\": %u, \"spp\": %u, \"ao_rays\": %u, \"tile\": %u, \"tiles_dealt\": %u, \"slot_elements\": %llu, "
                    "\"frames\": %u, \"loop_ms\": %.3f, \"us_per_frame\": %.3f, \"hit_pixels_last_frame\": %llu}\n",
                    rank, ranks, transport.c_str(), device, nidx / 3, width, height, spp, ao, tile, grid[2] * grid[3],
                    (unsigned long long)slot_elems, frames, loop_ms, loop_ms * 1e3 / frames, (unsigned long long)hits);
        inv_comm_destroy(comm);
        (void)hipFree(planes), (void)hipFree(gathered), (void)hipFree(frame);
        inv_scene_destroy(scene);
        inv_mesh_destroy(mesh);
        inv_ctx_destroy(ctx);
        return 0;
    }

    // one attachment set per frame in flight (the swapchain images and the depth image of the reference)
    const uint32_t nsets = in_flight * per_launch;
    std::vector<Attachments> sets(nsets);
    for (auto& s : sets)
        if (hipMalloc(&s.color, npix * 4) != hipSuccess || hipMalloc(&s.depth, npix * 4) != hipSuccess ||
            hipMalloc(&s.primid, npix * 4) != hipSuccess) {
            std::fprintf(stderr, "invert_headless: hipMalloc of the attachments failed\n");
            return 1;
        }

    inv_params prm = {};
    prm.width = width, prm.height = height, prm.spp = spp, prm.ao_rays = ao, prm.flags = flags;
    std::vector<inv_camera> cams(per_launch);
    std::vector<inv_params> prms(per_launch, prm);
    std::vector<inv_outputs> outs(per_launch);
    uint32_t last_set = 0;
    const double t_loop = now_ms();
    for (uint32_t first = 0, k = 0; first < frames; first += per_launch, k++) { // mainLoop
        const uint32_t slot = k % in_flight, n = std::min(per_launch, frames - first);
        if (inv_frame_wait(ctx, slot) != INV_OK) return die("inv_frame_wait"); // vkWaitForFences
        for (uint32_t u = 0; u < n; u++) {                                     // updateUniformBuffer
            const int rc = turntable ? inv_camera_turntable(width, height, (float)(first + u) / 60.0f, &cams[u])
                                     : inv_camera_reference(width, height, &cams[u]);
            if (rc != INV_OK) return die("camera");
            const Attachments& s = sets[slot * per_launch + u];
            outs[u] = {s.color, s.depth, s.primid, 1};
            last_set = slot * per_launch + u;
        }
        if (inv_render_batch(scene, slot, n, cams.data(), prms.data(), outs.data()) != INV_OK) return die("inv_render_batch");
    }
    for (uint32_t slot = 0; slot < in_flight; slot++)
        if (inv_frame_wait(ctx, slot) != INV_OK) return die("inv_frame_wait");
    const double t_done = now_ms();

    // the last frame, read back instead of presented
    std::vector<uint8_t> color(npix * 4);
    std::vector<float> depth(npix);
    std::vector<uint32_t> primid(npix);
    if (hipMemcpy(color.data(), sets[last_set].color, npix * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(depth.data(), sets[last_set].depth, npix * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(primid.data(), sets[last_set].primid, npix * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        std::fprintf(stderr, "invert_headless: reading the attachments back failed\n");
        return 1;
    }
    uint64_t hits = 0;
    for (uint32_t p : primid) hits += p != 0xFFFFFFFFu;
    if (!out_prefix.empty()) {
        if (inv_write_ppm((out_prefix + ".ppm").c_str(), color.data(), width, height, flags) != INV_OK ||
            inv_write_png((out_prefix + ".png").c_str(), color.data(), width, height, flags) != INV_OK ||
            inv_write_pfm((out_prefix + ".pfm").c_str(), depth.data(), width, height) != INV_OK ||
            inv_write_primid((out_prefix + ".pid").c_str(), primid.data(), width, height) != INV_OK)
            return die("writing the attachments");
    }
    const double loop_ms = t_done - t_loop;
    const double rays = ((double)npix * spp + (double)hits * ao) * frames;
    std::printf("{\"vertices\": %u, \"triangles\": %u, \"width\": %u, \"height\": %u, \"spp\": %u, \"ao_rays\": %u, \"flags\": %u, "
                "\"frames\": %u, \"frames_in_flight\": %u, \"frames_per_launch\": %u, \"load_ms\": %.3f, \"build_ms\": %.3f, "
                "\"loop_ms\": %.3f, \"us_per_frame\": %.3f, \"mrays_per_s\": %.1f, \"hit_pixels_last_frame\": %llu}\n",
                nv, nidx / 3, width, height, spp, ao, flags, frames, in_flight, per_launch, t_build - t_load, t_built - t_build,
                loop_ms, loop_ms * 1e3 / frames, rays / (loop_ms * 1e-3) / 1e6, (unsigned long long)hits);
    for (auto& s : sets) (void)hipFree(s.color), (void)hipFree(s.depth), (void)hipFree(s.primid);
    inv_scene_destroy(scene);
    inv_mesh_destroy(mesh);
    inv_ctx_destroy(ctx);
    return 0;
}
