// C ABI (include/invert.h): handles, argument checks, error strings and the host side of the draw.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

#include <hip/hip_runtime.h>

#include "internal.h"

namespace inv {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

static int hip_fail(hipError_t e, const char* what) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return fail(INV_ERR_HIP, buf);
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

// C++ exceptions must not cross the C ABI: entry points that allocate host memory end with this.
#define INV_CATCH(who)                                                            \
    catch (const std::bad_alloc&) { return fail(INV_ERR_NOMEM, who ": out of host memory"); } \
    catch (...) { return fail(INV_ERR_ARG, who ": unexpected C++ exception"); }

} // namespace inv

using namespace inv;

// ---- device memory of the entry point that allocates nothing (inv_render_noalloc) ----------------------------
// The occlusion queue of a draw (256-byte header with the counter, then per queued pixel a 48-byte record and a
// 4-byte accumulator) normally comes from hipMalloc on first use.  Callers that cannot allocate (HIP graph capture,
// serving loops) get it from this array instead: part of the code object, so every GPU the library runs on holds
// its own copy from the moment the library is loaded there until the process ends.  The name below is a host-side
// stand-in, not a device address: host code uses the per-device pointer hipGetSymbolAddress returns.
constexpr uint64_t kNoAllocQueueCap = INV_NOALLOC_QUEUE_PIXELS;       // queued pixels
constexpr size_t kNoAllocBytes = 256 + kNoAllocQueueCap * (48 + 4); // 13 631 744 + 256 bytes
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];
constexpr int kNoAllocMaxDevices = 64;
static void* g_noalloc_ptr[kNoAllocMaxDevices]; // per device: where that GPU's copy of `scratch` lives

struct inv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // frame slots (created on first use): a stream each, and an event the join records on it
    hipStream_t flight[INV_MAX_FRAMES_IN_FLIGHT] = {};
    hipEvent_t flight_done[INV_MAX_FRAMES_IN_FLIGHT] = {};

    // what a scene build needs only while it runs; kept for the next build of this context (inv_ctx_trim frees it)
    void* build_scratch = nullptr;
    size_t build_scratch_bytes = 0;
    // The allocation of the last scene destroyed, for the next build that fits it (a scene that is built again and
    // again, each build after the old scene is gone: allocating 0.8 GB takes 40 us, a twentieth of the build).
    void* spare_arena = nullptr;
    size_t spare_arena_bytes = 0;
    int build_sort_hint = 0;      // see build_lbvh: kept from one build to the next of as many triangles (a scene
    uint32_t build_sort_ntri = 0; // that is built again and again), forgotten when another scene comes

    unsigned char* noalloc = nullptr; // this GPU's copy of `scratch` (inv_render_noalloc)
    uint32_t live_scenes = 0;         // scenes built on this context and not yet destroyed
};

struct inv_mesh {
    std::vector<float> xyzw;
    std::vector<uint32_t> idx;
    // after inv_mesh_pin: the same arrays in page-locked host memory (the vectors are then empty)
    float* pinned_xyzw = nullptr;
    uint32_t* pinned_idx = nullptr;
    size_t nxyzw = 0, nidx = 0; // element counts of the pinned arrays
    const float* vertices() const { return pinned_xyzw ? pinned_xyzw : xyzw.data(); }
    const uint32_t* indices() const { return pinned_idx ? pinned_idx : idx.data(); }
    size_t vertex_floats() const { return pinned_xyzw ? nxyzw : xyzw.size(); }
    size_t index_count() const { return pinned_idx ? nidx : idx.size(); }
};

// What one draw needs beside the caller's attachments.  Draws on different streams must not share it, so a scene
// holds one for the context's stream and one per frame slot.
struct FrameScratch {
    // device staging for host-pointer outputs (grow-only)
    void* stage[3] = {nullptr, nullptr, nullptr};
    uint64_t stage_elems = 0;
    float* ao_queue = nullptr; // occlusion-pass queue (grow-only): 256-byte header with the counter, then records
    uint64_t ao_cap = 0;
    BinMem bins; // bins of the binning pass (grow-only)
    unsigned long long* vis = nullptr; // visibility buffer of frames drawn one lane per triangle (grow-only)
    uint64_t vis_elems = 0;
};

struct inv_scene {
    inv_ctx* ctx = nullptr;
    DeviceScene d;
    float max_abs = 0.0f;
    float* vs_stage = nullptr;
    uint64_t vs_stage_floats = 0;
    float vertex_diag = 0.0f; // diagonal of the vertices' bounding box: default reach of occlusion rays
    FrameScratch main;                             // draws on the context's stream
    FrameScratch flight[INV_MAX_FRAMES_IN_FLIGHT]; // draws on the frame slots
};

namespace {

template <class T>
void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_device_scene(DeviceScene& d) {
    dfree(d.arena); // every array of the scene lives in this one allocation
    dfree(d.wnodes); // (but the occlusion kernel's wide nodes, made on demand)
    d.xyzw = nullptr, d.idx = nullptr, d.triA = d.triB = d.triC = nullptr;
    d.morton = nullptr, d.order = d.child = d.parent = d.leaf_parent = nullptr, d.nodes = d.cut = d.cut_groups = d.fine = nullptr, d.slot_of = nullptr;
}

void free_frame_scratch(FrameScratch& fs) {
    for (auto& p : fs.stage) dfree(p);
    fs.stage_elems = 0;
    dfree(fs.ao_queue);
    fs.ao_cap = 0;
    dfree(fs.bins.base);
    fs.bins.nbins = 0;
    dfree(fs.vis);
    fs.vis_elems = 0;
}

// Tile sharding terms of a frame (one rank: row-major, several: tile slots).
int fill_layout(const inv_params& p, FrameK* f) {
    if (p.width == 0 || p.height == 0) return fail(INV_ERR_ARG, "params: width and height must be non-zero");
    // Culling takes pixel coordinates through float32 with 1/16 pixel of slack: that stays far above one rounding of
    // the normalised coordinate only while an extent is well below 2^19 pixels.
    if (p.width > 65536u || p.height > 65536u) return fail(INV_ERR_ARG, "params: width and height are at most 65536");
    if ((uint64_t)p.width * p.height > 0x7FFFFFFFull) return fail(INV_ERR_ARG, "params: more than 2^31 pixels");
    if (p.spp > 4096u) return fail(INV_ERR_ARG, "params: at most 4096 samples per pixel");
    if (p.nranks > 65536u) return fail(INV_ERR_ARG, "params: at most 65536 ranks");
    f->w = p.width, f->h = p.height;
    f->wf = (float)p.width, f->hf = (float)p.height;
    f->rwf = 1.0f / f->wf, f->rhf = 1.0f / f->hf;
    f->spp = p.spp ? p.spp : 1u;
    f->flags = p.flags;
    f->seed = p.seed;
    f->tile = p.tile ? p.tile : 64u;
    f->nranks = p.nranks ? p.nranks : 1u;
    f->rank = p.rank;
    if (f->rank >= f->nranks) return fail(INV_ERR_ARG, "params: rank must be below nranks");
    f->first_rank = 0;
    bool idle_root = false; // this is rank 0 of a frame whose rank 0 takes no tiles
    if (p.flags & INV_ROOT_IDLE) {
        if (f->nranks < 3u) return fail(INV_ERR_ARG, "params: INV_ROOT_IDLE needs at least three ranks");
        idle_root = f->rank == 0;
        f->first_rank = 1, f->nranks -= 1u;
        f->rank = idle_root ? 0xFFFFFFFFu : f->rank - 1u;
    }
    if (f->tile > 1024u || (f->tile & 31u)) return fail(INV_ERR_ARG, "params: tile must be a multiple of 32, at most 1024");
    f->tiles_x = (p.width + f->tile - 1) / f->tile;
    f->tiles_y = (p.height + f->tile - 1) / f->tile;
    f->gx0 = f->gy0 = 0;
    f->ntiles = f->tiles_x * f->tiles_y;
    f->slots = idle_root ? 0u : (f->ntiles + f->nranks - 1) / f->nranks;
    f->rowmajor = f->nranks == 1 && !f->first_rank;
    f->npix = f->rowmajor ? (uint64_t)p.width * p.height : (uint64_t)f->slots * f->tile * f->tile;
    return INV_OK;
}

// Sparse sharding (INV_SPARSE with several ranks): only the tiles the hit rectangle touches are dealt out; rank 0
// holds the row-major frame and clears everything else itself.
void apply_sparse_grid(const uint32_t rect[4], FrameK* f) {
    if (f->nranks + f->first_rank <= 1) return;
    const uint32_t T = f->tile;
    if (rect[2] > rect[0] && rect[3] > rect[1]) {
        f->gx0 = rect[0] / T, f->gy0 = rect[1] / T;
        f->tiles_x = (rect[2] + T - 1) / T - f->gx0;
        f->tiles_y = (rect[3] + T - 1) / T - f->gy0;
    } else {
        f->gx0 = f->gy0 = f->tiles_x = f->tiles_y = 0;
    }
    f->ntiles = f->tiles_x * f->tiles_y;
    const bool root = f->first_rank ? f->rank == 0xFFFFFFFFu : f->rank == 0;
    f->slots = (f->ntiles + f->nranks - 1) / f->nranks; // (an idle rank 0 keeps the others' slot count: it sizes the gather)
    f->rowmajor = root;
    f->npix = f->rowmajor ? (uint64_t)f->w * f->h : (uint64_t)f->slots * T * T;
}

int check_rect(const uint32_t rect[4], uint32_t w, uint32_t h, const char* who) {
    if (!rect) return fail(INV_ERR_ARG, std::string(who) + ": rect is null");
    if (rect[0] > rect[2] || rect[1] > rect[3] || rect[2] > w || rect[3] > h)
        return fail(INV_ERR_ARG, std::string(who) + ": rect must satisfy x0 <= x1 <= width and y0 <= y1 <= height");
    return INV_OK;
}

} // namespace

extern "C" {

const char* inv_last_error(void) { return g_last_error.c_str(); }
const char* inv_version(void) { return "invert-mi355x 0.1 (gfx950)"; }

// ---- context -------------------------------------------------------------------------------------
int inv_ctx_create(int device, void* hip_stream, inv_ctx** out) {
    if (!out) return fail(INV_ERR_ARG, "inv_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(INV_ERR_NO_DEVICE, "inv_ctx_create: no HIP device is visible; this library has no CPU path");
    if (device < 0 || device >= count) return fail(INV_ERR_ARG, "inv_ctx_create: device index out of range");
    HIP_TRY(hipSetDevice(device));
    inv_ctx* c = new (std::nothrow) inv_ctx;
    if (!c) return fail(INV_ERR_NOMEM, "inv_ctx_create: out of memory");
    c->device = device;
    // once per device, with that device current: each GPU holds the array at its own address
    if (device < kNoAllocMaxDevices) {
        if (!g_noalloc_ptr[device]) {
            void* p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                delete c;
                return hip_fail(e != hipSuccess ? e : hipErrorInvalidSymbol, "hipGetSymbolAddress(scratch)");
            }
            g_noalloc_ptr[device] = p;
        }
        c->noalloc = (unsigned char*)g_noalloc_ptr[device];
    }
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return hip_fail(e, "hipStreamCreateWithFlags");
        }
        c->own_stream = true;
    }
    *out = c;
    return INV_OK;
}

int inv_ctx_destroy(inv_ctx* ctx) {
    if (!ctx) return INV_OK;
    if (ctx->live_scenes)
        return fail(INV_ERR_ARG, "inv_ctx_destroy: scenes built on this context are still alive; destroy them first");
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (uint32_t i = 0; i < INV_MAX_FRAMES_IN_FLIGHT; i++) {
        if (ctx->flight[i]) (void)hipStreamSynchronize(ctx->flight[i]), (void)hipStreamDestroy(ctx->flight[i]);
        if (ctx->flight_done[i]) (void)hipEventDestroy(ctx->flight_done[i]);
    }
    if (ctx->build_scratch) (void)hipFree(ctx->build_scratch);
    if (ctx->spare_arena) (void)hipFree(ctx->spare_arena);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return INV_OK;
}

int inv_ctx_stream(inv_ctx* ctx, void** hip_stream) {
    if (!ctx || !hip_stream) return fail(INV_ERR_ARG, "inv_ctx_stream: null argument");
    *hip_stream = (void*)ctx->stream;
    return INV_OK;
}

int inv_ctx_device(inv_ctx* ctx, int* device) {
    if (!ctx || !device) return fail(INV_ERR_ARG, "inv_ctx_device: null argument");
    *device = ctx->device;
    return INV_OK;
}

int inv_ctx_trim(inv_ctx* ctx) {
    if (!ctx) return fail(INV_ERR_ARG, "inv_ctx_trim: ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->build_scratch) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(ctx->build_scratch));
    }
    ctx->build_scratch = nullptr, ctx->build_scratch_bytes = 0;
    if (ctx->spare_arena) HIP_TRY(hipFree(ctx->spare_arena));
    ctx->spare_arena = nullptr, ctx->spare_arena_bytes = 0;
    return INV_OK;
}

int inv_ctx_synchronize(inv_ctx* ctx) {
    if (!ctx) return fail(INV_ERR_ARG, "inv_ctx_synchronize: ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return INV_OK;
}

// ---- mesh ----------------------------------------------------------------------------------------
int inv_mesh_parse_obj(const char* text, uint64_t size, inv_mesh** out) try {
    if (!out) return fail(INV_ERR_ARG, "inv_mesh_parse_obj: out is null");
    *out = nullptr;
    std::unique_ptr<inv_mesh> m(new inv_mesh);
    int rc = parse_obj_text(text, size, m->xyzw, m->idx);
    if (rc != INV_OK) return rc;
    *out = m.release();
    return INV_OK;
}
INV_CATCH("inv_mesh_parse_obj")

int inv_mesh_load_obj(const char* path, inv_mesh** out) try {
    if (!out) return fail(INV_ERR_ARG, "inv_mesh_load_obj: out is null");
    *out = nullptr;
    if (!path) return fail(INV_ERR_ARG, "inv_mesh_load_obj: path is null");
    FILE* fp = std::fopen(path, "rb"); // whole-file read (src/invert.cpp:44-50); a missing file is an error here
    if (!fp) return fail(INV_ERR_IO, std::string("inv_mesh_load_obj: cannot open ") + path);
    std::vector<char> text;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, fp)) > 0) text.insert(text.end(), chunk, chunk + got);
    bool bad = std::ferror(fp) != 0;
    std::fclose(fp);
    if (bad) return fail(INV_ERR_IO, std::string("inv_mesh_load_obj: read error on ") + path);
    return inv_mesh_parse_obj(text.data(), text.size(), out);
}
INV_CATCH("inv_mesh_load_obj")

int inv_mesh_load_obj_general(const char* path, inv_mesh** out) try {
    if (!out) return fail(INV_ERR_ARG, "inv_mesh_load_obj_general: out is null");
    *out = nullptr;
    if (!path) return fail(INV_ERR_ARG, "inv_mesh_load_obj_general: path is null");
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return fail(INV_ERR_IO, std::string("inv_mesh_load_obj_general: cannot open ") + path);
    std::vector<char> text;
    char chunk[1 << 16]; // chunked reads (the TODO at src/invert.cpp:41)
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, fp)) > 0) text.insert(text.end(), chunk, chunk + got);
    bool bad = std::ferror(fp) != 0;
    std::fclose(fp);
    if (bad) return fail(INV_ERR_IO, std::string("inv_mesh_load_obj_general: read error on ") + path);
    std::unique_ptr<inv_mesh> m(new inv_mesh);
    int rc = parse_obj_general(text.data(), text.size(), m->xyzw, m->idx);
    if (rc != INV_OK) return rc;
    *out = m.release();
    return INV_OK;
}
INV_CATCH("inv_mesh_load_obj_general")

int inv_mesh_from_arrays(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, inv_mesh** out) try {
    if (!out) return fail(INV_ERR_ARG, "inv_mesh_from_arrays: out is null");
    *out = nullptr;
    if ((nv && !xyzw) || (nidx && !idx)) return fail(INV_ERR_ARG, "inv_mesh_from_arrays: null array");
    if (nidx % 3u) return fail(INV_ERR_ARG, "inv_mesh_from_arrays: nidx must be a multiple of 3 (triangle list)");
    for (uint32_t i = 0; i < nidx; i++)
        if (idx[i] >= nv) return fail(INV_ERR_ARG, "inv_mesh_from_arrays: index out of range");
    std::unique_ptr<inv_mesh> m(new inv_mesh);
    m->xyzw.assign(xyzw, xyzw + (size_t)nv * 4);
    m->idx.assign(idx, idx + nidx);
    *out = m.release();
    return INV_OK;
}
INV_CATCH("inv_mesh_from_arrays")

int inv_mesh_counts(const inv_mesh* mesh, uint32_t* nv, uint32_t* nidx) {
    if (!mesh) return fail(INV_ERR_ARG, "inv_mesh_counts: mesh is null");
    if (nv) *nv = (uint32_t)(mesh->vertex_floats() / 4);
    if (nidx) *nidx = (uint32_t)mesh->index_count();
    return INV_OK;
}

const float* inv_mesh_vertices(const inv_mesh* mesh) { return mesh ? mesh->vertices() : nullptr; }
const uint32_t* inv_mesh_indices(const inv_mesh* mesh) { return mesh ? mesh->indices() : nullptr; }

int inv_mesh_pin(inv_mesh* mesh) {
    if (!mesh) return fail(INV_ERR_ARG, "inv_mesh_pin: mesh is null");
    if (mesh->pinned_xyzw || mesh->pinned_idx) return INV_OK; // already pinned
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(INV_ERR_NO_DEVICE, "inv_mesh_pin: page-locked memory needs a HIP device");
    const size_t nx = mesh->xyzw.size(), ni = mesh->idx.size();
    float* px = nullptr;
    uint32_t* pi = nullptr;
    hipError_t e = hipHostMalloc((void**)&px, std::max<size_t>(nx * 4, 16), hipHostMallocDefault);
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc(vertices)");
    e = hipHostMalloc((void**)&pi, std::max<size_t>(ni * 4, 16), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipHostFree(px);
        return hip_fail(e, "hipHostMalloc(indices)");
    }
    if (nx) std::memcpy(px, mesh->xyzw.data(), nx * 4);
    if (ni) std::memcpy(pi, mesh->idx.data(), ni * 4);
    mesh->pinned_xyzw = px, mesh->pinned_idx = pi, mesh->nxyzw = nx, mesh->nidx = ni;
    std::vector<float>().swap(mesh->xyzw);
    std::vector<uint32_t>().swap(mesh->idx);
    return INV_OK;
}

int inv_mesh_destroy(inv_mesh* mesh) {
    if (!mesh) return INV_OK;
    if (mesh->pinned_xyzw) (void)hipHostFree(mesh->pinned_xyzw);
    if (mesh->pinned_idx) (void)hipHostFree(mesh->pinned_idx);
    delete mesh;
    return INV_OK;
}

// ---- camera --------------------------------------------------------------------------------------
int inv_camera_reference(uint32_t width, uint32_t height, inv_camera* cam) {
    if (!cam) return fail(INV_ERR_ARG, "inv_camera_reference: cam is null");
    if (width == 0 || height == 0) return fail(INV_ERR_ARG, "inv_camera_reference: zero extent");
    reference_camera(width, height, cam);
    return INV_OK;
}

int inv_camera_turntable(uint32_t width, uint32_t height, float elapsed_seconds, inv_camera* cam) {
    if (!cam) return fail(INV_ERR_ARG, "inv_camera_turntable: cam is null");
    if (width == 0 || height == 0) return fail(INV_ERR_ARG, "inv_camera_turntable: zero extent");
    if (!std::isfinite(elapsed_seconds)) return fail(INV_ERR_ARG, "inv_camera_turntable: elapsed_seconds is not finite");
    turntable_camera(width, height, elapsed_seconds, cam);
    return INV_OK;
}

// ---- scene ---------------------------------------------------------------------------------------
int inv_scene_build_instanced(inv_ctx* ctx, const inv_mesh* mesh, const float* translations_xyz, uint32_t ninst,
                              inv_scene** out) try {
    if (!out) return fail(INV_ERR_ARG, "inv_scene_build: out is null");
    *out = nullptr;
    if (!ctx || !mesh) return fail(INV_ERR_ARG, "inv_scene_build: ctx or mesh is null");
    if (ninst == 0) return fail(INV_ERR_ARG, "inv_scene_build: ninst must be at least 1");
    const uint32_t nv = (uint32_t)(mesh->vertex_floats() / 4);
    const uint32_t base_ntri = (uint32_t)(mesh->index_count() / 3);
    const float* hv = mesh->vertices();
    const uint32_t* hi = mesh->indices();
    const uint64_t ntri64 = (uint64_t)base_ntri * ninst;
    if (ntri64 >= 0x7FFFFFFFull) return fail(INV_ERR_ARG, "inv_scene_build: more than 2^31 triangles");
    HIP_TRY(hipSetDevice(ctx->device));
    inv_scene* s = new (std::nothrow) inv_scene;
    if (!s) return fail(INV_ERR_NOMEM, "inv_scene_build: out of memory");
    s->ctx = ctx;
    DeviceScene& d = s->d;
    d.nv = nv;
    d.ntri = (uint32_t)ntri64;
    // Two device allocations: what the scene keeps, and what only the build needs (the context keeps the latter for
    // its next build: allocating and freeing half a gigabyte costs as much as a tenth of a five-million-triangle build).
    const size_t tri_bytes = Arena::aligned((size_t)d.ntri * 16);
    const bool tree = d.ntri >= 2;
    Arena keep, scratch;
    keep.size = Arena::aligned((size_t)nv * 16) + Arena::aligned((size_t)base_ntri * 12) + 3 * tri_bytes +
                lbvh_keep_bytes(d.ntri) + Arena::aligned((size_t)ninst * 12) + 256;
    scratch.size = (tree ? Arena::aligned((size_t)d.ntri * 48) + lbvh_scratch_bytes(d.ntri) : 0) + Arena::aligned((size_t)ninst * 12) + 256;
    int rc = INV_OK;
    hipError_t e = hipSuccess;
    auto bail = [&](hipError_t err, const char* what) {
        rc = hip_fail(err, what);
        (void)hipStreamSynchronize(ctx->stream);
        free_device_scene(d);
        delete s;
        return rc;
    };
    if (ctx->spare_arena && ctx->spare_arena_bytes >= keep.size && ctx->spare_arena_bytes <= keep.size + keep.size / 2) {
        keep.base = (char*)ctx->spare_arena, d.arena_bytes = ctx->spare_arena_bytes; // (its last reader is long done)
        ctx->spare_arena = nullptr, ctx->spare_arena_bytes = 0;
    } else {
        if ((e = hipMalloc((void**)&keep.base, keep.size)) != hipSuccess) return bail(e, "hipMalloc(scene)");
        d.arena_bytes = keep.size;
    }
    d.arena = keep.base;
    if (ctx->build_scratch_bytes < scratch.size) {
        if (ctx->build_scratch) (void)hipFree(ctx->build_scratch); // no build is running: every build ends with a sync
        ctx->build_scratch = nullptr, ctx->build_scratch_bytes = 0;
        if ((e = hipMalloc(&ctx->build_scratch, scratch.size)) != hipSuccess) return bail(e, "hipMalloc(build scratch)");
        ctx->build_scratch_bytes = scratch.size;
    }
    scratch.base = (char*)ctx->build_scratch;
    // The Obj arrays go up first (src/invert.cpp:1136-1140, 1161-1164).  From a pinned mesh (inv_mesh_pin: the
    // host-visible staging buffer of transferData, src/invert.cpp:993-1082) the copies are DMA straight out of the
    // arrays and these calls return at once; the context's stream keeps them in order with the build's kernels.
    // (A stream of its own for the copies was tried: it brought nothing more and took one of the process's four
    // hardware queues away from the frame slots.)
    if (nv) {
        d.xyzw = keep.take<float>((size_t)nv * 16);
        if ((e = hipMemcpyAsync(d.xyzw, hv, (size_t)nv * 16, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return bail(e, "hipMemcpyAsync(vertices)");
    }
    if (base_ntri) {
        d.idx = keep.take<uint32_t>((size_t)base_ntri * 12);
        if ((e = hipMemcpyAsync(d.idx, hi, (size_t)base_ntri * 12, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return bail(e, "hipMemcpyAsync(indices)");
    }
    // While the uploads run (from a pinned mesh they are DMA in the background): largest |coordinate| the rays can
    // meet (feeds the guard pad) and the bounding box of the vertices as stored, over all instances.  Rounding is monotonic, so the extremes of fl(v + t) over vertices and instances are
    // fl(vmin + tmin) and fl(vmax + tmax) per axis: no need to visit every pair.
    float max_abs = 0.0f;
    if (nv) {
        float vlo[3] = {INFINITY, INFINITY, INFINITY}, vhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < nv; i++)
            for (int c = 0; c < 3; c++) {
                const float v = hv[(size_t)i * 4 + c];
                vlo[c] = v < vlo[c] ? v : vlo[c], vhi[c] = v > vhi[c] ? v : vhi[c]; // a NaN is skipped, as by fmin/fmax
            }
        float lo[3], hi[3];
        for (int c = 0; c < 3; c++) {
            lo[c] = vlo[c], hi[c] = vhi[c];
            if (translations_xyz) {
                float tlo = INFINITY, thi = -INFINITY;
                for (uint32_t j = 0; j < ninst; j++) {
                    const float t = translations_xyz[(size_t)j * 3 + c];
                    tlo = std::fmin(tlo, t), thi = std::fmax(thi, t);
                }
                lo[c] = vlo[c] + tlo, hi[c] = vhi[c] + thi;
            }
            max_abs = std::fmax(max_abs, std::fmax(std::fabs(lo[c]), std::fabs(hi[c])));
        }
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        s->vertex_diag = std::sqrt((ex * ex + ey * ey) + ez * ez);
    }
    s->max_abs = max_abs;
    if (d.ntri) {
        d.triA = keep.take<float>((size_t)d.ntri * 16);
        d.triB = keep.take<float>((size_t)d.ntri * 16);
        d.triC = keep.take<float>((size_t)d.ntri * 16);
        // with a tree the assembled triangles are scratch: the build stores them again in Morton order
        // (as 48-byte records: the gather into Morton order then pulls one or two cache lines per triangle, not three)
        // (a mesh of up to 2 MB stays in every XCD's 4 MB L2 while the kernels that read it run: the build then
        // assembles a triangle again wherever it needs one, and the records are not stored at all)
        const bool cached_mesh = (size_t)nv * 16 + (size_t)base_ntri * 12 <= (size_t)2 << 20;
        const bool stored = !(tree && cached_mesh);
        float* records = tree && stored ? scratch.take<float>((size_t)d.ntri * 48) : nullptr;
        float* bounds = tree ? lbvh_take_bounds(scratch, d.ntri) : nullptr;
        float* A0 = !stored ? nullptr : tree ? records : d.triA;
        float* B0 = !stored ? nullptr : tree ? records + 4 : d.triB;
        float* C0 = !stored ? nullptr : tree ? records + 8 : d.triC;
        float* d_tr = nullptr;
        if (translations_xyz) {
            d_tr = keep.take<float>((size_t)ninst * 12); // (kept: draws of a cached mesh assemble triangles from it too)
            if ((e = hipMemcpyAsync(d_tr, translations_xyz, (size_t)ninst * 12, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
                return bail(e, "hipMemcpyAsync(translations)");
        }
        if ((e = launch_assemble(ctx->stream, d.xyzw, nv, d.idx, base_ntri, d_tr, ninst, A0, B0, C0, tree ? 3u : 1u, bounds)) != hipSuccess)
            return bail(e, "assemble kernel");
        if (ctx->build_sort_ntri != d.ntri) ctx->build_sort_hint = 0, ctx->build_sort_ntri = d.ntri;
        const MeshSource mesh_src = mesh_source(cached_mesh ? (const float4*)d.xyzw : nullptr, d.idx, d_tr, base_ntri, ninst);
        d.translations = d_tr, d.base_ntri = base_ntri, d.ninst = ninst, d.mesh_cached = cached_mesh;
        if ((e = build_lbvh(ctx->stream, d, keep, scratch, records, bounds, mesh_src, &ctx->build_sort_hint)) != hipSuccess) return bail(e, "LBVH build");
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize(scene build)");
    ctx->live_scenes++;
    *out = s;
    return INV_OK;
}
INV_CATCH("inv_scene_build_instanced")

int inv_scene_build(inv_ctx* ctx, const inv_mesh* mesh, inv_scene** out) {
    return inv_scene_build_instanced(ctx, mesh, nullptr, 1, out);
}

int inv_scene_counts(const inv_scene* scene, uint32_t* ntri, uint32_t* nnodes, uint64_t* scene_bytes) {
    if (!scene) return fail(INV_ERR_ARG, "inv_scene_counts: scene is null");
    uint32_t n = scene->d.ntri;
    uint32_t nodes = (scene->d.has_bvh && n >= 2) ? n - 1 : 0;
    if (ntri) *ntri = n;
    if (nnodes) *nnodes = nodes;
    if (scene_bytes) *scene_bytes = (uint64_t)n * 48 + (uint64_t)nodes * 64;
    return INV_OK;
}

int inv_scene_destroy(inv_scene* scene) {
    if (!scene) return INV_OK;
    (void)hipSetDevice(scene->ctx->device);
    (void)hipStreamSynchronize(scene->ctx->stream);
    for (hipStream_t fs : scene->ctx->flight)
        if (fs) (void)hipStreamSynchronize(fs); // draws in flight still read the scene
    {   // the scene's allocation becomes the context's spare one if it is the larger of the two
        inv_ctx* ctx = scene->ctx;
        DeviceScene& d = scene->d;
        if (d.arena && d.arena_bytes > ctx->spare_arena_bytes) {
            if (ctx->spare_arena) (void)hipFree(ctx->spare_arena);
            ctx->spare_arena = d.arena, ctx->spare_arena_bytes = d.arena_bytes;
            d.arena = nullptr;
        }
    }
    free_device_scene(scene->d);
    free_frame_scratch(scene->main);
    for (auto& fsr : scene->flight) free_frame_scratch(fsr);
    dfree(scene->vs_stage);
    if (scene->ctx->live_scenes) scene->ctx->live_scenes--;
    delete scene;
    return INV_OK;
}

int inv_scene_read_lbvh(inv_scene* scene, uint64_t* morton, uint32_t* order, uint32_t* child, uint32_t* parent,
                        uint32_t* leaf_parent, float* box) {
    if (!scene) return fail(INV_ERR_ARG, "inv_scene_read_lbvh: scene is null");
    const DeviceScene& d = scene->d;
    if (!d.has_bvh) return fail(INV_ERR_ARG, "inv_scene_read_lbvh: the scene has no LBVH (fewer than 2 triangles)");
    HIP_TRY(hipSetDevice(scene->ctx->device));
    HIP_TRY(hipStreamSynchronize(scene->ctx->stream));
    size_t n = d.ntri;
    if (morton) HIP_TRY(hipMemcpy(morton, d.morton, n * 8, hipMemcpyDeviceToHost));
    if (order) HIP_TRY(hipMemcpy(order, d.order, n * 4, hipMemcpyDeviceToHost));
    if (child || parent || leaf_parent || box) try {
        // the node records are all the scene keeps of the hierarchy: child references in words 12 and 13 of each, and
        // every node but the root is somebody's child
        std::vector<float> nodes((n - 1) * 16);
        HIP_TRY(hipMemcpy(nodes.data(), d.nodes, (n - 1) * 64, hipMemcpyDeviceToHost));
        if (parent) parent[0] = 0xFFFFFFFFu;
        for (size_t i = 0; i + 1 < n; i++) {
            if (box) std::memcpy(box + 12 * i, nodes.data() + 16 * i, 48);
            uint32_t c[2];
            std::memcpy(c, nodes.data() + 16 * i + 12, 8);
            if (child) child[2 * i] = c[0], child[2 * i + 1] = c[1];
            for (int k = 0; k < 2; k++) {
                const uint32_t at = c[k] & 0x7FFFFFFFu;
                if (c[k] >> 31) {
                    if (leaf_parent && at < n) leaf_parent[at] = (uint32_t)i;
                } else if (parent && at + 1 < n) {
                    parent[at] = (uint32_t)i;
                }
            }
        }
    } catch (...) {
        return fail(INV_ERR_NOMEM, "inv_scene_read_lbvh: out of host memory");
    }
    return INV_OK;
}

static_assert(INV_FINE_CUT_LEAVES == kFineMax, "the header documents the fine cut's record size");
int inv_scene_read_fine_cut(inv_scene* scene, float* records, uint32_t cap, uint32_t* nrecords) {
    if (!scene || !nrecords) return fail(INV_ERR_ARG, "inv_scene_read_fine_cut: null argument");
    const DeviceScene& d = scene->d;
    *nrecords = d.has_bvh ? d.nfine : 0u;
    const uint32_t n = std::min(*nrecords, cap);
    if (records && n) {
        HIP_TRY(hipSetDevice(scene->ctx->device));
        HIP_TRY(hipStreamSynchronize(scene->ctx->stream));
        HIP_TRY(hipMemcpy(records, d.fine, (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    return INV_OK;
}

// ---- vertex stage ----------------------------------------------------------------------------------
int inv_vertex_shader(inv_scene* scene, const inv_camera* cam, float* clip4, float* color3, int on_device) {
    if (!scene || !cam) return fail(INV_ERR_ARG, "inv_vertex_shader: scene or cam is null");
    inv_ctx* ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t nv = scene->d.nv;
    if (on_device) {
        HIP_TRY(launch_vertex_shader(ctx->stream, scene->d.xyzw, nv, *cam, clip4, color3));
        return INV_OK;
    }
    uint64_t need = (uint64_t)nv * 7;
    if (need > scene->vs_stage_floats) {
        dfree(scene->vs_stage);
        HIP_TRY(hipMalloc((void**)&scene->vs_stage, need * 4));
        scene->vs_stage_floats = need;
    }
    float* dclip = clip4 ? scene->vs_stage : nullptr;
    float* dcol = color3 ? scene->vs_stage + (size_t)nv * 4 : nullptr;
    HIP_TRY(launch_vertex_shader(ctx->stream, scene->d.xyzw, nv, *cam, dclip, dcol));
    if (clip4) HIP_TRY(hipMemcpyAsync(clip4, dclip, (size_t)nv * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (color3) HIP_TRY(hipMemcpyAsync(color3, dcol, (size_t)nv * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return INV_OK;
}

// ---- the draw ----------------------------------------------------------------------------------------
namespace {

// One draw enqueued on `stream`, with `fs` as its staging and occlusion queue (nothing else may use `fs` on
// another stream at the same time).
// Argument checks of a draw and the per-frame constants the kernels take.
int prepare_frame(const inv_scene* scene, const inv_camera* cam, const inv_params* params, FrameK* fk) {
    if (params->ao_rays != 0 && params->spp > 1)
        return fail(INV_ERR_UNSUPPORTED, "draw: ambient occlusion is defined for one sample per pixel");
    if (params->ao_rays > 4096) return fail(INV_ERR_ARG, "draw: at most 4096 ambient-occlusion rays per pixel");
    {
        // Bits above the INV_* flags switch parts of the kernels off for timing experiments (tools/phase_timing.py,
        // tools/tile_stats.py) and make the images wrong: they are refused unless INV_TIMING_SWITCHES is set.
        constexpr uint32_t known = INV_LAMBERT | INV_INVERT | INV_RGBA | INV_BRUTE_FORCE | INV_LANE_TRAVERSAL | INV_SPARSE | INV_BLOCKS | INV_ROOT_IDLE | INV_TILE_TRAVERSAL;
#ifdef INV_TIMING_BUILD
        static const bool allow = std::getenv("INV_TIMING_SWITCHES") != nullptr;
#else
        constexpr bool allow = false; // the release kernels do not contain the switches
#endif
        if ((params->flags & ~known) && !allow) return fail(INV_ERR_ARG, "draw: unknown bits in params.flags");
    }
    FrameK& f = *fk;
    std::memset(&f, 0, sizeof f);
    int rc = fill_layout(*params, &f);
    if (rc != INV_OK) return rc;
    if ((rc = frame_from_camera(*cam, scene->max_abs, &f)) != INV_OK) return rc;
    uint32_t rect[4] = {0, 0, f.w, f.h};
    if (scene->d.has_bvh) hit_rect(f, scene->d.bbox, rect);
    f.rx0 = rect[0], f.ry0 = rect[1], f.rx1 = rect[2], f.ry1 = rect[3];
    {
        // Bound on the components of the object-space ray direction over the hit rectangle (a pixel wider on every
        // side): the direction is linear in the pixel position, so its components are largest at a corner.  The same
        // float32 arithmetic as the kernels' dir_at, then a relative margin far above its roundings.
        float worst = 0.0f;
        for (int c = 0; c < 4; c++) {
            const float x = (c & 1) ? (float)f.rx1 + 1.0f : (float)f.rx0 - 1.0f, y = (c & 2) ? (float)f.ry1 + 1.0f : (float)f.ry0 - 1.0f;
            const float xn = x * f.inv_w2 - 1.0f, yn = y * f.inv_h2 - 1.0f;
            const float dxv = xn * f.inv_p00, dyv = yn * f.inv_p11;
            const float d[3] = {(f.d0x * dxv + f.d1x * dyv) - f.d2x, (f.d0y * dxv + f.d1y * dyv) - f.d2y,
                                (f.d0z * dxv + f.d1z * dyv) - f.d2z};
            for (float v : d) worst = std::max(worst, std::fabs(v));
        }
        f.dir_abs_max = worst * 1.0001f;
    }
    if (f.flags & INV_SPARSE) apply_sparse_grid(rect, &f);
    return INV_OK;
}

// Room for `nbins` bins in the scratch of a stream (grow-only, with some slack so that a moving camera does not
// allocate every few frames); new memory is cleared once, after that the kernels keep the counters at zero.
int ensure_bins(hipStream_t stream, FrameScratch& fs, uint64_t nbins) {
    if (nbins <= fs.bins.nbins) return INV_OK;
    HIP_TRY(hipStreamSynchronize(stream)); // an earlier draw may still use the old bins
    dfree(fs.bins.base);
    fs.bins.nbins = 0;
    const uint64_t room = nbins + nbins / 4 + 64;
    HIP_TRY(hipMalloc(&fs.bins.base, bin_mem_bytes(room)));
    fs.bins.nbins = room;
    HIP_TRY(hipMemsetAsync(fs.bins.base, 0, bin_mem_bytes(room), stream));
    return INV_OK;
}

// A visibility buffer of at least `elems` entries in the scratch of a stream (grow-only); new memory is set to the
// miss key once, after that the kernels leave every entry they touch at the miss key again.
int ensure_vis(hipStream_t stream, FrameScratch& fs, uint64_t elems) {
    if (elems <= fs.vis_elems) return INV_OK;
    HIP_TRY(hipStreamSynchronize(stream)); // an earlier draw may still use the old buffer
    dfree(fs.vis);
    fs.vis_elems = 0;
    HIP_TRY(hipMalloc((void**)&fs.vis, elems * 8));
    fs.vis_elems = elems;
    HIP_TRY(launch_vis_fill(stream, fs.vis, elems));
    return INV_OK;
}

// With `noalloc` the draw takes its occlusion queue from the context's copy of `scratch` and refuses whatever would
// need an allocation (host attachments, more queued pixels than the array holds).
int render_on(inv_scene* scene, hipStream_t stream, FrameScratch& fs, const inv_camera* cam, const inv_params* params,
              const inv_outputs* out, bool noalloc = false) {
    inv_ctx* ctx = scene->ctx;
    FrameK f;
    int rc = prepare_frame(scene, cam, params, &f);
    if (rc != INV_OK) return rc;
    if (noalloc) {
        if (!out->on_device)
            return fail(INV_ERR_ARG, "inv_render_noalloc: attachments must be device pointers (host attachments need staging memory)");
        if (!ctx->noalloc) return fail(INV_ERR_UNSUPPORTED, "inv_render_noalloc: this device index has no static scratch memory");
        // only pixels of the hit rectangle can be queued
        const uint64_t rect_px = (uint64_t)(f.rx1 - f.rx0) * (f.ry1 - f.ry0);
        if (params->ao_rays && std::min<uint64_t>(rect_px, f.npix) > kNoAllocQueueCap) {
            char buf[200];
            std::snprintf(buf, sizeof buf,
                          "inv_render_noalloc: up to %llu pixels can be hit, the static occlusion queue holds %llu; use inv_render",
                          (unsigned long long)std::min<uint64_t>(rect_px, f.npix), (unsigned long long)kNoAllocQueueCap);
            return fail(INV_ERR_ARG, buf);
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    Outputs dev = {nullptr, nullptr, nullptr};
    if (out->on_device) {
        dev.color = (uint8_t*)out->color, dev.depth = (float*)out->depth, dev.primid = (uint32_t*)out->primid;
    } else {
        if (f.npix > fs.stage_elems) {
            HIP_TRY(hipStreamSynchronize(stream)); // an earlier draw may still write the old staging
            for (auto& p : fs.stage) dfree(p);
            fs.stage_elems = 0;
            for (auto& p : fs.stage) HIP_TRY(hipMalloc(&p, f.npix * 4));
            fs.stage_elems = f.npix;
        }
        if (out->color) dev.color = (uint8_t*)fs.stage[0];
        if (out->depth) dev.depth = (float*)fs.stage[1];
        if (out->primid) dev.primid = (uint32_t*)fs.stage[2];
    }
    AoK ao = {0, 0.0f, nullptr, nullptr, nullptr};
    if (params->ao_rays && noalloc) {
        // same layout as the allocated queue, in this GPU's copy of the static array (its last call's contents are
        // still there: the counter is cleared here, an accumulator by the lane that queues its record)
        float* q = (float*)ctx->noalloc;
        ao.rays = params->ao_rays;
        ao.tmax = params->ao_tmax > 0.0f ? params->ao_tmax : scene->vertex_diag;
        ao.count = (uint32_t*)q;
        ao.queue = q + 64;
        ao.acc = (uint32_t*)(ao.queue + (size_t)kNoAllocQueueCap * 12);
        HIP_TRY(hipMemsetAsync(ao.count, 0, 256, stream));
    } else if (params->ao_rays) {
        if (f.npix > fs.ao_cap) { // 256-byte header (counters), then at most one 48-byte record and one
            HIP_TRY(hipStreamSynchronize(stream)); // 4-byte accumulator per pixel; an earlier draw may still use the old queue
            dfree(fs.ao_queue);
            fs.ao_cap = 0;
            HIP_TRY(hipMalloc((void**)&fs.ao_queue, f.npix * 52 + 256));
            fs.ao_cap = f.npix;
        }
        ao.rays = params->ao_rays;
        ao.tmax = params->ao_tmax > 0.0f ? params->ao_tmax : scene->vertex_diag;
        ao.count = (uint32_t*)fs.ao_queue;
        ao.queue = fs.ao_queue + 64; // records start 256 bytes in
        ao.acc = (uint32_t*)(ao.queue + (size_t)fs.ao_cap * 12);
        HIP_TRY(hipMemsetAsync(ao.count, 0, 256, stream));
    }
    const bool brute = (params->flags & INV_BRUTE_FORCE) || !scene->d.has_bvh;
    if (brute) {
        HIP_TRY(launch_render_brute(stream, f, scene->d, ao, dev));
    } else {
        // (the entry point that allocates nothing draws from the coarse cut: same pixels, no bins)
        const bool scatter = !noalloc && scatter_wanted(f, scene->d, dev);
        const uint32_t nbins = noalloc || scatter ? 0u : tile_bins_wanted(f, scene->d, dev, false);
        if (nbins && (rc = ensure_bins(stream, fs, nbins)) != INV_OK) return rc;
        if (scatter && (rc = ensure_vis(stream, fs, (uint64_t)f.w * f.h)) != INV_OK) return rc;
        HIP_TRY(launch_render_bvh(stream, f, scene->d, ao, dev, nbins ? &fs.bins : nullptr, scatter ? fs.vis : nullptr));
    }
    if (ao.rays && !noalloc && scene->d.has_bvh && !scene->d.wnodes) {
        // the first draw of this scene with occlusion rays: the occlusion kernel's four-wide node records, derived
        // from the LBVH once (64 bytes per internal node; the entry point that allocates nothing walks the LBVH's own
        // two-wide records instead).  Other frame slots may draw with them next, so their making is waited for.
        // The primary kernel has already queued this frame's occlusion records, so nothing here may end the draw:
        // without the wide records (no memory for them) the occlusion kernel walks the LBVH's own two-wide records,
        // as it does for the entry point that allocates nothing, with the same counts.
        uint32_t* w = nullptr;
        if (hipMalloc((void**)&w, (size_t)(scene->d.ntri - 1u) * 64u) != hipSuccess) {
            (void)hipGetLastError(); // cleared: the draw goes on
            w = nullptr;
        } else {
            hipError_t e = launch_make_wnodes(stream, scene->d, w);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                (void)hipFree(w);
                return fail(INV_ERR_HIP, std::string("wide nodes: ") + hipGetErrorString(e));
            }
            scene->d.wnodes = w;
        }
    }
    HIP_TRY(launch_ao(stream, f, scene->d, ao, dev, noalloc || (ao.rays && !scene->d.wnodes)));
    if (!out->on_device) {
        if (out->color) HIP_TRY(hipMemcpyAsync(out->color, dev.color, f.npix * 4, hipMemcpyDeviceToHost, stream));
        if (out->depth) HIP_TRY(hipMemcpyAsync(out->depth, dev.depth, f.npix * 4, hipMemcpyDeviceToHost, stream));
        if (out->primid) HIP_TRY(hipMemcpyAsync(out->primid, dev.primid, f.npix * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return INV_OK;
}

// The slot's stream, created on first use (the context's device must be current).
int flight_stream(inv_ctx* ctx, uint32_t slot, hipStream_t* s) {
    if (slot >= INV_MAX_FRAMES_IN_FLIGHT) return fail(INV_ERR_ARG, "frame slot must be below INV_MAX_FRAMES_IN_FLIGHT");
    if (!ctx->flight[slot]) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(st);
            return hip_fail(e, "hipEventCreateWithFlags");
        }
        ctx->flight[slot] = st, ctx->flight_done[slot] = ev;
    }
    *s = ctx->flight[slot];
    return INV_OK;
}

} // namespace

int inv_render(inv_scene* scene, const inv_camera* cam, const inv_params* params, const inv_outputs* out) {
    if (!scene || !cam || !params || !out) return fail(INV_ERR_ARG, "inv_render: null argument");
    return render_on(scene, scene->ctx->stream, scene->main, cam, params, out);
}

int inv_draw_path(const inv_scene* scene, const inv_camera* cam, const inv_params* params, uint32_t* path) {
    if (!scene || !cam || !params || !path) return fail(INV_ERR_ARG, "inv_draw_path: null argument");
    FrameK f;
    int rc = prepare_frame(scene, cam, params, &f);
    if (rc != INV_OK) return rc;
    const DeviceScene& d = scene->d;
    const Outputs none = {nullptr, nullptr, nullptr}; // (null pointers count as 16-byte aligned)
    if ((params->flags & INV_BRUTE_FORCE) || !d.has_bvh)
        *path = INV_PATH_BRUTE;
    else if (params->flags & INV_LANE_TRAVERSAL)
        *path = INV_PATH_LANES;
    else if (scatter_wanted(f, d, none))
        *path = INV_PATH_SCATTER;
    else {
        const int mode = tile_mode_for(f, d, none, false);
        *path = (mode == 0 ? INV_PATH_BLOCKS : mode == 1 ? INV_PATH_TILES : INV_PATH_TILES_BATCHED) |
                (tile_bins_wanted(f, d, none, false) ? (uint32_t)INV_PATH_BINNED : 0u);
    }
    return INV_OK;
}

int inv_render_noalloc(inv_scene* scene, const inv_camera* cam, const inv_params* params, const inv_outputs* out) {
    if (!scene || !cam || !params || !out) return fail(INV_ERR_ARG, "inv_render_noalloc: null argument");
    return render_on(scene, scene->ctx->stream, scene->main, cam, params, out, true);
}

// ---- frames in flight ----------------------------------------------------------------------------------
int inv_render_in_flight(inv_scene* scene, uint32_t slot, const inv_camera* cam, const inv_params* params,
                         const inv_outputs* out) {
    if (!scene || !cam || !params || !out) return fail(INV_ERR_ARG, "inv_render_in_flight: null argument");
    HIP_TRY(hipSetDevice(scene->ctx->device));
    hipStream_t s = nullptr;
    int rc = flight_stream(scene->ctx, slot, &s);
    if (rc != INV_OK) return rc;
    return render_on(scene, s, scene->flight[slot], cam, params, out);
}

// ---- several draws in one submission ---------------------------------------------------------------------
int inv_render_batch(inv_scene* scene, uint32_t slot, uint32_t nframes, const inv_camera* cams, const inv_params* params,
                     const inv_outputs* outs) {
    if (!scene || (nframes && (!cams || !params || !outs))) return fail(INV_ERR_ARG, "inv_render_batch: null argument");
    inv_ctx* ctx = scene->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = ctx->stream;
    FrameScratch* fs = &scene->main;
    if (slot != INV_CONTEXT_STREAM) {
        int rc = flight_stream(ctx, slot, &stream);
        if (rc != INV_OK) return rc;
        fs = &scene->flight[slot];
    }
    // every frame is checked before anything is enqueued
    FrameK fk[INV_BATCH_MAX];
    Outputs dev[INV_BATCH_MAX];
    for (uint32_t i = 0; i < nframes; i++) {
        int rc = prepare_frame(scene, &cams[i], &params[i], &fk[0]);
        if (rc != INV_OK) return rc;
    }
    // Frames that can share a launch are collected (up to INV_BATCH_MAX); a frame that cannot (occlusion rays, host
    // attachments, another traversal, a variant of the tile kernel other than the blocks) is drawn on its own after
    // the launch of those collected before it.  Everything runs on one stream, and the attachments do not overlap.
    // (A frame left alone is drawn as a single draw: that picks the variant that finishes one frame soonest.)
    uint32_t pending = 0, last = 0; // frames collected; index of the latest of them
    uint64_t pending_bins = 0;      // bins their binning pass needs
    const auto flush = [&]() -> int {
        const uint32_t n = pending;
        const uint64_t nbins = pending_bins;
        pending = 0, pending_bins = 0;
        if (n == 1) return render_on(scene, stream, *fs, &cams[last], &params[last], &outs[last]);
        if (n) {
            if (nbins) {
                const int rc = ensure_bins(stream, *fs, nbins);
                if (rc != INV_OK) return rc;
            }
            HIP_TRY(launch_render_batch(stream, fk, dev, n, scene->d, nbins ? &fs->bins : nullptr));
        }
        return INV_OK;
    };
    for (uint32_t i = 0; i < nframes; i++) {
        FrameK& f = fk[pending];
        int rc = prepare_frame(scene, &cams[i], &params[i], &f);
        if (rc != INV_OK) return rc;
        const Outputs o = {(uint8_t*)outs[i].color, (float*)outs[i].depth, (uint32_t*)outs[i].primid};
        if (outs[i].on_device && params[i].ao_rays == 0 && tile_batchable(f, scene->d, o)) {
            pending_bins += tile_bins_wanted(f, scene->d, o, true);
            dev[pending++] = o, last = i;
            if (pending == INV_BATCH_MAX && (rc = flush()) != INV_OK) return rc;
        } else {
            if ((rc = flush()) != INV_OK) return rc;
            if ((rc = render_on(scene, stream, *fs, &cams[i], &params[i], &outs[i])) != INV_OK) return rc;
        }
    }
    return flush();
}

int inv_frame_wait(inv_ctx* ctx, uint32_t slot) {
    if (!ctx) return fail(INV_ERR_ARG, "inv_frame_wait: ctx is null");
    if (slot >= INV_MAX_FRAMES_IN_FLIGHT) return fail(INV_ERR_ARG, "inv_frame_wait: frame slot must be below INV_MAX_FRAMES_IN_FLIGHT");
    if (!ctx->flight[slot]) return INV_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->flight[slot]));
    return INV_OK;
}

int inv_frames_join(inv_ctx* ctx) {
    if (!ctx) return fail(INV_ERR_ARG, "inv_frames_join: ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < INV_MAX_FRAMES_IN_FLIGHT; i++) {
        if (!ctx->flight[i]) continue;
        HIP_TRY(hipEventRecord(ctx->flight_done[i], ctx->flight[i]));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->flight_done[i], 0));
    }
    return INV_OK;
}

int inv_frame_stream(inv_ctx* ctx, uint32_t slot, void** hip_stream) {
    if (!ctx || !hip_stream) return fail(INV_ERR_ARG, "inv_frame_stream: null argument");
    *hip_stream = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    int rc = flight_stream(ctx, slot, &s);
    if (rc != INV_OK) return rc;
    *hip_stream = (void*)s;
    return INV_OK;
}

int inv_write_ppm(const char* path, const void* color, uint32_t width, uint32_t height, uint32_t flags) {
    if (!path || !color || !width || !height) return fail(INV_ERR_ARG, "inv_write_ppm: null argument or zero extent");
    return write_ppm(path, (const uint8_t*)color, width, height, (flags & INV_RGBA) != 0);
}

int inv_write_png(const char* path, const void* color, uint32_t width, uint32_t height, uint32_t flags) try {
    if (!path || !color || !width || !height) return fail(INV_ERR_ARG, "inv_write_png: null argument or zero extent");
    return write_png(path, (const uint8_t*)color, width, height, (flags & INV_RGBA) != 0);
}
INV_CATCH("inv_write_png")

int inv_write_pfm(const char* path, const float* depth, uint32_t width, uint32_t height) {
    if (!path || !depth || !width || !height) return fail(INV_ERR_ARG, "inv_write_pfm: null argument or zero extent");
    return write_pfm(path, depth, width, height);
}

int inv_write_primid(const char* path, const uint32_t* primid, uint32_t width, uint32_t height) {
    if (!path || !primid || !width || !height) return fail(INV_ERR_ARG, "inv_write_primid: null argument or zero extent");
    return write_primid(path, primid, width, height);
}

int inv_tile_layout(const inv_params* params, uint64_t* npixels, uint32_t* tile_ids, uint32_t tile_ids_cap,
                    uint32_t* ntiles) {
    if (!params) return fail(INV_ERR_ARG, "inv_tile_layout: params is null");
    FrameK f;
    std::memset(&f, 0, sizeof f);
    int rc = fill_layout(*params, &f);
    if (rc != INV_OK) return rc;
    if (npixels) *npixels = f.npix;
    uint32_t count = 0;
    for (uint32_t slot = 0; slot < f.slots; slot++) {
        uint64_t id = (uint64_t)slot * f.nranks + f.rank;
        if (id >= f.ntiles) break;
        if (tile_ids && count < tile_ids_cap) tile_ids[count] = (uint32_t)id;
        count++;
    }
    if (ntiles) *ntiles = count;
    return INV_OK;
}

int inv_scene_hit_rect(const inv_scene* scene, const inv_camera* cam, uint32_t width, uint32_t height,
                       uint32_t rect[4]) {
    if (!scene || !cam || !rect) return fail(INV_ERR_ARG, "inv_scene_hit_rect: null argument");
    inv_params p;
    std::memset(&p, 0, sizeof p);
    p.width = width, p.height = height;
    FrameK f;
    std::memset(&f, 0, sizeof f);
    int rc = fill_layout(p, &f);
    if (rc != INV_OK) return rc;
    if ((rc = frame_from_camera(*cam, scene->max_abs, &f)) != INV_OK) return rc;
    rect[0] = 0, rect[1] = 0, rect[2] = f.w, rect[3] = f.h;
    if (scene->d.has_bvh) hit_rect(f, scene->d.bbox, rect);
    return INV_OK;
}

int inv_sparse_layout(const inv_params* params, const uint32_t rect[4], uint32_t grid[4], uint64_t* npixels,
                      uint32_t* tile_ids, uint32_t tile_ids_cap, uint32_t* ntiles) {
    if (!params) return fail(INV_ERR_ARG, "inv_sparse_layout: params is null");
    FrameK f;
    std::memset(&f, 0, sizeof f);
    int rc = fill_layout(*params, &f);
    if (rc != INV_OK) return rc;
    if ((rc = check_rect(rect, f.w, f.h, "inv_sparse_layout")) != INV_OK) return rc;
    const uint32_t frame_tiles_x = f.tiles_x;
    if (f.nranks + f.first_rank <= 1) return fail(INV_ERR_ARG, "inv_sparse_layout: the sparse composite needs at least two ranks");
    apply_sparse_grid(rect, &f);
    if (grid) grid[0] = f.gx0, grid[1] = f.gy0, grid[2] = f.tiles_x, grid[3] = f.tiles_y;
    if (npixels) *npixels = f.npix;
    uint32_t count = 0;
    for (uint64_t id = f.rank; id < f.ntiles; id += f.nranks) { // (an idle rank 0 has rank 0xFFFFFFFF here: no tiles)
        if (tile_ids && count < tile_ids_cap)
            tile_ids[count] = (f.gy0 + (uint32_t)id / f.tiles_x) * frame_tiles_x + f.gx0 + (uint32_t)id % f.tiles_x;
        count++;
    }
    if (ntiles) *ntiles = count;
    return INV_OK;
}

int inv_deinterleave_sparse(inv_ctx* ctx, const inv_params* params, const uint32_t rect[4], const void* gathered,
                            uint64_t rank_stride, uint64_t plane_stride, uint32_t nplanes, void* frame,
                            uint64_t frame_plane_stride) {
    if (!ctx || !params || !frame) return fail(INV_ERR_ARG, "inv_deinterleave_sparse: null argument");
    FrameK f;
    std::memset(&f, 0, sizeof f);
    inv_params p = *params;
    p.rank = 0;
    int rc = fill_layout(p, &f);
    if (rc != INV_OK) return rc;
    if ((rc = check_rect(rect, f.w, f.h, "inv_deinterleave_sparse")) != INV_OK) return rc;
    if (f.nranks + f.first_rank <= 1) return fail(INV_ERR_ARG, "inv_deinterleave_sparse: the sparse composite needs at least two ranks");
    apply_sparse_grid(rect, &f);
    if (f.ntiles == 0 || nplanes == 0) return INV_OK; // nothing was dealt out: rank 0 already holds the frame
    if (!gathered) return fail(INV_ERR_ARG, "inv_deinterleave_sparse: gathered is null");
    const uint64_t own = (uint64_t)f.slots * f.tile * f.tile;
    if (plane_stride < own || rank_stride < plane_stride * (nplanes - 1) + own)
        return fail(INV_ERR_ARG, "inv_deinterleave_sparse: strides are smaller than one rank's planes");
    if (frame_plane_stride < (uint64_t)f.w * f.h)
        return fail(INV_ERR_ARG, "inv_deinterleave_sparse: frame_plane_stride is smaller than the frame");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_deinterleave_sparse(ctx->stream, f, (const uint32_t*)gathered, rank_stride, plane_stride, nplanes,
                                       (uint32_t*)frame, frame_plane_stride));
    return INV_OK;
}

int inv_deinterleave(inv_ctx* ctx, const inv_params* params, const void* gathered, uint64_t rank_stride, void* frame,
                     uint32_t elem_size) {
    if (!ctx || !params || !gathered || !frame) return fail(INV_ERR_ARG, "inv_deinterleave: null argument");
    if (elem_size != 4) return fail(INV_ERR_ARG, "inv_deinterleave: elem_size must be 4");
    FrameK f;
    std::memset(&f, 0, sizeof f);
    inv_params p = *params;
    p.rank = 0;
    int rc = fill_layout(p, &f);
    if (rc != INV_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    // one drawing rank's buffer (with INV_ROOT_IDLE the layout of rank 0 itself has no slots, the others' do)
    const uint64_t own = (uint64_t)((f.ntiles + f.nranks - 1) / f.nranks) * f.tile * f.tile;
    if (rank_stride != 0 && rank_stride < own) return fail(INV_ERR_ARG, "inv_deinterleave: rank_stride is smaller than one rank's buffer");
    HIP_TRY(launch_deinterleave(ctx->stream, f, (const uint32_t*)gathered, rank_stride ? rank_stride : own, (uint32_t*)frame));
    return INV_OK;
}

} // extern "C"
