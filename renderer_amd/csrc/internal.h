// Host-side declarations shared by the C ABI, the loader, the camera and the kernel launchers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/invert.h"
#include "frame.h"

namespace inv {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// A1 -- obj_loader.cpp
int parse_obj_text(const char* text, uint64_t size, std::vector<float>& xyzw, std::vector<uint32_t>& idx);

// io_extras.cpp
int parse_obj_general(const char* text, uint64_t size, std::vector<float>& xyzw, std::vector<uint32_t>& idx);
int write_ppm(const char* path, const uint8_t* color, uint32_t w, uint32_t h, bool rgba_order);
int write_png(const char* path, const uint8_t* color, uint32_t w, uint32_t h, bool rgba_order);
int write_pfm(const char* path, const float* depth, uint32_t w, uint32_t h);
int write_primid(const char* path, const uint32_t* primid, uint32_t w, uint32_t h);

// A2 -- camera.cpp
void reference_camera(uint32_t width, uint32_t height, inv_camera* cam);
void turntable_camera(uint32_t width, uint32_t height, float elapsed_seconds, inv_camera* cam);
// view*model, its inverse (double Gauss-Jordan rounded to float) and the Lambert light direction.
int frame_from_camera(const inv_camera& cam, float max_abs_coord, FrameK* f);
// Pixel rectangle [x0, x1) x [y0, y1) outside which no ray can hit anything inside `bbox`.
void hit_rect(const FrameK& f, const float bbox[6], uint32_t rect[4]);

// kernels.hip -- launchers (all asynchronous on `stream`)
// Start set of the tile kernel: at most kCutCap subtrees (or single leaves) that together cover the scene, each
// with its box; subtrees of at most kRangeMax leaves are taken whole as a range of the sorted triangle array.
constexpr uint32_t kCutCap = 2048, kRangeMax = 16, kCutGroup = 16;
// Leaves per record of the fine cut (a power of two, at most kRangeMax).  Smaller records have tighter boxes, so a
// block is handed fewer triangles that cannot touch it, but there are more of them for the binning pass to place and
// a block's list (256 ranges) holds fewer triangles: with 4 leaves the tiles of the 5 M-triangle scene (some 700
// triangles each) overflowed their lists and the frame took 3.7 ms instead of 0.52 ms with 16.
#ifndef INV_FINE_MAX
#define INV_FINE_MAX 16
#endif
constexpr uint32_t kFineMax = INV_FINE_MAX;
static_assert(kFineMax <= kRangeMax && (kFineMax & (kFineMax - 1u)) == 0u, "a power of two, at most kRangeMax");

// One device allocation handed out in 256-byte aligned pieces (a scene build makes two: what the scene keeps and
// what only the build needs).
struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
    static size_t aligned(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    template <class T>
    T* take(size_t bytes) {
        T* p = (T*)(base + used);
        used += aligned(bytes ? bytes : 16);
        return used <= size ? p : nullptr;
    }
};

// The occlusion kernel's node records hold child boxes on a 16-bit grid over the scene's bounds, rounded outwards by
// more than a grid step (so the rounding of the quantisation and of the ray's folded plane terms stays inside the
// margin): 12 bytes per box instead of 24.  Index k of axis a stands for the coordinate org[a] + k * step[a].
struct QGrid {
    float lo[3], org[3], step[3], inv[3];
};
__host__ __device__ inline QGrid qgrid_for(const float* bbox) {
    QGrid g;
    const float ex = bbox[3] - bbox[0], ey = bbox[4] - bbox[1], ez = bbox[5] - bbox[2];
    const float emax = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-30f));
    for (int a = 0; a < 3; a++) {
        const float e = fmaxf(bbox[3 + a] - bbox[a], emax * (1.0f / 4096.0f)); // a flat scene still gets a grid
        g.lo[a] = bbox[a];
        g.step[a] = e / 65532.0f;
        g.inv[a] = 65532.0f / e;
        g.org[a] = bbox[a] - g.step[a];
    }
    return g;
}

struct DeviceScene {
    void* arena = nullptr; // every pointer below points into this one allocation
    uint32_t nv = 0, ntri = 0;
    float* xyzw = nullptr;    // nv float4 (mesh vertices as uploaded)
    uint32_t* idx = nullptr;  // 3 * base_ntri indices
    float* triA = nullptr;    // ntri float4 each, storage order (LBVH order once built)
    float* triB = nullptr;
    float* triC = nullptr;
    // LBVH
    uint64_t* morton = nullptr;    // ntri, sorted
    uint32_t* order = nullptr;     // ntri: storage slot -> primitive id
    uint32_t* child = nullptr;     // 2*(ntri-1)
    uint32_t* parent = nullptr;    // ntri-1
    uint32_t* leaf_parent = nullptr; // ntri
    uint32_t* slot_of = nullptr;   // ntri: primitive id -> storage slot (the inverse of `order`)
    float* nodes = nullptr;        // 16 floats per internal node: two child boxes, two child refs, leaf range
    uint32_t* wnodes = nullptr;    // 16 words per internal node: its (up to four) grandchildren, each a box on the grid
                                   // of qgrid_for(bbox) (see QGrid) and a child reference, for the occlusion kernel; an
                                   // allocation of its own, made by the first draw with occlusion rays (launch_make_wnodes)
    float* cut = nullptr;          // kCutCap records of 8 floats: lo xyz, first leaf | node, hi xyz, leaf count
    float* cut_groups = nullptr;   // boxes of kCutGroup consecutive cut records: 8 floats each
    uint32_t ncut = 0;             // records in use
    float* fine = nullptr;         // the fine cut: every subtree of at most kFineMax leaves under a larger parent,
                                   // 8 floats per record (lo xyz, first leaf, hi xyz, leaf count), in no particular order
    uint32_t nfine = 0;            // records of the fine cut; 0: the scene has none (a few triangles, or too unbalanced)
    size_t arena_bytes = 0;        // size of the one allocation that holds every array above
    bool has_bvh = false;
    uint32_t height = 0;           // height of the root (a node over two leaves has height 1)
    float bbox[6] = {0, 0, 0, 0, 0, 0};
    // What the triangles were assembled from, for kernels that assemble a triangle again instead of fetching its
    // record (see MeshSource): the instances' translations (xyz each; null: one instance, not moved), the triangles of
    // one instance, the instances, and whether the mesh is small enough to stay in the caches.
    float* translations = nullptr;
    uint32_t base_ntri = 0, ninst = 1;
    bool mesh_cached = false;
};

hipError_t launch_vertex_shader(hipStream_t s, const float* xyzw, uint32_t nv, const inv_camera& cam, float* clip4,
                                float* color3);
// What primitive assembly reads: the Obj arrays in device memory and the instances' translations.
struct MeshSource {
    const float4* xyzw;
    const uint32_t* idx;
    const float* translations; // xyz per instance, or null
    uint32_t base_ntri, ninst;
    // g / base_ntri for any 32-bit g without a division (T. Granlund, P. Montgomery, "Division by Invariant Integers
    // using Multiplication", 1994, figure 4.1): q = mulhi(div_magic, g); (q + ((g - q) >> div_sh1)) >> div_sh2
    uint32_t div_magic, div_sh1, div_sh2;
};
inline MeshSource mesh_source(const float4* xyzw, const uint32_t* idx, const float* translations, uint32_t base_ntri, uint32_t ninst) {
    MeshSource m = {xyzw, idx, translations, base_ntri, ninst, 1u, 0u, 0u};
    const uint32_t d = base_ntri ? base_ntri : 1u;
    uint32_t l = 0;
    while ((1ull << l) < d) l++; // ceil(log2 d)
    m.div_magic = (uint32_t)((((1ull << l) - d) << 32) / d + 1u), m.div_sh1 = l < 1u ? l : 1u, m.div_sh2 = l > 0u ? l - 1u : 0u;
    return m;
}
// Triangle g's three 16-byte items go to A[g * stride], B[g * stride], C[g * stride] (in float4 units): stride 1
// fills three arrays, stride 3 with B = A + 4 floats and C = A + 8 floats one array of 48-byte records.
// `bounds` (or null): bounds_blocks_for(triangles) entries of six floats, the box of a share of the triangles each
// (lo xyz, hi xyz), for build_lbvh.  A, B and C may be null with `bounds`: the records are then worked out for the
// bounds only and stored nowhere (a build that assembles them again from a cached mesh wherever it needs them).
inline uint32_t bounds_blocks_for(uint32_t n) { return n / 256u + 1u < 2048u ? n / 256u + 1u : 2048u; }
hipError_t launch_assemble(hipStream_t s, const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t base_ntri,
                           const float* translations, uint32_t ninst, float* A, float* B, float* C, uint32_t stride,
                           float* bounds);
hipError_t launch_render_brute(hipStream_t s, const FrameK& f, const DeviceScene& sc, const AoK& ao, const Outputs& out);
// Device memory for the bins of a frame's binning pass: `nbins` counters, then `nbins` lists of 256 entries.  The
// counters are zero between draws (whoever allocates the memory clears all of it once; every traced block leaves
// its counter at zero again).  One BinMem serves one stream at a time.
struct BinMem {
    void* base = nullptr;
    uint64_t nbins = 0;
};
size_t bin_mem_bytes(uint64_t nbins);
// Bins the draw of this frame would use (0: it starts from the scene's coarse cut and needs none).  `batch`: the
// frame shares a launch with others.
uint32_t tile_bins_wanted(const FrameK& f, const DeviceScene& sc, const Outputs& out, bool batch);
// `bins` may be null or too small: the draw then starts from the coarse cut (same pixels either way).
// `vis`: a visibility buffer of width * height 64-bit entries, all equal to the miss key (launch_vis_fill; the
// kernels leave it that way), for frames drawn one lane per triangle (scatter_wanted); null: such frames go through
// the tile kernel like any other.
bool scatter_wanted(const FrameK& f, const DeviceScene& sc, const Outputs& out);
// Whether the resolve pass of such frames assembles its winners again from a mesh that the caches hold (always,
// unless INV_MESH_DRAW=0 in the environment): the build then leaves `slot_of` out, which only that pass would read.
bool mesh_draw_enabled();
hipError_t launch_vis_fill(hipStream_t s, unsigned long long* vis, uint64_t n);
hipError_t launch_render_bvh(hipStream_t s, const FrameK& f, const DeviceScene& sc, const AoK& ao, const Outputs& out,
                             const BinMem* bins, unsigned long long* vis);
// Several frames of one scene in one launch of the tile kernel (at most INV_BATCH_MAX; every frame must be
// tile_batchable: LBVH scene, default traversal, not the dense variant; no occlusion pass).
#define INV_BATCH_MAX 8
bool tile_batchable(const FrameK& f, const DeviceScene& sc, const Outputs& out);
int tile_mode_for(const FrameK& f, const DeviceScene& sc, const Outputs& out, bool batch);
constexpr uint32_t kStackDepth = 40; // per-lane stack entries of the traversal kernels (a lane whose stack is full tests the far subtree leaf by leaf)
hipError_t launch_render_batch(hipStream_t s, const FrameK* frames, const Outputs* outs, uint32_t n,
                               const DeviceScene& sc, const BinMem* bins);
// Fills `wnodes` (64 bytes per internal node of `sc`) from sc.nodes.
hipError_t launch_make_wnodes(hipStream_t s, const DeviceScene& sc, uint32_t* wnodes);
// Occlusion pass over the records the primary kernel queued (ao.rays > 0).
// `narrow`: walk the two-wide records even if the wide ones exist.
hipError_t launch_ao(hipStream_t s, const FrameK& f, const DeviceScene& sc, const AoK& ao, const Outputs& out, bool narrow = false);
hipError_t launch_deinterleave_sparse(hipStream_t s, const FrameK& f, const uint32_t* gathered, uint64_t rank_stride,
                                      uint64_t plane_stride, uint32_t nplanes, uint32_t* frame,
                                      uint64_t frame_plane_stride);
hipError_t launch_deinterleave(hipStream_t s, const FrameK& f, const uint32_t* gathered, uint64_t rank_stride,
                               uint32_t* frame);
// LBVH build over the assembled triangles (48-byte records: the sort's gather then pulls one or two cache lines per
// triangle instead of three): fills sc.triA/B/C (Morton order) and the tree arrays from
// `keep`, temporaries from `scratch` (sizes from the two functions below).  Synchronises the stream once at the end.
size_t lbvh_keep_bytes(uint32_t ntri);
size_t lbvh_scratch_bytes(uint32_t ntri);
// `bounds`: what launch_assemble left for these records, in memory from lbvh_take_bounds (part of the scratch bytes).
float* lbvh_take_bounds(Arena& scratch, uint32_t n);
// `sort_hint` (or null): where in the build's list of sorts, shortest first, this build starts and where it ended
// (0 to begin with; the caller keeps it from build to build).
// `mesh`: what the records were assembled from, when that is small enough to stay in every L2 (the Morton keys and
// the sorted triangle arrays are then made from it instead of from the records, and `records` may be null: the
// assembled records are neither written nor read), or a MeshSource without arrays.
hipError_t build_lbvh(hipStream_t s, DeviceScene& sc, Arena& keep, Arena& scratch, const float* records, const float* bounds,
                      const MeshSource& mesh, int* sort_hint);

} // namespace inv
