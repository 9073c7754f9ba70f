// HIP kernels for gfx950: vertex stage, primitive assembly, brute-force render and
// tile de-interleave.  The LBVH build and the traversal render live in lbvh.hip and
// traverse.hip.
#include "internal.h"
#include "rt_core.h"

namespace inv {

// ---------------------------------------------------------------------------------------------
// vertexShader (src/shader.slang:22-29): clip = P*(V*(M*(pos,1))) as three successive products.
// ---------------------------------------------------------------------------------------------
struct Mats {
    float m[16], v[16], p[16];
};

__device__ __forceinline__ float4 mat_vec(const float* m, float4 x) {
    float4 r;
    r.x = ((m[0] * x.x + m[4] * x.y) + m[8] * x.z) + m[12] * x.w;
    r.y = ((m[1] * x.x + m[5] * x.y) + m[9] * x.z) + m[13] * x.w;
    r.z = ((m[2] * x.x + m[6] * x.y) + m[10] * x.z) + m[14] * x.w;
    r.w = ((m[3] * x.x + m[7] * x.y) + m[11] * x.z) + m[15] * x.w;
    return r;
}

__global__ __launch_bounds__(256) void k_vertex_shader(const float4* __restrict__ xyzw, uint32_t nv, Mats mats,
                                                       float4* __restrict__ clip, float* __restrict__ color3) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    float4 p = xyzw[i];
    p.w = 1.0f; // float4(input.pos, 1.0f): lane 3 of the vertex buffer is ignored
    float4 c = mat_vec(mats.p, mat_vec(mats.v, mat_vec(mats.m, p)));
    if (clip) clip[i] = c;
    if (color3) {
        color3[3 * (size_t)i + 0] = 1.0f;
        color3[3 * (size_t)i + 1] = 1.0f;
        color3[3 * (size_t)i + 2] = p.z * 10.0f;
    }
}

hipError_t launch_vertex_shader(hipStream_t s, const float* xyzw, uint32_t nv, const inv_camera& cam, float* clip4,
                                float* color3) {
    if (nv == 0) return hipSuccess;
    Mats mats;
    for (int i = 0; i < 16; i++) mats.m[i] = cam.model[i], mats.v[i] = cam.view[i], mats.p[i] = cam.proj[i];
    hipLaunchKernelGGL(k_vertex_shader, dim3((nv + 255) / 256), dim3(256), 0, s, (const float4*)xyzw, nv, mats,
                       (float4*)clip4, color3);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Primitive assembly (VK_PRIMITIVE_TOPOLOGY_TRIANGLE_LIST, src/invert.cpp:1307-1310): triangle k
// of instance j = vertices idx[3k..3k+2] (+ translation j) -> A/B/C records, primitive id
// j*base_ntri + k.
// ---------------------------------------------------------------------------------------------
// With `bounds`, every workgroup also leaves the box of its triangles there (lo xyz, hi xyz: the first stage of the
// scene bounds the tree build starts from), taken from the records as stored -- a corner is A + B, not the vertex
// it was made from -- so that the build does not have to read them all again for it.
__global__ __launch_bounds__(256) void k_assemble(MeshSource m, float4* __restrict__ A, float4* __restrict__ B,
                                                  float4* __restrict__ C, uint32_t stride, float* __restrict__ bounds) {
    const uint64_t total = (uint64_t)m.base_ntri * m.ninst;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        float4 ra, rb, rc;
        assemble_record(m, g, ra, rb, rc);
        const uint64_t at = g * stride;
        if (A) A[at] = ra, B[at] = rb, C[at] = rc; // (no arrays: only the bounds are wanted, see launch_assemble)
        if (bounds) {
            float l[3], h[3];
            tri_box(ra, rb, rc, l, h);
#pragma unroll
            for (int x = 0; x < 3; x++) lo[x] = fminf(lo[x], l[x]), hi[x] = fmaxf(hi[x], h[x]);
        }
    }
    if (bounds) block_bounds<256>(lo, hi, bounds + 6 * (size_t)blockIdx.x); // (block-uniform)
}

hipError_t launch_assemble(hipStream_t s, const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t base_ntri,
                           const float* translations, uint32_t ninst, float* A, float* B, float* C, uint32_t stride,
                           float* bounds) {
    (void)nv;
    uint64_t total = (uint64_t)base_ntri * ninst;
    if (total == 0) return hipSuccess;
    // (with bounds: as many workgroups as the bounds have first-stage entries; without, one triangle per lane)
    const unsigned blocks = bounds ? bounds_blocks_for((uint32_t)total) : (unsigned)((total + 255) / 256);
    const MeshSource m = mesh_source((const float4*)xyzw, idx, translations, base_ntri, ninst);
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, s, m, (float4*)A, (float4*)B, (float4*)C, stride, bounds);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Brute force: one lane per pixel, every triangle tested (INV_BRUTE_FORCE).  The triangle index is
// wave-uniform, so the records come in through scalar loads.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_render_brute(FrameK f, const float4* __restrict__ TA,
                                                      const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                      uint32_t ntri, AoK ao, Outputs out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.npix) return;
    uint32_t px, py;
    if (!work_to_pixel(f, i, px, py)) return;
    uint32_t pix = py * f.w + px;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    Hit first = miss_hit();
    Ray ray0 = make_ray(f, px, py, 0.5f, 0.5f);
    for (uint32_t s = 0; s < f.spp; s++) {
        float sx, sy;
        sample_offset(pix, s, f.seed, sx, sy);
        Ray ray = make_ray(f, px, py, sx, sy);
        Hit best = miss_hit();
        for (uint32_t k = 0; k < ntri; k++) mt_test(f, ray, TA[k], TB[k], TC[k], k, best);
        if (s == 0) first = best;
        float r, g, b;
        shade_sample(f, best, TA, TB, TC, r, g, b);
        sr = sr + r, sg = sg + g, sb = sb + b;
    }
    const uint32_t ws = first.key != kMissKey ? first.slot : 0u;
    finish_pixel(f, ao, out, i, pix, ray0, first, TB[ws], TC[ws], sr, sg, sb);
}

hipError_t launch_render_brute(hipStream_t s, const FrameK& f, const DeviceScene& sc, const AoK& ao,
                               const Outputs& out) {
    if (f.npix == 0) return hipSuccess;
    uint64_t blocks = (f.npix + 255) / 256;
    hipLaunchKernelGGL(k_render_brute, dim3((unsigned)blocks), dim3(256), 0, s, f, (const float4*)sc.triA,
                       (const float4*)sc.triB, (const float4*)sc.triC, sc.ntri, ao, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Rank 0 after the gather: tile-major buffers of all ranks -> one row-major frame.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deinterleave(FrameK f, const uint32_t* __restrict__ gathered,
                                                      uint64_t rank_stride, uint32_t* __restrict__ frame) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)f.w * f.h) return;
    uint32_t px = (uint32_t)(i % f.w), py = (uint32_t)(i / f.w);
    uint32_t id = (py / f.tile) * f.tiles_x + px / f.tile;
    uint32_t r = id % f.nranks + f.first_rank, slot = id / f.nranks;
    uint64_t src = (uint64_t)r * rank_stride + (uint64_t)slot * (f.tile * f.tile) + (py % f.tile) * f.tile + px % f.tile;
    frame[i] = gathered[src];
}

// Sparse composite: only the grid tiles dealt to ranks 1.. are copied; rank 0 rendered the rest in place.
__global__ __launch_bounds__(256) void k_deinterleave_sparse(FrameK f, const uint32_t* __restrict__ gathered,
                                                             uint64_t rank_stride, uint64_t plane_stride,
                                                             uint32_t nplanes, uint32_t* __restrict__ frame,
                                                             uint64_t frame_plane_stride) {
    const uint32_t tt = f.tile * f.tile;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)f.ntiles * tt) return;
    const uint32_t id = (uint32_t)(i / tt), rem = (uint32_t)(i % tt);
    const uint32_t r = id % f.nranks + f.first_rank;
    if (r == 0) return;
    const uint32_t px = (f.gx0 + id % f.tiles_x) * f.tile + rem % f.tile;
    const uint32_t py = (f.gy0 + id / f.tiles_x) * f.tile + rem / f.tile;
    if (px >= f.w || py >= f.h) return;
    const uint64_t src = (uint64_t)r * rank_stride + (uint64_t)(id / f.nranks) * tt + rem;
    const uint64_t dst = (uint64_t)py * f.w + px;
    for (uint32_t k = 0; k < nplanes; k++) frame[k * frame_plane_stride + dst] = gathered[src + k * plane_stride];
}

hipError_t launch_deinterleave_sparse(hipStream_t s, const FrameK& f, const uint32_t* gathered, uint64_t rank_stride,
                                      uint64_t plane_stride, uint32_t nplanes, uint32_t* frame,
                                      uint64_t frame_plane_stride) {
    const uint64_t n = (uint64_t)f.ntiles * f.tile * f.tile;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_deinterleave_sparse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f, gathered,
                       rank_stride, plane_stride, nplanes, frame, frame_plane_stride);
    return hipGetLastError();
}

hipError_t launch_deinterleave(hipStream_t s, const FrameK& f, const uint32_t* gathered, uint64_t rank_stride,
                               uint32_t* frame) {
    uint64_t n = (uint64_t)f.w * f.h;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f, gathered, rank_stride, frame);
    return hipGetLastError();
}

} // namespace inv
