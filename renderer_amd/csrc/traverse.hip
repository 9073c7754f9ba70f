// LBVH traversal render kernel for gfx950.
//
// One lane per pixel, one wave per 8x8 pixel block (4 waves = a 16x16 block per workgroup) so the
// lanes of a wave walk nearly the same nodes.  Each lane keeps the far child of a two-way hit on a
// short stack in LDS (one column per lane, so lanes never share a bank).  Nodes carry both child
// boxes, so one 64-byte fetch decides two subtrees.
//
// Culling is conservative: boxes are grown by twice the guard pad and the slab comparison carries
// a relative slack, so every triangle that can pass the Moller-Trumbore test (whose hit point is
// required to lie in the triangle's padded box) is visited.  The winner is the minimum
// (depth, primitive) key, so the visiting order cannot change the image.
#include "internal.h"
#include "rt_core.h"

namespace inv {

namespace {

constexpr int kStack = 40; // entries per lane; deeper trees use the brute-force kernel

struct BlockMap {
    uint32_t px, py;
    uint64_t out; // index into the output buffers
    bool valid;
};

// Workgroup b covers a 16x16 pixel block; wave w its (w&1, w>>1) 8x8 quadrant.
__device__ __forceinline__ BlockMap map_block(const FrameK& f, uint32_t b, uint32_t tid) {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t lx = ((wave & 1u) << 3) | (lane & 7u), ly = ((wave >> 1) << 3) | (lane >> 3);
    BlockMap m;
    if (f.nranks <= 1) {
        const uint32_t bx = (f.w + 15u) >> 4;
        m.px = (b % bx) * 16u + lx;
        m.py = (b / bx) * 16u + ly;
        m.valid = m.px < f.w && m.py < f.h;
        m.out = (uint64_t)m.py * f.w + m.px;
    } else {
        const uint32_t per = f.tile >> 4, bpt = per * per;
        const uint32_t slot = b / bpt, sub = b % bpt;
        const uint32_t id = slot * f.nranks + f.rank;
        const uint32_t tx = (sub % per) * 16u + lx, ty = (sub / per) * 16u + ly;
        m.px = (id % f.tiles_x) * f.tile + tx;
        m.py = (id / f.tiles_x) * f.tile + ty;
        m.valid = id < f.ntiles && m.px < f.w && m.py < f.h;
        m.out = (uint64_t)slot * f.tile * f.tile + (uint64_t)ty * f.tile + tx;
    }
    return m;
}

__device__ __forceinline__ bool slab(const Ray& r, float ix, float iy, float iz, float lx, float ly, float lz, float hx,
                                     float hy, float hz, float grow, float tmax, float& tnear) {
    float ax = ((lx - grow) - r.ox) * ix, bx = ((hx + grow) - r.ox) * ix;
    float ay = ((ly - grow) - r.oy) * iy, by = ((hy + grow) - r.oy) * iy;
    float az = ((lz - grow) - r.oz) * iz, bz = ((hz + grow) - r.oz) * iz;
    // fminf/fmaxf drop NaNs (0 * inf), which keeps the test conservative
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = t0;
    return t0 * 0.9999f <= t1 * 1.0001f;
}

__device__ __forceinline__ void trace(const FrameK& f, const Ray& r, const float4* __restrict__ TA,
                                      const float4* __restrict__ TB, const float4* __restrict__ TC,
                                      const float4* __restrict__ nodes, uint32_t* stack /* column of this lane */,
                                      Hit& best) {
    const float ix = 1.0f / r.dx, iy = 1.0f / r.dy, iz = 1.0f / r.dz;
    const float grow = 2.0f * f.pad;
    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        const float4 q0 = nodes[4 * (size_t)node + 0], q1 = nodes[4 * (size_t)node + 1];
        const float4 q2 = nodes[4 * (size_t)node + 2], q3 = nodes[4 * (size_t)node + 3];
        const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
        const float tmax = best.t * 1.0001f; // INFINITY until something is hit
        float tn0, tn1;
        bool h0 = slab(r, ix, iy, iz, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, grow, tmax, tn0);
        bool h1 = slab(r, ix, iy, iz, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, grow, tmax, tn1);
        if (h0 && (c0 & 0x80000000u)) {
            const uint32_t k = c0 & 0x7FFFFFFFu;
            mt_test(f, r, TA[k], TB[k], TC[k], k, best);
            h0 = false;
        }
        if (h1 && (c1 & 0x80000000u)) {
            const uint32_t k = c1 & 0x7FFFFFFFu;
            mt_test(f, r, TA[k], TB[k], TC[k], k, best);
            h1 = false;
        }
        if (h0 && h1) {
            const bool first1 = tn1 < tn0;
            stack[(size_t)sp * 256] = first1 ? c0 : c1; // far child waits
            sp++;
            node = first1 ? c1 : c0;
        } else if (h0) {
            node = c0;
        } else if (h1) {
            node = c1;
        } else {
            if (sp == 0) break;
            sp--;
            node = stack[(size_t)sp * 256];
        }
    }
}

__global__ __launch_bounds__(256) void k_render_bvh(FrameK f, const float4* __restrict__ TA,
                                                    const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                    const float4* __restrict__ nodes, Outputs out) {
    __shared__ uint32_t s_stack[kStack * 256];
    const BlockMap m = map_block(f, blockIdx.x, threadIdx.x);
    if (!m.valid) return;
    const uint32_t pix = m.py * f.w + m.px;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    Hit first = miss_hit();
    for (uint32_t s = 0; s < f.spp; s++) {
        float sx, sy;
        sample_offset(pix, s, f.seed, sx, sy);
        const Ray ray = make_ray(f, m.px, m.py, sx, sy);
        Hit best = miss_hit();
        trace(f, ray, TA, TB, TC, nodes, s_stack + threadIdx.x, best);
        if (s == 0) first = best;
        float r, g, b;
        shade_sample(f, best, TA, TB, TC, r, g, b);
        sr = sr + r, sg = sg + g, sb = sb + b;
    }
    const float n = (float)f.spp;
    if (out.depth) out.depth[m.out] = __uint_as_float((uint32_t)(first.key >> 32));
    if (out.primid) out.primid[m.out] = (uint32_t)first.key;
    if (out.color) ((uint32_t*)out.color)[m.out] = pack_pixel(f.flags, sr / n, sg / n, sb / n);
}

} // namespace

hipError_t launch_render_bvh(hipStream_t s, const FrameK& f, const DeviceScene& sc, const Outputs& out) {
    if (f.npix == 0) return hipSuccess;
    if (!sc.has_bvh || sc.height > (uint32_t)kStack) return launch_render_brute(s, f, sc, out);
    uint32_t blocks;
    if (f.nranks <= 1) {
        blocks = ((f.w + 15u) >> 4) * ((f.h + 15u) >> 4);
    } else {
        const uint32_t per = f.tile >> 4;
        blocks = f.slots * per * per;
    }
    hipLaunchKernelGGL(k_render_bvh, dim3(blocks), dim3(256), 0, s, f, (const float4*)sc.triA, (const float4*)sc.triB,
                       (const float4*)sc.triC, (const float4*)sc.nodes, out);
    return hipGetLastError();
}

} // namespace inv
