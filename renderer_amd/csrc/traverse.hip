// LBVH render kernels for gfx950.
//
// k_render_tile (primary rays, the default).  The grid has two parts.
//   Trace part: one workgroup of four waves per 8x8 pixel tile of the rectangle that can contain hits
//   (the projected scene bounds).
//     1. All 256 lanes walk the LBVH breadth-first, one node per lane, against the four side planes of
//        the tile's frustum, starting from a precomputed cut a few levels below the root; subtrees of at
//        most kRangeMax leaves are taken whole as a range of the sorted triangle array.  Frontier and
//        range lists live in LDS.
//     2. One candidate triangle per lane: its own box and its facing against the tile's frustum; the
//        survivors' records are compacted into LDS.
//     3. Every wave runs the Moller-Trumbore test of a quarter of the records for all 64 pixels (records
//        come out of LDS as broadcast reads); a 64-bit LDS min per pixel picks the winning
//        (depth, primitive) key and the lane that holds it shades and stores.
//   If a list overflows, the first wave falls back to the per-lane stack traversal below.
//   Fill part: 32x8 pixel blocks over the whole frame store the clear values outside the traced tiles,
//   two full 128-byte lines per wave and attachment.
//
// k_render_bvh (INV_LANE_TRAVERSAL, and the fallback)
//   One lane per pixel with a short stack in LDS (one column per lane).
//
// Both cull conservatively (boxes grown by twice the guard pad, frustum grown by 1/16 pixel, a
// slack on every comparison): every triangle that can pass the Moller-Trumbore test, whose hit point
// must lie in the triangle's padded box, reaches that test.  The winner is the minimum
// (depth, primitive) key, so neither the visiting order nor the superset of candidates changes a pixel.
#include "internal.h"
#include "rt_core.h"

namespace inv {

namespace {

constexpr int kStack = 40;        // per-lane stack entries; deeper trees use the brute-force kernel
constexpr int kFrontierCap = 768;  // nodes per breadth-first level kept in LDS
constexpr int kRangeCap = 512;     // leaf ranges per tile
constexpr int kTileCap = 256;      // candidate triangle records per tile
constexpr uint32_t kRangeMax = 16; // subtrees of at most this many leaves are not descended

struct BlockMap {
    uint32_t px, py;
    uint64_t out; // index into the output buffers
    bool valid;
};

// origin pixel of a 32x8 block and of its output run
struct BlockOrigin {
    uint32_t x0, y0;   // top-left pixel in the frame
    uint64_t out0;     // output index of that pixel
    uint32_t pitch;    // output elements per pixel row
    bool live;         // false for padding slots of a sharded frame
};

__device__ __forceinline__ BlockOrigin block_origin(const FrameK& f, uint32_t b) {
    BlockOrigin o;
    if (f.nranks <= 1) {
        const uint32_t bx = (f.w + 31u) >> 5;
        o.x0 = (b % bx) * 32u, o.y0 = (b / bx) * 8u;
        o.out0 = (uint64_t)o.y0 * f.w + o.x0;
        o.pitch = f.w;
        o.live = true;
    } else {
        const uint32_t perx = f.tile >> 5, pery = f.tile >> 3, bpt = perx * pery;
        const uint32_t slot = b / bpt, sub = b % bpt;
        const uint32_t id = slot * f.nranks + f.rank;
        const uint32_t tx = (sub % perx) * 32u, ty = (sub / perx) * 8u;
        o.x0 = (id % f.tiles_x) * f.tile + tx;
        o.y0 = (id / f.tiles_x) * f.tile + ty;
        o.out0 = (uint64_t)slot * f.tile * f.tile + (uint64_t)ty * f.tile + tx;
        o.pitch = f.tile;
        o.live = id < f.ntiles;
    }
    return o;
}

// Workgroup b covers a 16x16 pixel block; wave w its (w&1, w>>1) 8x8 quadrant.
__device__ __forceinline__ BlockMap map_block16(const FrameK& f, uint32_t b, uint32_t tid) {
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
                                      uint32_t stride /* lanes sharing the stack array */, Hit& best) {
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
            stack[(size_t)sp * stride] = first1 ? c0 : c1; // far child waits
            sp++;
            node = first1 ? c1 : c0;
        } else if (h0) {
            node = c0;
        } else if (h1) {
            node = c1;
        } else {
            if (sp == 0) break;
            sp--;
            node = stack[(size_t)sp * stride];
        }
    }
}

// all samples of one pixel through the per-lane traversal, then resolve and store
__device__ __forceinline__ void pixel_by_traversal(const FrameK& f, uint32_t px, uint32_t py, uint64_t out_index,
                                                   const float4* __restrict__ TA, const float4* __restrict__ TB,
                                                   const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                                   uint32_t* stack, uint32_t stride, const Outputs& out) {
    const uint32_t pix = py * f.w + px;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    Hit first = miss_hit();
    for (uint32_t s = 0; s < f.spp; s++) {
        float sx, sy;
        sample_offset(pix, s, f.seed, sx, sy);
        const Ray ray = make_ray(f, px, py, sx, sy);
        Hit best = miss_hit();
        trace(f, ray, TA, TB, TC, nodes, stack, stride, best);
        if (s == 0) first = best;
        float r, g, b;
        shade_sample(f, best, TA, TB, TC, r, g, b);
        sr = sr + r, sg = sg + g, sb = sb + b;
    }
    const float n = (float)f.spp;
    if (out.depth) out.depth[out_index] = __uint_as_float((uint32_t)(first.key >> 32));
    if (out.primid) out.primid[out_index] = (uint32_t)first.key;
    if (out.color) ((uint32_t*)out.color)[out_index] = pack_pixel(f.flags, sr / n, sg / n, sb / n);
}

__global__ __launch_bounds__(256) void k_render_bvh(FrameK f, const float4* __restrict__ TA,
                                                    const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                    const float4* __restrict__ nodes, Outputs out) {
    __shared__ uint32_t s_stack[kStack * 256];
    const BlockMap m = map_block16(f, blockIdx.x, threadIdx.x);
    if (!m.valid) return;
    pixel_by_traversal(f, m.px, m.py, m.out, TA, TB, TC, nodes, s_stack + threadIdx.x, 256, out);
}

// ------------------------------------------------------------------------------------------------
// tile kernel
// ------------------------------------------------------------------------------------------------
struct Plane {
    float a, b, c, d; // a*x + b*y + c*z + d >= 0 inside
};

struct Frustum {
    Plane p[4];
};

// Side planes, in object space, of the pyramid of rays through framebuffer rectangle
// [x0 - m, x1 + m] x [y0 - m, y1 + m] (pixels).  With clip = P * (V*M) * p and w_clip = -z_view:
//   x_ndc >= xl  <=>  P00 * x_view + xl * z_view >= 0, and so on for the other three sides.
__device__ __forceinline__ Frustum make_frustum(const FrameK& f, float x0, float y0, float x1, float y1) {
    const float m = 0.0625f;
    const float xl = (2.0f * (x0 - m)) / f.wf - 1.0f, xr = (2.0f * (x1 + m)) / f.wf - 1.0f;
    const float yl = (2.0f * (y0 - m)) / f.hf - 1.0f, yr = (2.0f * (y1 + m)) / f.hf - 1.0f;
    const float* r0 = f.mv;
    const float* r1 = f.mv + 4;
    const float* r2 = f.mv + 8;
    Frustum fr;
    fr.p[0] = {f.p00 * r0[0] + xl * r2[0], f.p00 * r0[1] + xl * r2[1], f.p00 * r0[2] + xl * r2[2], f.p00 * r0[3] + xl * r2[3]};
    fr.p[1] = {-(f.p00 * r0[0] + xr * r2[0]), -(f.p00 * r0[1] + xr * r2[1]), -(f.p00 * r0[2] + xr * r2[2]), -(f.p00 * r0[3] + xr * r2[3])};
    fr.p[2] = {f.p11 * r1[0] + yl * r2[0], f.p11 * r1[1] + yl * r2[1], f.p11 * r1[2] + yl * r2[2], f.p11 * r1[3] + yl * r2[3]};
    fr.p[3] = {-(f.p11 * r1[0] + yr * r2[0]), -(f.p11 * r1[1] + yr * r2[1]), -(f.p11 * r1[2] + yr * r2[2]), -(f.p11 * r1[3] + yr * r2[3])};
    return fr;
}

// false only if the box, grown by `grow`, lies wholly outside one side plane
__device__ __forceinline__ bool box_in_frustum(const Frustum& fr, float lx, float ly, float lz, float hx, float hy,
                                               float hz, float grow) {
    lx -= grow, ly -= grow, lz -= grow, hx += grow, hy += grow, hz += grow;
    bool in = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const Plane& p = fr.p[i];
        // largest value of the plane function over the box
        const float v = (p.a * (p.a >= 0.0f ? hx : lx) + p.b * (p.b >= 0.0f ? hy : ly)) + (p.c * (p.c >= 0.0f ? hz : lz) + p.d);
        in = in && !(v < 0.0f);
    }
    return in;
}

// direction through framebuffer position (x, y), the same arithmetic as make_ray
__device__ __forceinline__ void dir_at(const FrameK& f, float x, float y, float& dx, float& dy, float& dz) {
    const float xn = (2.0f * x) / f.wf - 1.0f, yn = (2.0f * y) / f.hf - 1.0f;
    const float dxv = xn / f.p00, dyv = yn / f.p11;
    dx = (f.d0x * dxv + f.d1x * dyv) - f.d2x;
    dy = (f.d0y * dxv + f.d1y * dyv) - f.d2y;
    dz = (f.d0z * dxv + f.d1z * dyv) - f.d2z;
}

// output index of frame pixel (px, py); false if another rank owns it
__device__ __forceinline__ bool out_index(const FrameK& f, uint32_t px, uint32_t py, uint64_t& idx) {
    if (f.nranks <= 1) {
        idx = (uint64_t)py * f.w + px;
        return true;
    }
    const uint32_t id = (py / f.tile) * f.tiles_x + px / f.tile;
    idx = (uint64_t)(id / f.nranks) * f.tile * f.tile + (uint64_t)(py % f.tile) * f.tile + px % f.tile;
    return id % f.nranks == f.rank;
}

struct TileLds {
    union {
        struct {
            uint32_t frontier[2][kFrontierCap];
            uint2 ranges[kRangeCap];      // (first sorted triangle, count <= kRangeMax)
            float4 rec[kTileCap][3];      // candidate records A, B, C that survived the per-triangle test
        } t;
        uint32_t stack[kStack * 64]; // fallback: per-lane traversal by the first wave
    } u;
    unsigned long long best[64]; // per pixel: smallest (depth, primitive) key of the current sample
    float acc[64][3];            // per pixel colour sum over samples
    uint32_t n_front[2];
    uint32_t n_ranges;
    uint32_t n_rec;
    uint32_t overflow;
};

// The trace part of the grid: one workgroup (4 waves) per 8x8 pixel tile of the rectangle that can contain hits.
__device__ __forceinline__ void trace_tile(const FrameK& f, TileLds& S, uint32_t tx0, uint32_t ty0,
                                           const float4* __restrict__ TA, const float4* __restrict__ TB,
                                           const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                           const uint32_t* __restrict__ cut, const Outputs& out) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const float grow = 2.0f * f.pad;
    const Frustum fr = make_frustum(f, (float)tx0, (float)ty0, (float)(tx0 + 8u), (float)(ty0 + 8u));

    // ---- phase 1: breadth-first descent from the cut, one node per lane ----
    const uint32_t ncut = cut[0], ncut_leaf = cut[1];
    if (tid == 0) S.n_front[0] = ncut, S.n_front[1] = 0, S.n_ranges = ncut_leaf, S.n_rec = 0, S.overflow = 0;
    for (uint32_t i = tid; i < ncut; i += 256) S.u.t.frontier[0][i] = cut[2 + i];
    for (uint32_t i = tid; i < ncut_leaf; i += 256) S.u.t.ranges[i] = make_uint2(cut[2 + kCutNodes + i], 1u);
    if (tid < 64) {
        S.best[tid] = kMissKey;
        S.acc[tid][0] = 0.0f, S.acc[tid][1] = 0.0f, S.acc[tid][2] = 0.0f;
    }
    __syncthreads();
    int cur = 0;
    for (;;) {
        const uint32_t n = S.n_front[cur];
        if (n == 0 || S.overflow) break;
        for (uint32_t base = 0; base < n; base += 256) {
            const uint32_t q = base + tid;
            if (q < n) {
                const uint32_t node = S.u.t.frontier[cur][q];
                const float4 q0 = nodes[4 * (size_t)node + 0], q1 = nodes[4 * (size_t)node + 1];
                const float4 q2 = nodes[4 * (size_t)node + 2], q3 = nodes[4 * (size_t)node + 3];
                const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                const uint32_t lo = __float_as_uint(q3.z), hi = __float_as_uint(q3.w);
                const uint32_t gamma = c0 & 0x7FFFFFFFu;
                const bool in0 = box_in_frustum(fr, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, grow);
                const bool in1 = box_in_frustum(fr, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, grow);
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    if (!(side ? in1 : in0)) continue;
                    const uint32_t c = side ? c1 : c0;
                    const uint32_t first = side ? gamma + 1u : lo;
                    const uint32_t count = side ? hi - gamma : gamma - lo + 1u;
                    if ((c & 0x80000000u) || count <= kRangeMax) {
                        const uint32_t at = atomicAdd(&S.n_ranges, 1u);
                        if (at < (uint32_t)kRangeCap)
                            S.u.t.ranges[at] = make_uint2(first, count);
                        else
                            S.overflow = 1;
                    } else {
                        const uint32_t at = atomicAdd(&S.n_front[cur ^ 1], 1u);
                        if (at < (uint32_t)kFrontierCap)
                            S.u.t.frontier[cur ^ 1][at] = c;
                        else
                            S.overflow = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) S.n_front[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
    __syncthreads();

    // ---- phase 2: one candidate triangle per lane against the tile's frustum and facing ----
    const uint32_t nranges = S.n_ranges;
    if (!S.overflow) {
        // directions through the four corners of the grown tile (for the facing test)
        float cdx[4], cdy[4], cdz[4];
        const float m = 0.0625f;
        dir_at(f, (float)tx0 - m, (float)ty0 - m, cdx[0], cdy[0], cdz[0]);
        dir_at(f, (float)(tx0 + 8u) + m, (float)ty0 - m, cdx[1], cdy[1], cdz[1]);
        dir_at(f, (float)tx0 - m, (float)(ty0 + 8u) + m, cdx[2], cdy[2], cdz[2]);
        dir_at(f, (float)(tx0 + 8u) + m, (float)(ty0 + 8u) + m, cdx[3], cdy[3], cdz[3]);
        const uint32_t nslots = nranges * kRangeMax;
        for (uint32_t base = 0; base < nslots; base += 256) { // block-uniform trip count
            const uint32_t q = base + tid;
            bool keep = false;
            float4 A, B, C;
            if (q < nslots) {
                const uint2 rg = S.u.t.ranges[q / kRangeMax];
                const uint32_t off = q % kRangeMax;
                if (off < rg.y) {
                    const uint32_t k = rg.x + off;
                    A = TA[k], B = TB[k], C = TC[k];
                    const float bx = A.x + B.x, by = A.y + B.y, bz = A.z + B.z;
                    const float cx = A.x + C.x, cy = A.y + C.y, cz = A.z + C.z;
                    keep = box_in_frustum(fr, fminf(A.x, fminf(bx, cx)), fminf(A.y, fminf(by, cy)),
                                          fminf(A.z, fminf(bz, cz)), fmaxf(A.x, fmaxf(bx, cx)), fmaxf(A.y, fmaxf(by, cy)),
                                          fmaxf(A.z, fmaxf(bz, cz)), grow);
                    // det = e1 . (d x e2) is linear in d and every sample direction of the tile is a convex
                    // combination of the corner directions: if det is clearly negative at all four corners the
                    // triangle faces away from every ray of the tile (cull BACK, src/invert.cpp:1334).
                    const float nx = C.y * B.z - C.z * B.y, ny = C.z * B.x - C.x * B.z, nz = C.x * B.y - C.y * B.x;
                    float dmax = -INFINITY, scale = 0.0f;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        dmax = fmaxf(dmax, (cdx[c] * nx + cdy[c] * ny) + cdz[c] * nz);
                        scale = fmaxf(scale, (fabsf(cdx[c] * nx) + fabsf(cdy[c] * ny)) + fabsf(cdz[c] * nz));
                    }
                    keep = keep && !(dmax < -1e-4f * scale);
                }
            }
            const uint64_t mask = __ballot(keep);
            uint32_t wbase = 0;
            if (lane == 0 && mask) wbase = atomicAdd(&S.n_rec, (uint32_t)__popcll(mask));
            wbase = __shfl(wbase, 0, 64);
            const uint32_t at = wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (keep) {
                if (at < (uint32_t)kTileCap) {
                    S.u.t.rec[at][0] = A;
                    S.u.t.rec[at][1] = B;
                    S.u.t.rec[at][2] = C;
                } else {
                    S.overflow = 1;
                }
            }
        }
    }
    __syncthreads();

    const uint32_t px = tx0 + (lane & 7u), py = ty0 + (lane >> 3);
    uint64_t oi = 0;
    const bool valid = px < f.w && py < f.h && out_index(f, px, py, oi);

    if (S.overflow) { // more candidates than the LDS lists hold: the first wave traverses per lane
        if (wave == 0 && valid) pixel_by_traversal(f, px, py, oi, TA, TB, TC, nodes, S.u.stack + lane, 64, out);
        return;
    }

    // ---- phase 3: every wave tests a quarter of the records for all 64 pixels; LDS min picks the winner ----
    const uint32_t nrec = S.n_rec;
    const uint32_t pix = py * f.w + px;
    Hit first = miss_hit();
    bool own_first = wave == 0;
    for (uint32_t s = 0; s < f.spp; s++) {
        float sx, sy;
        sample_offset(pix, s, f.seed, sx, sy);
        const Ray ray = make_ray(f, px, py, sx, sy);
        Hit best = miss_hit();
        for (uint32_t j = wave; j < nrec; j += 4)
            mt_test(f, ray, S.u.t.rec[j][0], S.u.t.rec[j][1], S.u.t.rec[j][2], j, best);
        if (best.key != kMissKey) atomicMin(&S.best[lane], (unsigned long long)best.key);
        __syncthreads();
        const uint64_t win = S.best[lane];
        // exactly one wave holds the winning record of a hit pixel (a triangle is listed once); wave 0 speaks for misses
        const bool mine = (win == kMissKey) ? wave == 0 : best.key == win;
        if (mine) {
            float r = 0.0f, g = 0.0f, b = 0.0f;
            if (win != kMissKey)
                shade_hit(f, best, S.u.t.rec[best.slot][0], S.u.t.rec[best.slot][1], S.u.t.rec[best.slot][2], r, g, b);
            S.acc[lane][0] = S.acc[lane][0] + r;
            S.acc[lane][1] = S.acc[lane][1] + g;
            S.acc[lane][2] = S.acc[lane][2] + b;
            if (s == 0) first = best, own_first = true;
        } else if (s == 0) {
            own_first = false;
        }
        __syncthreads();
        if (wave == 0) S.best[lane] = kMissKey;
        if (s + 1 < f.spp) __syncthreads();
    }
    if (!valid) return;
    if (own_first) { // the lane that won sample 0 stores depth and primitive id
        if (out.depth) out.depth[oi] = __uint_as_float((uint32_t)(first.key >> 32));
        if (out.primid) out.primid[oi] = (uint32_t)first.key;
    }
    if (wave == 0 && out.color) {
        const float n = (float)f.spp;
        ((uint32_t*)out.color)[oi] = pack_pixel(f.flags, S.acc[lane][0] / n, S.acc[lane][1] / n, S.acc[lane][2] / n);
    }
}

// Grid = [8x8 tiles of the hit rectangle] ++ [32x8 blocks of the whole frame].  The second part stores the clear
// values of every pixel whose 8x8 tile is not in the first part: two full 128-byte lines per wave and plane.
__global__ __launch_bounds__(256) void k_render_tile(FrameK f, uint32_t tiles_x, uint32_t ntiles,
                                                     const float4* __restrict__ TA, const float4* __restrict__ TB,
                                                     const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                                     const uint32_t* __restrict__ cut, Outputs out) {
    __shared__ TileLds S;
    if (blockIdx.x < ntiles) {
        const uint32_t tx0 = (f.rx0 & ~7u) + (blockIdx.x % tiles_x) * 8u;
        const uint32_t ty0 = (f.ry0 & ~7u) + (blockIdx.x / tiles_x) * 8u;
        if (f.nranks > 1) { // block-uniform: a tile belongs to one rank
            uint64_t dummy;
            if (!out_index(f, tx0, ty0, dummy)) return;
        }
        trace_tile(f, S, tx0, ty0, TA, TB, TC, nodes, cut, out);
        return;
    }
    const BlockOrigin o = block_origin(f, blockIdx.x - ntiles);
    if (!o.live) return;
    const uint32_t cx = o.x0 + (threadIdx.x & 31u), cy = o.y0 + (threadIdx.x >> 5);
    if (cx >= f.w || cy >= f.h) return;
    const uint32_t hx0 = f.rx0 & ~7u, hy0 = f.ry0 & ~7u, hx1 = hx0 + tiles_x * 8u, hy1 = hy0 + (ntiles / (tiles_x ? tiles_x : 1u)) * 8u;
    if (cx >= hx0 && cx < hx1 && cy >= hy0 && cy < hy1) return; // traced by the first part of the grid
    const uint64_t i = o.out0 + (uint64_t)(threadIdx.x >> 5) * o.pitch + (threadIdx.x & 31u);
    if (out.depth) out.depth[i] = 1.0f;
    if (out.primid) out.primid[i] = 0xFFFFFFFFu;
    if (out.color) ((uint32_t*)out.color)[i] = background_pixel(f.flags);
}

} // namespace

hipError_t launch_render_bvh(hipStream_t s, const FrameK& f, const DeviceScene& sc, const Outputs& out) {
    if (f.npix == 0) return hipSuccess;
    if (!sc.has_bvh || sc.height > (uint32_t)kStack) return launch_render_brute(s, f, sc, out);
    if (f.flags & INV_LANE_TRAVERSAL) {
        uint32_t blocks;
        if (f.nranks <= 1) {
            blocks = ((f.w + 15u) >> 4) * ((f.h + 15u) >> 4);
        } else {
            const uint32_t per = f.tile >> 4;
            blocks = f.slots * per * per;
        }
        hipLaunchKernelGGL(k_render_bvh, dim3(blocks), dim3(256), 0, s, f, (const float4*)sc.triA,
                           (const float4*)sc.triB, (const float4*)sc.triC, (const float4*)sc.nodes, out);
        return hipGetLastError();
    }
    uint32_t fill_blocks;
    if (f.nranks <= 1)
        fill_blocks = ((f.w + 31u) >> 5) * ((f.h + 7u) >> 3);
    else
        fill_blocks = f.slots * (f.tile >> 5) * (f.tile >> 3);
    uint32_t tiles_x = 0, tiles_y = 0;
    if (f.rx1 > f.rx0 && f.ry1 > f.ry0) {
        tiles_x = ((f.rx1 + 7u) >> 3) - (f.rx0 >> 3);
        tiles_y = ((f.ry1 + 7u) >> 3) - (f.ry0 >> 3);
    }
    const uint32_t ntiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(k_render_tile, dim3(ntiles + fill_blocks), dim3(256), 0, s, f, tiles_x, ntiles,
                       (const float4*)sc.triA, (const float4*)sc.triB, (const float4*)sc.triC, (const float4*)sc.nodes,
                       sc.cut, out);
    return hipGetLastError();
}

} // namespace inv
