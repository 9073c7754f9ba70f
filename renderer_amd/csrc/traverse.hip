// LBVH render kernels for gfx950.
//
// k_render_tile / k_render_tile_batch (primary rays, the default).  The grid of a frame has two parts.
//   Trace part: one workgroup of four waves per pixel square of the rectangle that can contain hits (the projected
//   scene bounds), in one of three forms chosen per frame (plan_tile_frame):
//     8x8 tiles (a draw on its own)
//       1. All 256 lanes walk the LBVH breadth-first, one box per lane, against the four side planes of the tile's
//          frustum, starting from a precomputed cut a few levels below the root; subtrees of at most kRangeMax leaves
//          are taken whole as a range of the sorted triangle array.  Frontier and range lists live in LDS.
//       2. One candidate triangle per lane: its own box and its facing against the tile's frustum; the survivors'
//          records are compacted into LDS.
//       3. Every wave runs the Moller-Trumbore test of a quarter of the records for all 64 pixels (records come out of
//          LDS as broadcast reads); a 64-bit LDS min per pixel picks the winning (depth, primitive) key and the lane
//          that holds it shades and stores.  With four or more samples per pixel the waves split the samples instead.
//     16x16 blocks (frames that share a launch: k_render_tile_batch; INV_BLOCKS)
//       Steps 1 and 2 once per 256 pixels, noting for every record which 8x8 quadrants its box can touch; in step 3 a
//       wave per quadrant walks that quadrant's records and shades and stores its own pixels.
//     8x8 tiles, batched (triangles far smaller than a pixel)
//       Step 3 works through the leaf ranges in batches that fit the record list.
//   If a list overflows, the workgroup falls back to the per-lane stack traversal below.
//   Fill part: stores the clear values outside the traced squares, 16 bytes per lane, attachment and store, a run of
//   consecutive chunks per wave, write-through.
//
// k_render_bvh (INV_LANE_TRAVERSAL, and the fallback)
//   One lane per pixel with a short stack in LDS (one column per lane).
//
// All of them cull conservatively (boxes grown by twice the guard pad, frustum grown by 1/16 pixel, a
// slack on every comparison): every triangle that can pass the Moller-Trumbore test, whose hit point
// must lie in the triangle's padded box, reaches that test.  The winner is the minimum
// (depth, primitive) key, so neither the visiting order nor the superset of candidates changes a pixel.
#include <cstdlib>
#include <cstring>

#include "internal.h"
#include "rt_core.h"

namespace inv {

namespace {

// Timing-only switches in the high flag bits (images are wrong when set; used by tools/phase_timing.py and
// tools/tile_stats.py).  They exist only in the library built with -DINV_TIMING_BUILD (make timing): in the release
// build every test below is a compile-time false and the kernels carry neither the branches nor the stamps.
#ifdef INV_TIMING_BUILD
constexpr uint32_t kDebugSkipAfter2 = 1u << 20, kDebugSkipAfter1 = 1u << 21, kDebugSkipAll = 1u << 22, kDebugCounts = 1u << 23, kDebugSkipExact = 1u << 24, kDebugSkipFill = 1u << 25;
#define INV_DBG(f, bits) (((f).flags & (bits)) != 0u)
#else
#define INV_DBG(f, bits) false
#endif

#ifndef INV_AO_PULL
#define INV_AO_PULL 8
#endif
#ifndef INV_AO_LEAF_NUM
#define INV_AO_LEAF_NUM 2
#endif
#ifndef INV_AO_PEND
#define INV_AO_PEND 1
#endif
constexpr bool kAoPend = INV_AO_PEND != 0;
constexpr int kAoPull = INV_AO_PULL;        // idle lanes at which a wave hands out buffered occlusion rays
constexpr int kAoLeafNum = INV_AO_LEAF_NUM; // eighths of the live lanes that must wait at a leaf before a leaf step
constexpr unsigned long long kAoChunk = 64; // occlusion rays a wave sets up at a time (one per lane)
constexpr int kStack = (int)kStackDepth; // per-lane stack entries; a deeper tree makes a lane test a far subtree leaf by leaf
constexpr int kFrontierCap = 256;  // nodes per breadth-first level kept in LDS
constexpr int kRangeCap = 256;     // leaf ranges per tile
// Candidate triangle records per tile (at most 64 per wave: one mask bit each).  The batched variant fills its list
// again and again, so it takes the larger one; the un-batched variant's smaller list leaves LDS for a fifth
// workgroup per CU (tiles with more candidates than that fall back to the per-lane traversal).
constexpr int kTileCapBatched = 256, kTileCapSparse = 128;
constexpr uint32_t kFillRun = 128;
constexpr uint32_t kBatchMax = INV_BATCH_MAX; // frames per launch of the batch kernel // 16-byte chunks a wave of the fill part stores per attachment and step

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
    if (f.rowmajor) {
        const uint32_t bx = (f.w + 31u) >> 5;
        o.x0 = (b % bx) * 32u, o.y0 = (b / bx) * 8u;
        o.out0 = (uint64_t)o.y0 * f.w + o.x0;
        o.pitch = f.w;
        o.live = owns_rowmajor(f, o.x0, o.y0); // a 32x8 block lies in one tile (tile is a multiple of 32)
    } else {
        const uint32_t perx = f.tile >> 5, pery = f.tile >> 3, bpt = perx * pery;
        const uint32_t slot = b / bpt, sub = b % bpt;
        const uint32_t id = slot * f.nranks + f.rank;
        const uint32_t tx = (sub % perx) * 32u, ty = (sub / perx) * 8u;
        o.x0 = (f.gx0 + id % f.tiles_x) * f.tile + tx;
        o.y0 = (f.gy0 + id / f.tiles_x) * f.tile + ty;
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
    if (f.rowmajor) {
        const uint32_t bx = (f.w + 15u) >> 4;
        m.px = (b % bx) * 16u + lx;
        m.py = (b / bx) * 16u + ly;
        m.valid = m.px < f.w && m.py < f.h && owns_rowmajor(f, m.px, m.py);
        m.out = (uint64_t)m.py * f.w + m.px;
    } else {
        const uint32_t per = f.tile >> 4, bpt = per * per;
        const uint32_t slot = b / bpt, sub = b % bpt;
        const uint32_t id = slot * f.nranks + f.rank;
        const uint32_t tx = (sub % per) * 16u + lx, ty = (sub / per) * 16u + ly;
        m.px = (f.gx0 + id % f.tiles_x) * f.tile + tx;
        m.py = (f.gy0 + id / f.tiles_x) * f.tile + ty;
        m.valid = id < f.ntiles && m.px < f.w && m.py < f.h;
        m.out = (uint64_t)slot * f.tile * f.tile + (uint64_t)ty * f.tile + tx;
    }
    return m;
}

// Ray against a box grown by `grow`, with the ray's terms folded once per ray:
// (l - grow - o) * i = l * i - (o + grow) * i, one fused multiply-add per plane.  Box tests only have to be conservative (the hit-point guard decides which triangles
// count), and the fold's rounding, about 2^-24 * |o * i|, is far inside the grow * |i| margin.
struct RaySlabs {
    float ix, iy, iz;    // 1 / direction
    float clx, cly, clz; // (origin + grow) / direction
    float chx, chy, chz; // (origin - grow) / direction
};

// A direction component of exactly zero (the centre ray of a frame with odd width and height runs along the view
// axis) would make the reciprocal infinite, and the folded form then computes inf - inf where the unfolded one has
// (l - o) * inf: a not-a-number that the minimum and maximum drop on the wrong side.  For the box tests such a
// component is taken as 2^-60 with its sign: over any ray length in a scene the ray then leaves its true line by
// less than a rounding error, and every product stays finite with the margin-to-rounding ratio unchanged.
__device__ __forceinline__ float slab_reciprocal(float d) {
    return 1.0f / (fabsf(d) >= 0x1p-60f ? d : copysignf(0x1p-60f, d));
}

__device__ __forceinline__ RaySlabs make_slabs(float ox, float oy, float oz, float dx, float dy, float dz, float grow) {
    RaySlabs s;
    s.ix = slab_reciprocal(dx), s.iy = slab_reciprocal(dy), s.iz = slab_reciprocal(dz);
    s.clx = (ox + grow) * s.ix, s.cly = (oy + grow) * s.iy, s.clz = (oz + grow) * s.iz;
    s.chx = (ox - grow) * s.ix, s.chy = (oy - grow) * s.iy, s.chz = (oz - grow) * s.iz;
    return s;
}

__device__ __forceinline__ bool slab_folded(const RaySlabs& s, float lx, float ly, float lz, float hx, float hy,
                                            float hz, float tmax, float& tnear) {
    const float ax = __builtin_fmaf(lx, s.ix, -s.clx), bx = __builtin_fmaf(hx, s.ix, -s.chx);
    const float ay = __builtin_fmaf(ly, s.iy, -s.cly), by = __builtin_fmaf(hy, s.iy, -s.chy);
    const float az = __builtin_fmaf(lz, s.iz, -s.clz), bz = __builtin_fmaf(hz, s.iz, -s.chz);
    // (the reciprocals are finite, see slab_reciprocal, so nothing here is infinite or not a number)
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = t0;
    return t0 * 0.9999f <= t1 * 1.0001f;
}

__device__ __forceinline__ void trace(const FrameK& f, const Ray& r, const float4* __restrict__ TA,
                                      const float4* __restrict__ TB, const float4* __restrict__ TC,
                                      const float4* __restrict__ nodes, uint32_t* stack /* column of this lane */,
                                      uint32_t stride /* lanes sharing the stack array */, Hit& best) {
    const RaySlabs rs = make_slabs(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, 2.0f * f.pad);
    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        const float4 q0 = nodes[4 * (size_t)node + 0], q1 = nodes[4 * (size_t)node + 1];
        const float4 q2 = nodes[4 * (size_t)node + 2], q3 = nodes[4 * (size_t)node + 3];
        const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
        const float tmax = best.t * 1.0001f; // INFINITY until something is hit
        float tn0, tn1;
        bool h0 = slab_folded(rs, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tmax, tn0);
        bool h1 = slab_folded(rs, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tmax, tn1);
        const bool leaf0 = h0 && (c0 & 0x80000000u), leaf1 = h1 && (c1 & 0x80000000u);
        h0 = h0 && !leaf0, h1 = h1 && !leaf1;
        const bool first1 = tn1 < tn0;
        // Both children are subtrees and the stack is full (only a tree deeper than it can do that): the far
        // subtree's leaves are tested one by one (an internal node's record carries its first and last sorted leaf).
        const bool spill = h0 && h1 && sp >= kStack;
        uint32_t far_lo = 1u, far_hi = 0u;
        if (spill) {
            const float4 rg = nodes[4 * (size_t)(first1 ? c0 : c1) + 3];
            far_lo = __float_as_uint(rg.z), far_hi = __float_as_uint(rg.w);
        }
        // one copy of the triangle test for the three things a step can have to test: leaf child 0, leaf child 1,
        // the leaves of a subtree that did not fit the stack
#pragma nounroll
        for (int part = 0; part < 3; part++) {
            uint32_t ka = 1u, kb = 0u;
            if (part == 0 && leaf0) ka = kb = c0 & 0x7FFFFFFFu;
            if (part == 1 && leaf1) ka = kb = c1 & 0x7FFFFFFFu;
            if (part == 2) ka = far_lo, kb = far_hi;
            for (uint32_t k = ka; k <= kb; k++) mt_test(f, r, TA[k], TB[k], TC[k], k, best);
        }
        if (h0 && h1) {
            if (!spill) {
                stack[(size_t)sp * stride] = first1 ? c0 : c1; // far child waits
                sp++;
            }
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
                                                   uint32_t* stack, uint32_t stride, const AoK& ao,
                                                   const Outputs& out) {
    const uint32_t pix = py * f.w + px;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    Hit first = miss_hit();
    const Ray ray0 = make_ray(f, px, py, 0.5f, 0.5f);
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
    const uint32_t ws = first.key != kMissKey ? first.slot : 0u;
    finish_pixel(f, ao, out, out_index, pix, ray0, first, TB[ws], TC[ws], sr, sg, sb);
}

__global__ __launch_bounds__(256) void k_render_bvh(FrameK f, const float4* __restrict__ TA,
                                                    const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                    const float4* __restrict__ nodes, AoK ao, Outputs out) {
    __shared__ uint32_t s_stack[kStack * 256];
    const BlockMap m = map_block16(f, blockIdx.x, threadIdx.x);
    if (!m.valid) return;
    pixel_by_traversal(f, m.px, m.py, m.out, TA, TB, TC, nodes, s_stack + threadIdx.x, 256, ao, out);
}

// ------------------------------------------------------------------------------------------------
// tile kernel
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 both(float x) {
    const f32x2 r = {x, x};
    return r;
}

// Four side planes a*x + b*y + c*z + d >= 0 (inside), prepared for boxes given as sum and difference of their
// corners: coefficients halved, their absolute values, and the offset raised by what growing the box adds.
// Planes 0 and 1 sit in the two halves of the *01 members, planes 2 and 3 in *23 (packed fp32 arithmetic).
struct Frustum {
    f32x2 a01, b01, c01, aa01, ab01, ac01, d01;
    f32x2 a23, b23, c23, aa23, ab23, ac23, d23;
};

// Side planes, in object space, of the pyramid of rays through framebuffer rectangle
// [x0 - m, x1 + m] x [y0 - m, y1 + m] (pixels), for boxes grown by `grow`.  With clip = P * (V*M) * p and
// w_clip = -z_view:   x_ndc >= xl  <=>  P00 * x_view + xl * z_view >= 0, and so on for the other three sides.
__device__ __forceinline__ Frustum make_frustum(const FrameK& f, float x0, float y0, float x1, float y1, float grow) {
    const float m = 0.0625f; // the 1/16 pixel of slack dwarfs the rounding of these reciprocal multiplies
    const float xl = (x0 - m) * f.inv_w2 - 1.0f, xr = (x1 + m) * f.inv_w2 - 1.0f;
    const float yl = (y0 - m) * f.inv_h2 - 1.0f, yr = (y1 + m) * f.inv_h2 - 1.0f;
    const float* r0 = f.mv;
    const float* r1 = f.mv + 4;
    const float* r2 = f.mv + 8;
    float p[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p[0][k] = f.p00 * r0[k] + xl * r2[k];
        p[1][k] = -(f.p00 * r0[k] + xr * r2[k]);
        p[2][k] = f.p11 * r1[k] + yl * r2[k];
        p[3][k] = -(f.p11 * r1[k] + yr * r2[k]);
    }
    float ha[4], hb[4], hc[4], d[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ha[i] = 0.5f * p[i][0], hb[i] = 0.5f * p[i][1], hc[i] = 0.5f * p[i][2]; // exact
        d[i] = p[i][3] + grow * ((fabsf(p[i][0]) + fabsf(p[i][1])) + fabsf(p[i][2]));
    }
    Frustum fr;
    fr.a01 = f32x2{ha[0], ha[1]}, fr.b01 = f32x2{hb[0], hb[1]}, fr.c01 = f32x2{hc[0], hc[1]};
    fr.aa01 = f32x2{fabsf(ha[0]), fabsf(ha[1])}, fr.ab01 = f32x2{fabsf(hb[0]), fabsf(hb[1])};
    fr.ac01 = f32x2{fabsf(hc[0]), fabsf(hc[1])}, fr.d01 = f32x2{d[0], d[1]};
    fr.a23 = f32x2{ha[2], ha[3]}, fr.b23 = f32x2{hb[2], hb[3]}, fr.c23 = f32x2{hc[2], hc[3]};
    fr.aa23 = f32x2{fabsf(ha[2]), fabsf(ha[3])}, fr.ab23 = f32x2{fabsf(hb[2]), fabsf(hb[3])};
    fr.ac23 = f32x2{fabsf(hc[2]), fabsf(hc[3])}, fr.d23 = f32x2{d[2], d[3]};
    return fr;
}

// false only if the box, grown by the frustum's `grow`, lies wholly outside one side plane.  The largest value of a
// plane function over the box is  a*(lo+hi)/2 + |a|*(hi-lo)/2 + ... + d: twelve packed fused multiply-adds for the
// four planes.  The test only has to be conservative: its roundings (about 2^-23 of |a| * |coordinate| per term)
// are thousands of times smaller than the 1/16 pixel the planes were moved outwards by, and a NaN (boxes with
// infinite corners) counts as inside.
__device__ __forceinline__ void box_planes(const Frustum& fr, float lx, float ly, float lz, float hx, float hy, float hz,
                                           f32x2& v01, f32x2& v23) {
    const f32x2 sx = both(lx + hx), sy = both(ly + hy), sz = both(lz + hz);
    const f32x2 ex = both(hx - lx), ey = both(hy - ly), ez = both(hz - lz);
#define INV_FMA2 __builtin_elementwise_fma
    v01 = INV_FMA2(fr.a01, sx, INV_FMA2(fr.b01, sy, INV_FMA2(fr.c01, sz,
          INV_FMA2(fr.aa01, ex, INV_FMA2(fr.ab01, ey, INV_FMA2(fr.ac01, ez, fr.d01))))));
    v23 = INV_FMA2(fr.a23, sx, INV_FMA2(fr.b23, sy, INV_FMA2(fr.c23, sz,
          INV_FMA2(fr.aa23, ex, INV_FMA2(fr.ab23, ey, INV_FMA2(fr.ac23, ez, fr.d23))))));
#undef INV_FMA2
}

__device__ __forceinline__ bool box_in_frustum(const Frustum& fr, float lx, float ly, float lz, float hx, float hy,
                                               float hz) {
    f32x2 v01, v23;
    box_planes(fr, lx, ly, lz, hx, hy, hz, v01, v23);
    return !(v01.x < 0.0f) && !(v01.y < 0.0f) && !(v23.x < 0.0f) && !(v23.y < 0.0f);
}

// Which 8x8 quadrants of a 16x16 pixel block a box can touch, given the planes of make_frustum(xm, ym, xm, ym) for
// the block's centre (xm, ym): plane 0 keeps x >= xm - slack (right half), 1 keeps x <= xm + slack (left), 2 keeps
// y >= ym - slack (lower rows), 3 keeps y <= ym + slack (upper rows).  Bit q = quadrant (q & 1 = right, q >> 1 = lower).
__device__ __forceinline__ uint32_t box_quadrants(const Frustum& mid, float lx, float ly, float lz, float hx, float hy,
                                                  float hz) {
    f32x2 v01, v23;
    box_planes(mid, lx, ly, lz, hx, hy, hz, v01, v23);
    const uint32_t right = !(v01.x < 0.0f), left = !(v01.y < 0.0f), lower = !(v23.x < 0.0f), upper = !(v23.y < 0.0f);
    return (left & upper) | ((right & upper) << 1) | ((left & lower) << 2) | ((right & lower) << 3);
}

// direction through framebuffer position (x, y) for culling: make_ray's formula with reciprocal multiplies
__device__ __forceinline__ void dir_at(const FrameK& f, float x, float y, float& dx, float& dy, float& dz) {
    const float xn = x * f.inv_w2 - 1.0f, yn = y * f.inv_h2 - 1.0f;
    const float dxv = xn * f.inv_p00, dyv = yn * f.inv_p11;
    dx = (f.d0x * dxv + f.d1x * dyv) - f.d2x;
    dy = (f.d0y * dxv + f.d1y * dyv) - f.d2y;
    dz = (f.d0z * dxv + f.d1z * dyv) - f.d2z;
}

// output index of frame pixel (px, py); false if another rank owns it
__device__ __forceinline__ bool out_index(const FrameK& f, uint32_t px, uint32_t py, uint64_t& idx) {
    if (f.rowmajor) {
        idx = (uint64_t)py * f.w + px;
        return owns_rowmajor(f, px, py);
    }
    uint32_t id;
    if (!grid_tile(f, px, py, id)) return false;
    idx = (uint64_t)(id / f.nranks) * f.tile * f.tile + (uint64_t)(py % f.tile) * f.tile + px % f.tile;
    return id % f.nranks == f.rank;
}

// A traced square of EDGE x EDGE pixels whose attachments can be stored in 16-byte pieces: the square lies inside the
// frame, the frame's rows and the buffers allow 16-byte accesses (vec4), and the colour is final here (no occlusion
// pass).  Block-uniform.  `o0` is the output index of the square's first pixel, `pitch` the output elements per row.
template <uint32_t EDGE>
__device__ __forceinline__ bool packed_square(const FrameK& f, int vec4, const AoK& ao, uint32_t tx0, uint32_t ty0,
                                              uint64_t& o0, uint32_t& pitch) {
    pitch = f.rowmajor ? f.w : f.tile;
    o0 = 0;
    if (!vec4 || ao.rays || tx0 + EDGE > f.w || ty0 + EDGE > f.h) return false;
    return out_index(f, tx0, ty0, o0); // owned by this rank (a square lies in one grid tile)
}

// One attachment of a finished square from LDS (row-major EDGE x EDGE values) to memory, `lanes` = EDGE * EDGE / 4
// consecutive lanes starting at lane `first`, 16 bytes each.
template <uint32_t EDGE>
__device__ __forceinline__ void store_square_rows(uint32_t* base, uint64_t o0, uint32_t pitch, const uint32_t* lds,
                                                  uint32_t k /* 0 .. EDGE * EDGE / 4 - 1 */) {
    constexpr uint32_t per_row = EDGE / 4u;
    const uint32_t row = k / per_row, seg = k % per_row;
    const u32x4_t v = *(const u32x4_t*)(lds + row * EDGE + seg * 4u);
    store_pixels4(base + o0 + (uint64_t)row * pitch + seg * 4u, v);
}

// The clear values of a square nothing can hit.
template <uint32_t EDGE>
__device__ __forceinline__ void clear_square(const FrameK& f, const Outputs& out, uint64_t o0, uint32_t pitch, uint32_t k,
                                             uint32_t which /* 0 colour, 1 depth, 2 primitive id */) {
    constexpr uint32_t per_row = EDGE / 4u;
    const uint32_t row = k / per_row, seg = k % per_row;
    const uint64_t at = o0 + (uint64_t)row * pitch + seg * 4u;
    const uint32_t bg = background_pixel(f.flags);
    if (which == 0u && out.color) store_pixels4((uint32_t*)out.color + at, u32x4_t{bg, bg, bg, bg});
    if (which == 1u && out.depth) store_pixels4((uint32_t*)out.depth + at, u32x4_t{0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u});
    if (which == 2u && out.primid) store_pixels4(out.primid + at, u32x4_t{~0u, ~0u, ~0u, ~0u});
}

template <int NW, int CAP>
struct TileLds {
    union {
        struct {
            uint32_t frontier[2][kFrontierCap];
            uint2 ranges[kRangeCap];      // (first sorted triangle, count <= kRangeMax)
            float4 rec[CAP][3];          // candidate records A, B, C that survived the per-triangle test
            float4 pre2[CAP / 2][5];     // folded terms of the cheap inside test, two records interleaved
                                         // (mt_inside_pair)
        } t;
        uint32_t stack[kStack * 64]; // fallback: per-lane traversal by the first wave
    } u;
    float srgb[256];             // copy of the sRGB threshold table
    unsigned long long best[64]; // per pixel: smallest (depth, primitive) key of the current sample
    float4 win[NW * 64];         // per pixel: (un, vn, det, record slot) of the lane that holds the winning key;
                                 // or, when waves split the samples, the colour of each wave's sample
    float win_t[64];             // and its ray parameter
    uint32_t groups[kCutCap / kCutGroup]; // cut groups whose box meets the tile's frustum
    uint32_t n_groups;
    uint32_t n_front[2];
    uint32_t n_ranges;
    uint32_t n_rec;
    uint32_t overflow;     // more subtrees or leaf ranges than the lists hold: per-lane traversal instead
    uint32_t rec_overflow; // more candidate records than the list holds: phase 3 runs in batches
    uint32_t rec_k[CAP]; // position of each record in the sorted triangle arrays
    float lambert[CAP];  // Lambert factor of each record's triangle
    uint32_t px_out[3][64]; // the tile's finished pixels (colour, depth bits, primitive id), row-major 8 x 8
};

// Append `flag` lanes' values to an LDS list: one counter update per wave, lanes ranked by ballot.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool flag, uint32_t lane) {
    const uint64_t mask = __ballot(flag);
    uint32_t base = 0;
    if (mask) {
        if (lane == (uint32_t)__ffsll((long long)mask) - 1u) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = __shfl(base, __ffsll((long long)mask) - 1, 64);
    }
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// One workgroup of NW waves (4 in every instantiation) per 8x8 pixel tile of the rectangle that can contain hits.
template <int NW, bool BATCHED>
__device__ __forceinline__ void trace_tile(const FrameK& f, TileLds<NW, BATCHED ? kTileCapBatched : kTileCapSparse>& S, uint32_t tx0, uint32_t ty0,
                                           const float4* __restrict__ TA, const float4* __restrict__ TB,
                                           const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                           const float4* __restrict__ cut, const float4* __restrict__ cut_groups,
                                           uint32_t ncut, uint32_t* __restrict__ bin_count,
                                           const uint2* __restrict__ bin_list, const float4* __restrict__ flat,
                                           uint32_t nflat, uint32_t ntri, int vec4, const AoK& ao, const Outputs& out) {
    constexpr uint32_t NT = NW * 64;
    static_assert(NT == (uint32_t)kRangeCap, "one lane per entry of a bin");
#ifndef INV_PACK_TILES
#define INV_PACK_TILES 0
#endif
    // Tiles of dense scenes store their pixels in 16-byte pieces through LDS, as the 16x16 blocks do.  The sparse
    // scenes' tiles do not: their rows are 32 bytes, and with the exchange the draw of the 1080p bunny took 11.9
    // instead of 11.1 us (4K, 16 samples: 36.4 instead of 35.5 us per frame).
    constexpr bool kPackTile = INV_PACK_TILES != 0 || BATCHED;
    constexpr int kTileCap = BATCHED ? kTileCapBatched : kTileCapSparse;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const float grow = 2.0f * f.pad;
    if (INV_DBG(f, kDebugSkipAll)) return;
    const Frustum fr = make_frustum(f, (float)tx0, (float)ty0, (float)(tx0 + 8u), (float)(ty0 + 8u), grow);
    const bool stamps = INV_DBG(f, kDebugCounts) && !INV_DBG(f, kDebugSkipAfter2); // diagnostic build of the image
    unsigned long long t0 = 0, t1 = 0, t2 = 0, r0 = 0;
    if (stamps) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    // ---- phase 1, binned: the frame's binning pass (k_bin_ranges) has listed the leaf ranges whose boxes can touch
    //      this tile; the list and its length arrive in one round trip (every lane fetches one entry whether the
    //      list reaches that far or not), and the tile leaves the counter at zero for the next frame's pass ----
    uint32_t bin_n = 0;
    uint2 bin_e = make_uint2(0u, 0u);
    if (bin_count) bin_n = __builtin_amdgcn_readfirstlane(*bin_count), bin_e = bin_list[tid]; // block-uniform
    const float srgb_entry = kSrgbT[tid & 255u]; // stored to LDS later: the load stays in flight behind the first steps
    if (tid < 64) S.best[tid] = kMissKey;
    if (bin_count) {
        if (tid == 0) {
            S.n_front[0] = 0, S.n_front[1] = 0, S.n_rec = 0, S.rec_overflow = 0, S.n_groups = 0;
            S.n_ranges = min(bin_n, (uint32_t)kRangeCap), S.overflow = bin_n > (uint32_t)kRangeCap;
        }
        if (tid < bin_n && tid < (uint32_t)kRangeCap) S.u.t.ranges[tid] = bin_e;
        __syncthreads();
        // only now: every wave has used the count it loaded (a wave that loaded after the reset would see zero)
        if (tid == 0) *bin_count = 0u;
    }
    // ---- phase 1 from the coarse cut (scenes without a fine cut, draws without a binning pass): groups of cut
    //      records, then the records of the surviving groups, one box per lane; whatever is still a subtree after
    //      that is opened breadth-first ----
    // ---- phase 1, flat (small scenes): every fine-cut record against the frustum, two per lane and trip
    const bool flat_mode = !bin_count && nflat != 0u; // block-uniform
    const uint32_t ngroups = bin_count || flat_mode ? 0u : (ncut + kCutGroup - 1u) / kCutGroup;
    // the first box this lane tests (a group of the coarse cut, or a record of the fine cut) is fetched before
    // anything else: the load is in flight across the first barrier
    float4 g_lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g_hi = g_lo, g2_lo = g_lo, g2_hi = g_lo;
    if (flat_mode) { // (two records per lane: both loads in flight from the start)
        if (tid < nflat) g_lo = flat[2 * (size_t)tid], g_hi = flat[2 * (size_t)tid + 1];
        // (8x8 tiles fetch only the first record ahead: with both, the 4K / 16 spp frame took 37.5 instead of 36.1 us
        // and a draw on its own 11.4 instead of 11.2 us: registers held across the whole kernel at five workgroups per CU)
#ifndef INV_TILE_PRELOAD2
#define INV_TILE_PRELOAD2 0
#endif
        if (INV_TILE_PRELOAD2 && tid + NT < nflat) g2_lo = flat[2 * (size_t)(tid + NT)], g2_hi = flat[2 * (size_t)(tid + NT) + 1];
    } else if (tid < ngroups) {
        g_lo = cut_groups[2 * (size_t)tid], g_hi = cut_groups[2 * (size_t)tid + 1];
    }
    if (!bin_count) {
        if (tid == 0) S.n_front[0] = 0, S.n_front[1] = 0, S.n_ranges = 0, S.n_rec = 0, S.overflow = 0, S.rec_overflow = 0, S.n_groups = 0;
        __syncthreads();
    }
    if (flat_mode) {
        for (uint32_t base = 0; base < nflat; base += 2u * NT) { // block-uniform trip count: one trip up to 512 records
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t q = base + (uint32_t)u * NT + tid;
                if (base + (uint32_t)u * NT >= nflat) break; // block-uniform
                bool in = false;
                uint32_t first = 0, count = 0;
                if (q < nflat) {
                    const bool pre = base == 0u && (INV_TILE_PRELOAD2 || u == 0); // (block-uniform)
                    const float4 lo = pre ? (u ? g2_lo : g_lo) : flat[2 * (size_t)q];
                    const float4 hi = pre ? (u ? g2_hi : g_hi) : flat[2 * (size_t)q + 1];
                    first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
                    in = box_in_frustum(fr, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
                }
                const uint32_t ar = wave_append(&S.n_ranges, in, lane);
                if (in) {
                    if (ar < (uint32_t)kRangeCap)
                        S.u.t.ranges[ar] = make_uint2(first, count);
                    else
                        S.overflow = 1;
                }
            }
        }
    }
    for (uint32_t base = 0; base < ngroups; base += NT) { // block-uniform trip count
        const uint32_t g = base + tid;
        bool in = false;
        if (g < ngroups) {
            const float4 lo = base ? cut_groups[2 * (size_t)g] : g_lo, hi = base ? cut_groups[2 * (size_t)g + 1] : g_hi;
            in = box_in_frustum(fr, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
        }
        const uint32_t at = wave_append(&S.n_groups, in, lane);
        if (in) S.groups[at] = g;
    }
    __syncthreads();
    const uint32_t nsel = S.n_groups * kCutGroup;
    for (uint32_t base = 0; base < nsel; base += NT) { // block-uniform trip count
        const uint32_t q = base + tid;
        bool to_range = false, to_front = false;
        uint32_t ref = 0, count = 0;
        if (q < nsel) {
            const uint32_t i = S.groups[q / kCutGroup] * kCutGroup + q % kCutGroup;
            if (i < ncut) {
                const float4 lo = cut[2 * (size_t)i], hi = cut[2 * (size_t)i + 1];
                ref = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
                const bool in = box_in_frustum(fr, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
                to_range = in && count <= kRangeMax;
                to_front = in && count > kRangeMax;
            }
        }
        const uint32_t ar = wave_append(&S.n_ranges, to_range, lane);
        if (to_range) {
            if (ar < (uint32_t)kRangeCap)
                S.u.t.ranges[ar] = make_uint2(ref, count);
            else
                S.overflow = 1;
        }
        const uint32_t af = wave_append(&S.n_front[0], to_front, lane);
        if (to_front) {
            if (af < (uint32_t)kFrontierCap)
                S.u.t.frontier[0][af] = ref;
            else
                S.overflow = 1;
        }
    }
    __syncthreads();
    int cur = 0;
    for (;;) { // only scenes whose cut is capped still have subtrees to open here
        const uint32_t n = S.n_front[cur];
        if (n == 0 || S.overflow) break;
        for (uint32_t base = 0; base < n; base += NT) {
            const uint32_t q = base + tid;
            bool in0 = false, in1 = false;
            uint32_t c0 = 0, c1 = 0, lo = 0, hi = 0;
            if (q < n) {
                const uint32_t node = S.u.t.frontier[cur][q];
                const float4 q0 = nodes[4 * (size_t)node + 0], q1 = nodes[4 * (size_t)node + 1];
                const float4 q2 = nodes[4 * (size_t)node + 2], q3 = nodes[4 * (size_t)node + 3];
                c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                lo = __float_as_uint(q3.z), hi = __float_as_uint(q3.w);
                in0 = box_in_frustum(fr, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                in1 = box_in_frustum(fr, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
            }
            const uint32_t gamma = c0 & 0x7FFFFFFFu;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const bool in = side ? in1 : in0;
                const uint32_t c = side ? c1 : c0;
                const uint32_t first = side ? gamma + 1u : lo;
                const uint32_t count = side ? hi - gamma : gamma - lo + 1u;
                const bool small = (c & 0x80000000u) || count <= kRangeMax;
                const uint32_t ar = wave_append(&S.n_ranges, in && small, lane);
                if (in && small) {
                    if (ar < (uint32_t)kRangeCap)
                        S.u.t.ranges[ar] = make_uint2(first, count);
                    else
                        S.overflow = 1;
                }
                const uint32_t af = wave_append(&S.n_front[cur ^ 1], in && !small, lane);
                if (in && !small) {
                    if (af < (uint32_t)kFrontierCap)
                        S.u.t.frontier[cur ^ 1][af] = c;
                    else
                        S.overflow = 1;
                }
            }
        }
        __syncthreads();
        if (tid == 0) S.n_front[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }

    const uint32_t px = tx0 + (lane & 7u), py = ty0 + (lane >> 3);
    uint64_t oi = 0;
    const bool valid = px < f.w && py < f.h && out_index(f, px, py, oi);
    if (INV_DBG(f, kDebugSkipAfter1)) return;
    if (S.n_ranges == 0 && !S.overflow && !INV_DBG(f, kDebugCounts)) { // nothing in this tile: clear values
        uint64_t sq0;
        uint32_t sq_pitch;
        if (kPackTile && packed_square<8u>(f, vec4, ao, tx0, ty0, sq0, sq_pitch)) { // in 16-byte pieces (block-uniform)
            if (tid < 48u) clear_square<8u>(f, out, sq0, sq_pitch, tid & 15u, tid >> 4);
        } else if (wave == 0 && valid) {
            if (out.depth) store_pixel((uint32_t*)out.depth + oi, 0x3F800000u); // 1.0f
            if (out.primid) store_pixel(out.primid + oi, 0xFFFFFFFFu);
            if (out.color) store_pixel((uint32_t*)out.color + oi, background_pixel(f.flags));
        }
        return;
    }
    if (stamps) t1 = __builtin_amdgcn_s_memtime();
    if (tid < 256) S.srgb[tid] = srgb_entry;
    // ---- phase 2: one candidate triangle per lane against the tile's frustum and facing ----
    const uint32_t nranges = S.n_ranges;
    // lanes per range: a bin's ranges are records of the fine cut (at most kFineMax leaves), the coarse cut's have up
    // to kRangeMax
    const uint32_t rshift = bin_count || flat_mode ? (uint32_t)__builtin_ctz(kFineMax) : (uint32_t)__builtin_ctz(kRangeMax);
    // the triangles of ranges [r0, r1) that can cover a sample of the tile -> S.u.t.rec (arguments block-uniform)
    const auto build_records = [&](uint32_t r0, uint32_t r1) {
        // directions through the four corners of the grown tile (for the facing test)
        float cdx[4], cdy[4], cdz[4];
        const float m = 0.0625f;
        dir_at(f, (float)tx0 - m, (float)ty0 - m, cdx[0], cdy[0], cdz[0]);
        dir_at(f, (float)(tx0 + 8u) + m, (float)ty0 - m, cdx[1], cdy[1], cdz[1]);
        dir_at(f, (float)tx0 - m, (float)(ty0 + 8u) + m, cdx[2], cdy[2], cdz[2]);
        dir_at(f, (float)(tx0 + 8u) + m, (float)(ty0 + 8u) + m, cdx[3], cdy[3], cdz[3]);
        // bound on |direction component| over the tile's samples: directions are affine in the sample position, so
        // the corners of the grown tile bound them (doubled: make_ray's own roundings are nowhere near that)
        float dabs = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) dabs = fmaxf(dabs, fmaxf(fmaxf(fabsf(cdx[c]), fabsf(cdy[c])), fabsf(cdz[c])));
        dabs = dabs + dabs;
        const uint32_t nslots = (r1 - r0) << rshift;
        for (uint32_t base = 0; base < nslots; base += 2u * NT) { // block-uniform trip count; two triangles per lane
            float4 A2[2], B2[2], C2[2];                           // and trip, all loads in flight before the first test
            uint32_t k2[2] = {0, 0};
            bool have[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t q = base + (uint32_t)u * NT + tid;
                have[u] = false;
                if (q < nslots) {
                    const uint2 rg = S.u.t.ranges[r0 + (q >> rshift)];
                    const uint32_t off = q & ((1u << rshift) - 1u);
                    if (off < rg.y && rg.x + off < ntri) { // (the second test never fails for a list built from this scene)
                        k2[u] = rg.x + off;
                        have[u] = true, A2[u] = TA[k2[u]], B2[u] = TB[k2[u]], C2[u] = TC[k2[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (base + (uint32_t)u * NT >= nslots) break; // block-uniform
                bool keep = false;
                const uint32_t k = k2[u];
                const float4 A = A2[u], B = B2[u], C = C2[u];
                if (have[u]) {
                    const float bx = A.x + B.x, by = A.y + B.y, bz = A.z + B.z;
                    const float cx = A.x + C.x, cy = A.y + C.y, cz = A.z + C.z;
                    keep = box_in_frustum(fr, fminf(A.x, fminf(bx, cx)), fminf(A.y, fminf(by, cy)),
                                          fminf(A.z, fminf(bz, cz)), fmaxf(A.x, fmaxf(bx, cx)), fmaxf(A.y, fmaxf(by, cy)),
                                          fmaxf(A.z, fmaxf(bz, cz)));
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
                const uint32_t at = wave_append(&S.n_rec, keep, lane);
                if (keep) {
                    if (at < (uint32_t)kTileCap) {
                        S.u.t.rec[at][0] = A;
                        S.u.t.rec[at][1] = B;
                        S.u.t.rec[at][2] = C;
                        float4 P0, P1, P2;
                        fold_triangle(f.ox, f.oy, f.oz, A, B, C, P0, P1, P2);
                        // interleaved with the pair's other record: this one fills half (at & 1) of every float pair
                        float* pp = (float*)&S.u.t.pre2[at >> 1][0] + (at & 1u);
                        pp[0] = P0.x, pp[2] = P0.y, pp[4] = P0.z, pp[6] = P1.x, pp[8] = P1.y, pp[10] = P1.z;
                        pp[12] = P2.x, pp[14] = P2.y, pp[16] = P2.z;
                        pp[18] = pair_eps(f.ox, f.oy, f.oz, dabs, A, B, C);
                        // once per record instead of once per shaded sample (the batched variant shades from the
                        // triangle arrays and would not use it)
                        if (!BATCHED && (f.flags & INV_LAMBERT)) S.lambert[at] = lambert_term(f, B, C);
                        if (BATCHED) S.rec_k[at] = k;
                    } else {
                        S.rec_overflow = 1;
                    }
                }
            }
        }
    };
    if (!BATCHED && !S.overflow) build_records(0, nranges);
    __syncthreads();

    if (S.overflow || (!BATCHED && S.rec_overflow)) { // more than the LDS lists hold: the first wave traverses per lane
        if (wave == 0 && valid) pixel_by_traversal(f, px, py, oi, TA, TB, TC, nodes, S.u.stack + lane, 64, ao, out);
        return;
    }

    if (INV_DBG(f, kDebugSkipAfter2)) {
        if (INV_DBG(f, kDebugCounts) && valid && wave == 0) { // list sizes instead of an image
            if (out.depth) out.depth[oi] = __uint_as_float(S.n_ranges);
            if (out.primid) out.primid[oi] = S.n_rec;
        }
        return;
    }
    if (stamps) t2 = __builtin_amdgcn_s_memtime();
    // ---- phase 3: every wave tests its share of the records for all 64 pixels; an LDS min per pixel picks the
    //      winning key; the lane that holds it hands its barycentric terms to the first wave, which shades.
    //      The BATCHED variant (scenes whose triangles are much smaller than a pixel: more candidates per tile
    //      than the record list holds) works through the ranges in batches that always fit, rebuilding the
    //      records for every batch; the winner is then named by its position in the sorted triangle arrays
    //      instead of its LDS slot. ----
    constexpr bool batched = BATCHED;
    const uint32_t kBatchRanges = (uint32_t)kTileCap >> rshift; // ranges whose triangles always fit the record list
    const uint32_t nbatches = batched ? (nranges + kBatchRanges - 1u) / kBatchRanges : 1u;
    const uint32_t pix = py * f.w + px;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f; // first wave only
    Hit first = miss_hit();
    const Ray ray0 = make_ray(f, px, py, 0.5f, 0.5f);
    // With at least one sample per wave the waves split the samples instead of the records: a wave carries its
    // sample through every record and shades it itself, so there is no per-sample exchange between waves; the
    // first wave adds the colours up in sample order, NW samples per round.
    const bool by_sample = !BATCHED && f.spp >= (uint32_t)NW;
    for (uint32_t s0 = 0; by_sample && s0 < f.spp; s0 += NW) { // block-uniform
        const uint32_t s = s0 + wave;
        if (s < f.spp) {
            float sx, sy;
            sample_offset(pix, s, f.seed, sx, sy);
            const Ray ray = make_ray(f, px, py, sx, sy);
            Hit best = miss_hit();
            const uint32_t nrec = S.n_rec;
            for (uint32_t c = 0; c < nrec; c += 64) { // 64 records per mask
                const uint32_t m = min(64u, nrec - c);
                uint64_t inside = 0;
                for (uint32_t i = 0; i < m; i += 2) { // two records per step
                    const float4* q = S.u.t.pre2[(c + i) >> 1];
                    inside |= (uint64_t)mt_inside_pair(ray, q[0], q[1], q[2], q[3], q[4]) << i;
                }
                if (m < 64u) inside &= (1ull << m) - 1ull; // the last pair may lack its second record
                while (inside) {
                    const uint32_t j = c + (uint32_t)__ffsll((long long)inside) - 1u;
                    inside &= inside - 1ull;
                    mt_test(f, ray, S.u.t.rec[j][0], S.u.t.rec[j][1], S.u.t.rec[j][2], j, best);
                }
            }
            float r = 0.0f, g = 0.0f, b = 0.0f; // clear colour (src/invert.cpp:742)
            if (best.key != kMissKey)
                shade_hit_with(f, best, S.u.t.rec[best.slot][0], S.u.t.rec[best.slot][1], S.u.t.rec[best.slot][2],
                               (f.flags & INV_LAMBERT) ? S.lambert[best.slot] : 1.0f, r, g, b);
            S.win[wave * 64u + lane] = make_float4(r, g, b, 0.0f);
            if (s == 0) first = best; // sample 0 is the first wave's
        }
        __syncthreads();
        if (wave == 0) {
            const uint32_t n = min((uint32_t)NW, f.spp - s0);
            for (uint32_t w = 0; w < n; w++) {
                const float4 c = S.win[w * 64u + lane];
                sr = sr + c.x, sg = sg + c.y, sb = sb + c.z;
            }
        }
        __syncthreads();
    }
    for (uint32_t s = 0; !by_sample && s < f.spp; s++) {
        float sx, sy;
        sample_offset(pix, s, f.seed, sx, sy);
        const Ray ray = make_ray(f, px, py, sx, sy);
        Hit best = miss_hit();
        for (uint32_t bt = 0; bt < nbatches; bt++) { // block-uniform
            if (batched) {
                __syncthreads(); // every wave is done with the previous records
                if (tid == 0) S.n_rec = 0;
                __syncthreads();
                build_records(bt * kBatchRanges, min((bt + 1u) * kBatchRanges, nranges));
                __syncthreads();
            }
            const uint32_t nrec = min(S.n_rec, (uint32_t)kTileCap);
            // cheap pass: which of this wave's records contain the sample (one bit per record, no divergence); the
            // waves share the record pairs: pair p = wave + k * NW gives bits 2k and 2k + 1 ...
            uint64_t inside = 0;
            const uint32_t npairs = (nrec + 1u) >> 1;
            for (uint32_t p = wave, k = 0; p < npairs; p += NW, k++) {
                const float4* q = S.u.t.pre2[p];
                inside |= (uint64_t)mt_inside_pair(ray, q[0], q[1], q[2], q[3], q[4]) << (2u * k);
            }
            // ... then the full test (depth, guard, key) only for those, each lane walking its own bits
            if (INV_DBG(f, kDebugSkipExact)) inside = 0; // timing experiment only
            while (inside) {
                const uint32_t i = (uint32_t)__ffsll((long long)inside) - 1u;
                inside &= inside - 1ull;
                const uint32_t j = 2u * (wave + (i >> 1) * NW) + (i & 1u);
                if (j < nrec) // the last pair may lack its second record
                    mt_test(f, ray, S.u.t.rec[j][0], S.u.t.rec[j][1], S.u.t.rec[j][2], batched ? S.rec_k[j] : j, best);
            }
        }
        if (best.key != kMissKey) atomicMin(&S.best[lane], (unsigned long long)best.key);
        __syncthreads();
        // exactly one wave holds the winning record of a hit pixel (a triangle is listed once)
        if (best.key != kMissKey && best.key == S.best[lane])
            S.win[lane] = make_float4(best.un, best.vn, best.det, __uint_as_float(best.slot)), S.win_t[lane] = best.t;
        __syncthreads();
        if (wave == 0) {
            const uint64_t key = S.best[lane];
            float r = 0.0f, g = 0.0f, b = 0.0f; // clear colour (src/invert.cpp:742)
            if (key != kMissKey) {
                const float4 wv = S.win[lane];
                Hit h;
                h.key = key, h.t = S.win_t[lane], h.un = wv.x, h.vn = wv.y, h.det = wv.z, h.slot = __float_as_uint(wv.w);
                if (batched)
                    shade_hit(f, h, TA[h.slot], TB[h.slot], TC[h.slot], r, g, b);
                else
                    shade_hit_with(f, h, S.u.t.rec[h.slot][0], S.u.t.rec[h.slot][1], S.u.t.rec[h.slot][2],
                                   (f.flags & INV_LAMBERT) ? S.lambert[h.slot] : 1.0f, r, g, b);
                if (s == 0) first = h;
            }
            sr = sr + r, sg = sg + g, sb = sb + b;
            S.best[lane] = kMissKey;
        }
        if (s + 1 < f.spp) __syncthreads();
    }
    if (stamps) { // cycles per phase of wave 0 instead of an image: lanes 0..3 of the tile's first row
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        if (wave == 0 && valid && out.depth) {
            const unsigned long long r3 = __builtin_amdgcn_s_memrealtime();
            const unsigned long long v = lane == 0 ? t1 - t0 : lane == 1 ? t2 - t1 : lane == 2 ? t3 - t2 : lane == 3 ? (unsigned long long)S.n_rec
                                       : lane == 4 ? r3 - r0 : lane == 5 ? (t0 & 0xFFFFFFFFull) : lane == 6 ? (r0 & 0xFFFFFFFFull) : 0ull;
            out.depth[oi] = __uint_as_float((uint32_t)v);
        }
        return;
    }
    // whether the tile's pixels leave in 16-byte pieces through LDS (block-uniform; worked out only here, so that it
    // does not occupy registers through the phases above)
    uint64_t sq0;
    uint32_t sq_pitch;
    if (kPackTile && packed_square<8u>(f, vec4, ao, tx0, ty0, sq0, sq_pitch)) { // the first wave holds the tile's 64 pixels
        if (wave == 0) {
            S.px_out[0][lane] = resolve_pixel(f, sr, sg, sb, S.srgb);
            S.px_out[1][lane] = (uint32_t)(first.key >> 32), S.px_out[2][lane] = (uint32_t)first.key;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t a = lane >> 4;
            uint32_t* const base = a == 0u ? (uint32_t*)out.color : a == 1u ? (uint32_t*)out.depth : out.primid;
            if (lane < 48u && base) store_square_rows<8u>(base, sq0, sq_pitch, S.px_out[a], lane & 15u);
        }
        return;
    }
    if (wave != 0 || !valid) return;
    const uint32_t ws = first.key != kMissKey ? first.slot : 0u;
    if (batched)
        finish_pixel(f, ao, out, oi, pix, ray0, first, TB[ws], TC[ws], sr, sg, sb, S.srgb);
    else
        finish_pixel(f, ao, out, oi, pix, ray0, first, S.u.t.rec[ws][1], S.u.t.rec[ws][2], sr, sg, sb, S.srgb);
}

// ------------------------------------------------------------------------------------------------
// 16x16 pixel blocks: the un-batched path of the tile kernel
// ------------------------------------------------------------------------------------------------
constexpr int kBlockCap = 256; // candidate triangle records per 16x16 block

struct BlockLds {
    union {
        struct {
            uint32_t frontier[2][kFrontierCap];
            uint2 ranges[kRangeCap];      // (first sorted triangle, count <= kRangeMax)
            float4 rec[kBlockCap][3];     // candidate records A, B, C that survived the per-triangle test
            float4 pre2[kBlockCap / 2][5]; // folded terms of the cheap inside test, two records interleaved
                                           // (mt_inside_pair)
        } t;
        uint32_t stack[kStack * 128]; // fallback: per-lane traversal, one column per lane, two waves at a time
    } u;
    float srgb[256];            // copy of the sRGB threshold table
    uint32_t quad[kBlockCap];   // per record: the quadrants (bit = wave) its box can touch
    float lambert[kBlockCap];   // per record: Lambert factor of the triangle
    uint32_t groups[kCutCap / kCutGroup]; // cut groups whose box meets the block's frustum
    uint32_t n_groups;
    uint32_t n_front[2];
    uint32_t n_ranges;
    uint32_t n_rec;
    uint32_t overflow; // more subtrees, leaf ranges or records than the lists hold: per-lane traversal instead
    uint32_t px_out[3][256]; // the block's finished pixels (colour, depth bits, primitive id), row-major 16 x 16
};

// One workgroup of four waves per 16x16 pixel block of the rectangle that can contain hits.  Phases 1 and 2 find the
// block's candidate triangles as a whole workgroup (cut groups -> cut records -> subtrees -> one triangle per lane
// against the block's frustum and facing) and note for every record which 8x8 quadrants its box can touch.  In phase
// 3 every wave owns one quadrant: it walks the records of that quadrant (wave-uniform loop, records read from LDS as
// broadcasts), runs the cheap inside test for its 64 pixels, the full Moller-Trumbore test where that says yes, and
// shades and stores its own pixels: no exchange between waves.
__device__ __forceinline__ void trace_block(const FrameK& f, BlockLds& S, uint32_t tx0, uint32_t ty0,
                                            const float4* __restrict__ TA, const float4* __restrict__ TB,
                                            const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                            const float4* __restrict__ cut, const float4* __restrict__ cut_groups,
                                            uint32_t ncut, uint32_t* __restrict__ bin_count,
                                            const uint2* __restrict__ bin_list, const float4* __restrict__ flat,
                                            uint32_t nflat, uint32_t ntri, int vec4, const AoK& ao,
                                            const Outputs& out) {
    constexpr uint32_t NT = 256;
    static_assert(NT == (uint32_t)kRangeCap, "one lane per entry of a bin");
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const float grow = 2.0f * f.pad;
    if (INV_DBG(f, kDebugSkipAll)) return;
    const Frustum fr = make_frustum(f, (float)tx0, (float)ty0, (float)(tx0 + 16u), (float)(ty0 + 16u), grow);
    const bool stamps = INV_DBG(f, kDebugCounts) && !INV_DBG(f, kDebugSkipAfter2); // diagnostic build of the image
    unsigned long long t0 = 0, t1 = 0, t2 = 0, r0 = 0, ta = 0, tb = 0, tc = 0;
    if (stamps) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    // ---- phase 1, binned: the frame's binning pass (k_bin_ranges) has listed the leaf ranges whose boxes can touch
    //      this block; the list and its length arrive in one round trip (every lane fetches one entry whether the
    //      list reaches that far or not), and the block leaves the counter at zero for the next frame's pass ----
    uint32_t bin_n = 0;
    uint2 bin_e = make_uint2(0u, 0u);
    if (bin_count) bin_n = __builtin_amdgcn_readfirstlane(*bin_count), bin_e = bin_list[tid]; // block-uniform
    const float srgb_entry = kSrgbT[tid]; // stored to LDS later: the load stays in flight behind the first steps
    if (bin_count) {
        if (tid == 0) {
            S.n_front[0] = 0, S.n_front[1] = 0, S.n_rec = 0, S.n_groups = 0;
            S.n_ranges = min(bin_n, (uint32_t)kRangeCap), S.overflow = bin_n > (uint32_t)kRangeCap;
        }
        if (tid < bin_n && tid < (uint32_t)kRangeCap) S.u.t.ranges[tid] = bin_e;
        __syncthreads();
        // only now: every wave has used the count it loaded (a wave that loaded after the reset would see zero)
        if (tid == 0) *bin_count = 0u;
    }
    // ---- phase 1 from the coarse cut (scenes without a fine cut, draws without a binning pass): groups of cut
    //      records, then the records of the surviving groups, one box per lane; whatever is still a subtree after
    //      that is opened breadth-first ----
    // ---- phase 1, flat (small scenes): every fine-cut record against the frustum, two per lane and trip
    const bool flat_mode = !bin_count && nflat != 0u; // block-uniform
    const uint32_t ngroups = bin_count || flat_mode ? 0u : (ncut + kCutGroup - 1u) / kCutGroup;
    // the first box this lane tests (a group of the coarse cut, or a record of the fine cut) is fetched before
    // anything else: the load is in flight across the first barrier
    float4 g_lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g_hi = g_lo, g2_lo = g_lo, g2_hi = g_lo;
    if (flat_mode) { // (two records per lane: both loads in flight from the start)
        if (tid < nflat) g_lo = flat[2 * (size_t)tid], g_hi = flat[2 * (size_t)tid + 1];
        if (tid + NT < nflat) g2_lo = flat[2 * (size_t)(tid + NT)], g2_hi = flat[2 * (size_t)(tid + NT) + 1];
    } else if (tid < ngroups) {
        g_lo = cut_groups[2 * (size_t)tid], g_hi = cut_groups[2 * (size_t)tid + 1];
    }
    if (!bin_count) {
        if (tid == 0) S.n_front[0] = 0, S.n_front[1] = 0, S.n_ranges = 0, S.n_rec = 0, S.overflow = 0, S.n_groups = 0;
        __syncthreads();
    }
    if (flat_mode) {
        for (uint32_t base = 0; base < nflat; base += 2u * NT) { // block-uniform trip count: one trip up to 512 records
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t q = base + (uint32_t)u * NT + tid;
                if (base + (uint32_t)u * NT >= nflat) break; // block-uniform
                bool in = false;
                uint32_t first = 0, count = 0;
                if (q < nflat) {
                    const bool pre = base == 0u; // (block-uniform)
                    const float4 lo = pre ? (u ? g2_lo : g_lo) : flat[2 * (size_t)q];
                    const float4 hi = pre ? (u ? g2_hi : g_hi) : flat[2 * (size_t)q + 1];
                    first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
                    in = box_in_frustum(fr, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
                }
                const uint32_t ar = wave_append(&S.n_ranges, in, lane);
                if (in) {
                    if (ar < (uint32_t)kRangeCap)
                        S.u.t.ranges[ar] = make_uint2(first, count);
                    else
                        S.overflow = 1;
                }
            }
        }
    }
    if (stamps) ta = __builtin_amdgcn_s_memtime();
    for (uint32_t base = 0; base < ngroups; base += NT) { // block-uniform trip count
        const uint32_t g = base + tid;
        bool in = false;
        if (g < ngroups) {
            const float4 lo = base ? cut_groups[2 * (size_t)g] : g_lo, hi = base ? cut_groups[2 * (size_t)g + 1] : g_hi;
            in = box_in_frustum(fr, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
        }
        const uint32_t at = wave_append(&S.n_groups, in, lane);
        if (in) S.groups[at] = g;
    }
    __syncthreads();
    if (stamps) tb = __builtin_amdgcn_s_memtime();
    const uint32_t nsel = S.n_groups * kCutGroup;
    for (uint32_t base = 0; base < nsel; base += 2u * NT) { // block-uniform trip count; two records per lane and
        float4 lo[2], hi[2];                                // trip, both loads in flight before the first test
        bool have[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t q = base + (uint32_t)u * NT + tid;
            have[u] = false;
            if (q < nsel) {
                const uint32_t i = S.groups[q / kCutGroup] * kCutGroup + q % kCutGroup;
                if (i < ncut) have[u] = true, lo[u] = cut[2 * (size_t)i], hi[u] = cut[2 * (size_t)i + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (base + (uint32_t)u * NT >= nsel) break; // block-uniform
            bool to_range = false, to_front = false;
            uint32_t ref = 0, count = 0;
            if (have[u]) {
                ref = __float_as_uint(lo[u].w), count = __float_as_uint(hi[u].w);
                const bool in = box_in_frustum(fr, lo[u].x, lo[u].y, lo[u].z, hi[u].x, hi[u].y, hi[u].z);
                to_range = in && count <= kRangeMax;
                to_front = in && count > kRangeMax;
            }
            const uint32_t ar = wave_append(&S.n_ranges, to_range, lane);
            if (to_range) {
                if (ar < (uint32_t)kRangeCap)
                    S.u.t.ranges[ar] = make_uint2(ref, count);
                else
                    S.overflow = 1;
            }
            const uint32_t af = wave_append(&S.n_front[0], to_front, lane);
            if (to_front) {
                if (af < (uint32_t)kFrontierCap)
                    S.u.t.frontier[0][af] = ref;
                else
                    S.overflow = 1;
            }
        }
    }
    __syncthreads();
    if (stamps) tc = __builtin_amdgcn_s_memtime();
    int cur = 0;
    for (;;) { // only scenes whose cut is capped still have subtrees to open here
        const uint32_t n = S.n_front[cur];
        if (n == 0 || S.overflow) break;
        for (uint32_t base = 0; base < n; base += NT) {
            const uint32_t q = base + tid;
            bool in0 = false, in1 = false;
            uint32_t c0 = 0, c1 = 0, lo = 0, hi = 0;
            if (q < n) {
                const uint32_t node = S.u.t.frontier[cur][q];
                const float4 q0 = nodes[4 * (size_t)node + 0], q1 = nodes[4 * (size_t)node + 1];
                const float4 q2 = nodes[4 * (size_t)node + 2], q3 = nodes[4 * (size_t)node + 3];
                c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                lo = __float_as_uint(q3.z), hi = __float_as_uint(q3.w);
                in0 = box_in_frustum(fr, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                in1 = box_in_frustum(fr, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
            }
            const uint32_t gamma = c0 & 0x7FFFFFFFu;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const bool in = side ? in1 : in0;
                const uint32_t c = side ? c1 : c0;
                const uint32_t first = side ? gamma + 1u : lo;
                const uint32_t count = side ? hi - gamma : gamma - lo + 1u;
                const bool small = (c & 0x80000000u) || count <= kRangeMax;
                const uint32_t ar = wave_append(&S.n_ranges, in && small, lane);
                if (in && small) {
                    if (ar < (uint32_t)kRangeCap)
                        S.u.t.ranges[ar] = make_uint2(first, count);
                    else
                        S.overflow = 1;
                }
                const uint32_t af = wave_append(&S.n_front[cur ^ 1], in && !small, lane);
                if (in && !small) {
                    if (af < (uint32_t)kFrontierCap)
                        S.u.t.frontier[cur ^ 1][af] = c;
                    else
                        S.overflow = 1;
                }
            }
        }
        __syncthreads();
        if (tid == 0) S.n_front[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }

    // this lane's pixel: wave w owns quadrant (w & 1, w >> 1) of the block
    const uint32_t px = tx0 + ((wave & 1u) << 3) + (lane & 7u), py = ty0 + ((wave >> 1) << 3) + (lane >> 3);
    uint64_t oi = 0;
    const bool valid = px < f.w && py < f.h && out_index(f, px, py, oi);
    if (INV_DBG(f, kDebugSkipAfter1)) return;
    if (S.n_ranges == 0 && !S.overflow && !INV_DBG(f, kDebugCounts)) { // nothing in this block: clear values
        uint64_t sq0;
        uint32_t sq_pitch;
        if (packed_square<16u>(f, vec4, ao, tx0, ty0, sq0, sq_pitch)) { // in 16-byte pieces (block-uniform)
            if (wave < 3u) clear_square<16u>(f, out, sq0, sq_pitch, lane, wave);
        } else if (valid) {
            if (out.depth) store_pixel((uint32_t*)out.depth + oi, 0x3F800000u); // 1.0f
            if (out.primid) store_pixel(out.primid + oi, 0xFFFFFFFFu);
            if (out.color) store_pixel((uint32_t*)out.color + oi, background_pixel(f.flags));
        }
        return;
    }
    if (stamps) t1 = __builtin_amdgcn_s_memtime();
    S.srgb[tid] = srgb_entry;
    // ---- phase 2: one candidate triangle per lane against the block's frustum and facing ----
    const uint32_t nranges = S.n_ranges;
    if (!S.overflow) {
        const Frustum mid = make_frustum(f, (float)(tx0 + 8u), (float)(ty0 + 8u), (float)(tx0 + 8u), (float)(ty0 + 8u), grow);
        // directions through the four corners of the grown block (for the facing test)
        float cdx[4], cdy[4], cdz[4];
        const float m = 0.0625f;
        dir_at(f, (float)tx0 - m, (float)ty0 - m, cdx[0], cdy[0], cdz[0]);
        dir_at(f, (float)(tx0 + 16u) + m, (float)ty0 - m, cdx[1], cdy[1], cdz[1]);
        dir_at(f, (float)tx0 - m, (float)(ty0 + 16u) + m, cdx[2], cdy[2], cdz[2]);
        dir_at(f, (float)(tx0 + 16u) + m, (float)(ty0 + 16u) + m, cdx[3], cdy[3], cdz[3]);
        // bound on |direction component| over the block's samples: directions are affine in the sample position, so
        // the corners of the grown block bound them (doubled: make_ray's own roundings are nowhere near that)
        float dabs = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) dabs = fmaxf(dabs, fmaxf(fmaxf(fabsf(cdx[c]), fabsf(cdy[c])), fabsf(cdz[c])));
        dabs = dabs + dabs;
        // lanes per range: a bin's ranges are records of the fine cut (at most kFineMax leaves), the coarse cut's
        // have up to kRangeMax
        const uint32_t rshift = bin_count || flat_mode ? (uint32_t)__builtin_ctz(kFineMax) : (uint32_t)__builtin_ctz(kRangeMax);
        const uint32_t nslots = nranges << rshift;
        for (uint32_t base = 0; base < nslots; base += 2u * NT) { // block-uniform trip count; two triangles per lane
            float4 A[2], B[2], C[2];                              // and trip, all loads in flight before the first test
            bool have[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t q = base + (uint32_t)u * NT + tid;
                have[u] = false;
                if (q < nslots) {
                    const uint2 rg = S.u.t.ranges[q >> rshift];
                    const uint32_t off = q & ((1u << rshift) - 1u);
                    if (off < rg.y && rg.x + off < ntri) { // (the second test never fails for a list built from this scene)
                        const uint32_t k = rg.x + off;
                        have[u] = true, A[u] = TA[k], B[u] = TB[k], C[u] = TC[k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (base + (uint32_t)u * NT >= nslots) break; // block-uniform
                bool keep = false;
                uint32_t quads = 0;
                const float4 a = A[u], b = B[u], c = C[u];
                if (have[u]) {
                    const float bx = a.x + b.x, by = a.y + b.y, bz = a.z + b.z;
                    const float cx = a.x + c.x, cy = a.y + c.y, cz = a.z + c.z;
                    const float lx = fminf(a.x, fminf(bx, cx)), ly = fminf(a.y, fminf(by, cy)), lz = fminf(a.z, fminf(bz, cz));
                    const float hx = fmaxf(a.x, fmaxf(bx, cx)), hy = fmaxf(a.y, fmaxf(by, cy)), hz = fmaxf(a.z, fmaxf(bz, cz));
                    keep = box_in_frustum(fr, lx, ly, lz, hx, hy, hz);
                    // det = e1 . (d x e2) is linear in d and every sample direction of the block is a convex
                    // combination of the corner directions: if det is clearly negative at all four corners the
                    // triangle faces away from every ray of the block (cull BACK, src/invert.cpp:1334).
                    const float nx = c.y * b.z - c.z * b.y, ny = c.z * b.x - c.x * b.z, nz = c.x * b.y - c.y * b.x;
                    float dmax = -INFINITY, scale = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        dmax = fmaxf(dmax, (cdx[k] * nx + cdy[k] * ny) + cdz[k] * nz);
                        scale = fmaxf(scale, (fabsf(cdx[k] * nx) + fabsf(cdy[k] * ny)) + fabsf(cdz[k] * nz));
                    }
                    keep = keep && !(dmax < -1e-4f * scale);
                    if (keep) quads = box_quadrants(mid, lx, ly, lz, hx, hy, hz);
                }
                const uint32_t at = wave_append(&S.n_rec, keep, lane);
                if (keep) {
                    if (at < (uint32_t)kBlockCap) {
                        S.u.t.rec[at][0] = a;
                        S.u.t.rec[at][1] = b;
                        S.u.t.rec[at][2] = c;
                        float4 P0, P1, P2;
                        fold_triangle(f.ox, f.oy, f.oz, a, b, c, P0, P1, P2);
                        // interleaved with the pair's other record: this one fills half (at & 1) of every float pair
                        float* pp = (float*)&S.u.t.pre2[at >> 1][0] + (at & 1u);
                        pp[0] = P0.x, pp[2] = P0.y, pp[4] = P0.z, pp[6] = P1.x, pp[8] = P1.y, pp[10] = P1.z;
                        pp[12] = P2.x, pp[14] = P2.y, pp[16] = P2.z;
                        pp[18] = pair_eps(f.ox, f.oy, f.oz, dabs, a, b, c);
                        if (f.flags & INV_LAMBERT) S.lambert[at] = lambert_term(f, b, c); // once per record, not per shaded sample
                        S.quad[at] = quads;
                    } else {
                        S.overflow = 1;
                    }
                }
            }
        }
    }
    __syncthreads();

    if (S.overflow) { // more than the LDS lists hold: every lane traverses the LBVH for its own pixel
        for (uint32_t half = 0; half < 2u; half++) { // the stacks of two waves fit at a time
            if ((wave >> 1) == half && valid)
                pixel_by_traversal(f, px, py, oi, TA, TB, TC, nodes, S.u.stack + (tid & 127u), 128, ao, out);
            __syncthreads();
        }
        return;
    }
    if (INV_DBG(f, kDebugSkipAfter2)) {
        if (INV_DBG(f, kDebugCounts) && valid) { // list sizes instead of an image
            if (out.depth) out.depth[oi] = __uint_as_float(S.n_ranges);
            if (out.primid) out.primid[oi] = S.n_rec;
        }
        return;
    }
    if (stamps) t2 = __builtin_amdgcn_s_memtime();
    // ---- phase 3: this wave's quadrant ----
    const uint32_t nrec = S.n_rec;

    const uint32_t pix = py * f.w + px;
    // records whose box can touch this quadrant, as wave-uniform bit masks (64 records each)
    uint64_t rel[kBlockCap / 64];
#pragma unroll
    for (int c = 0; c < kBlockCap / 64; c++) {
        const uint32_t j = (uint32_t)c * 64u + lane;
        rel[c] = __ballot(j < nrec && ((S.quad[j] >> wave) & 1u));
    }
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    Hit first = miss_hit();
    const Ray ray0 = make_ray(f, px, py, 0.5f, 0.5f);
    for (uint32_t s = 0; s < f.spp; s++) {
        // (sample 0 is the pixel centre; a ray copied from ray0 under a branch ends up in scratch memory)
        float sx = 0.5f, sy = 0.5f;
        if (s) sample_offset(pix, s, f.seed, sx, sy);
        const Ray ray = make_ray(f, px, py, sx, sy);
        Hit best = miss_hit();
#pragma unroll
        for (int c = 0; c < kBlockCap / 64; c++) {
            if (!rel[c]) continue;
            // cheap pass: which of the quadrant's records contain the sample (one bit per record, no divergence), two
            // records per step: the pairs with at least one record of this quadrant (wave-uniform) ...
            uint64_t todo = (rel[c] | (rel[c] >> 1)) & 0x5555555555555555ull;
            uint64_t inside = 0;
            while (todo) {
                const uint32_t i = (uint32_t)__builtin_ctzll(todo); // even: the pair's first record
                todo &= todo - 1ull;
                const float4* q = S.u.t.pre2[((uint32_t)c * 64u + i) >> 1];
                inside |= (uint64_t)mt_inside_pair(ray, q[0], q[1], q[2], q[3], q[4]) << i;
            }
            inside &= rel[c]; // a pair's other record may belong to other quadrants only, or (last pair) not exist
            // ... then the full test (depth, guard, key) only for those, each lane walking its own bits
            if (INV_DBG(f, kDebugSkipExact)) inside = 0; // timing experiment only
            while (inside) {
                const uint32_t j = (uint32_t)c * 64u + (uint32_t)__ffsll((long long)inside) - 1u;
                inside &= inside - 1ull;
                mt_test(f, ray, S.u.t.rec[j][0], S.u.t.rec[j][1], S.u.t.rec[j][2], j, best);
            }
        }
        float r = 0.0f, g = 0.0f, b = 0.0f; // clear colour (src/invert.cpp:742)
        if (best.key != kMissKey)
            shade_hit_with(f, best, S.u.t.rec[best.slot][0], S.u.t.rec[best.slot][1], S.u.t.rec[best.slot][2],
                           (f.flags & INV_LAMBERT) ? S.lambert[best.slot] : 1.0f, r, g, b);
        sr = sr + r, sg = sg + g, sb = sb + b;
        if (s == 0) first = best;
    }
    if (stamps) { // cycles per phase of wave 0 instead of an image: lanes 0..6 of the block's first row
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        if (wave == 0 && valid && out.depth) {
            const unsigned long long r3 = __builtin_amdgcn_s_memrealtime();
            const unsigned long long v = lane == 0 ? t1 - t0 : lane == 1 ? t2 - t1 : lane == 2 ? t3 - t2 : lane == 3 ? (unsigned long long)S.n_rec
                                       : lane == 4 ? r3 - r0 : lane == 5 ? (t0 & 0xFFFFFFFFull) : lane == 6 ? (r0 & 0xFFFFFFFFull)
                                       : lane == 7 ? ta - t0 : lane == 8 ? tb - ta : lane == 9 ? tc - tb : lane == 10 ? (unsigned long long)S.n_groups : 0ull;
            out.depth[oi] = __uint_as_float((uint32_t)v);
        }
        return;
    }
    // whether the block's pixels leave in 16-byte pieces through LDS (block-uniform; worked out only here, so that it
    // does not occupy registers through the phases above)
    uint64_t sq0;
    uint32_t sq_pitch;
    if (packed_square<16u>(f, vec4, ao, tx0, ty0, sq0, sq_pitch)) { // every wave's quadrant into the block's image
        const uint32_t li = ((((wave >> 1) << 3) | (lane >> 3)) << 4) | ((wave & 1u) << 3) | (lane & 7u);
        S.px_out[0][li] = resolve_pixel(f, sr, sg, sb, S.srgb);
        S.px_out[1][li] = (uint32_t)(first.key >> 32), S.px_out[2][li] = (uint32_t)first.key;
        __syncthreads();
        uint32_t* const base = wave == 0u ? (uint32_t*)out.color : wave == 1u ? (uint32_t*)out.depth : out.primid;
        if (wave < 3u && base) store_square_rows<16u>(base, sq0, sq_pitch, S.px_out[wave], lane);
        return;
    }
    if (!valid) return;
    const uint32_t ws = first.key != kMissKey ? first.slot : 0u;
    finish_pixel(f, ao, out, oi, pix, ray0, first, S.u.t.rec[ws][1], S.u.t.rec[ws][2], sr, sg, sb, S.srgb);
}

// What the launcher works out for one frame of the tile kernel (host arithmetic, see plan_tile_frame).
struct TilePlan {
    uint32_t block_px;          // edge of a traced block: 16 (a wave per quadrant) or 8 (the tile variants)
    uint32_t tiles_x, ntiles;   // blocks of the traced rectangle
    uint32_t fill_blocks;       // workgroups of the fill part
    uint32_t pixel_blocks;      // 32x8 pixel blocks of the frame (fill part without 16-byte stores)
    int vec4;                   // rows and addresses allow 16-byte stores
    uint32_t row4_magic;        // floor(2^32 / chunks per row)
    // Bins of the frame's binning pass, one per traced block: bin_count[b] entries of bin_list[b * kRangeCap ...],
    // each the (first sorted triangle, count) of a leaf range whose box can touch block b.  Null: no binning pass
    // ran for this frame and the blocks start from the scene's coarse cut.
    uint32_t* bin_count;
    uint2* bin_list;
    uint32_t ntri;              // triangles of the scene: no list entry is followed past the arrays
    // Scenes of a few hundred fine-cut records (flat != null, no bins): a block tests every record against its own
    // frustum, two or four per lane, instead of walking the coarse cut's two levels: its first list of leaf ranges
    // is one round trip away instead of two.
    const float4* flat;
    uint32_t nflat;
};

// How a frame's traced rectangle is worked through (chosen per frame by plan_tile_frame):
//   kBlocks   16x16 pixel blocks, a wave per quadrant (trace_block): the least arithmetic per pixel
//   kTiles    8x8 pixel tiles whose four waves split the samples (trace_tile<4, false>): frames with four or more
//             samples per pixel, where a wave per quadrant would carry too long a chain of samples
//   kBatched  8x8 pixel tiles worked through in batches of leaf ranges (trace_tile<4, true>): triangles far smaller
//             than a pixel
enum TraceMode { kBlocks = 0, kTiles = 1, kBatched = 2 };

template <int MODE>
struct TraceLds {
    typedef BlockLds type;
};
template <>
struct TraceLds<kTiles> {
    typedef TileLds<4, kTileCapSparse> type;
};
template <>
struct TraceLds<kBatched> {
    typedef TileLds<4, kTileCapBatched> type;
};

// One trace workgroup: block `tb` of the trace part of a frame's grid.
template <int MODE>
__device__ __forceinline__ void tile_block(const FrameK& f, const TilePlan& pl, uint32_t tb,
                                           typename TraceLds<MODE>::type& S,
                                           const float4* __restrict__ TA, const float4* __restrict__ TB,
                                           const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                           const float4* __restrict__ cut, const float4* __restrict__ cut_groups,
                                           uint32_t ncut, const AoK& ao, const Outputs& out) {
    constexpr uint32_t kEdge = MODE == kBlocks ? 16u : 8u;
    const uint32_t tx0 = (f.rx0 & ~(kEdge - 1u)) + (tb % pl.tiles_x) * kEdge;
    const uint32_t ty0 = (f.ry0 & ~(kEdge - 1u)) + (tb / pl.tiles_x) * kEdge;
    if (f.nranks > 1) { // block-uniform: a block lies in one grid tile (their edge is a multiple of 32), so in one rank
        uint64_t dummy;
        if (!out_index(f, tx0, ty0, dummy)) return;
    }
    uint32_t* const bin_count = pl.bin_count ? pl.bin_count + tb : nullptr;
    const uint2* const bin_list = pl.bin_list + (size_t)tb * (uint32_t)kRangeCap;
    if constexpr (MODE == kBlocks)
        trace_block(f, S, tx0, ty0, TA, TB, TC, nodes, cut, cut_groups, ncut, bin_count, bin_list, pl.flat, pl.nflat, pl.ntri,
                    pl.vec4, ao, out);
    else
        trace_tile<4, MODE == kBatched>(f, S, tx0, ty0, TA, TB, TC, nodes, cut, cut_groups, ncut, bin_count, bin_list, pl.flat,
                                        pl.nflat, pl.ntri, pl.vec4, ao, out);
}

// One workgroup of the fill part: block `fb` of pl.fill_blocks.  The fill blocks store the clear values of every
// pixel whose 8x8 tile is not in the trace part, grid-stride: 16 bytes per lane when the row length and the buffer
// addresses allow it (vec4), else one pixel per lane in 32x8 pixel blocks (two 128-byte lines per wave-instruction).
__device__ __forceinline__ void fill_part(const FrameK& f, const TilePlan& pl, uint32_t fb, const Outputs& out) {
    const uint32_t tiles_x = pl.tiles_x, ntiles = pl.ntiles, fill_blocks = pl.fill_blocks;
    const uint32_t pixel_blocks = pl.pixel_blocks, row4_magic = pl.row4_magic;
    const int vec4 = pl.vec4;
    if (INV_DBG(f, kDebugSkipFill)) return;
    const uint32_t edge = pl.block_px;
    const uint32_t hx0 = f.rx0 & ~(edge - 1u), hy0 = f.ry0 & ~(edge - 1u);
    const uint32_t hx1 = hx0 + tiles_x * edge, hy1 = hy0 + (tiles_x ? ntiles / tiles_x : 0u) * edge; // traced rectangle
    const uint32_t bg = background_pixel(f.flags);
    if (vec4) {
        // 16 bytes (four pixels of a row) per lane, attachment and store: 1 KiB per wave-instruction.  A wave takes runs
        // of kFillRun consecutive chunks; a run that lies in a row-major frame and touches no row of the traced
        // rectangle (most do) is stored with one address per attachment and immediate offsets, without any per-lane
        // pixel arithmetic.
        // Write-through, non-temporal stores (sc1 nt).  Write-through: the clear values leave the XCD's L2 as they
        // are written instead of staying behind as dirty lines that push the scene out and are written back later
        // (12.8 -> 12.1 us per frame alone, 7.8 -> 6.8 us with two frames in flight).  Non-temporal: they are not
        // kept in the Infinity Cache either.  With sc1 alone a job whose attachment sets fit that cache (eight 1080p
        // sets: 199 MB) clears a frame in 3.2 us and one whose sets do not in 4.25 us; with sc1 nt both take 3.8 us,
        // and whole frames (clear and traced blocks) take no longer than the clear alone, 3.76 us = 6.8 TB/s into
        // HBM, where sc1 alone took 4.6 us once the sets outgrew the cache.  The trailing s_nop keeps the next
        // instruction off the data registers until the store has read them.
        const uint32_t row4 = (f.rowmajor ? f.w : f.tile) >> 2; // chunks per stored row
        const uint32_t nchunks = (uint32_t)(f.npix >> 2);          // the host keeps npix near 2^31 at most, so c0 + stride cannot wrap
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const u32x4 c4 = {bg, bg, bg, bg}, p4 = {~0u, ~0u, ~0u, ~0u};
        const f32x4 d4 = {1.0f, 1.0f, 1.0f, 1.0f};
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wave0 = __builtin_amdgcn_readfirstlane(fb * 4u + (threadIdx.x >> 6));
        const uint32_t run_stride = fill_blocks * 4u * kFillRun;
#ifndef INV_FILL_MODS
#define INV_FILL_MODS "sc1 nt"
#endif
#define INV_FILL_ST(v, p) asm volatile("global_store_dwordx4 %0, %1, off " INV_FILL_MODS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory")
#define INV_FILL_AT(v, p, off) \
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2 " INV_FILL_MODS "\n\ts_nop 1" ::"v"(p), "v"(v), "n"(off) : "memory")
#define INV_FILL_RUN(v, p)                                             \
    do {                                                               \
        INV_FILL_AT(v, p, 0);                                          \
        if (kFillRun > 64u) INV_FILL_AT(v, p, 1024);                   \
        if (kFillRun > 128u) {                                         \
            INV_FILL_AT(v, p, 2048);                                   \
            INV_FILL_AT(v, p, 3072);                                   \
        }                                                              \
    } while (0)
        static_assert(kFillRun == 64 || kFillRun == 128 || kFillRun == 256, "immediate offsets reach 3 KiB");
        for (uint32_t c0 = wave0 * kFillRun; c0 < nchunks; c0 += run_stride) { // wave-uniform
            bool whole = f.rowmajor && c0 + kFillRun <= nchunks;
            if (whole && hx1 > hx0 && hy1 > hy0) {
                // first and last row of the run: c / row4 without a division (row4_magic = floor(2^32 / row4) gives
                // the quotient or one less)
                const uint32_t cl = c0 + kFillRun - 1u;
                uint32_t q0 = __umulhi(c0, row4_magic), q1 = __umulhi(cl, row4_magic);
                if (c0 - q0 * row4 >= row4) q0++;
                if (cl - q1 * row4 >= row4) q1++;
                whole = q1 < hy0 || q0 >= hy1;
            }
            if (whole) {
                const uint32_t c = c0 + lane;
                if (out.depth) INV_FILL_RUN(d4, (f32x4*)out.depth + c);
                if (out.primid) INV_FILL_RUN(p4, (u32x4*)out.primid + c);
                if (out.color) INV_FILL_RUN(c4, (u32x4*)out.color + c);
                continue;
            }
            for (uint32_t j = 0; j < kFillRun; j += 64u) {
                const uint32_t c = c0 + j + lane;
                if (c >= nchunks) continue;
                uint32_t px, py;
                if (f.rowmajor) {
                    uint32_t q = __umulhi(c, row4_magic), r = c - q * row4;
                    if (r >= row4) q++, r -= row4;
                    px = r << 2, py = q;
                } else {
                    const uint32_t per_tile = (f.tile * f.tile) >> 2;
                    const uint32_t slot = c / per_tile, rem = c % per_tile;
                    const uint32_t id = slot * f.nranks + f.rank;
                    if (id >= f.ntiles) continue;
                    px = (f.gx0 + id % f.tiles_x) * f.tile + ((rem % row4) << 2);
                    py = (f.gy0 + id / f.tiles_x) * f.tile + rem / row4;
                    if (px >= f.w || py >= f.h) continue;
                }
                if (px >= hx0 && px < hx1 && py >= hy0 && py < hy1) continue; // traced by the first part of the grid
                if (out.depth) INV_FILL_ST(d4, (f32x4*)out.depth + c);
                if (out.primid) INV_FILL_ST(p4, (u32x4*)out.primid + c);
                if (out.color) INV_FILL_ST(c4, (u32x4*)out.color + c);
            }
        }
#undef INV_FILL_ST
#undef INV_FILL_RUN
#undef INV_FILL_AT
        return;
    }
    for (uint32_t b = fb; b < pixel_blocks; b += fill_blocks) {
        const BlockOrigin o = block_origin(f, b);
        if (!o.live) continue;
        const uint32_t cx = o.x0 + (threadIdx.x & 31u), cy = o.y0 + (threadIdx.x >> 5);
        if (cx >= f.w || cy >= f.h) continue;
        if (cx >= hx0 && cx < hx1 && cy >= hy0 && cy < hy1) continue;
        const uint64_t i = o.out0 + (uint64_t)(threadIdx.x >> 5) * o.pitch + (threadIdx.x & 31u);
        if (out.depth) out.depth[i] = 1.0f;
        if (out.primid) out.primid[i] = 0xFFFFFFFFu;
        if (out.color) ((uint32_t*)out.color)[i] = bg;
    }
}

constexpr uint32_t kXcdRun = 16; // consecutive tiles one XCD takes (see k_render_tile)
__host__ __device__ inline uint32_t xcd_run_grid(uint32_t ntiles) { // tile workgroups launched: whole runs for all XCDs
    return (ntiles + 8u * kXcdRun - 1u) / (8u * kXcdRun) * (8u * kXcdRun);
}

// Grid = [blocks of the hit rectangle] ++ [fill blocks]; workgroups of four waves.
#ifndef INV_TILES_PER_CU
#define INV_TILES_PER_CU 5
#endif
template <int MODE>
__global__ __launch_bounds__(256, MODE == kBatched ? 1 : (MODE == kTiles ? INV_TILES_PER_CU : 4)) void k_render_tile(FrameK f, TilePlan pl,
                                                         const float4* __restrict__ TA, const float4* __restrict__ TB,
                                                         const float4* __restrict__ TC, const float4* __restrict__ nodes,
                                                         const float4* __restrict__ cut,
                                                         const float4* __restrict__ cut_groups, uint32_t ncut,
                                                         AoK ao, Outputs out) {
    __shared__ typename TraceLds<MODE>::type S;
    // Workgroups go to the eight XCDs in turn, each with an L2 of its own.  Neighbouring tiles walk the same
    // subtrees and read the same triangles, so an XCD takes runs of kXcdRun consecutive tiles instead of every eighth
    // tile: what a tile has pulled into its L2 is there for its neighbours (the scene of a single bunny fits every L2
    // whole; a scene of millions of triangles does not).  Runs, not an eighth of the rectangle each: the tiles' work
    // is uneven (the bunny's body against its ears), and eight bands would leave the XCDs unevenly loaded.  Only the
    // variant for dense scenes does this: with a few hundred tiles per frame even runs load the XCDs unevenly (two
    // overlapping draws of the bunny took 10.0 instead of 6.3 us per frame), and such scenes gain nothing from it.
    const uint32_t tile_grid = MODE == kBatched ? xcd_run_grid(pl.ntiles) : pl.ntiles;
    if (blockIdx.x < tile_grid) {
        uint32_t t = blockIdx.x;
        if (MODE == kBatched) {
            const uint32_t k = blockIdx.x >> 3; // position in this XCD's sequence
            t = ((k / kXcdRun) * 8u + (blockIdx.x & 7u)) * kXcdRun + k % kXcdRun;
        }
        if (t < pl.ntiles) tile_block<MODE>(f, pl, t, S, TA, TB, TC, nodes, cut, cut_groups, ncut, ao, out);
    } else {
        fill_part(f, pl, blockIdx.x - tile_grid, out);
    }
}

// Several frames in one launch (different cameras, extents and attachments, one scene).
// Grid = [tiles of frame 0] ++ [fill blocks of frame 0] ++ [tiles of frame 1] ++ ... : workgroups start in grid
// order, so the stores of one frame's fill part overlap the traced tiles of the next, and the dispatch gap and the
// wait for the slowest tiles are paid once per launch.
struct BatchFrame {
    FrameK f;
    TilePlan pl;
    Outputs out;
};
struct Batch {
    uint32_t nframes;
    uint32_t block_end[kBatchMax]; // running sum of ntiles + fill_blocks
    BatchFrame fr[kBatchMax];
};
static_assert(sizeof(Batch) <= 3072, "the batch travels in the kernel arguments");

__global__ __launch_bounds__(256, 4) void k_render_tile_batch(Batch b, const float4* __restrict__ TA,
                                                                  const float4* __restrict__ TB,
                                                                  const float4* __restrict__ TC,
                                                                  const float4* __restrict__ nodes,
                                                                  const float4* __restrict__ cut,
                                                                  const float4* __restrict__ cut_groups, uint32_t ncut) {
    __shared__ BlockLds S;
    const AoK ao = {0, 0.0f, nullptr, nullptr, nullptr};
    // block-uniform: frame of this block and the first block of that frame.  The running sums are read in one go and
    // compared without branches (entries past the last frame hold the grid size, which no block index reaches).
    uint32_t i = 0, first = 0;
#pragma unroll
    for (uint32_t k = 0; k + 1u < kBatchMax; k++) {
        const uint32_t e = b.block_end[k];
        const bool past = blockIdx.x >= e;
        i += past ? 1u : 0u;
        first = past ? e : first;
    }
    const uint32_t local = blockIdx.x - first;
    if (local < b.fr[i].pl.ntiles)
        tile_block<kBlocks>(b.fr[i].f, b.fr[i].pl, local, S, TA, TB, TC, nodes, cut, cut_groups, ncut, ao, b.fr[i].out);
    else
        fill_part(b.fr[i].f, b.fr[i].pl, local - b.fr[i].pl.ntiles, b.fr[i].out);
}

// ------------------------------------------------------------------------------------------------
// binning pass: the fine cut's leaf ranges -> per-block lists, once per frame
// ------------------------------------------------------------------------------------------------
// One lane per record of the scene's fine cut (a subtree of at most kFineMax leaves with its box).  The box, grown as
// the tile kernel grows it, is carried to the framebuffer corner by corner with the frame's own matrices; the blocks
// its pixel rectangle (widened by 1/8 pixel) touches get the record's leaf range appended to their lists.  A box lies
// outside a block's frustum exactly when its projected rectangle misses the block, so the lists are what the blocks'
// own frustum tests would have kept.  Why they are conservative: a triangle that passes the Moller-Trumbore test has
// its hit point inside its own box grown by one guard pad, hence inside the subtree's box grown by two; that point
// lies on the sample's ray and so projects to the sample position inside the pixel, to within the roundings of this
// projection and of the ray's inverse matrix (about 10^-3 pixel at the largest extents the API accepts).  A box that
// reaches the eye's plane is listed for every block.
// Lists are appended to in scheduling order: the tile kernel's
// result is a minimum over the candidates and does not depend on their order.  A list that is full keeps counting;
// its block sees the count and falls back to the per-lane traversal.
__device__ __forceinline__ bool bin_owned(const FrameK& f, const TilePlan& pl, uint32_t bx, uint32_t by) {
    if (f.nranks <= 1) return true;
    const uint32_t edge = pl.block_px;
    uint64_t dummy;
    return out_index(f, (f.rx0 & ~(edge - 1u)) + bx * edge, (f.ry0 & ~(edge - 1u)) + by * edge, dummy);
}

// `rpw` records per workgroup (a power of two, at most 256: 256 for scenes with hundreds of thousands of records,
// fewer for small scenes, whose few thousand (record, block) pairs then spread over more workgroups).  The lanes first
// find their records' block rectangles, a scan in LDS numbers the (record, block) pairs of the workgroup, and then
// every lane takes pairs: no lane waits for one atomic after another on behalf of a record that touches many blocks,
// and a lane's four pairs per step are in flight together.
__global__ __launch_bounds__(256) void k_bin_ranges(Batch b, const float4* __restrict__ fine, uint32_t nfine,
                                                    uint32_t rpw) {
    __shared__ uint4 s_rect[256];
    __shared__ uint2 s_entry[256];
    __shared__ uint32_t s_off[257]; // pairs before record r of this workgroup; [rpw] = all pairs
    __shared__ uint32_t s_wave[4];
    // bins the workgroup's pairs fall into: bin + 1 (0: free), pairs counted, places reserved in the bin's list
    constexpr uint32_t kBinHashBits = 9, kBinHash = 1u << kBinHashBits;
    __shared__ uint32_t s_hkey[kBinHash], s_hcnt[kBinHash], s_hbase[kBinHash];
    const FrameK& f = b.fr[blockIdx.y].f;
    const TilePlan& pl = b.fr[blockIdx.y].pl;
    if (!pl.bin_count || pl.ntiles == 0) return; // block-uniform
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i = blockIdx.x * rpw + tid;
    const uint32_t tiles_x = pl.tiles_x, tiles_y = pl.ntiles / pl.tiles_x;
    for (uint32_t h = tid; h < kBinHash; h += 256u) s_hkey[h] = 0u, s_hcnt[h] = 0u; // (barriers below before its first use)
    uint32_t bx0 = 1, bx1 = 0, by0 = 1, by1 = 0; // empty
    uint2 entry = make_uint2(0u, 0u);
    if (tid < rpw && i < nfine) {
        const float4 lo = fine[2 * (size_t)i], hi = fine[2 * (size_t)i + 1];
        entry = make_uint2(__float_as_uint(lo.w), __float_as_uint(hi.w));
        const float grow = 2.0f * f.pad;
        const float cx[2] = {lo.x - grow, hi.x + grow}, cy[2] = {lo.y - grow, hi.y + grow}, cz[2] = {lo.z - grow, hi.z + grow};
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, wmin = INFINITY;
        bool whole = false;
        const float half_w = 0.5f * f.wf, half_h = 0.5f * f.hf;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float x = cx[c & 1], y = cy[(c >> 1) & 1], z = cz[c >> 2];
            // (fused multiply-adds and the hardware's reciprocal, a unit in the last place off at most: what this feeds is
            // widened by an eighth of a pixel below)
            const float xv = __builtin_fmaf(f.mv[0], x, __builtin_fmaf(f.mv[1], y, __builtin_fmaf(f.mv[2], z, f.mv[3])));
            const float yv = __builtin_fmaf(f.mv[4], x, __builtin_fmaf(f.mv[5], y, __builtin_fmaf(f.mv[6], z, f.mv[7])));
            const float zv = __builtin_fmaf(f.mv[8], x, __builtin_fmaf(f.mv[9], y, __builtin_fmaf(f.mv[10], z, f.mv[11])));
            const float w = -zv; // w_clip of a GLM perspective
            const float rw = __builtin_amdgcn_rcpf(w);
            const float fx = __builtin_fmaf(xv * rw, half_w * f.p00, half_w); // ((p00 xv / w) + 1) w / 2
            const float fy = __builtin_fmaf(yv * rw, half_h * f.p11, half_h);
            // (a position that is not a number: a minimum would drop it; a w that is not a number makes such a position)
            whole = whole | __builtin_isunordered(fx, fy);
            wmin = fminf(wmin, w);
            xmin = fminf(xmin, fx), xmax = fmaxf(xmax, fx), ymin = fminf(ymin, fy), ymax = fmaxf(ymax, fy);
        }
        whole = whole | !(wmin > 1e-30f);
        const float edge = (float)pl.block_px, redge = 1.0f / edge; // 8 or 16: exact
        const float hx0 = (float)(f.rx0 & ~(pl.block_px - 1u)), hy0 = (float)(f.ry0 & ~(pl.block_px - 1u));
        const float m = 0.125f;
        float fx0 = floorf(((xmin - m) - hx0) * redge), fx1 = floorf(((xmax + m) - hx0) * redge);
        float fy0 = floorf(((ymin - m) - hy0) * redge), fy1 = floorf(((ymax + m) - hy0) * redge);
        if (whole) fx0 = 0.0f, fy0 = 0.0f, fx1 = (float)tiles_x, fy1 = (float)tiles_y;
        if (fx1 >= 0.0f && fy1 >= 0.0f && fx0 <= (float)(tiles_x - 1u) && fy0 <= (float)(tiles_y - 1u)) {
            bx0 = (uint32_t)fmaxf(fx0, 0.0f), by0 = (uint32_t)fmaxf(fy0, 0.0f);
            bx1 = (uint32_t)fminf(fx1, (float)(tiles_x - 1u)), by1 = (uint32_t)fminf(fy1, (float)(tiles_y - 1u));
        }
    }
    const bool any = bx0 <= bx1 && by0 <= by1;
    const uint32_t nb = any ? (bx1 - bx0 + 1u) * (by1 - by0 + 1u) : 0u;
    // exclusive scan of nb over the workgroup's lanes
    uint32_t incl = nb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) s_wave[wave] = incl;
    s_rect[tid] = make_uint4(bx0, by0, bx1, by1), s_entry[tid] = entry;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
    s_off[tid] = before + incl - nb;
    if (tid == 255u) s_off[256] = before + incl;
    __syncthreads();
    const uint32_t npairs = s_off[256];
    // The records of a workgroup lie together, so its pairs fall into few bins, and so do its neighbours' at the same
    // moment: an update of a bin's counter per pair made the pass wait for some sixty updates of one address after
    // another.  The pairs are counted per bin in LDS first (a small hash table; a pair that finds no place in it
    // updates the bin's counter itself, as before), then one update per bin and workgroup reserves the places.
    for (uint32_t base = 0; base < npairs; base += 1024u) { // block-uniform
        uint32_t bin[4], at[4], slot[4];
        uint2 e[4];
        bool go[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = base + (uint32_t)u * 256u + tid;
            go[u] = false, bin[u] = 0, at[u] = 0, slot[u] = kBinHash, e[u] = make_uint2(0u, 0u);
            if (p < npairs) {
                // the record this pair belongs to: the last r with s_off[r] <= p
                uint32_t r = 0;
#pragma unroll
                for (uint32_t step = 128u; step != 0u; step >>= 1)
                    if (s_off[r + step] <= p) r += step;
                const uint4 rc = s_rect[r];
                const uint32_t k = p - s_off[r], wx = rc.z - rc.x + 1u;
                const uint32_t bx = rc.x + k % wx, by = rc.y + k / wx;
                go[u] = bin_owned(f, pl, bx, by);
                bin[u] = by * tiles_x + bx, e[u] = s_entry[r];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (go[u]) {
                uint32_t h = (bin[u] * 2654435761u) >> (32 - kBinHashBits);
                for (int probe = 0; probe < 8; probe++) {
                    const uint32_t was = atomicCAS(&s_hkey[h], 0u, bin[u] + 1u);
                    if (was == 0u || was == bin[u] + 1u) {
                        slot[u] = h;
                        break;
                    }
                    h = (h + 1u) & (kBinHash - 1u);
                }
                at[u] = slot[u] != kBinHash ? atomicAdd(&s_hcnt[slot[u]], 1u) : atomicAdd(&pl.bin_count[bin[u]], 1u);
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < kBinHash; i += 256u)
            if (s_hkey[i] != 0u) s_hbase[i] = atomicAdd(&pl.bin_count[s_hkey[i] - 1u], s_hcnt[i]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (go[u]) {
                const uint32_t place = slot[u] != kBinHash ? s_hbase[slot[u]] + at[u] : at[u];
                if (place < (uint32_t)kRangeCap) pl.bin_list[(size_t)bin[u] * (uint32_t)kRangeCap + place] = e[u];
            }
        }
        if (base + 1024u < npairs) { // (block-uniform) the table again empty for the next pairs
            __syncthreads();
            for (uint32_t i = tid; i < kBinHash; i += 256u) s_hkey[i] = 0u, s_hcnt[i] = 0u;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// triangles much smaller than a pixel: one lane per triangle instead of lists per tile
// ------------------------------------------------------------------------------------------------
// Where an 8x8 tile meets hundreds of triangles (the 32 x 32 grid of bunnies: 700 per covered tile, eleven per
// pixel) the tile kernel spends its time building and testing lists of candidates that mostly miss.  Such frames
// are drawn the other way round: every triangle finds the few pixel centres its own box can contain, runs the
// Moller-Trumbore test of exactly those pixels' rays (same function, same operands, so the same bits as any other
// kernel computes) and offers its (depth, primitive) key to the pixel's entry of a visibility buffer with a 64-bit
// atomic minimum: the depth test LESS with the lower primitive winning ties (src/invert.cpp:1384 + draw order), and
// independent of the order in which triangles arrive.  A second pass reads each pixel's winner, looks up where that
// primitive sits in the sorted arrays (or, when the scene was made from a mesh small enough for the caches, assembles
// the triangle again from that mesh: two fetches that the L2 answers instead of a chain of three all over memory),
// repeats the test once for the barycentric terms, shades and stores, and puts the entry back to the miss key for the
// next frame.
// Which pixels a triangle must try: a triangle that passes the test at a pixel has its hit point inside its own box
// grown by one guard pad, and that point projects to the pixel centre (to within the roundings of this projection and
// of the ray's inverse matrix: about 10^-3 pixel at the largest extents the API accepts); so the centre lies in the
// rectangle that the eight corners of the box, grown by two pads, project to, widened by 1/8 pixel.  A box that
// reaches the eye's plane tries every pixel of the frame's hit rectangle.
// One sample per pixel only (frames with more take the tile kernel).
#ifndef INV_SCATTER_PER_LANE
#define INV_SCATTER_PER_LANE 2
#endif
constexpr uint32_t kScatterPerLane = INV_SCATTER_PER_LANE; // triangles per lane of the triangle pass
// (triangle, pixel) pairs a workgroup of the triangle pass lists before it tests them (some 150 where triangles are much
// smaller than a pixel; what no longer fits is tested on the spot).  The list decides how many workgroups share a CU's
// LDS, and the pass is a chain of round trips that wants every wave it can get: with 1024 entries the workgroup was 40
// bytes above a fifth of a CU's LDS (58.2 us per frame of five million triangles), with 512 five fit (52.7 us), with 256
// six (49.4 us).  Six are taken while the triangle arrays fit the Infinity Cache; a scene of 69 million triangles, which
// waits for HBM, was 1.4 % slower with six than with five.
template <int WAVES>
constexpr uint32_t kPairCap = WAVES >= 6 ? 256u : 512u;
constexpr uint64_t kScatterSixBytes = 256ull << 20; // triangle arrays up to this size: six workgroups per CU

// The exact test of one (triangle, pixel) pair and the key minimum in the visibility buffer.
__device__ __forceinline__ void scatter_pair(const FrameK& f, uint32_t px, uint32_t py, const float4& A,
                                             const float4& B, const float4& C, unsigned long long* __restrict__ vis) {
    const Ray ray = make_ray(f, px, py, 0.5f, 0.5f);
    Hit best = miss_hit();
    mt_test(f, ray, A, B, C, 0u, best); // (the key carries the primitive id: A.w)
    if (best.key != kMissKey) atomicMin(&vis[(size_t)py * f.w + px], (unsigned long long)best.key);
}

// Part one, a lane per triangle: the back-face test; the triangles that pass are listed in LDS.  Part two, a lane per
// listed triangle (about half as many): the pixels its projection can cover and the cheap inside test on each of them;
// what passes is listed as (triangle, pixel) pairs.  Part three, a lane per listed pair: the exact ray and the full
// test.  (All in one loop per triangle, a wave ran the full test once per loop trip for whichever lanes had a
// candidate in that trip, and everything behind the back-face test with half its lanes: 760 wave-instructions per
// wave at 42 % of lanes.)
// (Assembling the triangles again from a mesh that the caches hold, as the resolve pass does with its winners, was
// no faster here: 72 against 71 us for five million triangles.  The pass is not waiting for its 48 bytes per triangle.)
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_scatter_triangles(FrameK f, const float4* __restrict__ TA,
                                                           const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                           uint32_t ntri, unsigned long long* __restrict__ vis) {
    __shared__ uint32_t s_nfront, s_npairs;
    // the listed triangles' records (read once from memory)
    __shared__ float4 s_recA[256 * kScatterPerLane], s_recB[256 * kScatterPerLane], s_recC[256 * kScatterPerLane];
    __shared__ uint2 s_pairs[kPairCap<WAVES>]; // (place in the list, px | py << 16)
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) s_nfront = 0, s_npairs = 0;
    __syncthreads();
    // kScatterPerLane triangles per lane, their records all in flight before the first is looked at: a workgroup is a
    // chain of three memory round trips whatever it holds, so it should hold enough to keep the memory system busy
    float4 rA[kScatterPerLane], rB[kScatterPerLane], rC[kScatterPerLane];
#pragma unroll
    for (uint32_t u = 0; u < kScatterPerLane; u++) {
        const uint32_t k = (blockIdx.x * kScatterPerLane + u) * 256u + threadIdx.x;
        rA[u] = rB[u] = rC[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (k < ntri) rA[u] = TA[k], rB[u] = TB[k], rC[u] = TC[k];
    }
    bool front[kScatterPerLane];
#pragma unroll
    for (uint32_t u = 0; u < kScatterPerLane; u++) {
        const uint32_t k = (blockIdx.x * kScatterPerLane + u) * 256u + threadIdx.x;
        front[u] = false;
        if (k < ntri) {
            // cull BACK (src/invert.cpp:1334).  For a ray from the origin o that meets the triangle's plane at
            // p = o + t d, t > 0, det = e1 . (d x e2) = (p - o) . (e2 x e1) / t, and (p - o) . n is the same for every p
            // of the plane: its sign at v0 decides for all rays at once.  With P0 = e2 x e1 as fold_triangle forms it
            // and t0 = o - v0, det = d . P0 = -(t0 . P0) / t: back-facing for every ray when t0 . P0 is clearly
            // positive (a margin of the products' rounding keeps the doubtful).
            const float4 A = rA[u], B = rB[u], C = rC[u];
            float4 P0, P1, P2;
            fold_triangle(f.ox, f.oy, f.oz, A, B, C, P0, P1, P2);
            const float tx = f.ox - A.x, ty = f.oy - A.y, tz = f.oz - A.z;
            const float sd = (tx * P0.x + ty * P0.y) + tz * P0.z;
            const float mag = (fabsf(tx * P0.x) + fabsf(ty * P0.y)) + fabsf(tz * P0.z);
            front[u] = !(sd > 1e-4f * mag);
        }
    }
    {
        // the wave's places in the list for all its triangles at once: one update of the counter, the lanes ranked by ballot
        uint64_t mask[kScatterPerLane];
        uint32_t n = 0;
#pragma unroll
        for (uint32_t u = 0; u < kScatterPerLane; u++) mask[u] = __ballot(front[u]), n += (uint32_t)__popcll(mask[u]);
        uint32_t at = 0;
        if (n) { // (the same for the whole wave)
            if (lane == 0) at = atomicAdd(&s_nfront, n);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        }
#pragma unroll
        for (uint32_t u = 0; u < kScatterPerLane; u++) {
            const uint32_t mine = at + (uint32_t)__popcll(mask[u] & ((1ull << lane) - 1ull));
            if (front[u]) s_recA[mine] = rA[u], s_recB[mine] = rB[u], s_recC[mine] = rC[u];
            at += (uint32_t)__popcll(mask[u]);
        }
    }
    __syncthreads();
    const uint32_t nfront = s_nfront;
    for (uint32_t base = threadIdx.x & ~63u; base < nfront; base += 256u) { // wave-uniform: 64 listed triangles a trip
        const bool have = base + lane < nfront;
        const uint32_t at = have ? base + lane : 0u;
        const float4 A = s_recA[at], B = s_recB[at], C = s_recC[at];
        float4 P0, P1, P2;
        fold_triangle(f.ox, f.oy, f.oz, A, B, C, P0, P1, P2);
        // The pixels whose centre ray can pass the full test lie inside the triangle's projection, give or take the
        // roundings of the test and of the ray (a few 1e-4 pixel at most): the three vertices through the frame's
        // matrix, their pixel bounds widened by an eighth of a pixel.  A vertex at or behind the eye's plane (the
        // launcher only takes scenes that lie in front of it) makes the triangle claim the whole frame.
        const float vx[3] = {A.x, A.x + B.x, A.x + C.x}, vy[3] = {A.y, A.y + B.y, A.y + C.y}, vz[3] = {A.z, A.z + B.z, A.z + C.z};
        const float half_w = 0.5f * f.wf, half_h = 0.5f * f.hf;
        float fxs[3], fys[3], ws[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // (fused multiply-adds and the hardware's reciprocal, a unit in the last place off at most: what this feeds is
            // widened by an eighth of a pixel below; the separate roundings cost twice the instructions and an exact
            // quotient ten)
            const float xv = __builtin_fmaf(f.mv[0], vx[c], __builtin_fmaf(f.mv[1], vy[c], __builtin_fmaf(f.mv[2], vz[c], f.mv[3])));
            const float yv = __builtin_fmaf(f.mv[4], vx[c], __builtin_fmaf(f.mv[5], vy[c], __builtin_fmaf(f.mv[6], vz[c], f.mv[7])));
            const float zv = __builtin_fmaf(f.mv[8], vx[c], __builtin_fmaf(f.mv[9], vy[c], __builtin_fmaf(f.mv[10], vz[c], f.mv[11])));
            ws[c] = -zv; // w_clip of a GLM perspective
            const float rw = __builtin_amdgcn_rcpf(ws[c]);
            fxs[c] = __builtin_fmaf(xv * rw, half_w * f.p00, half_w); // ((p00 xv / w) + 1) w / 2
            fys[c] = __builtin_fmaf(yv * rw, half_h * f.p11, half_h);
        }
        const float xmin = fminf(fminf(fxs[0], fxs[1]), fxs[2]), xmax = fmaxf(fmaxf(fxs[0], fxs[1]), fxs[2]);
        const float ymin = fminf(fminf(fys[0], fys[1]), fys[2]), ymax = fmaxf(fmaxf(fys[0], fys[1]), fys[2]);
        // a w that is not clearly positive, or a position that is not a number (a minimum would drop it; a w that is not a
        // number makes such a position): one comparison per vertex says whether either of its two is
        const bool whole = !(fminf(fminf(ws[0], ws[1]), ws[2]) > 1e-30f) | __builtin_isunordered(fxs[0], fys[0]) |
                           __builtin_isunordered(fxs[1], fys[1]) | __builtin_isunordered(fxs[2], fys[2]);
        // pixel centres px + 0.5 inside [xmin - m, xmax + m], clipped to the rectangle that can contain hits
        const float m = 0.125f;
        float fx0 = ceilf(xmin - m - 0.5f), fx1 = floorf(xmax + m - 0.5f);
        float fy0 = ceilf(ymin - m - 0.5f), fy1 = floorf(ymax + m - 0.5f);
        if (whole) fx0 = 0.0f, fy0 = 0.0f, fx1 = f.wf, fy1 = f.hf;
        uint32_t x0 = 1, x1 = 0, y0 = 1, y1 = 0; // an empty rectangle
        if (have && fx1 >= (float)f.rx0 && fy1 >= (float)f.ry0 && fx0 <= (float)f.rx1 - 1.0f && fy0 <= (float)f.ry1 - 1.0f) {
            x0 = (uint32_t)fmaxf(fx0, (float)f.rx0), y0 = (uint32_t)fmaxf(fy0, (float)f.ry0);
            x1 = (uint32_t)fminf(fx1, (float)f.rx1 - 1.0f), y1 = (uint32_t)fminf(fy1, (float)f.ry1 - 1.0f);
        }
        const bool some = x0 <= x1 && y0 <= y1;
        // The cheap inside test of the tile kernel (mt_inside_pair, one record): the three dot products of the folded
        // terms with the pixel's direction, every comparison widened by a bound on the roundings (of these products,
        // of mt_test's own, and of the reciprocal multiplies in dir_at; f.dir_abs_max bounds the direction's
        // components over the frame's hit rectangle).  Only pixels it lets through get their exact ray and the full
        // test, which alone decides.
        const float eps = pair_eps(f.ox, f.oy, f.oz, f.dir_abs_max + f.dir_abs_max, A, B, C);
        // the pixel loop, every lane of the wave in step: trip i is the lane's i-th pixel, if it has that many
        const uint32_t cols = some ? x1 - x0 + 1u : 0u, mine = some ? cols * (y1 - y0 + 1u) : 0u;
        uint32_t px = x0, py = y0;
        // (dir_at at the pixel's centre with its steps folded into multiply-adds: view-space x = px * sx + tx; roundings of
        // the size eps already allows for)
        const float sx = f.inv_w2 * f.inv_p00, tx = __builtin_fmaf(0.5f, sx, -f.inv_p00);
        const float sy = f.inv_h2 * f.inv_p11, ty = __builtin_fmaf(0.5f, sy, -f.inv_p11);
        for (uint32_t i = 0; __ballot(i < mine) != 0ull; i++) { // as many trips as the lane with the most pixels has
            bool pass = false;
            if (i < mine) {
                const float dxv = __builtin_fmaf((float)px, sx, tx), dyv = __builtin_fmaf((float)py, sy, ty);
                const float dx = __builtin_fmaf(f.d0x, dxv, __builtin_fmaf(f.d1x, dyv, -f.d2x));
                const float dy = __builtin_fmaf(f.d0y, dxv, __builtin_fmaf(f.d1y, dyv, -f.d2y));
                const float dz = __builtin_fmaf(f.d0z, dxv, __builtin_fmaf(f.d1z, dyv, -f.d2z));
                const float det = __builtin_fmaf(dx, P0.x, __builtin_fmaf(dy, P0.y, dz * P0.z));
                const float un = __builtin_fmaf(dx, P1.x, __builtin_fmaf(dy, P1.y, dz * P1.z));
                const float vn = __builtin_fmaf(dx, P2.x, __builtin_fmaf(dy, P2.y, dz * P2.z));
                const float e2 = eps + eps, hi = det + e2;
                const float worst = fminf(fminf(fminf(det + eps, un + eps), fminf(hi - un, vn + eps)), (hi + eps) - (un + vn));
                pass = !(worst < 0.0f); // (a NaN goes on to the exact test)
                if (pass && f.nranks > 1) { // another rank's tile
                    uint64_t dummy;
                    pass = out_index(f, px, py, dummy);
                }
            }
            const uint32_t ap = wave_append(&s_npairs, pass, lane);
            if (pass) {
                if (ap < kPairCap<WAVES>)
                    s_pairs[ap] = make_uint2(at, px | (py << 16));
                else { // the list is full: this pair right away
                    // (the records read again, from a place the optimiser cannot see through: what the test computes from
                    // them alone is then not computed ahead of the loop for every triangle, and they need not stay in
                    // registers through it)
                    uint32_t at2 = at;
                    asm volatile("" : "+v"(at2));
                    scatter_pair(f, px, py, s_recA[at2], s_recB[at2], s_recC[at2], vis);
                }
            }
            if (++px > x1) px = x0, py++;
        }
    }
    __syncthreads();
    const uint32_t npairs = min(s_npairs, kPairCap<WAVES>);
    for (uint32_t q = threadIdx.x; q < npairs; q += 256u) {
        const uint2 e = s_pairs[q];
        scatter_pair(f, e.y & 0xFFFFu, e.y >> 16, s_recA[e.x], s_recB[e.x], s_recC[e.x], vis);
    }
}

// One lane per output pixel: the winner of the pixel's visibility entry is shaded and stored, and the entry is put
// back to the miss key.  `slot_of`: primitive id -> position in the sorted triangle arrays.
__global__ __launch_bounds__(256) void k_scatter_resolve(FrameK f, MeshSource mesh, const float4* __restrict__ TA,
                                                         const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                         const uint32_t* __restrict__ slot_of, uint32_t ntri,
                                                         unsigned long long* __restrict__ vis, AoK ao, Outputs out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < f.npix;
    uint32_t px = 0, py = 0;
    if (live) live = work_to_pixel(f, i, px, py);
    // (every lane of the wave goes through finish_pixel together: it ranks the lanes that queue occlusion records)
    const uint32_t pix = py * f.w + px;
    const Ray ray0 = make_ray(f, px, py, 0.5f, 0.5f);
    Hit first = miss_hit();
    float r = 0.0f, g = 0.0f, b = 0.0f; // clear colour (src/invert.cpp:742)
    float4 Bw = make_float4(0.0f, 0.0f, 0.0f, 0.0f), Cw = Bw;
    if (live) {
        const unsigned long long key = vis[pix];
        if (key != kMissKey) vis[pix] = kMissKey;
        if (key != kMissKey && (uint32_t)key < ntri) { // (the second test never fails for keys the first pass wrote)
            // (with a cached mesh the winner is assembled again from it: no look-up of its place in the sorted arrays)
            uint32_t slot = 0u;
            float4 A;
            if (mesh.xyzw) {
                assemble_record(mesh, (uint32_t)key, A, Bw, Cw);
            } else {
                slot = slot_of[(uint32_t)key];
                A = TA[slot], Bw = TB[slot], Cw = TC[slot];
            }
            mt_test(f, ray0, A, Bw, Cw, slot, first); // the same test again: the same key, now with its terms
            if (first.key != kMissKey) shade_hit(f, first, A, Bw, Cw, r, g, b);
        }
    }
    if (ao.rays) {
        // finish_pixel queues hit pixels with a wave-wide ranking: lanes without a pixel take part with a miss
        Outputs o = out;
        if (!live) o.color = nullptr, o.depth = nullptr, o.primid = nullptr;
        finish_pixel(f, ao, o, i, pix, ray0, first, Bw, Cw, r, g, b);
        return;
    }
    if (live) finish_pixel(f, ao, out, i, pix, ray0, first, Bw, Cw, r, g, b);
}

// The same for whole row-major frames whose rows and buffers allow 16-byte accesses, without an occlusion pass: four
// pixels of a row per lane, 16 bytes per attachment and store (write-through and non-temporal, like the clear: five
// sixths of these pixels are background).
// A wave takes a 16 x 16 pixel square, a lane four pixels of one of its rows: neighbouring pixels are won by
// triangles that lie close together in the sorted arrays, so the square's gathers (slot_of, the winners' records)
// share cache lines where a run of 256 pixels along one row would not.
// Where the scene covers the square, two in five of its pixels have a winner: a lane that shaded its own four one
// after the other ran every step for whichever of the wave's lanes had a hit in that place.  The wave lists its hit
// pixels in LDS instead and shades them 64 at a time, a lane per listed pixel, whosever it is; the colours go back
// through LDS to the lanes that store them.
__global__ __launch_bounds__(256) void k_scatter_resolve4(FrameK f, MeshSource mesh, const float4* __restrict__ TA,
                                                          const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                          const uint32_t* __restrict__ slot_of, uint32_t ntri,
                                                          unsigned long long* __restrict__ vis, Outputs out) {
    // per wave: the square's winners by pixel (the primitive id of the key; pixel 4 * lane + j is lane's j-th), the list
    // of the pixels that have one, and the three values of every pixel
    __shared__ u32x4_t s_prim[4][64];
    __shared__ uint8_t s_list[4][256];
    __shared__ u32x4_t s_val[4][3][64];
    const uint32_t squares_x = (f.w + 15u) >> 4, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sq = blockIdx.x * 4u + wave;
    const uint32_t sx0 = (sq % squares_x) * 16u, sy0 = (sq / squares_x) * 16u;
    const uint32_t px0 = sx0 + (lane & 3u) * 4u, py = sy0 + (lane >> 2);
    const bool in_frame = px0 < f.w && py < f.h; // (the frame's width is a multiple of four)
    const uint32_t pix0 = py * f.w + px0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2* const vp = (u64x2*)(vis + pix0);
    const u64x2 miss2 = {kMissKey, kMissKey};
    u64x2 k01 = miss2, k23 = miss2;
    if (in_frame) k01 = vp[0], k23 = vp[1];
    const unsigned long long keys[4] = {k01.x, k01.y, k23.x, k23.y};
    const uint32_t bg = background_pixel(f.flags);
    u32x4_t col = {bg, bg, bg, bg}, dep = {0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u}, pid = {~0u, ~0u, ~0u, ~0u};
    const bool some = k01.x != kMissKey || k01.y != kMissKey || k23.x != kMissKey || k23.y != kMissKey;
    if (some) vp[0] = miss2, vp[1] = miss2;
    if (__ballot(some) != 0ull) { // some pixel of the square was hit (the same for the whole wave)
        uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool hit = keys[j] != kMissKey && (uint32_t)keys[j] < ntri; // (the second test never fails for keys the first pass wrote)
            const uint64_t m = __ballot(hit);
            if (hit) s_list[wave][n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)(4u * lane + (uint32_t)j);
            n += (uint32_t)__popcll(m);
        }
        s_prim[wave][lane] = u32x4_t{(uint32_t)k01.x, (uint32_t)k01.y, (uint32_t)k23.x, (uint32_t)k23.y};
        s_val[wave][0][lane] = col, s_val[wave][1][lane] = dep, s_val[wave][2][lane] = pid;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t e0 = 0; e0 < n; e0 += 64u) { // (wave-uniform)
            if (e0 + lane < n) {
                const uint32_t at = s_list[wave][e0 + lane];
                const uint32_t prim = ((const uint32_t*)&s_prim[wave][0])[at];
                const uint32_t px = sx0 + ((at >> 2) & 3u) * 4u + (at & 3u), pyy = sy0 + (at >> 4);
                // (with a cached mesh the winner is assembled again from it: no look-up of its place in the sorted arrays)
                uint32_t slot = 0u;
                float4 A, B, C;
                if (mesh.xyzw) { // (the same for every lane)
                    assemble_record(mesh, prim, A, B, C);
                } else {
                    slot = slot_of[prim];
                    A = TA[slot], B = TB[slot], C = TC[slot];
                }
                float r = 0.0f, g = 0.0f, b = 0.0f; // clear colour (src/invert.cpp:742)
                Hit first = miss_hit();
                const Ray ray0 = make_ray(f, px, pyy, 0.5f, 0.5f);
                mt_test(f, ray0, A, B, C, slot, first); // the same test again: the same key, now with its terms
                if (first.key != kMissKey) shade_hit(f, first, A, B, C, r, g, b);
                uint32_t* const v = (uint32_t*)&s_val[wave][0][0];
                v[at] = pack_pixel(f.flags, r, g, b), v[256u + at] = (uint32_t)(first.key >> 32), v[512u + at] = (uint32_t)first.key;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        col = s_val[wave][0][lane], dep = s_val[wave][1][lane], pid = s_val[wave][2][lane];
    }
    if (!in_frame) return;
#define INV_ST16_NT(p, v) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory")
    if (out.color) INV_ST16_NT((uint32_t*)out.color + pix0, col);
    if (out.depth) INV_ST16_NT((uint32_t*)out.depth + pix0, dep);
    if (out.primid) INV_ST16_NT(out.primid + pix0, pid);
#undef INV_ST16_NT
}

__global__ __launch_bounds__(256) void k_vis_fill(unsigned long long* __restrict__ vis, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) vis[i] = kMissKey;
}

// ------------------------------------------------------------------------------------------------
// occlusion pass over the pixels the primary kernel queued
// ------------------------------------------------------------------------------------------------
// Scenes without an LBVH (fewer than two triangles): one wave per queued pixel,
// one lane per ray, every triangle tested.
__global__ __launch_bounds__(256) void k_ao_brute(FrameK f, AoK ao, const float4* __restrict__ TA,
                                                  const float4* __restrict__ TB, const float4* __restrict__ TC,
                                                  uint32_t ntri, Outputs out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    const uint32_t count = *ao.count;
    const float4* queue = (const float4*)ao.queue;
    for (uint32_t rec = wave; rec < count; rec += nwaves) { // wave-uniform
        const float4 q0 = queue[3 * (size_t)rec], q1 = queue[3 * (size_t)rec + 1], q2 = queue[3 * (size_t)rec + 2];
        const float n[3] = {q1.x, q1.y, q1.z};
        const uint32_t pix = __float_as_uint(q2.y);
        const uint64_t oi = (uint64_t)__float_as_uint(q2.z) | ((uint64_t)__float_as_uint(q2.w) << 32);
        const uint32_t hp = hash32(pix ^ hash32(f.seed ^ 0xA0761D65u));
        uint32_t occ = 0;
        for (uint32_t base = 0; base < ao.rays; base += 64) { // wave-uniform trip count
            const uint32_t k = base + lane;
            bool hit = false;
            if (k < ao.rays) {
                float w[3];
                ao_direction(n, hp, k, ao.rays, w);
                for (uint32_t t = 0; t < ntri && !hit; t++)
                    hit = mt_any(f, q0.x, q0.y, q0.z, w[0], w[1], w[2], ao.tmax, TA[t], TB[t], TC[t]);
            }
            occ += (uint32_t)__popcll(__ballot(hit));
        }
        if (lane == 0 && out.color) {
            const float a = 1.0f - (float)occ / (float)ao.rays;
            ((uint32_t*)out.color)[oi] = pack_pixel(f.flags, q0.w * a, q1.w * a, q2.x * a);
        }
    }
}

// LBVH scenes: a stream of rays, 64 at a time ("chunks", dealt round-robin to the waves of a resident grid).
//  * A wave sets a chunk's rays up at full width (record, direction, reciprocals) into its own LDS buffer; a lane whose
//    ray has ended (the first occluder ends it) takes the next ray out of that buffer as soon as kAoPull lanes are
//    idle: taking a ray is ten LDS reads, so the threshold can be low and few lanes wait for work.
//  * Each step the wave does one thing: a node (two child boxes) for the lanes that stand at a node, or a leaf (one
//    triangle).  A lane that reaches a leaf sets it aside (one leaf per lane) and goes on with the next node from its
//    stack; only a lane that reaches a second leaf has to wait.  A leaf step is taken when kAoLeafNum eighths of the
//    live lanes wait, and then every lane with a leaf set aside tests it.
// The set of leaves a ray reaches does not depend on the visiting order (a ray that is occluded by a leaf it set
// aside has only looked at a few more boxes than it had to), and the per-pixel result is a sum of hits, so the image
// does not depend on the scheduling.
#ifdef INV_AO_STATS
__device__ unsigned long long g_ao_stats[16];
#define AO_STAT(i, v) st[i] += (v)
#else
#define AO_STAT(i, v) ((void)0)
#endif
constexpr uint32_t kAoRayWords = 10; // per buffered ray: origin, direction, reciprocal direction, record
constexpr uint32_t kNoNode = 0x7FFFFFFFu;

// The lanes of the wave for which `p` holds (the compiler's own ballot: a scalar AND with the execution mask where
// __ballot goes through a vector compare).
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// One entry of a node record (a box on the scene's 16-bit grid and a child reference) against the ray's slab terms.
__device__ __forceinline__ bool slab_entry(const RaySlabs& rs, const uint4& e, float tmax, float& tnear) {
    return slab_folded(rs, (float)(e.x & 0xFFFFu), (float)(e.x >> 16), (float)(e.y & 0xFFFFu), (float)(e.y >> 16),
                       (float)(e.z & 0xFFFFu), (float)(e.z >> 16), tmax, tnear);
}

// WIDE: the node records are the four-wide ones (an internal node's grandchildren, or a child that is a leaf, in up
// to four entries on the scene's 16-bit grid; unused entries refer to kNoNode): half as many node steps per ray, each
// with four box tests.  Otherwise they are the LBVH's own two-wide float records (a caller that may not allocate
// the wide ones).
template <typename StackT, bool WIDE>
__global__ __launch_bounds__(256) void k_ao(FrameK f, AoK ao, const float4* __restrict__ TA,
                                            const float4* __restrict__ TB, const float4* __restrict__ TC,
                                            const void* __restrict__ node_records, QGrid grid, Outputs out) {
    extern __shared__ uint32_t s_dyn[]; // [4 waves][kAoRayWords][64] ray buffers, then the stacks: one column per lane
    const uint32_t lane = threadIdx.x & 63u;
    float* const rb = (float*)s_dyn + (threadIdx.x >> 6) * (kAoRayWords * 64u);
    StackT* const stack = (StackT*)(s_dyn + 4u * kAoRayWords * 64u) + threadIdx.x;
    // a child reference on the stack: scenes of fewer than 32 768 triangles keep it in 16 bits (leaf flag in bit 15)
    const auto packed = [](uint32_t c) -> StackT {
        return sizeof(StackT) == 4 ? (StackT)c : (StackT)((c & 0x7FFFu) | ((c >> 16) & 0x8000u));
    };
    const auto unpacked = [](StackT v) -> uint32_t {
        return sizeof(StackT) == 4 ? (uint32_t)v : (((uint32_t)v & 0x7FFFu) | (((uint32_t)v & 0x8000u) << 16));
    };
    const uint64_t total = (uint64_t)(*ao.count) * ao.rays;
    const uint64_t nchunks = (total + (kAoChunk - 1)) / kAoChunk;
    const bool narrow = total < 0xFFFFFFFFull; // ray index -> (record, ray) without a 64-bit division
    const float4* queue = (const float4*)ao.queue;
    const float grow = 2.0f * f.pad, tlim = ao.tmax * 1.0001f;
    const uint32_t hseed = hash32(f.seed ^ 0xA0761D65u);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4u;
    // the chunk this wave sets up next: the same in every lane, and said so, so that it and what follows from it (the
    // buffer's fill state, the tests on it) are kept in scalar registers
    uint64_t chunk = (uint64_t)blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t bcur = 0, bend = 0;   // rays of the wave's buffer not yet handed to a lane (wave-uniform)
    uint32_t active = 0; // 1 while the lane holds a ray (an integer: its ballot is then a single vector compare)
#ifdef INV_AO_STATS
    unsigned long long st[12] = {};
#endif
    uint32_t rec = 0, node = 0, sp = 0, pend = 0;
    float ox = 0, oy = 0, oz = 0, wx = 0, wy = 0, wz = 0;
    RaySlabs rs = {}; // in grid units
    for (;;) {
        const uint64_t idle = ballot64(active == 0u);
        const uint32_t nidle = (uint32_t)__builtin_popcount((uint32_t)idle) + (uint32_t)__builtin_popcount((uint32_t)(idle >> 32));
        if (nidle >= (uint32_t)kAoPull && (bcur < bend || chunk < nchunks)) {
            AO_STAT(6, 1);
            if (bcur == bend) { // set the next chunk's rays up, every lane one
                AO_STAT(7, 1);
                const uint64_t base = chunk * kAoChunk;
                const uint32_t n = total - base < kAoChunk ? (uint32_t)(total - base) : (uint32_t)kAoChunk;
                if (lane < n) {
                    const uint64_t g = base + lane;
                    uint32_t r, k;
                    if (narrow)
                        r = (uint32_t)g / ao.rays, k = (uint32_t)g % ao.rays;
                    else
                        r = (uint32_t)(g / ao.rays), k = (uint32_t)(g % ao.rays);
                    const float4 q0 = queue[3 * (size_t)r], q1 = queue[3 * (size_t)r + 1];
                    const uint32_t pix = __float_as_uint(queue[3 * (size_t)r + 2].y);
                    const float nrm[3] = {q1.x, q1.y, q1.z};
                    float w[3];
                    ao_direction(nrm, hash32(pix ^ hseed), k, ao.rays, w);
                    rb[0 * 64 + lane] = q0.x, rb[1 * 64 + lane] = q0.y, rb[2 * 64 + lane] = q0.z;
                    rb[3 * 64 + lane] = w[0], rb[4 * 64 + lane] = w[1], rb[5 * 64 + lane] = w[2];
                    rb[6 * 64 + lane] = slab_reciprocal(w[0]);
                    rb[7 * 64 + lane] = slab_reciprocal(w[1]);
                    rb[8 * 64 + lane] = slab_reciprocal(w[2]);
                    rb[9 * 64 + lane] = __uint_as_float(r);
                }
                bcur = 0, bend = n;
                chunk += nwaves;
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t idx = bcur + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!active && idx < bend) {
                ox = rb[0 * 64 + idx], oy = rb[1 * 64 + idx], oz = rb[2 * 64 + idx];
                wx = rb[3 * 64 + idx], wy = rb[4 * 64 + idx], wz = rb[5 * 64 + idx];
                const float ix = rb[6 * 64 + idx], iy = rb[7 * 64 + idx], iz = rb[8 * 64 + idx];
                rec = __float_as_uint(rb[9 * 64 + idx]);
                // plane k of axis a is met at (org + k * step - o) / d = k * (step / d) - (o - org) / d
                rs.ix = grid.step[0] * ix, rs.iy = grid.step[1] * iy, rs.iz = grid.step[2] * iz;
                rs.clx = ((ox + grow) - grid.org[0]) * ix, rs.cly = ((oy + grow) - grid.org[1]) * iy;
                rs.clz = ((oz + grow) - grid.org[2]) * iz;
                rs.chx = ((ox - grow) - grid.org[0]) * ix, rs.chy = ((oy - grow) - grid.org[1]) * iy;
                rs.chz = ((oz - grow) - grid.org[2]) * iz;
                node = 0, sp = 0, pend = 0;
                active = 1u;
            }
            bcur = bcur + nidle < bend ? bcur + nidle : bend;
            __builtin_amdgcn_wave_barrier();
        }
        const uint64_t live = ballot64(active != 0u);
        if (!live) {
            if (bcur == bend && chunk >= nchunks) break;
            continue;
        }
        AO_STAT(0, 1), AO_STAT(1, __popcll(live));
        const bool at_node = active && !(node & 0x80000000u) && node != kNoNode;
        const bool waits = active && !at_node; // stands at a leaf with another set aside, or has only that one left
        const bool leaf_step = 8u * (uint32_t)__popcll(ballot64(waits)) > (uint32_t)kAoLeafNum * (uint32_t)__popcll(live);
        bool ended = false, hit = false;
        if (leaf_step)
            AO_STAT(2, 1), AO_STAT(3, __popcll(ballot64(kAoPend ? (active && pend) : waits)));
        else
            AO_STAT(4, 1), AO_STAT(5, __popcll(ballot64(at_node)));
        if (leaf_step) {
            if (kAoPend ? (active && pend) : waits) {
                const uint32_t k = (kAoPend ? pend : node) & 0x7FFFFFFFu;
                if (mt_any(f, ox, oy, oz, wx, wy, wz, ao.tmax, TA[k], TB[k], TC[k])) hit = true, ended = true;
                pend = 0;
                if (!kAoPend) node = sp ? unpacked(stack[(size_t)(--sp) * 256]) : kNoNode;
            }
        } else if (at_node) {
            if constexpr (WIDE) {
                const uint4* const w = (const uint4*)node_records + 4 * (size_t)node;
                const uint4 e0 = w[0], e1 = w[1], e2 = w[2], e3 = w[3];
                float t0, t1, t2, t3;
                const bool h0 = slab_entry(rs, e0, tlim, t0), h1 = slab_entry(rs, e1, tlim, t1) && e1.w != kNoNode;
                const bool h2 = slab_entry(rs, e2, tlim, t2) && e2.w != kNoNode, h3 = slab_entry(rs, e3, tlim, t3) && e3.w != kNoNode;
                // the nearest of the hit children goes on (an occluder close to the surface ends the ray soonest), the
                // others wait on the stack
                float tb = INFINITY;
                uint32_t nb = kNoNode;
                if (h0) tb = t0, nb = e0.w;
                if (h1 && t1 < tb) tb = t1, nb = e1.w;
                if (h2 && t2 < tb) tb = t2, nb = e2.w;
                if (h3 && t3 < tb) tb = t3, nb = e3.w;
                // every entry is written at the stack's top and the top moves only past those that wait: the next write
                // lands on an entry that does not (the stack has one entry of slack for the last)
                stack[(size_t)sp * 256] = packed(e0.w), sp += (uint32_t)(h0 && e0.w != nb);
                stack[(size_t)sp * 256] = packed(e1.w), sp += (uint32_t)(h1 && e1.w != nb);
                stack[(size_t)sp * 256] = packed(e2.w), sp += (uint32_t)(h2 && e2.w != nb);
                stack[(size_t)sp * 256] = packed(e3.w), sp += (uint32_t)(h3 && e3.w != nb);
                if (nb != kNoNode)
                    node = nb;
                else if (sp == 0)
                    node = kNoNode;
                else
                    node = unpacked(stack[(size_t)(--sp) * 256]);
            } else {
                // the LBVH's own node records (float boxes; the launcher passes the grid that leaves coordinates as they are)
                const float4* const nd = (const float4*)node_records + 4 * (size_t)node;
                const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
                const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                float tn0, tn1;
                const bool h0 = slab_folded(rs, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tlim, tn0);
                const bool h1 = slab_folded(rs, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tlim, tn1);
                if (h0 && h1) { // nearer child first: an occluder close to the surface ends the ray soonest
                    const bool first1 = tn1 < tn0;
                    stack[(size_t)sp * 256] = packed(first1 ? c0 : c1);
                    sp++;
                    node = first1 ? c1 : c0;
                } else if (h0) {
                    node = c0;
                } else if (h1) {
                    node = c1;
                } else if (sp == 0) {
                    node = kNoNode;
                } else {
                    node = unpacked(stack[(size_t)(--sp) * 256]);
                }
            }
        }
        if (active && !ended) {
            if (kAoPend && (node & 0x80000000u) && pend == 0) { // a leaf, and none set aside yet: set it aside and go on
                pend = node;
                node = sp ? unpacked(stack[(size_t)(--sp) * 256]) : kNoNode;
            }
            if (node == kNoNode && pend == 0) ended = true;
        }
        if (ended) {
            // the result is not read back here, so the wave does not wait for the atomic
            if (hit) __hip_atomic_fetch_add(&ao.acc[rec], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            active = 0u;
            AO_STAT(8, 1);
        }
    }
#ifdef INV_AO_STATS
    if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_ao_stats[i], st[i]);
    atomicAdd(&g_ao_stats[8], st[8]);
#endif
}

// After the ray stream: one lane per queued pixel turns its occlusion count into the stored colour.
__global__ __launch_bounds__(256) void k_ao_resolve(FrameK f, AoK ao, Outputs out) {
    const uint32_t count = *ao.count;
    const float4* queue = (const float4*)ao.queue;
    for (uint32_t rec = blockIdx.x * 256u + threadIdx.x; rec < count; rec += gridDim.x * 256u) {
        const float4 q0 = queue[3 * (size_t)rec], q1 = queue[3 * (size_t)rec + 1], q2 = queue[3 * (size_t)rec + 2];
        const uint64_t oi = (uint64_t)__float_as_uint(q2.z) | ((uint64_t)__float_as_uint(q2.w) << 32);
        const float a = 1.0f - (float)ao.acc[rec] / (float)ao.rays;
        ((uint32_t*)out.color)[oi] = pack_pixel(f.flags, q0.w * a, q1.w * a, q2.x * a);
    }
}

} // namespace

// Small scenes: blocks test the whole fine cut themselves (TilePlan::flat).  INV_FLAT=0 / 1 in the environment: never /
// up to 2048 records (measurements); otherwise up to kFlatMax records.
constexpr uint32_t kFlatMax = 1024;
static bool flat_wanted(const DeviceScene& sc) {
    static const int env = [] {
        const char* e = std::getenv("INV_FLAT");
        return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    if (!sc.has_bvh || sc.nfine == 0 || env == 0) return false;
    return sc.nfine <= (env == 1 ? 2048u : kFlatMax);
}

// Grid terms of one frame of the tile kernel and the way its traced rectangle is worked through.
static TilePlan plan_tile_frame(const FrameK& f, const DeviceScene& sc, const Outputs& out, bool batch, int* mode) {
    TilePlan pl;
    pl.bin_count = nullptr, pl.bin_list = nullptr, pl.ntri = sc.ntri;
    pl.flat = nullptr, pl.nflat = 0;
    if (flat_wanted(sc)) pl.flat = (const float4*)sc.fine, pl.nflat = sc.nfine;
    if (f.rowmajor)
        pl.pixel_blocks = ((f.w + 31u) >> 5) * ((f.h + 7u) >> 3);
    else
        pl.pixel_blocks = f.slots * (f.tile >> 5) * (f.tile >> 3);
    const auto aligned16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    pl.vec4 = (f.w % 4u == 0) && aligned16(out.color) && aligned16(out.depth) && aligned16(out.primid);
    const bool any = f.rx1 > f.rx0 && f.ry1 > f.ry0;
    // Where triangles are much smaller than a pixel a block has more candidates than the record list holds: such
    // frames take the variant that works through them in batches, in 8x8 tiles.
    const double rect_px = (double)(f.rx1 - f.rx0) * (double)(f.ry1 - f.ry0);
    const double per_tile = any ? (double)sc.ntri * 64.0 / rect_px : 0.0; // triangles per 8x8 pixels, if spread evenly
    // Blocks do the least arithmetic per pixel but a wave per quadrant is a longer chain of work than a tile's four
    // waves sharing it: a draw on its own is over sooner with tiles (12 against 21 us for the 1080p bunny), several
    // frames sharing a launch get through sooner with blocks (4.0 against 5.1 us per frame).
    // (INV_BLOCKS asks for blocks whatever the frame looks like: always correct, since a block whose lists overflow
    // hands its pixels to the per-lane traversal, but slow where a block meets hundreds of triangles.)
    const bool blocks = (f.flags & INV_BLOCKS) || (batch && f.spp < 4u && per_tile <= 24.0);
    *mode = blocks ? kBlocks : per_tile > 48.0 ? kBatched : kTiles;
    pl.block_px = *mode == kBlocks ? 16u : 8u;
    const uint32_t sh = *mode == kBlocks ? 4u : 3u, e1 = pl.block_px - 1u;
    const uint32_t tiles_x = any ? ((f.rx1 + e1) >> sh) - (f.rx0 >> sh) : 0u;
    const uint32_t tiles_y = any ? ((f.ry1 + e1) >> sh) - (f.ry0 >> sh) : 0u;
    pl.tiles_x = tiles_x, pl.ntiles = tiles_x * tiles_y;
    const uint64_t fill_work = pl.vec4 ? ((f.npix >> 2) + kFillRun * 4u - 1u) / (kFillRun * 4u) : pl.pixel_blocks;
    pl.fill_blocks = (uint32_t)std::min<uint64_t>(fill_work, 2048u);
    const uint32_t row4 = std::max(1u, (f.rowmajor ? f.w : f.tile) >> 2);
    pl.row4_magic = row4 == 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / row4);
    return pl;
}

// ---- the binning pass: when it runs, and where its bins live ----
// INV_BINS=0 / 1 in the environment: never / whenever possible (measurements); otherwise scenes whose coarse cut is
// capped (their blocks would walk the tree level by level from it) are binned, and small scenes are not: the pass is a
// launch of its own, some five microseconds in front of draws that take four to ten, and what it saves their blocks
// (one of three dependent round trips) is less than that (measured: 3.7 -> 4.4 us per frame for batches of the
// 1080p bunny, 11 -> 16 us for a draw on its own; 0.67 -> 0.52 ms for the 5 M-triangle scene).
static int bins_env() {
    static const int v = [] {
        const char* e = std::getenv("INV_BINS");
        return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    return v;
}

// Every corner of the scene's box (grown) in front of the eye: otherwise most boxes reach the eye's plane and would
// be listed for every block.
static bool eye_outside_scene(const FrameK& f, const DeviceScene& sc) {
    const float grow = 4.0f * f.pad;
    for (int c = 0; c < 8; c++) {
        const float x = (c & 1) ? sc.bbox[3] + grow : sc.bbox[0] - grow, y = (c & 2) ? sc.bbox[4] + grow : sc.bbox[1] - grow;
        const float z = (c & 4) ? sc.bbox[5] + grow : sc.bbox[2] - grow;
        const float zv = (f.mv[8] * x + f.mv[9] * y) + (f.mv[10] * z + f.mv[11]);
        if (!(-zv > 1e-6f * ((std::fabs(x) + std::fabs(y)) + (std::fabs(z) + 1.0f)))) return false;
    }
    return true;
}

static size_t bin_counts_bytes(uint64_t nbins) { return Arena::aligned((size_t)nbins * 4); }
size_t bin_mem_bytes(uint64_t nbins) { return nbins ? bin_counts_bytes(nbins) + (size_t)nbins * kRangeCap * sizeof(uint2) : 0; }

uint32_t tile_bins_wanted(const FrameK& f, const DeviceScene& sc, const Outputs& out, bool batch) {
    if (f.npix == 0 || !sc.has_bvh || sc.nfine == 0) return 0;
    if (f.flags & (INV_LANE_TRAVERSAL | INV_BRUTE_FORCE)) return 0;
    const int env = bins_env();
    (void)batch;
    if (env == 0 || (env < 0 && sc.ntri <= kCutCap * 8u)) return 0; // (a cut of 2048 subtrees of 8 leaves is not capped)
    if (!eye_outside_scene(f, sc)) return 0;
    int mode = kBlocks;
    const TilePlan pl = plan_tile_frame(f, sc, out, batch, &mode);
    if (bin_mem_bytes(pl.ntiles) > ((size_t)1 << 30)) return 0; // a gigabyte of lists: start from the coarse cut instead
    return pl.ntiles;
}

// Points the plan at bins [first, first + pl.ntiles) of `bins` (sized by the caller from tile_bins_wanted).
static bool attach_bins(TilePlan& pl, const BinMem* bins, uint32_t first, uint32_t wanted) {
    if (!bins || !bins->base || wanted == 0 || wanted != pl.ntiles || (uint64_t)first + wanted > bins->nbins) return false;
    pl.bin_count = (uint32_t*)bins->base + first;
    pl.bin_list = (uint2*)((char*)bins->base + bin_counts_bytes(bins->nbins)) + (size_t)first * kRangeCap;
    return true;
}

static void launch_bin_ranges(hipStream_t s, const Batch& b, const DeviceScene& sc) {
    // small scenes: fewer records per workgroup, so that their (record, block) pairs spread over many workgroups
    const uint32_t rpw = sc.nfine > 65536u ? 256u : sc.nfine > 8192u ? 64u : 16u;
    hipLaunchKernelGGL(k_bin_ranges, dim3((sc.nfine + rpw - 1u) / rpw, b.nframes), dim3(256), 0, s, b, (const float4*)sc.fine,
                       sc.nfine, rpw);
}

// ---- one lane per triangle: when ----
// INV_SCATTER=0 / 1 in the environment: never / whenever possible (tests, measurements); otherwise the frames for
// which the tile kernel would work through its candidates in batches (triangles far smaller than a pixel).
static int scatter_env() {
    static const int v = [] {
        const char* e = std::getenv("INV_SCATTER");
        return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    return v;
}

bool mesh_draw_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("INV_MESH_DRAW");
        return !(e && e[0] == '0');
    }();
    return on;
}

bool scatter_wanted(const FrameK& f, const DeviceScene& sc, const Outputs& out) {
    // (the resolve pass finds a winner's triangle through slot_of or in the cached mesh)
    if (f.npix == 0 || !sc.has_bvh || !(sc.slot_of || (sc.mesh_cached && mesh_draw_enabled())) || f.spp != 1u) return false;
    if (f.flags & (INV_LANE_TRAVERSAL | INV_BRUTE_FORCE | INV_BLOCKS | INV_TILE_TRAVERSAL)) return false;
    const int env = scatter_env();
    if (env == 0 || !eye_outside_scene(f, sc)) return false;
    if (env > 0) return true;
    int mode = kBlocks;
    (void)plan_tile_frame(f, sc, out, false, &mode);
    return mode == kBatched;
}

hipError_t launch_vis_fill(hipStream_t s, unsigned long long* vis, uint64_t n) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_vis_fill, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, vis, n);
    return hipGetLastError();
}

hipError_t launch_render_bvh(hipStream_t s, const FrameK& f, const DeviceScene& sc, const AoK& ao,
                             const Outputs& out, const BinMem* bins, unsigned long long* vis) {
    if (f.npix == 0) return hipSuccess;
    if (!sc.has_bvh) return launch_render_brute(s, f, sc, ao, out);
    if (f.flags & INV_LANE_TRAVERSAL) {
        uint32_t blocks;
        if (f.rowmajor) {
            blocks = ((f.w + 15u) >> 4) * ((f.h + 15u) >> 4);
        } else {
            const uint32_t per = f.tile >> 4;
            blocks = f.slots * per * per;
        }
        hipLaunchKernelGGL(k_render_bvh, dim3(blocks), dim3(256), 0, s, f, (const float4*)sc.triA,
                           (const float4*)sc.triB, (const float4*)sc.triC, (const float4*)sc.nodes, ao, out);
        return hipGetLastError();
    }
    if (vis && scatter_wanted(f, sc, out)) {
        // a mesh that the caches hold: the resolve pass assembles the winners from it instead of looking them up in the
        // sorted arrays (INV_MESH_DRAW=0 in the environment: never; for measurements)
        const MeshSource mesh = mesh_source(sc.mesh_cached && mesh_draw_enabled() ? (const float4*)sc.xyzw : nullptr, sc.idx,
                                                 sc.translations, sc.base_ntri, sc.ninst);
        // (nothing to try when the scene lies outside the frame: the resolve pass then stores clear values only)
        if (f.rx1 > f.rx0 && f.ry1 > f.ry0) {
            const dim3 grid((sc.ntri + 256u * kScatterPerLane - 1u) / (256u * kScatterPerLane));
            if ((uint64_t)sc.ntri * 48u <= kScatterSixBytes)
                hipLaunchKernelGGL(k_scatter_triangles<6>, grid, dim3(256), 0, s, f, (const float4*)sc.triA, (const float4*)sc.triB,
                                   (const float4*)sc.triC, sc.ntri, vis);
            else
                hipLaunchKernelGGL(k_scatter_triangles<5>, grid, dim3(256), 0, s, f, (const float4*)sc.triA, (const float4*)sc.triB,
                                   (const float4*)sc.triC, sc.ntri, vis);
        }
        const auto aligned16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
        if (f.rowmajor && f.nranks == 1 && !ao.rays && f.w % 4u == 0 && aligned16(out.color) && aligned16(out.depth) &&
            aligned16(out.primid)) {
            const unsigned resolve_blocks = (((f.w + 15u) >> 4) * ((f.h + 15u) >> 4) + 3u) / 4u; // a wave per 16 x 16 square
            hipLaunchKernelGGL(k_scatter_resolve4, dim3(resolve_blocks), dim3(256), 0, s, f, mesh,
                               (const float4*)sc.triA, (const float4*)sc.triB, (const float4*)sc.triC,
                               (const uint32_t*)sc.slot_of, sc.ntri, vis, out);
        } else
            hipLaunchKernelGGL(k_scatter_resolve, dim3((unsigned)((f.npix + 255) / 256)), dim3(256), 0, s, f, mesh,
                               (const float4*)sc.triA, (const float4*)sc.triB, (const float4*)sc.triC,
                               (const uint32_t*)sc.slot_of, sc.ntri, vis, ao, out);
        return hipGetLastError();
    }
    int mode = kBlocks;
    TilePlan pl = plan_tile_frame(f, sc, out, false, &mode);
    if (pl.ntiles + pl.fill_blocks == 0) return hipSuccess;
    if (attach_bins(pl, bins, 0, tile_bins_wanted(f, sc, out, false))) {
        Batch b;
        std::memset(&b, 0, sizeof b);
        b.nframes = 1, b.fr[0].f = f, b.fr[0].pl = pl, b.fr[0].out = out;
        launch_bin_ranges(s, b, sc);
    }
#define INV_LAUNCH_TILE(MODE)                                                                                        \
    hipLaunchKernelGGL((k_render_tile<MODE>), dim3((MODE == kBatched ? xcd_run_grid(pl.ntiles) : pl.ntiles) + pl.fill_blocks), dim3(256), 0, s, f, pl, \
                       (const float4*)sc.triA, (const float4*)sc.triB, (const float4*)sc.triC, (const float4*)sc.nodes, \
                       (const float4*)sc.cut, (const float4*)sc.cut_groups, sc.ncut, ao, out)
    if (mode == kBatched)
        INV_LAUNCH_TILE(kBatched);
    else if (mode == kTiles)
        INV_LAUNCH_TILE(kTiles);
    else
        INV_LAUNCH_TILE(kBlocks);
#undef INV_LAUNCH_TILE
    return hipGetLastError();
}

int tile_mode_for(const FrameK& f, const DeviceScene& sc, const Outputs& out, bool batch) {
    int mode = kBlocks;
    (void)plan_tile_frame(f, sc, out, batch, &mode);
    return mode; // 0: 16x16 blocks, 1: 8x8 tiles, 2: 8x8 tiles worked through in batches
}

bool tile_batchable(const FrameK& f, const DeviceScene& sc, const Outputs& out) {
    if (f.npix == 0 || !sc.has_bvh) return false;
    if (f.flags & (INV_LANE_TRAVERSAL | INV_BRUTE_FORCE)) return false;
    int mode = kBlocks;
    (void)plan_tile_frame(f, sc, out, true, &mode);
    return mode == kBlocks;
}

hipError_t launch_render_batch(hipStream_t s, const FrameK* frames, const Outputs* outs, uint32_t n,
                               const DeviceScene& sc, const BinMem* bins) {
    if (n == 0) return hipSuccess;
    if (n > kBatchMax) return hipErrorInvalidValue;
    Batch b;
    std::memset(&b, 0, sizeof b);
    b.nframes = n;
    uint32_t blocks = 0, first_bin = 0;
    bool binned = false;
    for (uint32_t i = 0; i < n; i++) {
        int mode = kBlocks;
        b.fr[i].f = frames[i];
        b.fr[i].out = outs[i];
        b.fr[i].pl = plan_tile_frame(frames[i], sc, outs[i], true, &mode);
        if (mode != kBlocks) return hipErrorInvalidValue; // the caller asks tile_batchable first
        const uint32_t wanted = tile_bins_wanted(frames[i], sc, outs[i], true);
        if (attach_bins(b.fr[i].pl, bins, first_bin, wanted)) binned = true, first_bin += wanted;
        blocks += b.fr[i].pl.ntiles + b.fr[i].pl.fill_blocks;
        b.block_end[i] = blocks;
    }
    for (uint32_t i = n; i < kBatchMax; i++) b.block_end[i] = blocks;
    if (blocks == 0) return hipSuccess;
    if (binned) launch_bin_ranges(s, b, sc); // one pass for all frames of the launch
    hipLaunchKernelGGL(k_render_tile_batch, dim3(blocks), dim3(256), 0, s, b, (const float4*)sc.triA,
                       (const float4*)sc.triB, (const float4*)sc.triC, (const float4*)sc.nodes, (const float4*)sc.cut,
                       (const float4*)sc.cut_groups, sc.ncut);
    return hipGetLastError();
}

} // namespace inv

namespace inv {
namespace {
// A child box on the 16-bit grid of QGrid: lower bounds rounded down and one step further, upper bounds rounded up
// and one step further, so that the quantised box holds the float box with room for the roundings of the index
// arithmetic and of the ray's folded plane terms.
__device__ __forceinline__ uint4 quantised_entry(const QGrid& g, const float lo[3], const float hi[3], uint32_t ref) {
    uint32_t ql[3], qh[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        ql[a] = (uint32_t)fminf(fmaxf(floorf((lo[a] - g.lo[a]) * g.inv[a]), 0.0f), 65532.0f);
        qh[a] = (uint32_t)fminf(fmaxf(ceilf((hi[a] - g.lo[a]) * g.inv[a]), 0.0f), 65533.0f) + 2u;
    }
    return make_uint4(ql[0] | (ql[1] << 16), ql[2] | (qh[0] << 16), qh[1] | (qh[2] << 16), ref);
}

// The two entries (child box, child reference) of an LBVH node record.
__device__ __forceinline__ void node_entries(const float4* __restrict__ nodes, uint32_t i, const QGrid& g, uint4& e0, uint4& e1) {
    const float4 a = nodes[4 * (size_t)i], b = nodes[4 * (size_t)i + 1], c = nodes[4 * (size_t)i + 2], d = nodes[4 * (size_t)i + 3];
    const float lo0[3] = {a.x, a.y, a.z}, hi0[3] = {a.w, b.x, b.y}, lo1[3] = {b.z, b.w, c.x}, hi1[3] = {c.y, c.z, c.w};
    e0 = quantised_entry(g, lo0, hi0, __float_as_uint(d.x));
    e1 = quantised_entry(g, lo1, hi1, __float_as_uint(d.y));
}

// An internal node's four-wide record: each child that is a leaf as it stands in the node's own record, each child
// that is an internal node replaced by that node's two children.
__global__ __launch_bounds__(256) void k_make_wnodes(const float4* __restrict__ nodes, uint32_t n_internal, QGrid g,
                                                     uint4* __restrict__ wnodes) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_internal) return;
    uint4* const dst = wnodes + 4 * (size_t)i;
    uint4 c[2];
    node_entries(nodes, i, g, c[0], c[1]);
    uint32_t pos = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (c[k].w & 0x80000000u) {
            dst[pos++] = c[k];
        } else {
            uint4 e0, e1;
            node_entries(nodes, c[k].w, g, e0, e1);
            dst[pos++] = e0;
            dst[pos++] = e1;
        }
    }
    while (pos < 4u) dst[pos++] = make_uint4(0u, 0u, 0u, kNoNode);
}
} // namespace

hipError_t launch_make_wnodes(hipStream_t s, const DeviceScene& sc, uint32_t* wnodes) {
    if (!sc.has_bvh || sc.ntri < 2u) return hipSuccess;
    hipLaunchKernelGGL(k_make_wnodes, dim3((sc.ntri - 1u + 255u) / 256u), dim3(256), 0, s, (const float4*)sc.nodes, sc.ntri - 1u,
                       qgrid_for(sc.bbox), (uint4*)wnodes);
    return hipGetLastError();
}

hipError_t launch_ao(hipStream_t s, const FrameK& f, const DeviceScene& sc, const AoK& ao, const Outputs& out, bool narrow) {
    if (!ao.rays || f.npix == 0) return hipSuccess;
    // fixed grids walk the queue (its length is only known on the device)
    // The occlusion kernel's per-lane stacks hold as many entries as a ray can have waiting, so they cannot overflow:
    // one per level with the two-wide records, up to three per two levels (and one of slack) with the four-wide ones.
    // A tree so deep that they do not fit the 64 KiB a workgroup may ask for goes the way of scenes without a tree.
    const bool wide = !narrow && sc.wnodes != nullptr;
    const bool small = sc.ntri < 32768u; // child references fit 16 bits
    const uint32_t entries = wide ? 3u * ((sc.height + 2u) / 2u) + 2u : sc.height + 1u;
    const size_t lds = (size_t)entries * 256u * (small ? 2u : 4u) + 4u * kAoRayWords * 64u * sizeof(uint32_t);
    if (!sc.has_bvh || lds > 64u * 1024u) {
        hipLaunchKernelGGL(k_ao_brute, dim3(1024), dim3(256), 0, s, f, ao, (const float4*)sc.triA,
                           (const float4*)sc.triB, (const float4*)sc.triC, sc.ntri, out);
        return hipGetLastError();
    }
    // as many workgroups as stay resident at once: 256 CUs x what the stack's LDS share allows (at most 8)
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / (lds + 512u)));
    const dim3 grid(256u * per_cu), block(256);
    const float4 *A = (const float4*)sc.triA, *B = (const float4*)sc.triB, *C = (const float4*)sc.triC;
    const void* rec = wide ? (const void*)sc.wnodes : (const void*)sc.nodes;
    QGrid qg = qgrid_for(sc.bbox);
    if (!wide) // the LBVH's float records: index = coordinate
        for (int a = 0; a < 3; a++) qg.lo[a] = 0.0f, qg.org[a] = 0.0f, qg.step[a] = 1.0f, qg.inv[a] = 1.0f;
    if (small && wide)
        hipLaunchKernelGGL((k_ao<uint16_t, true>), grid, block, lds, s, f, ao, A, B, C, rec, qg, out);
    else if (small)
        hipLaunchKernelGGL((k_ao<uint16_t, false>), grid, block, lds, s, f, ao, A, B, C, rec, qg, out);
    else if (wide)
        hipLaunchKernelGGL((k_ao<uint32_t, true>), grid, block, lds, s, f, ao, A, B, C, rec, qg, out);
    else
        hipLaunchKernelGGL((k_ao<uint32_t, false>), grid, block, lds, s, f, ao, A, B, C, rec, qg, out);
    if (out.color) hipLaunchKernelGGL(k_ao_resolve, dim3(512), dim3(256), 0, s, f, ao, out);
    return hipGetLastError();
}
} // namespace inv

#ifdef INV_AO_STATS
// measurement builds only (tools/ao_stats.py): the counters of the occlusion kernel since the last call
extern "C" int inv_debug_ao_stats(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(inv::g_ao_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long zero[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(inv::g_ao_stats), zero, sizeof(zero)) != hipSuccess;
}
#endif
