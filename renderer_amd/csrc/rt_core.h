// Device-side arithmetic shared by every render kernel: sample positions, primary
// rays, the Moller-Trumbore test with its (depth, primitive) key, shading and the
// sRGB store.  Every expression is written with an explicit operation order and
// the translation units are compiled with -ffp-contract=off, so results do not
// depend on which kernel evaluates them.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/invert.h"
#include "frame.h"

namespace inv {

struct Ray {
    float ox, oy, oz, dx, dy, dz;
};

struct Hit {
    uint64_t key; // (depth bits << 32) | primitive id ; miss = (bits of 1.0f, 0xFFFFFFFF)
    float t, un, vn, det;
    uint32_t slot; // storage slot of the winning triangle (for the attribute fetch)
};

static constexpr uint64_t kMissKey = ((uint64_t)0x3F800000u << 32) | 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// Sample 0 is the pixel centre, the rasteriser's single sample point
// (src/invert.cpp:1375-1378); further samples are hashed from (pixel, sample, seed).
__device__ __forceinline__ void sample_offset(uint32_t pixel, uint32_t sample, uint32_t seed, float& sx, float& sy) {
    if (sample == 0) {
        sx = 0.5f;
        sy = 0.5f;
        return;
    }
    uint32_t k = hash32(seed ^ 0x9E3779B9u);
    uint32_t a = hash32(pixel + k);
    uint32_t b = hash32(a ^ (sample * 0x85EBCA6Bu));
    uint32_t hx = hash32(b);
    uint32_t hy = hash32(b ^ 0x68E31DA4u);
    sx = (float)(hx >> 8) * 0x1p-24f;
    sy = (float)(hy >> 8) * 0x1p-24f;
}

// Framebuffer position -> NDC (viewport (0,0,w,h), src/invert.cpp:784-791) -> view-space
// direction (x_ndc/P00, y_ndc/P11, -1) -> object space.
// a / b, bit for bit, for a divisor b whose reciprocal rb = 1.0f / b the host has computed (one correctly rounded
// division per frame instead of eleven instructions per quotient).  q0 = a * rb is within 1.5 ulp of a / b; the
// residual e = a - b * q is exact in a fused multiply-add, so q1 = q0 + e * rb is off by a rounding and 2^-47 of the
// quotient: less than one ulp; and from an approximation that close and a correctly rounded reciprocal the same
// step once more rounds the exact quotient (Markstein, "Computation of elementary functions on the IBM RISC
// System/6000 processor", 1990, theorem on the final correction).  No intermediate over- or underflows for the
// operands a ray has (pixel coordinates over extents, normalised coordinates over projection terms of ordinary size).
__device__ __forceinline__ float div_by(float a, float b, float rb) {
    float q = a * rb;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, rb, q);
    e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, rb, q);
}

__device__ __forceinline__ Ray make_ray(const FrameK& f, uint32_t px, uint32_t py, float sx, float sy) {
    float fx = (float)px + sx;
    float fy = (float)py + sy;
    float xn = div_by(2.0f * fx, f.wf, f.rwf) - 1.0f;
    float yn = div_by(2.0f * fy, f.hf, f.rhf) - 1.0f;
    float dxv = div_by(xn, f.p00, f.inv_p00);
    float dyv = div_by(yn, f.p11, f.inv_p11);
    Ray r;
    r.ox = f.ox, r.oy = f.oy, r.oz = f.oz;
    r.dx = (f.d0x * dxv + f.d1x * dyv) - f.d2x;
    r.dy = (f.d0y * dxv + f.d1y * dyv) - f.d2y;
    r.dz = (f.d0z * dxv + f.d1z * dyv) - f.d2z;
    return r;
}

// The inside test of Moller-Trumbore on its own (same expressions and order as mt_test, no early exits):
// true iff mt_test would go on to compute t for this triangle.
__device__ __forceinline__ bool mt_inside(const Ray& r, const float4& A, const float4& B, const float4& C) {
    float px = r.dy * C.z - r.dz * C.y;
    float py = r.dz * C.x - r.dx * C.z;
    float pz = r.dx * C.y - r.dy * C.x;
    float det = (B.x * px + B.y * py) + B.z * pz;
    float tx = r.ox - A.x, ty = r.oy - A.y, tz = r.oz - A.z;
    float un = (tx * px + ty * py) + tz * pz;
    float qx = ty * B.z - tz * B.y;
    float qy = tz * B.x - tx * B.z;
    float qz = tx * B.y - ty * B.x;
    float vn = (r.dx * qx + r.dy * qy) + r.dz * qz;
    return (det > 0.0f) & !(un < 0.0f) & !(un > det) & !(vn < 0.0f) & !(un + vn > det);
}

// Terms of that decision folded once per triangle, for rays that share the origin o (every primary ray does):
// with t = o - v0,  det = e1 . (d x e2) = d . (e2 x e1),  un = t . (d x e2) = d . (e2 x t),  vn = d . (t x e1):
// P0 = e2 x e1, P1 = e2 x t, P2 = t x e1, so that each of the three is one dot product with the direction.
__device__ __forceinline__ void fold_triangle(float ox, float oy, float oz, const float4& A, const float4& B,
                                              const float4& C, float4& P0, float4& P1, float4& P2) {
    const float tx = ox - A.x, ty = oy - A.y, tz = oz - A.z;
    P0.x = C.y * B.z - C.z * B.y, P0.y = C.z * B.x - C.x * B.z, P0.z = C.x * B.y - C.y * B.x;
    P1.x = C.y * tz - C.z * ty, P1.y = C.z * tx - C.x * tz, P1.z = C.x * ty - C.y * tx;
    P2.x = ty * B.z - tz * B.y, P2.y = tz * B.x - tx * B.z, P2.z = tx * B.y - ty * B.x; // mt_test's qvec
    P0.w = 0.0f, P1.w = 0.0f, P2.w = 0.0f;
}

// The decision for two records at once in packed fp32 arithmetic: true (per record) wherever mt_inside is true and
// almost nowhere else.  The records' folded terms are stored
// interleaved (record a in the low, record b in the high half of every pair of floats):
//   q0 = (P0.x, P0.y)   q1 = (P0.z, P1.x)   q2 = (P1.y, P1.z)   q3 = (P2.x, P2.y)   q4 = (P2.z, eps)
// Fused multiply-adds and another summation order than mt_test's: every value differs from mt_test's by rounding only,
// so all comparisons are widened by eps (pair_eps below bounds the products of all three dot products), and the
// five conditions become one minimum.  A NaN is dropped by the minimum (conservative).  Bit 0: record a, bit 1: b.
typedef float pk2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pair_eps(float ox, float oy, float oz, float dmax, const float4& A, const float4& B,
                                          const float4& C) {
    const float tx = ox - A.x, ty = oy - A.y, tz = oz - A.z;
    const float c1 = (fabsf(C.x) + fabsf(C.y)) + fabsf(C.z);
    const float b1 = (fabsf(B.x) + fabsf(B.y)) + fabsf(B.z), t1 = (fabsf(tx) + fabsf(ty)) + fabsf(tz);
    // det: products below dmax*c1*b1, un: dmax*c1*t1, vn: dmax*t1*b1; a dozen roundings of 2^-24 each
    return 0x1p-18f * dmax * fmaxf(c1 * fmaxf(b1, t1), t1 * b1);
}

__device__ __forceinline__ uint32_t mt_inside_pair(const Ray& r, const float4& q0, const float4& q1, const float4& q2,
                                                   const float4& q3, const float4& q4) {
    const pk2 dx = {r.dx, r.dx}, dy = {r.dy, r.dy}, dz = {r.dz, r.dz};
    const pk2 p0x = {q0.x, q0.y}, p0y = {q0.z, q0.w}, p0z = {q1.x, q1.y};
    const pk2 p1x = {q1.z, q1.w}, p1y = {q2.x, q2.y}, p1z = {q2.z, q2.w};
    const pk2 p2x = {q3.x, q3.y}, p2y = {q3.z, q3.w}, p2z = {q4.x, q4.y};
    const pk2 e = {q4.z, q4.w};
    const pk2 det = __builtin_elementwise_fma(dx, p0x, __builtin_elementwise_fma(dy, p0y, dz * p0z));
    const pk2 un = __builtin_elementwise_fma(dx, p1x, __builtin_elementwise_fma(dy, p1y, dz * p1z));
    const pk2 vn = __builtin_elementwise_fma(dx, p2x, __builtin_elementwise_fma(dy, p2y, dz * p2z));
    const pk2 e2 = e + e, hi = det + e2;
    const pk2 t1 = det + e, t2 = un + e, t3 = hi - un, t4 = vn + e, t5 = (hi + e) - (un + vn);
    const float ma = fminf(fminf(fminf(t1.x, t2.x), fminf(t3.x, t4.x)), t5.x);
    const float mb = fminf(fminf(fminf(t1.y, t2.y), fminf(t3.y, t4.y)), t5.y);
    return (uint32_t)!(ma < 0.0f) | ((uint32_t)!(mb < 0.0f) << 1);
}

// Triangle record: three float4
//   A = (v0.xyz, primitive id bits)   B = (e1.xyz, z of vertex 1)   C = (e2.xyz, z of vertex 2)
// with e1 = v1 - v0, e2 = v2 - v0 rounded once at assembly.
__device__ __forceinline__ void mt_test(const FrameK& f, const Ray& r, const float4& A, const float4& B, const float4& C,
                                        uint32_t slot, Hit& best) {
    // pvec = d x e2
    float px = r.dy * C.z - r.dz * C.y;
    float py = r.dz * C.x - r.dx * C.z;
    float pz = r.dx * C.y - r.dy * C.x;
    float det = (B.x * px + B.y * py) + B.z * pz;
    if (!(det > 0.0f)) return; // cull BACK with front = the side the OBJ winding faces (src/invert.cpp:1334-1335)
    float tx = r.ox - A.x, ty = r.oy - A.y, tz = r.oz - A.z;
    float un = (tx * px + ty * py) + tz * pz;
    if (un < 0.0f || un > det) return;
    // qvec = tvec x e1
    float qx = ty * B.z - tz * B.y;
    float qy = tz * B.x - tx * B.z;
    float qz = tx * B.y - ty * B.x;
    float vn = (r.dx * qx + r.dy * qy) + r.dz * qz;
    if (vn < 0.0f || un + vn > det) return;
    float tn = (C.x * qx + C.y * qy) + C.z * qz;
    float t = tn / det;
    if (!(t > 0.0f)) return;
    // z_view = -t, so z_clip = P22*(-t) + P32 and w_clip = t (GLM RH zero-to-one, src/invert.cpp:8, 639)
    float zc = f.p22 * (-t) + f.p32;
    float depth = zc / t;
    if (!(depth >= 0.0f && depth < 1.0f)) return; // near/far clip; LESS against the 1.0 clear (src/invert.cpp:793, 1384)
    const uint64_t key = ((uint64_t)__float_as_uint(depth) << 32) | (uint64_t)__float_as_uint(A.w);
    if (!(key < best.key)) return; // depth LESS, lower primitive wins ties (src/invert.cpp:1384 + draw order)
    // guard: the hit point lies inside the triangle's bounding box grown by the pad
    float bx = A.x + B.x, by = A.y + B.y, bz = A.z + B.z;
    float cx = A.x + C.x, cy = A.y + C.y, cz = A.z + C.z;
    float hx = r.ox + t * r.dx, hy = r.oy + t * r.dy, hz = r.oz + t * r.dz;
    if (hx < fminf(A.x, fminf(bx, cx)) - f.pad || hx > fmaxf(A.x, fmaxf(bx, cx)) + f.pad) return;
    if (hy < fminf(A.y, fminf(by, cy)) - f.pad || hy > fmaxf(A.y, fmaxf(by, cy)) + f.pad) return;
    if (hz < fminf(A.z, fminf(bz, cz)) - f.pad || hz > fmaxf(A.z, fmaxf(bz, cz)) + f.pad) return;
    best.key = key;
    best.t = t, best.un = un, best.vn = vn, best.det = det;
    best.slot = slot;
}

// g / base_ntri (see MeshSource)
__device__ __forceinline__ uint32_t mesh_instance_of(const MeshSource& m, uint32_t g) {
    const uint32_t q = __umulhi(m.div_magic, g);
    return (q + ((g - q) >> m.div_sh1)) >> m.div_sh2;
}

// Primitive assembly of triangle g: instance g / base_ntri of template triangle g % base_ntri, as the three 16-byte
// items of its record (A with the primitive id, B = v1 - A, C = v2 - A, each with the vertex's own z in w).  Every
// kernel that wants the record without reading it back gets the same floats from here.
__device__ __forceinline__ void assemble_record(const MeshSource& m, uint64_t g, float4& ra, float4& rb, float4& rc) {
    const uint32_t inst = mesh_instance_of(m, (uint32_t)g), k = (uint32_t)g - inst * m.base_ntri; // (g is below 2^32)
    float4 a = m.xyzw[m.idx[3 * (size_t)k + 0]];
    float4 b = m.xyzw[m.idx[3 * (size_t)k + 1]];
    float4 c = m.xyzw[m.idx[3 * (size_t)k + 2]];
    if (m.translations) {
        const float tx = m.translations[3 * (size_t)inst + 0], ty = m.translations[3 * (size_t)inst + 1],
                    tz = m.translations[3 * (size_t)inst + 2];
        a.x = a.x + tx, a.y = a.y + ty, a.z = a.z + tz;
        b.x = b.x + tx, b.y = b.y + ty, b.z = b.z + tz;
        c.x = c.x + tx, c.y = c.y + ty, c.z = c.z + tz;
    }
    ra = make_float4(a.x, a.y, a.z, __uint_as_float((uint32_t)g));
    rb = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, b.z);
    rc = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, c.z);
}

// The same for N triangles at once (their ids below 2^32): what a stage needs from memory -- the indices, then the
// vertices and the translations -- is asked for together, so a lane waits three times for N triangles, not 3 N times.
template <int N>
__device__ __forceinline__ void assemble_records(const MeshSource& m, const uint32_t (&g)[N], float4 (&ra)[N], float4 (&rb)[N],
                                                 float4 (&rc)[N]) {
    uint32_t inst[N], i0[N], i1[N], i2[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
        inst[u] = mesh_instance_of(m, g[u]);
        const uint32_t k = g[u] - inst[u] * m.base_ntri;
        i0[u] = m.idx[3 * (size_t)k + 0], i1[u] = m.idx[3 * (size_t)k + 1], i2[u] = m.idx[3 * (size_t)k + 2];
    }
    float t[N][3];
#pragma unroll
    for (int u = 0; u < N; u++) {
        t[u][0] = t[u][1] = t[u][2] = 0.0f;
        if (m.translations) // (the same for every lane)
            t[u][0] = m.translations[3 * (size_t)inst[u] + 0], t[u][1] = m.translations[3 * (size_t)inst[u] + 1],
            t[u][2] = m.translations[3 * (size_t)inst[u] + 2];
    }
    float4 a[N], b[N], c[N];
#pragma unroll
    for (int u = 0; u < N; u++) a[u] = m.xyzw[i0[u]], b[u] = m.xyzw[i1[u]], c[u] = m.xyzw[i2[u]];
#pragma unroll
    for (int u = 0; u < N; u++) {
        if (m.translations) {
            a[u].x = a[u].x + t[u][0], a[u].y = a[u].y + t[u][1], a[u].z = a[u].z + t[u][2];
            b[u].x = b[u].x + t[u][0], b[u].y = b[u].y + t[u][1], b[u].z = b[u].z + t[u][2];
            c[u].x = c[u].x + t[u][0], c[u].y = c[u].y + t[u][1], c[u].z = c[u].z + t[u][2];
        }
        ra[u] = make_float4(a[u].x, a[u].y, a[u].z, __uint_as_float(g[u]));
        rb[u] = make_float4(b[u].x - a[u].x, b[u].y - a[u].y, b[u].z - a[u].z, b[u].z);
        rc[u] = make_float4(c[u].x - a[u].x, c[u].y - a[u].y, c[u].z - a[u].z, c[u].z);
    }
}

// Box of a triangle record as stored (A, B = v1 - A, C = v2 - A): the corners are A, A + B, A + C, not the vertices
// they were made from.
__device__ __forceinline__ void tri_box(const float4& A, const float4& B, const float4& C, float lo[3], float hi[3]) {
    float bx = A.x + B.x, by = A.y + B.y, bz = A.z + B.z;
    float cx = A.x + C.x, cy = A.y + C.y, cz = A.z + C.z;
    lo[0] = fminf(A.x, fminf(bx, cx)), lo[1] = fminf(A.y, fminf(by, cy)), lo[2] = fminf(A.z, fminf(bz, cz));
    hi[0] = fmaxf(A.x, fmaxf(bx, cx)), hi[1] = fmaxf(A.y, fmaxf(by, cy)), hi[2] = fmaxf(A.z, fmaxf(bz, cz));
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// The box over the boxes of all lanes of a workgroup of NT threads, to out[0..5] (lo xyz, hi xyz).  Call from every
// thread (one barrier); not twice in a row without a barrier in between.
template <int NT>
__device__ __forceinline__ void block_bounds(const float lo[3], const float hi[3], float* __restrict__ out) {
    __shared__ float red[NT / 64][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float l = wave_min(lo[a]), h = wave_max(hi[a]);
        if (lane == 0) red[wave][a] = l, red[wave][3 + a] = h;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < NT / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        out[threadIdx.x] = v;
    }
}

__device__ __forceinline__ Hit miss_hit() {
    Hit h;
    h.key = kMissKey;
    h.t = INFINITY, h.un = 0.0f, h.vn = 0.0f, h.det = 1.0f;
    h.slot = 0;
    return h;
}

// linear -> sRGB8 as a count of float32 thresholds (see tools/gen_srgb_table.py)
static __device__ const float kSrgbT[256] = {
#include "srgb_thresholds.inc"
};

// The result is the unique k with T[k] <= x < T[k+1]; a hardware log2/exp2 estimate of the transfer
// function lands on or next to it and the table decides, so the estimate's accuracy never shows.
// `T` may point at the table in global memory or at a copy in LDS.
__device__ __forceinline__ uint32_t srgb8(float x, const float* T) {
    if (!(x >= 0x1.3e4568p-13f)) return 0u; // below T[1]; also NaN and negatives
    if (x >= 0x1.fdb81cp-1f) return 255u;  // T[255]
    const float s = x <= 0.0031308f ? 12.92f * x : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f)) - 0.055f;
    int k = (int)(255.0f * s + 0.5f);
    k = k < 1 ? 1 : (k > 254 ? 254 : k);
    // (both neighbours of the estimate are asked for together: one wait where the estimate is right, which is the rule)
    const float t0 = T[k], t1 = T[k + 1];
    if (x < t0) {
        do k--; while (x < T[k]); // stops at k = 1 at the latest: x >= T[1]
    } else if (x >= t1) {
        do k++; while (x >= T[k + 1]); // stops at k = 254 at the latest: x < T[255]
    }
    return (uint32_t)k;
}

// Lambert factor of a triangle: max(0, unit geometric normal . direction towards the light); it depends on the
// triangle only, so the tile kernel computes it once per candidate record instead of once per shaded sample.
__device__ __forceinline__ float lambert_term(const FrameK& f, const float4& B, const float4& C) {
    float nx = B.y * C.z - B.z * C.y;
    float ny = B.z * C.x - B.x * C.z;
    float nz = B.x * C.y - B.y * C.x;
    float nn = (nx * nx + ny * ny) + nz * nz;
    float L = 0.0f;
    if (nn > 0.0f) {
        float inv = 1.0f / sqrtf(nn);
        float hx = nx * inv, hy = ny * inv, hz = nz * inv;
        L = (hx * f.lx + hy * f.ly) + hz * f.lz;
        if (!(L > 0.0f)) L = 0.0f;
    }
    return L;
}

// fragmentShader + interpolated vertex colour (src/shader.slang:27, 31-36), optional Lambert term `L`
// (lambert_term of the same triangle).  A, B, C are the winning triangle's records.
__device__ __forceinline__ void shade_hit_with(const FrameK& f, const Hit& h, const float4& A, const float4& B,
                                               const float4& C, float L, float& r, float& g, float& b) {
    float u = h.un / h.det, v = h.vn / h.det;
    float w0 = (1.0f - u) - v;
    float c0 = A.z * 10.0f, c1 = B.w * 10.0f, c2 = C.w * 10.0f;
    float cb = (w0 * c0 + u * c1) + v * c2;
    r = 1.0f, g = 1.0f, b = cb;
    if (f.flags & INV_LAMBERT) r = r * L, g = g * L, b = b * L;
}

__device__ __forceinline__ void shade_hit(const FrameK& f, const Hit& h, const float4& A, const float4& B,
                                          const float4& C, float& r, float& g, float& b) {
    shade_hit_with(f, h, A, B, C, (f.flags & INV_LAMBERT) ? lambert_term(f, B, C) : 1.0f, r, g, b);
}

__device__ __forceinline__ void shade_sample(const FrameK& f, const Hit& h, const float4* __restrict__ TA,
                                             const float4* __restrict__ TB, const float4* __restrict__ TC, float& r,
                                             float& g, float& b) {
    if (h.key == kMissKey) {
        r = g = b = 0.0f; // clear colour (src/invert.cpp:742)
        return;
    }
    shade_hit(f, h, TA[h.slot], TB[h.slot], TC[h.slot], r, g, b);
}

// ---- ambient occlusion (cosine-weighted any-hit rays about the geometric normal) ----
struct float2c {
    float c, s;
};
static __device__ const float2c kAoDirs[256] = {
#include "ao_dirs.inc"
};

// two-sided Moller-Trumbore with the same hit-point guard as the primary test
__device__ __forceinline__ bool mt_any(const FrameK& f, float ox, float oy, float oz, float wx, float wy, float wz,
                                       float tmax, const float4& A, const float4& B, const float4& C) {
    float px = wy * C.z - wz * C.y;
    float py = wz * C.x - wx * C.z;
    float pz = wx * C.y - wy * C.x;
    float det = (B.x * px + B.y * py) + B.z * pz;
    if (!(fabsf(det) > 0.0f)) return false;
    float inv = 1.0f / det;
    float tx = ox - A.x, ty = oy - A.y, tz = oz - A.z;
    float u = ((tx * px + ty * py) + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float qx = ty * B.z - tz * B.y;
    float qy = tz * B.x - tx * B.z;
    float qz = tx * B.y - ty * B.x;
    float v = ((wx * qx + wy * qy) + wz * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = ((C.x * qx + C.y * qy) + C.z * qz) * inv;
    if (!(t > 0.0f && t < tmax)) return false;
    float bx = A.x + B.x, by = A.y + B.y, bz = A.z + B.z;
    float cx = A.x + C.x, cy = A.y + C.y, cz = A.z + C.z;
    float hx = ox + t * wx, hy = oy + t * wy, hz = oz + t * wz;
    if (hx < fminf(A.x, fminf(bx, cx)) - f.pad || hx > fmaxf(A.x, fmaxf(bx, cx)) + f.pad) return false;
    if (hy < fminf(A.y, fminf(by, cy)) - f.pad || hy > fmaxf(A.y, fmaxf(by, cy)) + f.pad) return false;
    if (hz < fminf(A.z, fminf(bz, cz)) - f.pad || hz > fmaxf(A.z, fmaxf(bz, cz)) + f.pad) return false;
    return true;
}

// Unit normal of the winning triangle and the ray origin lifted off the surface by 8 guard pads.
// false for a zero-area triangle (no occlusion is computed for it).
__device__ __forceinline__ bool ao_origin(const FrameK& f, const Ray& r, float t, const float4& B, const float4& C,
                                          float o[3], float n[3]) {
    float nx = B.y * C.z - B.z * C.y;
    float ny = B.z * C.x - B.x * C.z;
    float nz = B.x * C.y - B.y * C.x;
    float nn = (nx * nx + ny * ny) + nz * nz;
    if (!(nn > 0.0f)) return false;
    float inv = 1.0f / sqrtf(nn);
    n[0] = nx * inv, n[1] = ny * inv, n[2] = nz * inv;
    float eps = 8.0f * f.pad;
    o[0] = (r.ox + t * r.dx) + n[0] * eps;
    o[1] = (r.oy + t * r.dy) + n[1] * eps;
    o[2] = (r.oz + t * r.dz) + n[2] * eps;
    return true;
}

// Direction of occlusion ray k of a pixel: an orthonormal basis around n (Duff et al., JCGT 2017), a stratified
// angle from the table and a hashed radius (Malley's method).
__device__ __forceinline__ void ao_direction(const float n[3], uint32_t hp, uint32_t k, uint32_t rays, float w[3]) {
    float sign = copysignf(1.0f, n[2]);
    float a = -1.0f / (sign + n[2]);
    float bb = (n[0] * n[1]) * a;
    float Tx = 1.0f + ((sign * n[0]) * n[0]) * a, Ty = sign * bb, Tz = -(sign * n[0]);
    float Bx = bb, By = sign + (n[1] * n[1]) * a, Bz = -n[1];
    uint32_t idx = ((k * 256u) / rays + (hp & 255u)) & 255u;
    uint32_t hk = hash32(hp + k * 0x9E3779B1u + 1u);
    float u1 = (float)(hk >> 8) * 0x1p-24f;
    float rr = sqrtf(u1);
    float x = rr * kAoDirs[idx].c, y = rr * kAoDirs[idx].s;
    float z = sqrtf(1.0f - u1);
    w[0] = (x * Tx + y * Bx) + z * n[0];
    w[1] = (x * Ty + y * By) + z * n[1];
    w[2] = (x * Tz + y * Bz) + z * n[2];
}

// Queue one hit pixel for the occlusion pass: one counter update per wave, lanes ranked by ballot.
// Call with the whole wave converged; `push` selects the lanes that have a record.
__device__ __forceinline__ void ao_push(const AoK& ao, bool push, const float o[3], const float n[3], float r, float g,
                                        float b, uint32_t pix, uint64_t out_index) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t mask = __ballot(push);
    if (!mask) return;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(ao.count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (push) {
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        ao.acc[slot] = 0u;
        float4* q = (float4*)ao.queue + 3 * (size_t)slot;
        q[0] = make_float4(o[0], o[1], o[2], r);
        q[1] = make_float4(n[0], n[1], n[2], g);
        q[2] = make_float4(b, __uint_as_float(pix), __uint_as_float((uint32_t)out_index),
                           __uint_as_float((uint32_t)(out_index >> 32)));
    }
}

// colour attachment store: sRGB encode, optional invert post-pass, B,G,R,A byte order
// unless INV_RGBA (src/invert.cpp:519)
__device__ __forceinline__ uint32_t pack_pixel(uint32_t flags, float r, float g, float b, const float* T = kSrgbT) {
    const uint32_t R0 = srgb8(r, T), G0 = (g == r) ? R0 : srgb8(g, T); // (a grey pixel: one lookup)
    uint32_t R = R0, G = G0, B = (b == r) ? R0 : (b == g) ? G0 : srgb8(b, T);
    if (flags & INV_INVERT) R = 255u - R, G = 255u - G, B = 255u - B;
    uint32_t lo = (flags & INV_RGBA) ? R : B;
    uint32_t hi = (flags & INV_RGBA) ? B : R;
    return lo | (G << 8) | (hi << 16) | (255u << 24);
}

__device__ __forceinline__ uint32_t background_pixel(uint32_t flags) {
    return (flags & INV_INVERT) ? 0xFFFFFFFFu : 0xFF000000u;
}

// A pixel's attachment values are stored write-through (sc1), like the clear values: they leave the XCD's L2 as
// they are written, so the end of the kernel has no dirty lines to write back.
__device__ __forceinline__ void store_pixel(uint32_t* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// The colour attachment's value of a pixel whose samples' linear colours add up to (sr, sg, sb).
__device__ __forceinline__ uint32_t resolve_pixel(const FrameK& f, float sr, float sg, float sb, const float* T = kSrgbT) {
    if (f.spp > 1) { // x / 1 is x: the resolve division is only needed for several samples
        const float n = (float)f.spp;
        sr = sr / n, sg = sg / n, sb = sb / n;
    }
    return pack_pixel(f.flags, sr, sg, sb, T);
}

// Sixteen bytes (four pixels of one attachment row) in one write-through store.  A four-byte write-through store is a
// memory-side write of its own, so a traced block that stored its pixels one lane at a time paid several times the
// price per byte that the clear pays; blocks and tiles therefore put their finished pixels together in LDS and store
// them in whole 16-byte pieces (store_block_rows below).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#ifndef INV_PIXEL_MODS
#define INV_PIXEL_MODS "sc1"
#endif
__device__ __forceinline__ void store_pixels4(uint32_t* p, u32x4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off " INV_PIXEL_MODS ::"v"(p), "v"(v) : "memory");
}

// Resolve and store one pixel whose samples were traced by this lane.  sr/sg/sb are the summed linear
// colours.  With ambient occlusion on (one sample per pixel) a hit pixel is queued for the occlusion pass,
// which writes its colour, instead.  Every active lane of the wave must call this together.
__device__ __forceinline__ void finish_pixel(const FrameK& f, const AoK& ao, const Outputs& out, uint64_t oi,
                                             uint32_t pix, const Ray& ray0, const Hit& first, const float4& Bw,
                                             const float4& Cw /* records B, C of the winner; unused for a miss */,
                                             float sr, float sg, float sb, const float* T = kSrgbT) {
    if (out.depth) store_pixel((uint32_t*)out.depth + oi, (uint32_t)(first.key >> 32));
    if (out.primid) store_pixel(out.primid + oi, (uint32_t)first.key);
    if (ao.rays) {
        float o[3] = {0.0f, 0.0f, 0.0f}, n[3] = {0.0f, 0.0f, 0.0f};
        bool push = false;
        if (first.key != kMissKey) push = ao_origin(f, ray0, first.t, Bw, Cw, o, n);
        ao_push(ao, push, o, n, sr, sg, sb, pix, oi);
        if (push) return;
    }
    if (out.color) store_pixel((uint32_t*)out.color + oi, resolve_pixel(f, sr, sg, sb, T));
}

// Grid tile of a pixel; false when the pixel lies outside the tile grid the shards are dealt from.
__device__ __forceinline__ bool grid_tile(const FrameK& f, uint32_t px, uint32_t py, uint32_t& id) {
    const uint32_t tx = px / f.tile - f.gx0, ty = py / f.tile - f.gy0; // wrap to huge values left of / above the grid
    id = ty * f.tiles_x + tx;
    return tx < f.tiles_x && ty < f.tiles_y;
}

// Row-major frame shared with other ranks (rank 0 of a sparse composite): this rank renders everything except
// the grid tiles dealt to the others.
__device__ __forceinline__ bool owns_rowmajor(const FrameK& f, uint32_t px, uint32_t py) {
    if (f.nranks <= 1) return true;
    uint32_t id;
    return !grid_tile(f, px, py, id) || id % f.nranks == f.rank;
}

// Work item -> pixel.  Row-major frames: item = pixel.  Slot buffers: this rank owns the grid tiles with
// id % nranks == rank and stores them one after another, tile*tile pixels each (partial edge tiles keep their
// full slot).
__device__ __forceinline__ bool work_to_pixel(const FrameK& f, uint64_t i, uint32_t& px, uint32_t& py) {
    if (f.rowmajor) {
        px = (uint32_t)(i % f.w);
        py = (uint32_t)(i / f.w);
        return owns_rowmajor(f, px, py);
    }
    uint32_t tt = f.tile * f.tile;
    uint32_t slot = (uint32_t)(i / tt), rem = (uint32_t)(i % tt);
    uint32_t id = slot * f.nranks + f.rank;
    if (id >= f.ntiles) return false;
    px = (f.gx0 + id % f.tiles_x) * f.tile + rem % f.tile;
    py = (f.gy0 + id / f.tiles_x) * f.tile + rem / f.tile;
    return px < f.w && py < f.h;
}

} // namespace inv
