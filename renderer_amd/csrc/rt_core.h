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
__device__ __forceinline__ Ray make_ray(const FrameK& f, uint32_t px, uint32_t py, float sx, float sy) {
    float fx = (float)px + sx;
    float fy = (float)py + sy;
    float xn = (2.0f * fx) / f.wf - 1.0f;
    float yn = (2.0f * fy) / f.hf - 1.0f;
    float dxv = xn / f.p00;
    float dyv = yn / f.p11;
    Ray r;
    r.ox = f.ox, r.oy = f.oy, r.oz = f.oz;
    r.dx = (f.d0x * dxv + f.d1x * dyv) - f.d2x;
    r.dy = (f.d0y * dxv + f.d1y * dyv) - f.d2y;
    r.dz = (f.d0z * dxv + f.d1z * dyv) - f.d2z;
    return r;
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
    while (x < T[k]) k--;      // stops at k = 1 at the latest: x >= T[1]
    while (x >= T[k + 1]) k++; // stops at k = 254 at the latest: x < T[255]
    return (uint32_t)k;
}

// fragmentShader + interpolated vertex colour (src/shader.slang:27, 31-36), optional Lambert term.
// A, B, C are the winning triangle's records.
__device__ __forceinline__ void shade_hit(const FrameK& f, const Hit& h, const float4& A, const float4& B,
                                          const float4& C, float& r, float& g, float& b) {
    float u = h.un / h.det, v = h.vn / h.det;
    float w0 = (1.0f - u) - v;
    float c0 = A.z * 10.0f, c1 = B.w * 10.0f, c2 = C.w * 10.0f;
    float cb = (w0 * c0 + u * c1) + v * c2;
    r = 1.0f, g = 1.0f, b = cb;
    if (f.flags & INV_LAMBERT) {
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
        r = r * L, g = g * L, b = b * L;
    }
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

// colour attachment store: sRGB encode, optional invert post-pass, B,G,R,A byte order
// unless INV_RGBA (src/invert.cpp:519)
__device__ __forceinline__ uint32_t pack_pixel(uint32_t flags, float r, float g, float b, const float* T = kSrgbT) {
    uint32_t R = srgb8(r, T), G = (g == r) ? R : srgb8(g, T), B = srgb8(b, T);
    if (flags & INV_INVERT) R = 255u - R, G = 255u - G, B = 255u - B;
    uint32_t lo = (flags & INV_RGBA) ? R : B;
    uint32_t hi = (flags & INV_RGBA) ? B : R;
    return lo | (G << 8) | (hi << 16) | (255u << 24);
}

__device__ __forceinline__ uint32_t background_pixel(uint32_t flags) {
    return (flags & INV_INVERT) ? 0xFFFFFFFFu : 0xFF000000u;
}

// Work item -> pixel.  One rank: row-major.  Several ranks: this rank owns the
// tiles with id % nranks == rank and stores them one after another, tile*tile
// pixels each (partial edge tiles keep their full slot).
__device__ __forceinline__ bool work_to_pixel(const FrameK& f, uint64_t i, uint32_t& px, uint32_t& py) {
    if (f.nranks <= 1) {
        px = (uint32_t)(i % f.w);
        py = (uint32_t)(i / f.w);
        return true;
    }
    uint32_t tt = f.tile * f.tile;
    uint32_t slot = (uint32_t)(i / tt), rem = (uint32_t)(i % tt);
    uint32_t id = slot * f.nranks + f.rank;
    if (id >= f.ntiles) return false;
    px = (id % f.tiles_x) * f.tile + rem % f.tile;
    py = (id / f.tiles_x) * f.tile + rem / f.tile;
    return px < f.w && py < f.h;
}

} // namespace inv
