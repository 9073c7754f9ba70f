// CPU oracle for the per-pixel hot path -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (oracle/Makefile).
// All pixel arithmetic is IEEE float32 with a fixed operation order; the HIP
// kernels evaluate the same expressions in the same order, so primitive ID,
// depth and colour can be compared bit for bit.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// small float32 helpers (every expression is written out so the op order is fixed)
// ---------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

const float kSrgbT[256] = {
#include "srgb_thresholds.inc"
};

uint8_t srgb8(float x) {
    // count of thresholds <= x; NaN and negatives give 0, x >= T[255] gives 255
    int lo = 0, hi = 255; // invariant: T[lo] <= x (or lo == 0), T[hi+1] > x
    if (!(x >= kSrgbT[1])) return 0;
    lo = 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (x >= kSrgbT[mid])
            lo = mid;
        else
            hi = mid - 1;
    }
    return (uint8_t)lo;
}

uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

void sample_offset(uint32_t pixel, uint32_t sample, uint32_t seed, float* sx, float* sy) {
    if (sample == 0) { // sample 0 is always the exact pixel centre (the rasteriser's sample point)
        *sx = 0.5f;
        *sy = 0.5f;
        return;
    }
    uint32_t k = hash32(seed ^ 0x9E3779B9u);
    uint32_t a = hash32(pixel + k);
    uint32_t b = hash32(a ^ (sample * 0x85EBCA6Bu));
    uint32_t hx = hash32(b);
    uint32_t hy = hash32(b ^ 0x68E31DA4u);
    *sx = (float)(hx >> 8) * 0x1p-24f;
    *sy = (float)(hy >> 8) * 0x1p-24f;
}

// ---------------------------------------------------------------------------
// A1  OBJ loader -- follows src/invert.cpp:42-209
// ---------------------------------------------------------------------------
int parse_obj(const char* buffer, uint64_t size, float** out_xyzw, uint32_t* out_nv, uint32_t** out_idx,
              uint32_t* out_nidx) {
    // The reference scans with no bounds checks; it is only defined for files
    // whose every line ends in '\n'.  Anything else is reported as an error here.
    if (size == 0 || buffer[size - 1] != '\n') return 2;
    uint32_t nvl = 0, nfl = 0;
    uint64_t off = 0;
    while (off < size) { // pass 1: src/invert.cpp:57-75 (first byte decides the line kind)
        if (buffer[off] == 'v')
            nvl++;
        else if (buffer[off] == 'f')
            nfl++;
        while (buffer[off] != '\n') off++;
        off++;
    }
    float* vd = (float*)std::calloc((size_t)nvl * 4 + 1, sizeof(float)); // lane 3 stays 0 here (uninitialised in the reference)
    uint32_t* id = (uint32_t*)std::calloc((size_t)nfl * 3 + 1, sizeof(uint32_t));
    if (!vd || !id) return 3;
    off = 0;
    uint32_t vi = 0, fi = 0;
    auto need = [&](uint64_t n) { return off + n <= size; };
    while (off < size) { // pass 2: src/invert.cpp:90-208
        if (buffer[off] == 'v') {
            off += 2;
            for (int i = 0; i < 3; i++) {
                if (!need(14)) goto bad;
                bool positive = buffer[off] != '-';
                if (!positive) off++;
                unsigned integral = (unsigned)(buffer[off] - '0'); // src/invert.cpp:105
                off += 2;
                unsigned fractional = 0, mult = 1000000; // src/invert.cpp:108-116
                for (int j = 0; j < 7; j++) {
                    fractional += (unsigned)(buffer[off] - '0') * mult;
                    mult /= 10;
                    off++;
                }
                off++; // 'e'
                if (!need(4)) goto bad;
                bool pos_exp = buffer[off] != '-'; // src/invert.cpp:120-125 ('+' is not handled)
                if (!pos_exp) off++;
                if (!need(4)) goto bad;
                unsigned exponent = 0, emult = 100; // src/invert.cpp:127-135
                for (int j = 0; j < 3; j++) {
                    exponent += (unsigned)(buffer[off] - '0') * emult;
                    emult /= 10;
                    off++;
                }
                off++; // space / newline
                float normaliser = 1.0f; // src/invert.cpp:139-150
                for (unsigned j = 0; j < exponent; j++) {
                    if (pos_exp)
                        normaliser *= 10.0f;
                    else
                        normaliser /= 10.0f;
                }
                // src/invert.cpp:152 -- float32, left to right
                float result = ((float)integral + ((float)fractional / 10000000.0f)) * normaliser * (positive ? 1.0f : -1.0f);
                vd[(size_t)vi * 4 + i] = result;
            }
            vi++;
        } else if (buffer[off] == 'f') {
            off += 2;
            for (int i = 0; i < 3; i++) { // src/invert.cpp:165-193
                unsigned digits[8] = {};
                unsigned count = 0;
                while (off < size && buffer[off] <= '9' && buffer[off] >= '0') {
                    if (count >= 8) goto bad;
                    digits[count++] = (unsigned)(buffer[off] - '0');
                    off++;
                }
                while (off < size && (buffer[off] == ' ' || buffer[off] == 0x0A || buffer[off] == 0x0D)) off++;
                unsigned result = 0, mult = 1;
                for (unsigned j = 0; j < count; j++) {
                    result += digits[count - 1 - j] * mult;
                    mult *= 10;
                }
                id[(size_t)fi * 3 + i] = result - 1;
            }
            fi++;
        } else {
            while (buffer[off] != '\n') off++;
            off++;
        }
    }
    *out_xyzw = vd;
    *out_nv = nvl;
    *out_idx = id;
    *out_nidx = nfl * 3;
    return 0;
bad:
    std::free(vd);
    std::free(id);
    return 4;
}

// ---------------------------------------------------------------------------
// A2  camera -- src/invert.cpp:621-642 with GLM's published formulas
// ---------------------------------------------------------------------------
// libm results are taken as the correctly rounded float32 of the true value
// (double libm, then one rounding) so two machines agree.
inline float cr_sin(float a) { return (float)std::sin((double)a); }
inline float cr_cos(float a) { return (float)std::cos((double)a); }
inline float cr_tan(float a) { return (float)std::tan((double)a); }
inline float glm_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float glm_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // tmp.x + tmp.y + tmp.z
inline V3 glm_cross(V3 x, V3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline V3 glm_normalize(V3 v) {
    float inv = 1.0f / std::sqrt(glm_dot(v, v));
    return {v.x * inv, v.y * inv, v.z * inv};
}

void mat_identity(float* m) {
    for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

void glm_lookat_rh(V3 eye, V3 center, V3 up, float* m) {
    V3 f = glm_normalize(sub(center, eye));
    V3 s = glm_normalize(glm_cross(f, up));
    V3 u = glm_cross(s, f);
    mat_identity(m);
    m[0] = s.x, m[4] = s.y, m[8] = s.z;
    m[1] = u.x, m[5] = u.y, m[9] = u.z;
    m[2] = -f.x, m[6] = -f.y, m[10] = -f.z;
    m[12] = -glm_dot(s, eye), m[13] = -glm_dot(u, eye), m[14] = glm_dot(f, eye);
}

void glm_rotate(float* m, float angle, V3 v) {
    float c = cr_cos(angle), s = cr_sin(angle);
    V3 axis = glm_normalize(v);
    V3 temp = {(1.0f - c) * axis.x, (1.0f - c) * axis.y, (1.0f - c) * axis.z};
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    float r[16];
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 4; row++)
            r[col * 4 + row] = m[0 + row] * R[col][0] + m[4 + row] * R[col][1] + m[8 + row] * R[col][2];
    for (int row = 0; row < 4; row++) r[12 + row] = m[12 + row];
    std::memcpy(m, r, sizeof r);
}

void glm_perspective_rh_zo(float fovy, float aspect, float z_near, float z_far, float* m) {
    float t = cr_tan(fovy / 2.0f);
    for (int i = 0; i < 16; i++) m[i] = 0.0f;
    m[0] = 1.0f / (aspect * t);
    m[5] = 1.0f / t;
    m[10] = z_far / (z_near - z_far);
    m[11] = -1.0f;
    m[14] = -(z_far * z_near) / (z_far - z_near);
}

// column-major mat * vec4, left-to-right sum (glm operator*, slang mul(M, v))
inline void mat_vec(const float* m, const float* v, float* r) {
    for (int row = 0; row < 4; row++)
        r[row] = ((m[0 + row] * v[0] + m[4 + row] * v[1]) + m[8 + row] * v[2]) + m[12 + row] * v[3];
}

// ---------------------------------------------------------------------------
// frame set-up (X1): object-space rays
// ---------------------------------------------------------------------------
struct Frame {
    V3 o;          // ray origin in object space
    V3 dc0, dc1, dc2; // columns of the 3x3 that maps (dxv, dyv, dzv) to object space
    float p00, p11, p22, p32;
    float pad;     // guard pad G
    uint32_t w, h;
    float wf, hf;
};

bool invert4_double(const double* a_in, double* inv) {
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            a[r][c] = a_in[c * 4 + r];
            a[r][4 + c] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return false;
        if (piv != col)
            for (int c = 0; c < 8; c++) std::swap(a[piv][c], a[col][c]);
        double d = a[col][col];
        for (int c = 0; c < 8; c++) a[col][c] = a[col][c] / d;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            double f = a[r][col];
            if (f == 0.0) continue;
            for (int c = 0; c < 8; c++) a[r][c] = a[r][c] - f * a[col][c];
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv[c * 4 + r] = a[r][4 + c];
    return true;
}

int frame_setup(const float* xyzw, uint32_t nv, const orc_camera* cam, uint32_t w, uint32_t h, Frame* f) {
    const float* P = cam->proj;
    // Only GLM-style perspective matrices (diagonal + P[2][3] = -1, P[3][2]) are supported.
    const int zero_idx[] = {1, 2, 3, 4, 6, 7, 8, 9, 12, 13, 15};
    for (int i : zero_idx)
        if (P[i] != 0.0f) return 5;
    if (P[11] != -1.0f || !(P[0] > 0.0f) || !(P[5] > 0.0f)) return 5;
    // MV = view * model, glm operator* order
    float mv[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            mv[c * 4 + r] = ((cam->view[0 + r] * cam->model[c * 4 + 0] + cam->view[4 + r] * cam->model[c * 4 + 1]) +
                             cam->view[8 + r] * cam->model[c * 4 + 2]) +
                            cam->view[12 + r] * cam->model[c * 4 + 3];
    double mvd[16], invd[16];
    for (int i = 0; i < 16; i++) mvd[i] = (double)mv[i];
    if (!invert4_double(mvd, invd)) return 6;
    float inv[16];
    for (int i = 0; i < 16; i++) inv[i] = (float)invd[i];
    f->o = {inv[12], inv[13], inv[14]};
    f->dc0 = {inv[0], inv[1], inv[2]};
    f->dc1 = {inv[4], inv[5], inv[6]};
    f->dc2 = {inv[8], inv[9], inv[10]};
    f->p00 = P[0], f->p11 = P[5], f->p22 = P[10], f->p32 = P[14];
    float R = std::max(std::fabs(f->o.x), std::max(std::fabs(f->o.y), std::fabs(f->o.z)));
    for (uint32_t i = 0; i < nv; i++)
        for (int c = 0; c < 3; c++) R = std::max(R, std::fabs(xyzw[(size_t)i * 4 + c]));
    f->pad = R * 0x1p-16f;
    f->w = w, f->h = h;
    f->wf = (float)w, f->hf = (float)h;
    return 0;
}

struct Ray {
    V3 o, d;
};

inline Ray make_ray(const Frame& f, uint32_t px, uint32_t py, float sx, float sy) {
    float fx = (float)px + sx;
    float fy = (float)py + sy;
    float xn = (2.0f * fx) / f.wf - 1.0f;
    float yn = (2.0f * fy) / f.hf - 1.0f;
    float dxv = xn / f.p00;
    float dyv = yn / f.p11;
    Ray r;
    r.o = f.o;
    r.d.x = (f.dc0.x * dxv + f.dc1.x * dyv) - f.dc2.x;
    r.d.y = (f.dc0.y * dxv + f.dc1.y * dyv) - f.dc2.y;
    r.d.z = (f.dc0.z * dxv + f.dc1.z * dyv) - f.dc2.z;
    return r;
}

// ---------------------------------------------------------------------------
// primitive assembly (A4): triangle k = indices 3k..3k+2 (src/invert.cpp:1307-1310)
// ---------------------------------------------------------------------------
struct Tri {
    V3 v0, e1, e2;
    float z1, z2; // exact object-space z of vertices 1 and 2 (for the colour attribute)
};

inline V3 vtx(const float* xyzw, uint32_t i) { return {xyzw[(size_t)i * 4], xyzw[(size_t)i * 4 + 1], xyzw[(size_t)i * 4 + 2]}; }

int assemble(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, std::vector<Tri>& tris) {
    uint32_t n = nidx / 3;
    tris.resize(n);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t i0 = idx[3 * k], i1 = idx[3 * k + 1], i2 = idx[3 * k + 2];
        if (i0 >= nv || i1 >= nv || i2 >= nv) return 7;
        V3 a = vtx(xyzw, i0), b = vtx(xyzw, i1), c = vtx(xyzw, i2);
        tris[k] = {a, sub(b, a), sub(c, a), b.z, c.z};
    }
    return 0;
}

inline void tri_bounds(const Tri& t, V3* lo, V3* hi) {
    V3 b = add(t.v0, t.e1), c = add(t.v0, t.e2);
    *lo = {std::min(t.v0.x, std::min(b.x, c.x)), std::min(t.v0.y, std::min(b.y, c.y)), std::min(t.v0.z, std::min(b.z, c.z))};
    *hi = {std::max(t.v0.x, std::max(b.x, c.x)), std::max(t.v0.y, std::max(b.y, c.y)), std::max(t.v0.z, std::max(b.z, c.z))};
}

// ---------------------------------------------------------------------------
// X2  Moller-Trumbore, culling branch; candidate key = (depth bits, primitive id)
// ---------------------------------------------------------------------------
struct Best {
    uint64_t key; // (depth_bits << 32) | primID ; miss = (1.0f bits, 0xFFFFFFFF)
    float t, un, vn, det;
};
const uint64_t kMissKey = ((uint64_t)0x3F800000u << 32) | 0xFFFFFFFFu;

inline void mt_test(const Frame& f, const Ray& r, const Tri& tr, uint32_t prim, Best* best) {
    V3 p = cross(r.d, tr.e2);
    float det = dot(tr.e1, p);
    if (!(det > 0.0f)) return; // back faces, degenerate triangles, NaN (cull BACK, src/invert.cpp:1334-1335)
    V3 tv = sub(r.o, tr.v0);
    float un = dot(tv, p);
    if (un < 0.0f || un > det) return;
    V3 q = cross(tv, tr.e1);
    float vn = dot(r.d, q);
    if (vn < 0.0f || un + vn > det) return;
    float tn = dot(tr.e2, q);
    float t = tn / det;
    if (!(t > 0.0f)) return;
    // depth = z_clip / w_clip with z_view = -t (GLM RH zero-to-one projection, src/invert.cpp:8,639)
    float zc = f.p22 * (-t) + f.p32;
    float depth = zc / t;
    if (!(depth >= 0.0f && depth < 1.0f)) return; // near/far clip and LESS against the 1.0 clear (src/invert.cpp:793,1384)
    // guard: the hit point must lie in the triangle's bounding box grown by the pad
    V3 lo, hi;
    tri_bounds(tr, &lo, &hi);
    float hx = r.o.x + t * r.d.x, hy = r.o.y + t * r.d.y, hz = r.o.z + t * r.d.z;
    if (hx < lo.x - f.pad || hx > hi.x + f.pad) return;
    if (hy < lo.y - f.pad || hy > hi.y + f.pad) return;
    if (hz < lo.z - f.pad || hz > hi.z + f.pad) return;
    uint64_t key = ((uint64_t)f2u(depth) << 32) | prim;
    if (key < best->key) { // LESS, earlier primitive wins ties (src/invert.cpp:1384, draw order)
        best->key = key;
        best->t = t, best->un = un, best->vn = vn, best->det = det;
    }
}

// ---------------------------------------------------------------------------
// A5 + X3 + X4  shading, resolve and store
// ---------------------------------------------------------------------------
struct Shade {
    uint32_t flags;
    V3 light; // -normalize(M3^-T.. see below) : direction towards the light in object space
    bool light_ok;
};

inline void shade_sample(const Shade& sh, const std::vector<Tri>& tris, const Best& b, float* rgb) {
    if (b.key == kMissKey) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f; // clear colour (src/invert.cpp:742)
        return;
    }
    const Tri& tr = tris[(uint32_t)b.key];
    float u = b.un / b.det, v = b.vn / b.det;
    float w0 = (1.0f - u) - v;
    float c0 = tr.v0.z * 10.0f, c1 = tr.z1 * 10.0f, c2 = tr.z2 * 10.0f; // src/shader.slang:27
    float cb = (w0 * c0 + u * c1) + v * c2;
    float r = 1.0f, g = 1.0f, bl = cb;
    if (sh.flags & ORC_LAMBERT) {
        V3 n = cross(tr.e1, tr.e2);
        float nn = dot(n, n);
        float L = 0.0f;
        if (nn > 0.0f && sh.light_ok) {
            float inv = 1.0f / std::sqrt(nn);
            V3 nh = {n.x * inv, n.y * inv, n.z * inv};
            L = dot(nh, sh.light);
            if (!(L > 0.0f)) L = 0.0f;
        }
        r = r * L, g = g * L, bl = bl * L;
    }
    rgb[0] = r, rgb[1] = g, rgb[2] = bl;
}

inline void store_pixel(uint32_t flags, const float* rgb, uint8_t* out) {
    uint8_t r = srgb8(rgb[0]), g = srgb8(rgb[1]), b = srgb8(rgb[2]);
    if (flags & ORC_INVERT) r = 255 - r, g = 255 - g, b = 255 - b;
    if (flags & ORC_RGBA)
        out[0] = r, out[1] = g, out[2] = b;
    else
        out[0] = b, out[1] = g, out[2] = r; // B8G8R8A8 memory order (src/invert.cpp:519)
    out[3] = 255;
}

Shade make_shade(const orc_camera* cam, uint32_t flags) {
    Shade sh;
    sh.flags = flags;
    // cam_angle lives in world space; normals are taken in object space and the
    // light direction is carried there with the transpose of the model 3x3
    // (exact for rigid model matrices; the reference uses the identity).
    const float* M = cam->model;
    V3 c = {cam->cam_angle[0], cam->cam_angle[1], cam->cam_angle[2]};
    V3 lo = {(M[0] * c.x + M[1] * c.y) + M[2] * c.z, (M[4] * c.x + M[5] * c.y) + M[6] * c.z, (M[8] * c.x + M[9] * c.y) + M[10] * c.z};
    float ll = dot(lo, lo);
    sh.light_ok = ll > 0.0f;
    if (sh.light_ok) {
        float inv = 1.0f / std::sqrt(ll);
        sh.light = {-(lo.x * inv), -(lo.y * inv), -(lo.z * inv)};
    } else {
        sh.light = {0.0f, 0.0f, 0.0f};
    }
    return sh;
}

// ---------------------------------------------------------------------------
// X5  ambient occlusion: cosine-weighted any-hit rays about the geometric normal
// ---------------------------------------------------------------------------
const float kAoDirs[256][2] = {
#include "ao_dirs.inc"
};

// two-sided Moller-Trumbore with the same hit-point guard as the primary test
inline bool mt_any(const Frame& f, V3 o, V3 w, float tmax, const Tri& tr) {
    V3 p = cross(w, tr.e2);
    float det = dot(tr.e1, p);
    if (!(std::fabs(det) > 0.0f)) return false;
    float inv = 1.0f / det;
    V3 tv = sub(o, tr.v0);
    float u = dot(tv, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 q = cross(tv, tr.e1);
    float v = dot(w, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = dot(tr.e2, q) * inv;
    if (!(t > 0.0f && t < tmax)) return false;
    V3 lo, hi;
    tri_bounds(tr, &lo, &hi);
    float hx = o.x + t * w.x, hy = o.y + t * w.y, hz = o.z + t * w.z;
    if (hx < lo.x - f.pad || hx > hi.x + f.pad) return false;
    if (hy < lo.y - f.pad || hy > hi.y + f.pad) return false;
    if (hz < lo.z - f.pad || hz > hi.z + f.pad) return false;
    return true;
}

struct AoParams {
    uint32_t rays;
    float tmax;
    uint32_t seed;
};

// Fraction of the hemisphere that is open, 1 - occluded / rays.  `occluded(o, w, tmax)` is the any-hit query.
template <class AnyHit>
inline float ao_factor(const Frame& f, const AoParams& ao, uint32_t pix, const Ray& r, const Best& b, const Tri& tr,
                       AnyHit occluded) {
    V3 n = cross(tr.e1, tr.e2);
    float nn = dot(n, n);
    if (!(nn > 0.0f)) return 1.0f;
    float inv = 1.0f / std::sqrt(nn);
    V3 nh = {n.x * inv, n.y * inv, n.z * inv};
    float eps = 8.0f * f.pad;
    V3 o = {(r.o.x + b.t * r.d.x) + nh.x * eps, (r.o.y + b.t * r.d.y) + nh.y * eps, (r.o.z + b.t * r.d.z) + nh.z * eps};
    // orthonormal basis around nh (Duff et al., "Building an Orthonormal Basis, Revisited", JCGT 2017)
    float sign = std::copysign(1.0f, nh.z);
    float a = -1.0f / (sign + nh.z);
    float bb = (nh.x * nh.y) * a;
    V3 T = {1.0f + ((sign * nh.x) * nh.x) * a, sign * bb, -(sign * nh.x)};
    V3 B = {bb, sign + (nh.y * nh.y) * a, -nh.y};
    uint32_t hp = hash32(pix ^ hash32(ao.seed ^ 0xA0761D65u));
    uint32_t occ = 0;
    for (uint32_t k = 0; k < ao.rays; k++) {
        uint32_t idx = ((k * 256u) / ao.rays + (hp & 255u)) & 255u;
        uint32_t hk = hash32(hp + k * 0x9E3779B1u + 1u);
        float u1 = (float)(hk >> 8) * 0x1p-24f;
        float rr = std::sqrt(u1);
        float x = rr * kAoDirs[idx][0], y = rr * kAoDirs[idx][1];
        float z = std::sqrt(1.0f - u1);
        V3 w = {(x * T.x + y * B.x) + z * nh.x, (x * T.y + y * B.y) + z * nh.y, (x * T.z + y * B.z) + z * nh.z};
        if (occluded(o, w, ao.tmax)) occ++;
    }
    return 1.0f - (float)occ / (float)ao.rays;
}

// ---------------------------------------------------------------------------
// LBVH (restatement of the GPU build, sequential)
// ---------------------------------------------------------------------------
struct Lbvh {
    uint32_t n = 0;
    std::vector<uint64_t> morton; // sorted
    std::vector<uint32_t> order;  // sorted position -> triangle
    std::vector<uint32_t> child;  // 2*(n-1); bit 31 = leaf
    std::vector<uint32_t> parent; // n-1 (root: 0xFFFFFFFF)
    std::vector<uint32_t> leaf_parent;
    std::vector<float> box;       // (n-1)*12
    V3 lo, hi;
};

inline uint64_t expand21(uint32_t v) {
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

inline uint32_t grid21(float c, float lo, float ext) {
    float q = (ext > 0.0f) ? (c - lo) / ext : 0.0f;
    float s = q * 2097152.0f;
    if (!(s > 0.0f)) return 0;
    if (s >= 2097151.0f) return 2097151u;
    return (uint32_t)s;
}

inline int delta(const Lbvh& b, int64_t i, int64_t j) {
    if (j < 0 || j >= (int64_t)b.n) return -1;
    uint64_t a = b.morton[i], c = b.morton[j];
    if (a != c) return __builtin_clzll(a ^ c);
    return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}

void lbvh_build(const std::vector<Tri>& tris, Lbvh& b) {
    uint32_t n = (uint32_t)tris.size();
    b.n = n;
    std::vector<V3> lo(n), hi(n);
    const float inf = std::numeric_limits<float>::infinity();
    b.lo = {inf, inf, inf}, b.hi = {-inf, -inf, -inf};
    for (uint32_t k = 0; k < n; k++) {
        tri_bounds(tris[k], &lo[k], &hi[k]);
        b.lo = {std::min(b.lo.x, lo[k].x), std::min(b.lo.y, lo[k].y), std::min(b.lo.z, lo[k].z)};
        b.hi = {std::max(b.hi.x, hi[k].x), std::max(b.hi.y, hi[k].y), std::max(b.hi.z, hi[k].z)};
    }
    // One grid step for all three axes (the largest extent): cubic cells, whatever the shape of the scene.  (With a
    // step per axis a flat scene -- a grid of instances in a plane -- gets cells that are thin slabs, and consecutive
    // keys then walk around an instance instead of across a patch of its surface.)
    V3 ext = sub(b.hi, b.lo);
    ext.x = ext.y = ext.z = std::max(ext.x, std::max(ext.y, ext.z));
    std::vector<std::pair<uint64_t, uint32_t>> keys(n);
    for (uint32_t k = 0; k < n; k++) {
        float cx = (lo[k].x + hi[k].x) * 0.5f, cy = (lo[k].y + hi[k].y) * 0.5f, cz = (lo[k].z + hi[k].z) * 0.5f;
        uint64_t m = (expand21(grid21(cx, b.lo.x, ext.x)) << 2) | (expand21(grid21(cy, b.lo.y, ext.y)) << 1) |
                     expand21(grid21(cz, b.lo.z, ext.z));
        keys[k] = {m, k};
    }
    std::sort(keys.begin(), keys.end());
    b.morton.resize(n), b.order.resize(n);
    for (uint32_t i = 0; i < n; i++) b.morton[i] = keys[i].first, b.order[i] = keys[i].second;
    if (n < 2) return;
    b.child.assign(2 * (size_t)(n - 1), 0);
    b.parent.assign(n - 1, 0xFFFFFFFFu);
    b.leaf_parent.assign(n, 0xFFFFFFFFu);
    b.box.assign((size_t)(n - 1) * 12, 0.0f);
    // Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", Alg. in sec. 4
    for (int64_t i = 0; i < (int64_t)n - 1; i++) {
        int d = (delta(b, i, i + 1) - delta(b, i, i - 1)) >= 0 ? 1 : -1;
        int dmin = delta(b, i, i - d);
        int64_t lmax = 2;
        while (delta(b, i, i + lmax * d) > dmin) lmax *= 2;
        int64_t l = 0;
        for (int64_t t = lmax / 2; t >= 1; t /= 2)
            if (delta(b, i, i + (l + t) * d) > dmin) l += t;
        int64_t j = i + l * d;
        int dnode = delta(b, i, j);
        int64_t s = 0;
        int64_t t = l;
        do {
            t = (t + 1) / 2;
            if (delta(b, i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        int64_t gamma = i + s * d + std::min(d, 0);
        uint32_t c0, c1;
        if (std::min(i, j) == gamma) {
            c0 = (uint32_t)gamma | 0x80000000u;
            b.leaf_parent[gamma] = (uint32_t)i;
        } else {
            c0 = (uint32_t)gamma;
            b.parent[gamma] = (uint32_t)i;
        }
        if (std::max(i, j) == gamma + 1) {
            c1 = (uint32_t)(gamma + 1) | 0x80000000u;
            b.leaf_parent[gamma + 1] = (uint32_t)i;
        } else {
            c1 = (uint32_t)(gamma + 1);
            b.parent[gamma + 1] = (uint32_t)i;
        }
        b.child[2 * i] = c0, b.child[2 * i + 1] = c1;
    }
    // bottom-up refit, iterative post-order from the root
    std::vector<V3> nlo(n - 1), nhi(n - 1);
    std::vector<uint32_t> stack;
    std::vector<uint8_t> state(n - 1, 0);
    stack.push_back(0);
    auto child_box = [&](uint32_t c, V3* l, V3* h) {
        if (c & 0x80000000u) {
            uint32_t k = b.order[c & 0x7FFFFFFFu];
            *l = lo[k], *h = hi[k];
        } else {
            *l = nlo[c], *h = nhi[c];
        }
    };
    while (!stack.empty()) {
        uint32_t i = stack.back();
        if (state[i] == 0) {
            state[i] = 1;
            for (int s = 0; s < 2; s++) {
                uint32_t c = b.child[2 * (size_t)i + s];
                if (!(c & 0x80000000u)) stack.push_back(c);
            }
        } else {
            stack.pop_back();
            V3 l0, h0, l1, h1;
            child_box(b.child[2 * (size_t)i], &l0, &h0);
            child_box(b.child[2 * (size_t)i + 1], &l1, &h1);
            float* bx = &b.box[(size_t)i * 12];
            bx[0] = l0.x, bx[1] = l0.y, bx[2] = l0.z, bx[3] = h0.x, bx[4] = h0.y, bx[5] = h0.z;
            bx[6] = l1.x, bx[7] = l1.y, bx[8] = l1.z, bx[9] = h1.x, bx[10] = h1.y, bx[11] = h1.z;
            nlo[i] = {std::min(l0.x, l1.x), std::min(l0.y, l1.y), std::min(l0.z, l1.z)};
            nhi[i] = {std::max(h0.x, h1.x), std::max(h0.y, h1.y), std::max(h0.z, h1.z)};
        }
    }
}

// conservative slab test: box grown by `grow`, NaN-ignoring min/max
inline bool slab(const Ray& r, V3 inv, const float* lo, const float* hi, float grow, float tmax, float* tnear) {
    float t0 = 0.0f, t1 = tmax;
    const float o[3] = {r.o.x, r.o.y, r.o.z}, iv[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; a++) {
        float ta = ((lo[a] - grow) - o[a]) * iv[a];
        float tb = ((hi[a] + grow) - o[a]) * iv[a];
        t0 = std::fmax(t0, std::fmin(ta, tb));
        t1 = std::fmin(t1, std::fmax(ta, tb));
    }
    *tnear = t0;
    // relative slack absorbs the rounding of the slab arithmetic itself
    return t0 * 0.9999f <= t1 * 1.0001f;
}

inline void trace_bvh(const Frame& f, const Ray& r, const std::vector<Tri>& tris, const Lbvh& b, Best* best,
                      orc_counters* cnt) {
    V3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    const float grow = 2.0f * f.pad;
    const float inf = std::numeric_limits<float>::infinity();
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        uint32_t node = stack[--sp];
        const float* bx = &b.box[(size_t)node * 12];
        float tn[2];
        bool hit[2];
        float tcur = (best->key == kMissKey) ? inf : best->t * 1.0001f;
        for (int s = 0; s < 2; s++) {
            hit[s] = slab(r, inv, bx + 6 * s, bx + 6 * s + 3, grow, tcur, &tn[s]);
            cnt->box_tests++;
        }
        int first = (hit[0] && hit[1] && tn[1] < tn[0]) ? 1 : 0;
        for (int q = 0; q < 2; q++) {
            int s = q == 0 ? 1 - first : first; // push far first, near last (popped first)
            if (!hit[s]) continue;
            uint32_t c = b.child[2 * (size_t)node + s];
            if (c & 0x80000000u) {
                uint32_t k = b.order[c & 0x7FFFFFFFu];
                mt_test(f, r, tris[k], k, best);
                cnt->tri_tests++;
            } else {
                stack[sp++] = c;
            }
        }
    }
}

inline bool any_bvh(const Frame& f, V3 o, V3 w, float tmax, const std::vector<Tri>& tris, const Lbvh& b,
                    orc_counters* cnt) {
    Ray r = {o, w};
    V3 inv = {1.0f / w.x, 1.0f / w.y, 1.0f / w.z};
    const float grow = 2.0f * f.pad;
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        uint32_t node = stack[--sp];
        const float* bx = &b.box[(size_t)node * 12];
        for (int s = 0; s < 2; s++) {
            float tn;
            cnt->box_tests++;
            if (!slab(r, inv, bx + 6 * s, bx + 6 * s + 3, grow, tmax * 1.0001f, &tn)) continue;
            uint32_t c = b.child[2 * (size_t)node + s];
            if (c & 0x80000000u) {
                cnt->tri_tests++;
                if (mt_any(f, o, w, tmax, tris[b.order[c & 0x7FFFFFFFu]])) return true;
            } else {
                stack[sp++] = c;
            }
        }
    }
    return false;
}

// ---------------------------------------------------------------------------
// frame driver
// ---------------------------------------------------------------------------
template <class TraceFn, class AnyFn>
void render_rows(const Frame& f, const Shade& sh, const orc_params* prm, const AoParams& ao,
                 const std::vector<Tri>& tris, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, int tid,
                 int nthreads, uint8_t* color, float* depth, uint32_t* primid, TraceFn trace, AnyFn occluded) {
    uint32_t spp = prm->spp ? prm->spp : 1;
    float sppf = (float)spp;
    for (uint32_t py = y0 + (uint32_t)tid; py < y1; py += (uint32_t)nthreads) {
        for (uint32_t px = x0; px < x1; px++) {
            uint32_t pix = py * f.w + px;
            float sum[3] = {0.0f, 0.0f, 0.0f};
            for (uint32_t s = 0; s < spp; s++) {
                float sx, sy;
                sample_offset(pix, s, prm->seed, &sx, &sy);
                Ray r = make_ray(f, px, py, sx, sy);
                Best best = {kMissKey, 0.0f, 0.0f, 0.0f, 1.0f};
                trace(r, &best);
                if (s == 0) {
                    uint32_t db = (uint32_t)(best.key >> 32);
                    if (depth) std::memcpy(&depth[pix], &db, 4);
                    if (primid) primid[pix] = (uint32_t)best.key;
                }
                float rgb[3];
                shade_sample(sh, tris, best, rgb);
                if (ao.rays && best.key != kMissKey) { // one sample per pixel in this mode (checked by the caller)
                    float a = ao_factor(f, ao, pix, r, best, tris[(uint32_t)best.key], occluded);
                    rgb[0] = rgb[0] * a, rgb[1] = rgb[1] * a, rgb[2] = rgb[2] * a;
                }
                sum[0] = sum[0] + rgb[0], sum[1] = sum[1] + rgb[1], sum[2] = sum[2] + rgb[2];
            }
            float rgb[3] = {sum[0] / sppf, sum[1] / sppf, sum[2] / sppf};
            if (color) store_pixel(prm->flags, rgb, &color[(size_t)pix * 4]);
        }
    }
}

template <class Body>
void run_threads(int nthreads, Body body) {
    if (nthreads <= 1) {
        body(0, 1);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(body, t, nthreads);
    for (auto& t : th) t.join();
}

AoParams make_ao(const orc_params* prm, const float* xyzw, uint32_t nv) {
    AoParams ao = {prm->ao_rays, prm->ao_tmax, prm->seed};
    if (ao.rays && !(ao.tmax > 0.0f)) { // default reach: diagonal of the vertices' bounding box
        const float inf = std::numeric_limits<float>::infinity();
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        for (uint32_t i = 0; i < nv; i++)
            for (int c = 0; c < 3; c++) {
                lo[c] = std::min(lo[c], xyzw[(size_t)i * 4 + c]);
                hi[c] = std::max(hi[c], xyzw[(size_t)i * 4 + c]);
            }
        float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        ao.tmax = nv ? std::sqrt((ex * ex + ey * ey) + ez * ez) : 0.0f;
    }
    return ao;
}

int check_rect(const orc_params* prm, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    if (prm->width == 0 || prm->height == 0) return 8;
    if (x0 > x1 || y0 > y1 || x1 > prm->width || y1 > prm->height) return 8;
    if (prm->ao_rays != 0 && prm->spp > 1) return 9; // ambient occlusion is defined for one sample per pixel
    return 0;
}

} // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

void orc_free(void* p) { std::free(p); }

int orc_parse_obj(const char* buf, uint64_t size, float** xyzw, uint32_t* nv, uint32_t** idx, uint32_t* nidx) {
    return parse_obj(buf, size, xyzw, nv, idx, nidx);
}

int orc_load_obj(const char* path, float** xyzw, uint32_t* nv, uint32_t** idx, uint32_t* nidx) {
    FILE* file = std::fopen(path, "rb"); // src/invert.cpp:44-50
    if (!file) return 1;
    std::fseek(file, 0, SEEK_END);
    long size = std::ftell(file);
    std::fseek(file, 0, SEEK_SET);
    std::vector<char> buffer((size_t)std::max(size, 0L) + 16, '\n');
    size_t got = size > 0 ? std::fread(buffer.data(), 1, (size_t)size, file) : 0;
    std::fclose(file);
    if ((long)got != size) return 1;
    return parse_obj(buffer.data(), (uint64_t)size, xyzw, nv, idx, nidx);
}

void orc_camera_reference(uint32_t width, uint32_t height, orc_camera* cam) {
    V3 eye = {0.0f, 0.0f, 1.0f}, focus = {0.0f, 0.0f, 0.0f}, up = {0.0f, 1.0f, 0.0f}; // src/invert.cpp:630-632
    V3 ang = sub(focus, eye);                                                          // src/invert.cpp:633
    cam->cam_angle[0] = ang.x, cam->cam_angle[1] = ang.y, cam->cam_angle[2] = ang.z;
    mat_identity(cam->model);                                                          // src/invert.cpp:636
    glm_lookat_rh(eye, focus, up, cam->view);                                          // src/invert.cpp:637
    glm_rotate(cam->view, glm_radians(180.0f), V3{0.0f, 0.0f, 1.0f});                  // src/invert.cpp:638
    glm_perspective_rh_zo(glm_radians(45.0f), (float)width / (float)height, 0.1f, 10.0f, cam->proj); // :639
}

// Turntable: the reference camera with model = rotate(identity, elapsed * radians(90), up) -- a use for the
// `elapsed` that src/invert.cpp:623-626 measures and ignores.
void orc_camera_turntable(uint32_t width, uint32_t height, float elapsed_seconds, orc_camera* cam) {
    orc_camera_reference(width, height, cam);
    mat_identity(cam->model);
    glm_rotate(cam->model, elapsed_seconds * glm_radians(90.0f), V3{0.0f, 1.0f, 0.0f});
}

void orc_vertex_shader(const float* xyzw, uint32_t nv, const orc_camera* cam, float* clip4, float* color3) {
    for (uint32_t i = 0; i < nv; i++) {
        float p[4] = {xyzw[(size_t)i * 4], xyzw[(size_t)i * 4 + 1], xyzw[(size_t)i * 4 + 2], 1.0f};
        float a[4], b[4];
        mat_vec(cam->model, p, a); // src/shader.slang:25 -- three successive products
        mat_vec(cam->view, a, b);
        mat_vec(cam->proj, b, &clip4[(size_t)i * 4]);
        if (color3) {
            color3[(size_t)i * 3] = 1.0f, color3[(size_t)i * 3 + 1] = 1.0f;
            color3[(size_t)i * 3 + 2] = p[2] * 10.0f; // src/shader.slang:27
        }
    }
}

int orc_frame_setup(const float* xyzw, uint32_t nv, const orc_camera* cam, float* out17) {
    Frame f;
    int rc = frame_setup(xyzw, nv, cam, 1, 1, &f);
    if (rc) return rc;
    const float v[17] = {f.o.x, f.o.y, f.o.z, f.dc0.x, f.dc0.y, f.dc0.z, f.dc1.x, f.dc1.y, f.dc1.z,
                         f.dc2.x, f.dc2.y, f.dc2.z, f.p00, f.p11, f.p22, f.p32, f.pad};
    std::memcpy(out17, v, sizeof v);
    return 0;
}

int orc_render_bruteforce(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, const orc_camera* cam,
                          const orc_params* prm, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint8_t* color,
                          float* depth, uint32_t* primid, int nthreads) {
    int rc = check_rect(prm, x0, y0, x1, y1);
    if (rc) return rc;
    Frame f;
    if ((rc = frame_setup(xyzw, nv, cam, prm->width, prm->height, &f))) return rc;
    std::vector<Tri> tris;
    if ((rc = assemble(xyzw, nv, idx, nidx, tris))) return rc;
    Shade sh = make_shade(cam, prm->flags);
    AoParams ao = make_ao(prm, xyzw, nv);
    run_threads(nthreads, [&](int tid, int nt) {
        render_rows(
            f, sh, prm, ao, tris, x0, y0, x1, y1, tid, nt, color, depth, primid,
            [&](const Ray& r, Best* best) {
                for (uint32_t k = 0; k < (uint32_t)tris.size(); k++) mt_test(f, r, tris[k], k, best);
            },
            [&](V3 o, V3 w, float tmax) {
                for (uint32_t k = 0; k < (uint32_t)tris.size(); k++)
                    if (mt_any(f, o, w, tmax, tris[k])) return true;
                return false;
            });
    });
    return 0;
}

int orc_render_bvh(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, const orc_camera* cam,
                   const orc_params* prm, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint8_t* color,
                   float* depth, uint32_t* primid, int nthreads, orc_counters* counters) {
    int rc = check_rect(prm, x0, y0, x1, y1);
    if (rc) return rc;
    Frame f;
    if ((rc = frame_setup(xyzw, nv, cam, prm->width, prm->height, &f))) return rc;
    std::vector<Tri> tris;
    if ((rc = assemble(xyzw, nv, idx, nidx, tris))) return rc;
    Shade sh = make_shade(cam, prm->flags);
    Lbvh bvh;
    lbvh_build(tris, bvh);
    if (nthreads < 1) nthreads = 1;
    std::vector<orc_counters> cnt((size_t)nthreads, orc_counters{0, 0, 0, 0});
    AoParams ao = make_ao(prm, xyzw, nv);
    run_threads(nthreads, [&](int tid, int nt) {
        orc_counters* c = &cnt[(size_t)tid];
        render_rows(
            f, sh, prm, ao, tris, x0, y0, x1, y1, tid, nt, color, depth, primid,
            [&](const Ray& r, Best* best) {
                c->rays++;
                if (tris.size() < 2) {
                    for (uint32_t k = 0; k < (uint32_t)tris.size(); k++) mt_test(f, r, tris[k], k, best);
                    c->tri_tests += tris.size();
                } else {
                    trace_bvh(f, r, tris, bvh, best, c);
                }
                if (best->key != kMissKey) c->hits++;
            },
            [&](V3 o, V3 w, float tmax) {
                c->rays++;
                if (tris.size() < 2) {
                    for (uint32_t k = 0; k < (uint32_t)tris.size(); k++)
                        if (mt_any(f, o, w, tmax, tris[k])) return true;
                    return false;
                }
                return any_bvh(f, o, w, tmax, tris, bvh, c);
            });
    });
    if (counters) {
        *counters = orc_counters{0, 0, 0, 0};
        for (auto& c : cnt) {
            counters->rays += c.rays, counters->box_tests += c.box_tests;
            counters->tri_tests += c.tri_tests, counters->hits += c.hits;
        }
    }
    return 0;
}

int orc_raster(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, const orc_camera* cam,
               uint32_t width, uint32_t height, double* depth, uint32_t* primid, uint32_t* n_front) {
    std::vector<float> clip((size_t)nv * 4);
    orc_vertex_shader(xyzw, nv, cam, clip.data(), nullptr);
    for (size_t i = 0; i < (size_t)width * height; i++) depth[i] = 1.0, primid[i] = 0xFFFFFFFFu; // clear (src/invert.cpp:793)
    uint32_t front = 0;
    for (uint32_t k = 0; k < nidx / 3; k++) {
        double sx[3], sy[3], sz[3];
        bool ok = true;
        for (int c = 0; c < 3; c++) {
            uint32_t i = idx[3 * k + c];
            if (i >= nv) return 7;
            double w = clip[(size_t)i * 4 + 3];
            if (!(w > 0.0)) ok = false; // would need clipping; not the case for the reference scene
            double xn = clip[(size_t)i * 4] / w, yn = clip[(size_t)i * 4 + 1] / w, zn = clip[(size_t)i * 4 + 2] / w;
            sx[c] = (xn + 1.0) * (double)width / 2.0; // viewport (0,0,w,h,0..1), src/invert.cpp:784-791
            sy[c] = (yn + 1.0) * (double)height / 2.0;
            sz[c] = zn;
        }
        if (!ok) continue;
        // Shoelace sum in y-down framebuffer coordinates; VK_FRONT_FACE_CLOCKWISE + cull BACK keeps area > 0
        // (src/invert.cpp:1334-1335).
        double area2 = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
        if (!(area2 > 0.0)) continue;
        front++;
        double minx = std::min(sx[0], std::min(sx[1], sx[2])), maxx = std::max(sx[0], std::max(sx[1], sx[2]));
        double miny = std::min(sy[0], std::min(sy[1], sy[2])), maxy = std::max(sy[0], std::max(sy[1], sy[2]));
        long px0 = std::max(0L, (long)std::floor(minx - 0.5)), px1 = std::min((long)width - 1, (long)std::ceil(maxx - 0.5));
        long py0 = std::max(0L, (long)std::floor(miny - 0.5)), py1 = std::min((long)height - 1, (long)std::ceil(maxy - 0.5));
        for (long py = py0; py <= py1; py++)
            for (long px = px0; px <= px1; px++) {
                double cx = px + 0.5, cy = py + 0.5; // one sample at the pixel centre (src/invert.cpp:1375-1378)
                double w0 = (sx[1] - cx) * (sy[2] - cy) - (sx[2] - cx) * (sy[1] - cy);
                double w1 = (sx[2] - cx) * (sy[0] - cy) - (sx[0] - cx) * (sy[2] - cy);
                double w2 = (sx[0] - cx) * (sy[1] - cy) - (sx[1] - cx) * (sy[0] - cy);
                if (w0 < 0.0 || w1 < 0.0 || w2 < 0.0) continue;
                double z = (w0 * sz[0] + w1 * sz[1] + w2 * sz[2]) / area2; // linear in screen space
                size_t pix = (size_t)py * width + (size_t)px;
                if (z >= 0.0 && z < depth[pix]) depth[pix] = z, primid[pix] = k; // LESS (src/invert.cpp:1384)
            }
    }
    if (n_front) *n_front = front;
    return 0;
}

int orc_lbvh_build(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, uint64_t* morton,
                   uint32_t* order, uint32_t* child, uint32_t* parent, uint32_t* leaf_parent, float* box) {
    std::vector<Tri> tris;
    int rc = assemble(xyzw, nv, idx, nidx, tris);
    if (rc) return rc;
    Lbvh b;
    lbvh_build(tris, b);
    uint32_t n = b.n;
    std::memcpy(morton, b.morton.data(), (size_t)n * 8);
    std::memcpy(order, b.order.data(), (size_t)n * 4);
    if (n >= 2) {
        std::memcpy(child, b.child.data(), (size_t)(n - 1) * 8);
        std::memcpy(parent, b.parent.data(), (size_t)(n - 1) * 4);
        std::memcpy(leaf_parent, b.leaf_parent.data(), (size_t)n * 4);
        std::memcpy(box, b.box.data(), (size_t)(n - 1) * 48);
    }
    return 0;
}

// The occlusion rays of one pixel: origin offset and the `rays` directions ao_factor would trace for a hit on a
// triangle with edges e1, e2 (hit point at the origin, guard pad 0), without tracing anything.
void orc_ao_directions(const float e1[3], const float e2[3], uint32_t pix, uint32_t seed, uint32_t rays, float* w3) {
    Frame f = {};
    f.pad = 0.0f;
    AoParams ao = {rays, 1.0f, seed};
    Ray r = {};
    Best b = {};
    b.t = 0.0f;
    Tri tr = {};
    tr.e1 = {e1[0], e1[1], e1[2]}, tr.e2 = {e2[0], e2[1], e2[2]};
    uint32_t k = 0;
    (void)ao_factor(f, ao, pix, r, b, tr, [&](const V3&, const V3& w, float) {
        w3[3 * k] = w.x, w3[3 * k + 1] = w.y, w3[3 * k + 2] = w.z;
        k++;
        return false;
    });
}

uint8_t orc_srgb8(float x) { return srgb8(x); }
uint32_t orc_hash32(uint32_t x) { return hash32(x); }
void orc_sample_offset(uint32_t pixel, uint32_t sample, uint32_t seed, float* sx, float* sy) {
    sample_offset(pixel, sample, seed, sx, sy);
}

} // extern "C"
