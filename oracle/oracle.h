/*
 * CPU oracle for the per-pixel hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar restatement, in portable C++ float32 arithmetic with FMA
 * contraction disabled, of what the reference's pipeline defines for every
 * pixel.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; nothing in renderer_amd/ links or calls it.
 *
 * PARITY UNPINNED: the reference (a Win32 + Vulkan GUI program) ships no test,
 * golden image or known-answer vector, never reads its framebuffer back and
 * cannot be built in this environment (needs <windows.h>, the Vulkan SDK, GLM,
 * slangc, MSVC).  The oracle is therefore pinned only by (a) the reference's
 * source-level definitions cited next to each function, and (b) facts derived
 * from res/bunny.obj that the tests freeze as fixtures.
 *
 * Third-party arithmetic restated here: GLM (glm::lookAt, glm::rotate,
 * glm::perspective with GLM_FORCE_DEPTH_ZERO_TO_ONE, glm::radians; version
 * unpinned by the reference, src/CMakeLists.txt:3) -- the published
 * right-handed, zero-to-one formulas are used.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same memory layout as inv_camera / inv_params in include/invert.h. */
typedef struct orc_camera {
    float model[16]; /* column-major, as GLM uploads them (src/invert.cpp:266-271,641) */
    float view[16];
    float proj[16];
    float cam_angle[3]; /* push constant (src/invert.cpp:633,817) */
} orc_camera;

typedef struct orc_params {
    uint32_t width, height;
    uint32_t spp;
    uint32_t ao_rays;
    uint32_t flags;
    uint32_t seed;
    uint32_t tile;
    uint32_t rank, nranks;
    float ao_tmax;
} orc_params;

enum {
    ORC_LAMBERT = 1u,     /* multiply colour by max(0, n . -cam_angle) */
    ORC_INVERT = 2u,      /* c' = 255 - c on R,G,B after the sRGB encode */
    ORC_RGBA = 4u,        /* byte order R,G,B,A instead of the reference's B,G,R,A */
    ORC_BRUTE_FORCE = 8u, /* ignored by the oracle's brute-force entry point */
};

typedef struct orc_counters {
    uint64_t rays;
    uint64_t box_tests;
    uint64_t tri_tests;
    uint64_t hits;
} orc_counters;

/* A1: src/invert.cpp:42-209.  Returns 0 on success; arrays are malloc'd. */
int orc_load_obj(const char* path, float** xyzw, uint32_t* nv, uint32_t** idx, uint32_t* nidx);
int orc_parse_obj(const char* buf, uint64_t size, float** xyzw, uint32_t* nv, uint32_t** idx, uint32_t* nidx);
void orc_free(void* p);

/* A2: src/invert.cpp:621-642. */
void orc_camera_reference(uint32_t width, uint32_t height, orc_camera* cam);
void orc_camera_turntable(uint32_t width, uint32_t height, float elapsed_seconds, orc_camera* cam);

/* A3: src/shader.slang:22-29.  clip = P*(V*(M*(pos,1))), colour = (1,1,10*z). */
void orc_vertex_shader(const float* xyzw, uint32_t nv, const orc_camera* cam, float* clip4, float* color3);

/* Ray set-up shared by both renderers: object-space origin, the 3x3 direction
 * matrix, projection terms and the guard pad.  out[0..2] origin, out[3..11]
 * direction matrix columns, out[12..15] P00,P11,P22,P32, out[16] guard pad. */
int orc_frame_setup(const float* xyzw, uint32_t nv, const orc_camera* cam, float* out17);

/* X1+X2+A4+A5: brute force over all triangles for the pixel rectangle
 * [x0,x1) x [y0,y1).  Buffers are full-frame (width*height); pixels outside the
 * rectangle are left untouched. */
int orc_render_bruteforce(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx,
                          const orc_camera* cam, const orc_params* prm,
                          uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                          uint8_t* color, float* depth, uint32_t* primid, int nthreads);

/* Same result through the CPU LBVH (validated against the brute force). */
int orc_render_bvh(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx,
                   const orc_camera* cam, const orc_params* prm,
                   uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   uint8_t* color, float* depth, uint32_t* primid, int nthreads,
                   orc_counters* counters);

/* A3+A4 as a double-precision scanline rasteriser (cross-check of the ray
 * formulation against the rasteriser the reference actually drives). */
int orc_raster(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx,
               const orc_camera* cam, uint32_t width, uint32_t height,
               double* depth, uint32_t* primid, uint32_t* n_front);

/* LBVH restatement (same algorithm as the GPU build) for structural comparison.
 * morton[n] (u64), order[n] (sorted position -> original triangle),
 * node arrays of n-1 internal nodes: child[2*(n-1)], parent[n-1], leaf_parent[n],
 * box[(n-1)*12] = child0 lo xyz, hi xyz, child1 lo xyz, hi xyz. */
int orc_lbvh_build(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx,
                   uint64_t* morton, uint32_t* order, uint32_t* child, uint32_t* parent,
                   uint32_t* leaf_parent, float* box);

void orc_ao_directions(const float e1[3], const float e2[3], uint32_t pix, uint32_t seed, uint32_t rays, float* w3);
uint8_t orc_srgb8(float x);
uint32_t orc_hash32(uint32_t x);
void orc_sample_offset(uint32_t pixel, uint32_t sample, uint32_t seed, float* sx, float* sy);

#ifdef __cplusplus
}
#endif
#endif
