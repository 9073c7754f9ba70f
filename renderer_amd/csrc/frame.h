// Per-frame constants and output pointers shared by host and device code.
#pragma once

#include <stdint.h>

namespace inv {

// Per-frame constants: what the reference passes as the uniform buffer, the push
// constant and the dynamic viewport (src/invert.cpp:266-271, 779-791, 817),
// reduced on the host to the terms the kernels use.
struct FrameK {
    float ox, oy, oz;    // ray origin in object space (inverse of view*model applied to the eye)
    float d0x, d0y, d0z; // columns of the 3x3 that carries a view-space direction to object space
    float d1x, d1y, d1z;
    float d2x, d2y, d2z;
    float p00, p11, p22, p32; // GLM perspective terms proj[0][0], [1][1], [2][2], [3][2]
    float pad;                // guard pad: 2^-16 * max(|origin|, |scene coordinates|)
    float lx, ly, lz;         // unit direction towards the light in object space (Lambert); 0 if undefined
    float wf, hf;
    float rwf, rhf;           // 1.0f / wf, 1.0f / hf, correctly rounded (host division): see div_by
    float mv[12];             // rows 0..2 of view*model (object -> view space), 4 floats each
    float inv_p00, inv_p11;   // 1.0f / P00, 1.0f / P11, correctly rounded (div_by; also used by the culling)
    // approximate helpers for conservative culling only (never used for a pixel's value)
    float inv_w2, inv_h2;     // 2/width, 2/height
    float dir_abs_max;        // bound on |component| of the object-space ray direction over the frame's hit rectangle
    uint32_t rx0, ry0, rx1, ry1; // pixel rectangle that can contain hits (projected scene bounds); rest is clear
    uint32_t w, h;
    uint32_t spp, flags, seed;
    // Tile sharding.  The shards are dealt from a grid of tiles_x * tiles_y tiles of tile * tile pixels whose first
    // tile is (gx0, gy0) in tile units: the whole frame (dense), or only the tiles the hit rectangle touches
    // (sparse).  Grid tile id belongs to rank id % nranks and is that rank's slot id / nranks.
    uint32_t tile, rank, nranks, tiles_x, tiles_y, ntiles, slots; // slots = tiles stored per rank
    uint32_t gx0, gy0;
    // INV_ROOT_IDLE: `nranks` counts the ranks that tiles are dealt to (all but rank 0) and `rank` is this rank's place
    // among them (rank 0 itself: a value no tile id maps to); rank k's buffer is the (k + first_rank)-th of a gather.
    uint32_t first_rank;
    uint32_t rowmajor; // outputs are a row-major width * height frame (one rank, or rank 0 of a sparse composite);
                       // otherwise this rank's slots one after another, tile * tile pixels each
    uint64_t npix;                                       // pixels this launch covers
};

// Ambient-occlusion pass: hit pixels are queued by the primary kernel and finished by k_ao.
// A queue record is three float4: (origin.xyz, r) (normal.xyz, g) (b, pixel, out index low, out index high).
// The occlusion kernel counts occluded rays per record in `acc`; a last small kernel turns the counts into colours.
struct AoK {
    uint32_t rays;   // 0 = off
    float tmax;
    float* queue;    // 12 floats per record
    uint32_t* count; // records queued (zeroed before the primary kernel)
    uint32_t* acc;   // per record: rays occluded (zeroed by the lane that queues the record)
};

struct Outputs {
    uint8_t* color;
    float* depth;
    uint32_t* primid;
};

} // namespace inv
