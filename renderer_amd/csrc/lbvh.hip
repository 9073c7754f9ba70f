// GPU LBVH build for gfx950: scene bounds -> 63-bit Morton codes of the triangle box centres ->
// LDS-staged 8-bit radix sort (stable, so ties keep primitive order; the lower digits by a kernel that
// orders short runs in place when it can) -> Karras hierarchy -> bottom-up refit: chunks of leaves in
// LDS (which also gather the triangles into sorted order), then workgroups that climb from the nodes
// across chunk boundaries to the root, one arrival counter per internal node.
//
// The build is deterministic: every float that ends up in a node is a min/max of input floats,
// the sort is stable and the hierarchy depends only on the sorted keys, so the tree is identical
// on every rank and equal to a sequential restatement of the same steps.
#include <cstdlib>
#include <cstring>

#include "internal.h"
#include "rt_core.h"

namespace inv {

namespace {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems; // keys per workgroup per pass

// ------------------------------------------------------------------------------------------
// 1. scene bounds (two-stage min/max reduction, no float atomics; the first stage is the assembly kernel's)
// ------------------------------------------------------------------------------------------
// Workgroups go to the eight XCDs in turn, each with an L2 of its own.  Where neighbouring blocks of work touch
// neighbouring memory, XCD x takes the x-th eighth of the blocks instead of every eighth block: launch
// xcd_grid(nblocks) workgroups and leave if xcd_block() is not below nblocks.
__device__ __forceinline__ uint32_t xcd_block(uint32_t nblocks) {
    return (blockIdx.x & 7u) * ((nblocks + 7u) >> 3) + (blockIdx.x >> 3);
}
static inline uint32_t xcd_grid(uint32_t nblocks) { return ((nblocks + 7u) >> 3) * 8u; }

// The scene bounds from their first stage: partial[block][6] = lo xyz, hi xyz over a share of the triangles, which
// the assembly kernel leaves on its way (no float atomics).  On the side it clears the build's counters (`nzero` words
// from `zero`): a fill of their own would be one more launch, and a launch of anything is 4 us behind the one before.
__global__ __launch_bounds__(1024) void k_bounds_final(const float* __restrict__ partial, uint32_t nblocks,
                                                       float* __restrict__ bbox, uint32_t* __restrict__ zero, uint32_t nzero) {
    for (uint32_t i = threadIdx.x; i < nzero; i += 1024u) zero[i] = 0u;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t b = threadIdx.x; b < nblocks; b += 1024u) {
#pragma unroll
        for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], partial[b * 6 + a]), hi[a] = fmaxf(hi[a], partial[b * 6 + 3 + a]);
    }
    block_bounds<1024>(lo, hi, bbox);
}

// ------------------------------------------------------------------------------------------
// 2. Morton keys: 21 bits per axis of the box centre, x in the highest interleaved position
// ------------------------------------------------------------------------------------------
// Bit j of an 11-bit value to bit 3 j of a 32-bit word.
__device__ __forceinline__ uint32_t spread11(uint32_t v) {
    v = (v | (v << 16)) & 0x070000FFu;
    v = (v | (v << 8)) & 0x0700F00Fu;
    v = (v | (v << 4)) & 0x430C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}
// The key of a cell (21 bits per axis): bit i of x at bit 3 i + 2, of y at 3 i + 1, of z at 3 i.  Made as two 32-bit
// words (the key's low word holds bits 0-10 of z and y and bits 0-9 of x), a third of the instructions of shifting
// and masking 64-bit values.
__device__ __forceinline__ uint64_t morton63(uint32_t x, uint32_t y, uint32_t z) {
    const uint32_t lo = spread11(z & 0x7FFu) | (spread11(y & 0x7FFu) << 1) | (spread11(x & 0x3FFu) << 2);
    const uint32_t hi = spread11(x >> 10) | (spread11(z >> 11) << 1) | (spread11(y >> 11) << 2);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t grid21(float c, float lo, float ext) {
    float q = (ext > 0.0f) ? (c - lo) / ext : 0.0f;
    float s = q * 2097152.0f;
    if (!(s > 0.0f)) return 0u;
    if (s >= 2097151.0f) return 2097151u;
    return (uint32_t)s;
}

// The record comes from memory, or -- with `mesh` -- is assembled again from a mesh that the caches hold (the same
// floats: assemble_record), which saves reading 48 bytes per triangle.
// A workgroup makes the keys of one tile of the sort (kSortTile triangles) and counts their digits of the sort's first
// pass on the way: that pass then needs no histogram kernel of its own (`counts` as k_sort_hist leaves them).
constexpr int kMortonThreads = 1024; // a workgroup per sort tile: two triangles per lane
__global__ __launch_bounds__(kMortonThreads) void k_morton(const float4* __restrict__ T, MeshSource mesh, uint32_t n,
                                                           const float* __restrict__ bbox, uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ vals, int shift, uint32_t ntiles,
                                                           uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[256];
    // (tile t writes counter t of each of the 256 digit rows: neighbouring tiles on one XCD fill whole lines)
    const uint32_t tile = xcd_block(ntiles);
    if (tile >= ntiles) return; // block-uniform
    if (threadIdx.x < 256u) hist[threadIdx.x] = 0;
    __syncthreads();
    // one grid step for all three axes (the largest extent): cubic cells, whatever the shape of the scene -- with a
    // step per axis a flat scene (a grid of instances in a plane) gets cells that are thin slabs, and consecutive keys
    // then walk around an instance instead of across a patch of its surface
    const float ex = fmaxf(bbox[3] - bbox[0], fmaxf(bbox[4] - bbox[1], bbox[5] - bbox[2]));
    // What a triangle needs from memory comes in three dependent steps (indices, vertices, translation), and five
    // million triangles are only 2 484 tiles: two triangles per lane, both in flight at once, instead of eight one after
    // the other.  (The kernel's time is mostly its arithmetic: without the gathers it takes 26 us of 33, without the
    // stores 29, without the counts 31.)
    constexpr uint32_t kPer = (uint32_t)kSortTile / (uint32_t)kMortonThreads;
    static_assert(kPer * kMortonThreads == kSortTile, "a tile is a whole number of triangles per lane");
    uint32_t g[kPer];
    float4 ra[kPer], rb[kPer], rc[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; u++) g[u] = min(tile * (uint32_t)kSortTile + u * (uint32_t)kMortonThreads + threadIdx.x, n - 1u);
    if (mesh.xyzw) { // (the same for every lane)
        assemble_records<(int)kPer>(mesh, g, ra, rb, rc);
    } else {
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) ra[u] = T[3 * (size_t)g[u]], rb[u] = T[3 * (size_t)g[u] + 1], rc[u] = T[3 * (size_t)g[u] + 2];
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; u++) {
        const uint32_t i = tile * (uint32_t)kSortTile + u * (uint32_t)kMortonThreads + threadIdx.x;
        float lo[3], hi[3];
        tri_box(ra[u], rb[u], rc[u], lo, hi);
        const float cx = (lo[0] + hi[0]) * 0.5f, cy = (lo[1] + hi[1]) * 0.5f, cz = (lo[2] + hi[2]) * 0.5f;
        const uint64_t key = morton63(grid21(cx, bbox[0], ex), grid21(cy, bbox[1], ex), grid21(cz, bbox[2], ex));
        if (i < n) {
            keys[i] = key;
            vals[i] = i;
            atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256u) counts[threadIdx.x * ntiles + tile] = hist[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// 3. radix sort, 8 bits per pass: per-tile digit histogram -> exclusive scan over
//    (digit, tile) -> LDS-staged stable scatter
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n, int shift,
                                                            uint32_t ntiles, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[256];
    // (tile t writes counter t of each of the 256 digit rows: neighbouring tiles on one XCD fill whole lines)
    const uint32_t tile = xcd_block(ntiles);
    if (tile >= ntiles) return; // block-uniform
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = tile * kSortTile;
    uint64_t key[kSortItems]; // (all asked for at once, from places that exist: loads under a condition wait for each other)
#pragma unroll
    for (int r = 0; r < kSortItems; r++) key[r] = keys[min(base + r * kSortThreads + threadIdx.x, n - 1u)];
#pragma unroll
    for (int r = 0; r < kSortItems; r++)
        if (base + r * kSortThreads + threadIdx.x < n) atomicAdd(&hist[(uint32_t)(key[r] >> shift) & 255u], 1u);
    __syncthreads();
    counts[threadIdx.x * ntiles + tile] = hist[threadIdx.x];
}

// Exclusive scan of `len` counters.  Small arrays: one workgroup of 1024 threads, a chunk per thread.
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t* __restrict__ counts, uint32_t len) {
    __shared__ uint32_t part[1024];
    uint32_t chunk = (len + 1023u) / 1024u;
    uint32_t b = threadIdx.x * chunk, e = min(b + chunk, len);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += counts[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
        uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (uint32_t i = b; i < e; i++) {
        uint32_t c = counts[i];
        counts[i] = run;
        run += c;
    }
}

// Large arrays: two levels.  (a) every workgroup scans its own kScanBlock counters in place and writes its total,
// (b) one workgroup scans the totals (k_sort_scan); the scatter adds a counter's block offset itself.
constexpr uint32_t kScanBlock = 4096; // 1024 threads x 4 counters

__global__ __launch_bounds__(1024) void k_scan_local(uint32_t* __restrict__ counts, uint32_t len,
                                                     uint32_t* __restrict__ totals) {
    __shared__ uint32_t part[1024];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = base + k < len ? counts[base + k] : 0u;
        sum += v[k];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t a = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < len) counts[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 1023) totals[blockIdx.x] = part[1023];
}

// One pass's scatter.  Wave w of a tile takes the keys [512 w, 512 w + 512) of the tile, 64 at a time, so the rank
// of a key among the tile's keys of the same digit (in input order: the sort is stable) is
//     (keys of that digit in lower waves) + (in this wave's earlier rounds) + (in lower lanes of this round),
// and the last two need no exchange between waves: per round eight ballots find the lanes with the same digit and
// the wave's own counters in LDS carry the rounds.  One barrier, then thread d turns the four waves' counts of digit
// d into their prefix.  `totals`: the offsets' second scan level (offset of each block of kScanBlock counters), or
// null when the offsets are already complete; with `raw_blocks` they are that many block totals not yet added up
// (at most kSortThreads of them), and every workgroup adds up the ones before its counters itself (a scan over the
// workgroup's lanes, on the side), against a launch of one workgroup between the passes' other kernels.
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t* __restrict__ keys_in,
                                                               const uint32_t* __restrict__ vals_in, uint32_t n,
                                                               int shift, uint32_t ntiles,
                                                               const uint32_t* __restrict__ offsets,
                                                               const uint32_t* __restrict__ totals, uint32_t raw_blocks,
                                                               uint64_t* __restrict__ keys_out,
                                                               uint32_t* __restrict__ vals_out) {
    __shared__ uint64_t s_key[kSortTile];
    __shared__ uint32_t s_val[kSortTile];
    __shared__ uint32_t s_wave[4][256]; // per wave: keys of each digit so far, later the count in lower waves
    __shared__ uint32_t s_run[256];     // the tile's keys of each digit, later the tile-local digit start
    __shared__ uint32_t s_goff[256];    // global start of (digit, this tile)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // Neighbouring tiles write neighbouring short runs (a tile has eight keys per digit on average): handled by the
    // same XCD one after the other, the runs meet in one L2 and leave it as whole lines.
    const uint32_t tile = xcd_block(ntiles);
    if (tile >= ntiles) return; // block-uniform
    const uint32_t base = tile * kSortTile;
    const uint32_t tile_n = min((uint32_t)kSortTile, n - base);
    __shared__ uint32_t s_tot[kSortThreads], s_wtot[4];
    const uint32_t at = tid * ntiles + tile; // this thread's digit, this tile
    uint32_t goff = offsets[at], block_total = 0;
    if (raw_blocks)
        block_total = tid < raw_blocks ? totals[tid] : 0u;
    else if (totals)
        goff += totals[at / kScanBlock];
#pragma unroll
    for (int w = 0; w < 4; w++) s_wave[w][tid] = 0;
    __syncthreads();

    uint64_t key[kSortItems];
    uint32_t val[kSortItems], rank[kSortItems];
    uint32_t* mine = s_wave[wave];
    // (all of a lane's keys are asked for at once, from places that exist whether the key does or not: loads under a
    // condition each wait for the one before)
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t j = min(wave * (kSortItems * 64u) + r * 64u + lane, tile_n - 1u);
        key[r] = keys_in[base + j], val[r] = vals_in[base + j];
    }
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t j = wave * (kSortItems * 64u) + r * 64u + lane;
        const bool valid = j < tile_n;
        if (!valid) key[r] = ~0ull, val[r] = 0u;
        const uint32_t digit = (uint32_t)(key[r] >> shift) & 255u;
        // lanes of this wave holding the same digit: per bit of the digit, the lanes whose bit differs from this lane's
        // are the ballot of the bit, complemented where this lane's bit is set (an exclusive or with 0 or ~0)
        uint32_t dlo = 0, dhi = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)digit, b, 1); // ~0 if the bit is set, else 0
            const uint64_t bal = __ballot(m != 0u);
            dlo |= (uint32_t)bal ^ m, dhi |= (uint32_t)(bal >> 32) ^ m;
        }
        const uint64_t same = ~(((uint64_t)dhi << 32) | dlo) & __ballot(valid);
        const uint32_t before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        // a wave's LDS accesses execute in program order: every lane reads the count, then one lane per digit adds
        const uint32_t earlier = mine[digit];
        rank[r] = earlier + before;
        if (valid && before == 0) mine[digit] = earlier + (uint32_t)__popcll(same);
    }
    __syncthreads();
    {
        const uint32_t c0 = s_wave[0][tid], c1 = s_wave[1][tid], c2 = s_wave[2][tid], c3 = s_wave[3][tid];
        s_wave[0][tid] = 0, s_wave[1][tid] = c0, s_wave[2][tid] = c0 + c1, s_wave[3][tid] = c0 + c1 + c2;
        s_run[tid] = c0 + c1 + c2 + c3;
        // on the side: the block totals before each block, within the wave here, the waves before it added below
        uint32_t incl = block_total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= (unsigned)o) incl += v;
        }
        s_tot[tid] = incl - block_total;
        if (lane == 63) s_wtot[wave] = incl;
    }
    __syncthreads();
    if (raw_blocks) {
        const uint32_t blk = at / kScanBlock;
        goff += s_tot[blk];
        for (uint32_t w = 0; w < (blk >> 6); w++) goff += s_wtot[w];
    }
    s_goff[tid] = goff; // (read after the barriers below)
    // exclusive scan of the tile's digit counts -> tile-local digit start (s_run becomes the start)
    {
        uint32_t c = s_run[tid];
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= (unsigned)o) incl += v;
        }
        __shared__ uint32_t s_wsum[4];
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t prev = 0;
        for (uint32_t w = 0; w < wave; w++) prev += s_wsum[w];
        s_run[tid] = prev + incl - c;
        __syncthreads();
    }
    // stage into LDS in digit order, then write runs out with consecutive lanes on consecutive addresses
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t j = wave * (kSortItems * 64u) + r * 64u + lane;
        if (j < tile_n) {
            const uint32_t digit = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t p = s_run[digit] + mine[digit] + rank[r];
            s_key[p] = key[r];
            s_val[p] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t j = r * kSortThreads + tid;
        if (j < tile_n) {
            const uint64_t k = s_key[j];
            const uint32_t digit = (uint32_t)(k >> shift) & 255u;
            const uint32_t g = s_goff[digit] + (j - s_run[digit]);
            keys_out[g] = k;
            vals_out[g] = s_val[j];
        }
    }
}

// The short forms of the sort: the passes over the upper digits only order the keys by their bits from `shift` up
// (stably, so keys that agree there stay in index order); what is left is to order every run of keys with equal upper
// bits by the lower ones.  Such runs are triangles whose box centres share a cell of the scene's extent -- from bit 32
// up a cell is a 2048th of it in x and a 1024th in y and z, from bit 24 up an 8192nd per axis -- one or a few, unless
// the scene is very large or triangles repeat.  Every key finds its run by looking at its neighbours and its place in
// it by counting the keys of the run that go before it (smaller, or equal and earlier).  A run of more than kRunMax
// keys is copied as it is and the flag is raised: the host then sorts again with more passes.  (A key looks at most
// kRunMax places either way, so all keys of a run come to the same verdict: a shorter run is seen whole by each of
// them, a longer one is seen whole and counted, or not seen whole.)
constexpr uint32_t kRunMax = 48;
__global__ __launch_bounds__(256) void k_sort_runs(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t n,
                                                   int shift, uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                   uint32_t* __restrict__ too_long) {
    // the workgroup's 256 keys and kRunMax on either side, in LDS: a key's looks at its neighbours are dependent steps,
    // a dozen or more of them where cells hold several triangles
    __shared__ uint64_t s_k[256 + 2 * kRunMax];
    const uint32_t base = blockIdx.x * 256u, t = threadIdx.x, i = base + t;
    const uint32_t v = i < n ? vin[i] : 0u; // (asked for now, wanted at the end)
    for (uint32_t j = t; j < 256u + 2u * kRunMax; j += 256u) {
        const int64_t g = (int64_t)base - (int64_t)kRunMax + j;
        s_k[j] = g >= 0 && g < (int64_t)n ? kin[g] : 0ull; // (places outside the array are never looked at)
    }
    __syncthreads();
    const bool valid = i < n;
    const uint32_t lane = t & 63u, p = t + kRunMax; // key i sits at s_k[p]
    const uint64_t k = s_k[p], hi = k >> shift;
    // Where the key's run begins and ends: first within the wave, from two ballots (does the key before, the key after,
    // belong to the same run?) -- no lane looks further than its neighbours unless its run reaches the wave's first or
    // last lane, and then it goes on from there, step by step.
    const bool ep = valid && i > 0u && (s_k[p - 1u] >> shift) == hi;
    const bool en = valid && i + 1u < n && (s_k[p + 1u] >> shift) == hi;
    const uint64_t mp = __ballot(ep), mn = __ballot(en);
    const uint64_t zp = ~mp & ((2ull << lane) - 1ull); // lanes up to this one whose run begins with them
    const uint64_t zn = ~mn & (~0ull << lane);         // lanes from this one on whose run ends with them
    uint32_t a = i - lane, b = i + (64u - lane); // (the wave's first key; the key after its last)
    if (zp)
        a = i - (lane - (63u - (uint32_t)__clzll((long long)zp)));
    else
        while (a > 0u && i - a < kRunMax && (s_k[p - (i - a) - 1u] >> shift) == hi) a--;
    if (zn)
        b = i + ((uint32_t)__ffsll((long long)zn) - 1u - lane) + 1u;
    else
        while (b < n && b - i <= kRunMax && (s_k[p + (b - i)] >> shift) == hi) b++; // (b - i counts key i itself)
    if (!valid) return;
    uint32_t at = i;
    if (b - a > kRunMax) {
        *too_long = 1u; // (every thread that comes here writes the same value)
    } else if (b - a > 1u) {
        uint32_t before = 0;
        for (uint32_t j = a; j < b; j++) {
            const uint64_t kj = s_k[p + j - i];
            before += (kj < k || (kj == k && j < i)) ? 1u : 0u;
        }
        at = a + before;
    }
    kout[at] = k, vout[at] = v;
}

// ------------------------------------------------------------------------------------------
// 4. hierarchy: T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and
//    k-d Trees", HPG 2012, section 4.  Keys are made unique by appending the sorted position.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int delta(const uint64_t* __restrict__ m, uint32_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= (int64_t)n) return -1;
    uint64_t a = m[i], b = m[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

__global__ __launch_bounds__(256) void k_karras(const uint64_t* __restrict__ m, uint32_t n,
                                                uint32_t* __restrict__ child, uint32_t* __restrict__ parent,
                                                uint32_t* __restrict__ leaf_parent, uint2* __restrict__ range) {
    // (a node's searches probe keys around its own: one L2 for a stretch of nodes)
    int64_t i = (int64_t)xcd_block((n - 1 + 255u) / 256u) * 256 + threadIdx.x;
    if (i >= (int64_t)n - 1) return;
    if (i == 0) parent[0] = 0xFFFFFFFFu;
    int d = (delta(m, n, i, i + 1) - delta(m, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(m, n, i, i - d);
    int64_t lmax = 2;
    while (delta(m, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (delta(m, n, i, i + (l + t) * d) > dmin) l += t;
    int64_t j = i + l * d;
    int dnode = delta(m, n, i, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(m, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int64_t gamma = i + s * d + (d < 0 ? d : 0);
    int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    uint32_t c0, c1;
    if (lo == gamma) {
        c0 = (uint32_t)gamma | 0x80000000u;
        leaf_parent[gamma] = (uint32_t)i;
    } else {
        c0 = (uint32_t)gamma;
        parent[gamma] = (uint32_t)i;
    }
    if (hi == gamma + 1) {
        c1 = (uint32_t)(gamma + 1) | 0x80000000u;
        leaf_parent[gamma + 1] = (uint32_t)i;
    } else {
        c1 = (uint32_t)(gamma + 1);
        parent[gamma + 1] = (uint32_t)i;
    }
    child[2 * i] = c0;
    child[2 * i + 1] = c1;
    // sorted leaves covered by this node: [lo, hi]; child 0 covers [lo, gamma], child 1 [gamma + 1, hi].  Kept in an
    // array of its own: the refit writes them with the rest of the node's 64-byte record, in one piece
    range[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
}

// ------------------------------------------------------------------------------------------
// 5. refit: the chunks of leaves first, each in LDS, then the nodes that cross chunk boundaries.
//    Node i (16 floats): child0 lo xyz, hi xyz, child1 lo xyz, hi xyz, child0, child1, first and
//    last sorted leaf under the node (from the hierarchy kernel's range array).  union_box[i] (8 floats)
//    carries the node's own box and height upwards.
// ------------------------------------------------------------------------------------------
// The fine cut: every subtree (or single leaf) of at most kFineMax leaves whose parent covers more.  These subtrees
// hold every leaf exactly once; a record is 8 floats, (lo.xyz, first sorted leaf) (hi.xyz, leaf count), the box being
// the child box the parent's node record carries.  The per-frame binning pass reads them instead of walking the tree.
// A finished node emits the records of its small children; the order of the records depends on scheduling: nothing reads them in order.  The records' counter is one address
// for the whole grid (a few hundred thousand updates of one address take longer than the rest of the refit), so a
// workgroup of NT threads updates it once for all of its lanes: ballots rank the lanes within a wave, LDS ranks the
// waves.  Call from every thread (three barriers).
template <int NT>
__device__ __forceinline__ void emit_fine(bool finished, uint32_t c0, uint2 leaves, const float b0[6], const float b1[6],
                                          float* __restrict__ fine, uint32_t* __restrict__ fine_count, uint32_t fine_cap) {
    __shared__ uint32_t s_wave[NT / 64];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gamma = c0 & 0x7FFFFFFFu;
    const uint32_t n0 = gamma - leaves.x + 1u, n1 = leaves.y - gamma;
    const bool big = finished && leaves.y - leaves.x + 1u > kFineMax;
    const bool f0 = big && n0 <= kFineMax, f1 = big && n1 <= kFineMax;
    const uint64_t m0 = __ballot(f0), m1 = __ballot(f1);
    if (lane == 0) s_wave[wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < NT / 64; w++) total += s_wave[w];
        s_base = total ? atomicAdd(fine_count, total) : 0u;
    }
    __syncthreads();
    uint32_t at = s_base;
    for (uint32_t w = 0; w < wave; w++) at += s_wave[w];
    const uint64_t below = (1ull << lane) - 1ull;
    const uint32_t at0 = at + (uint32_t)__popcll(m0 & below), at1 = at + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & below);
    if (f0 && at0 < fine_cap) {
        float4* r = (float4*)fine + 2 * (size_t)at0;
        r[0] = make_float4(b0[0], b0[1], b0[2], __uint_as_float(leaves.x));
        r[1] = make_float4(b0[3], b0[4], b0[5], __uint_as_float(n0));
    }
    if (f1 && at1 < fine_cap) {
        float4* r = (float4*)fine + 2 * (size_t)at1;
        r[0] = make_float4(b1[0], b1[1], b1[2], __uint_as_float(gamma + 1u));
        r[1] = make_float4(b1[3], b1[4], b1[5], __uint_as_float(n1));
    }
    __syncthreads(); // the next call overwrites s_wave
}

// First the part of the tree that needs no exchange between workgroups.  Internal node i covers a range of sorted
// leaves that contains leaf i, so every node whose range lies inside one chunk of kChunk consecutive leaves has its
// index, its leaves and all of its descendants inside that chunk: one workgroup per chunk finishes those nodes in
// LDS, sweep by sweep (a node is finished in the sweep after the one that finished its later child, so the sweep
// number is the node's height).  What remains are the nodes whose range crosses a chunk boundary (a root-to-leaf path
// per boundary); the workgroup that owns such a node's index sets its arrival counter to the number of children that
// are already complete (leaves, and nodes finished inside a chunk) and puts it in a slot to be climbed from if both
// are.
constexpr uint32_t kChunk = 512; // (1024 leaves per workgroup: 12 % slower at five million triangles)
static_assert(kChunk == 512, "k_build_chunks packs positions within a chunk into nine bits");
constexpr uint32_t kOpen = 0xFFFFFFFFu;

// The slot of a node that waits to be climbed from: two per unit of leaves (a chunk, a span), the one for the node
// that reaches across the unit's lower end and the one for the node that reaches across its upper end (see
// k_refit_climb); `index` is the node's own, which is one of its leaves'.
__device__ __forceinline__ uint32_t boundary_slot(uint32_t index, uint2 leaves, uint32_t unit) {
    const uint32_t u = index / unit;
    return 2u * u + (leaves.x < u * unit ? 0u : 1u);
}


__global__ __launch_bounds__(kChunk) void k_refit_chunks(const float4* __restrict__ T0, MeshSource mesh, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ slot_of, float4* __restrict__ TA,
                                                         float4* __restrict__ TB, float4* __restrict__ TC,
                                                         const uint32_t* __restrict__ child,
                                                         const uint32_t* __restrict__ parent,
                                                         const uint2* __restrict__ range, uint32_t n,
                                                         float* __restrict__ union_box,
                                                         float* __restrict__ nodes, uint32_t* __restrict__ arrive,
                                                         uint4* __restrict__ slots, uint32_t* __restrict__ root_done,
                                                         uint32_t* __restrict__ finished_count,
                                                         float* __restrict__ fine, uint32_t* __restrict__ fine_count,
                                                         uint32_t fine_cap) {
    __shared__ float s_leaf[6][kChunk]; // boxes of the chunk's leaves
    __shared__ float s_box[6][kChunk];  // boxes of the internal nodes finished here
    __shared__ uint32_t s_sweep[kChunk]; // sweep that finished the node, kOpen while it waits
    const uint32_t chunk = blockIdx.x; // (taking the chunks XCD by XCD instead makes no difference)
    const uint32_t t = threadIdx.x, base = chunk * kChunk, i = base + t;
    // Everything this lane reads by its own index is asked for at once, and what hangs on those answers (the record,
    // the leaves of children in other chunks) in a second round: a workgroup lives for some thirty microseconds, most
    // of them waiting, and every round of loads that follows another is one or two of them.
    const bool valid = i + 1u < n;
    const uint32_t src = i < n ? order[i] : 0u;
    uint32_t c0 = 0, c1 = 0, up = 0xFFFFFFFFu;
    uint2 leaves = make_uint2(0u, 0u);
    if (valid) {
        const uint2 c = ((const uint2*)child)[i];
        c0 = c.x, c1 = c.y;
        leaves = range[i];
        up = parent[i]; // (the root has none: 0xFFFFFFFF)
    }
    const bool leaf0 = (c0 >> 31) != 0u, leaf1 = (c1 >> 31) != 0u;
    const uint32_t k0 = (c0 & 0x7FFFFFFFu) - base, k1 = (c1 & 0x7FFFFFFFu) - base; // >= kChunk: in another chunk
    uint2 far0 = make_uint2(0u, 1u), far1 = make_uint2(0u, 1u); // leaves of a child node of another chunk
    if (valid && !leaf0 && k0 >= kChunk) far0 = range[c0];
    if (valid && !leaf1 && k1 >= kChunk) far1 = range[c1];
    if (i < n) {
        // The gather into sorted order happens here, on the way: the triangle at sorted position i is the assembled
        // record order[i].  (A kernel of its own for it read the records, wrote them sorted, and this one read them
        // again: a sixth of a five-million-triangle build.)
        // With `mesh` (one that the caches hold) the record is assembled again from it, the same floats: the assembled
        // records are then never stored at all, which saves writing 48 bytes per triangle and fetching them here from
        // all over memory.
        float4 a, b, c;
        if (mesh.xyzw) { // (the same for every lane)
            assemble_record(mesh, src, a, b, c);
        } else {
            const float4* rec = T0 + 3 * (size_t)src;
            a = rec[0], b = rec[1], c = rec[2];
        }
        TA[i] = a, TB[i] = b, TC[i] = c;
        if (slot_of) slot_of[src] = i; // the inverse of `order`: primitive id -> sorted position (null: nobody will ask)
        float lo[3], hi[3];
        tri_box(a, b, c, lo, hi);
#pragma unroll
        for (int a = 0; a < 3; a++) s_leaf[a][t] = lo[a], s_leaf[3 + a][t] = hi[a];
    }
    s_sweep[t] = kOpen;
    bool open = valid && k0 < kChunk && k1 < kChunk;
    float b0[6] = {0, 0, 0, 0, 0, 0}, b1[6] = {0, 0, 0, 0, 0, 0}, u[6];
    uint32_t height = kOpen;
    __syncthreads();
    for (uint32_t sweep = 1;; sweep++) {
        bool did = false;
        // a child finished in this very sweep reads as kOpen or as `sweep`: not ready either way, and its box is
        // not looked at before the barrier below
        if (open && (leaf0 || s_sweep[k0] < sweep) && (leaf1 || s_sweep[k1] < sweep)) {
#pragma unroll
            for (int a = 0; a < 6; a++) {
                b0[a] = leaf0 ? s_leaf[a][k0] : s_box[a][k0];
                b1[a] = leaf1 ? s_leaf[a][k1] : s_box[a][k1];
            }
#pragma unroll
            for (int a = 0; a < 3; a++) u[a] = fminf(b0[a], b1[a]), u[3 + a] = fmaxf(b0[3 + a], b1[3 + a]);
#pragma unroll
            for (int a = 0; a < 6; a++) s_box[a][t] = u[a];
            s_sweep[t] = height = sweep;
            open = false, did = true;
        }
        if (!__syncthreads_or(did)) break;
    }
    // the finished nodes' records, all at once: neighbouring lanes write neighbouring 64-byte records, each whole
    if (height != kOpen) {
        float4* nd = (float4*)nodes + 4 * (size_t)i;
        nd[0] = make_float4(b0[0], b0[1], b0[2], b0[3]);
        nd[1] = make_float4(b0[4], b0[5], b1[0], b1[1]);
        nd[2] = make_float4(b1[2], b1[3], b1[4], b1[5]);
        nd[3] = make_float4(__uint_as_float(c0), __uint_as_float(c1), __uint_as_float(leaves.x), __uint_as_float(leaves.y));
        // the node's own box is read again only by a parent that is not finished here (or, the root's, by the host)
        const uint32_t p = up - base; // the root has no parent: 0xFFFFFFFF is in no chunk
        if (p >= kChunk || s_sweep[p] == kOpen) {
            float4* ub = (float4*)union_box + 2 * (size_t)i;
            ub[0] = make_float4(u[0], u[1], u[2], u[3]);
            ub[1] = make_float4(u[4], u[5], __uint_as_float(height), 0.0f);
        }
    }
    emit_fine<(int)kChunk>(height != kOpen, c0, leaves, b0, b1, fine, fine_count, fine_cap);
    const bool crossing = valid && height == kOpen;
    uint32_t arrived = 0;
    if (crossing) {
        // a leaf counts (the climb reads its triangle itself); a node of this chunk if it was finished here; a node
        // of another chunk if its leaves all lie in that chunk: it is finished there
        arrived += leaf0 ? 1u : k0 < kChunk ? (s_sweep[k0] != kOpen ? 1u : 0u) : (far0.x / kChunk == far0.y / kChunk ? 1u : 0u);
        arrived += leaf1 ? 1u : k1 < kChunk ? (s_sweep[k1] != kOpen ? 1u : 0u) : (far1.x / kChunk == far1.y / kChunk ? 1u : 0u);
        arrive[i] = arrived;
    }
    const uint32_t finished = (uint32_t)__syncthreads_count(valid && !crossing);
    if (t == 0 && finished) atomicAdd(finished_count, finished);
    // the nodes that are ready to be climbed from, with their links, in the chunk's two slots (see k_refit_climb)
    __shared__ uint32_t s_taken[2];
    if (t < 2u) s_taken[t] = 0u;
    __syncthreads();
    const bool ready = crossing && arrived == 2u;
    if (ready) {
        const uint32_t at = boundary_slot(i, leaves, kChunk);
        s_taken[at & 1u] = 1u;
        slots[2 * (size_t)at] = make_uint4(i + 1u, c0, c1, up);
        slots[2 * (size_t)at + 1] = make_uint4(leaves.x, leaves.y, 0u, 0u);
    }
    __syncthreads();
    if (t < 2u && !s_taken[t]) slots[2 * (size_t)(2u * chunk + t)] = make_uint4(0u, 0u, 0u, 0u);
    if (i == 0u && height != kOpen) *root_done = 1u; // one chunk holds it all
}

// ------------------------------------------------------------------------------------------
// 5b. hierarchy and boxes of a chunk in one pass, bottom-up (C. Apetrei, "Fast and Simple Agglomerative LBVH
//     Construction", 2014), in LDS.  A node over the sorted leaves [l, r] is the child of the node that also holds
//     whichever neighbour, l - 1 or r + 1, shares the longer key prefix with it: with delta(p) the common prefix of
//     the keys at p and p + 1 (index bits appended, so no two boundaries of one node tie), it is the left child of the
//     node split at r if delta(r) > delta(l - 1), else the right child of the node split at l - 1.  Two children meet at
//     their parent's split: the first to arrive leaves the far end of its range there and stops, the second merges
//     and goes on.  No node searches for anything.  In T. Karras's numbering (which the rest of the build and the
//     sequential restatement use) a left child is named after the last leaf of its range and a right child after the
//     first, so a node learns its own name when it learns which side its parent is on, and the children of a node
//     split at g are g and g + 1.
//     What a chunk cannot finish are the nodes whose leaves reach into another chunk.  It hands up its "tops": the
//     subtrees (or single leaves) it made whose sibling lies elsewhere, a dozen or so, in leaf order, each with its box,
//     its leaves, its height and the prefixes at its two ends.  k_build_level does the same merging over the tops of
//     32 units at a time, again in LDS, and hands up its own tops, until one workgroup holds what is left: three
//     further launches for five million triangles, no node ever searching or waiting for anything in memory.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMerged = 0xFFFFFFFFu;
typedef float v4f __attribute__((ext_vector_type(4)));
// min and max of box coordinates as single instructions (fminf and fmaxf first turn each operand into its canonical form,
// an instruction each, which changes nothing for the numbers boxes are made of)
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Put the 32 bytes m0, m1 at `mine`, swap `word` into *flag and read the 32 bytes at `slot` behind it, all in LDS (the
// slots 16-byte aligned), without waiting for the swap's answer in between: returns what the flag held.  Written out
// because the order of the five is the point -- left to itself the compiler moves the reads to where their values are
// used, behind the wait for the answer.  (The low half of an LDS pointer is the offset the instructions take.)
// A slot's two halves lie HALF bytes apart (an array of first halves, then one of second halves: lanes that take
// neighbouring slots then touch neighbouring 16-byte pieces, where 32-byte slots made every access a two-way bank
// conflict).
template <int HALF>
__device__ __forceinline__ uint32_t lds_put_swap_then_read(float4* mine, v4f m0, v4f m1, uint32_t* flag, uint32_t word, const float4* slot,
                                                           v4f& s0, v4f& s1) {
    uint32_t other;
    const uint32_t ma = (uint32_t)(uintptr_t)mine, fa = (uint32_t)(uintptr_t)flag, sa = (uint32_t)(uintptr_t)slot;
    asm volatile("ds_write_b128 %6, %7\n\tds_write_b128 %6, %8 offset:%9\n\tds_wrxchg_rtn_b32 %0, %3, %4\n\tds_read_b128 %1, %5\n\t"
                 "ds_read_b128 %2, %5 offset:%9\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(other), "=&v"(s0), "=&v"(s1)
                 : "v"(fa), "v"(word), "v"(sa), "v"(ma), "v"(m0), "v"(m1), "n"(HALF)
                 : "memory");
    return other;
}
// (the same with four more bytes of one's own, `mw` to `mine_w`, and of the sibling's, from `word_at`)
template <int HALF>
__device__ __forceinline__ uint32_t lds_put_swap_then_read(float4* mine, v4f m0, v4f m1, uint32_t* mine_w, uint32_t mw, uint32_t* flag,
                                                           uint32_t word, const float4* slot, const uint32_t* word_at, v4f& s0,
                                                           v4f& s1, uint32_t& w) {
    uint32_t other;
    const uint32_t ma = (uint32_t)(uintptr_t)mine, mwa = (uint32_t)(uintptr_t)mine_w, fa = (uint32_t)(uintptr_t)flag,
                   sa = (uint32_t)(uintptr_t)slot, wa = (uint32_t)(uintptr_t)word_at;
    asm volatile("ds_write_b128 %8, %9\n\tds_write_b128 %8, %10 offset:%13\n\tds_write_b32 %11, %12\n\tds_wrxchg_rtn_b32 %0, %4, %5\n\t"
                 "ds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:%13\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(other), "=&v"(s0), "=&v"(s1), "=&v"(w)
                 : "v"(fa), "v"(word), "v"(sa), "v"(wa), "v"(ma), "v"(m0), "v"(m1), "v"(mwa), "v"(mw), "n"(HALF)
                 : "memory");
    return other;
}
constexpr uint32_t kTopCap = 64;       // tops a unit may hand up (more: the build starts again the other way)
constexpr uint32_t kLevelUnits = 32;   // units whose tops one workgroup of k_build_level merges
constexpr uint32_t kLevelItems = 768;  // tops it can hold (32 units hand up some 400; two such workgroups share a CU's LDS)
constexpr uint32_t kLevelThreads = 1024; // a lane per top: a lane with two to carry would start the second when the first has stopped
constexpr uint32_t kLevelFine = 256;   // records of the fine cut it collects before it writes them out
// A top: (box lo xyz, first leaf) (box hi xyz, last leaf) (prefix at the lower end + 1 | prefix at the upper end + 1 << 8
// | height << 16, -, -, -)

__device__ __forceinline__ int key_delta(uint64_t a, uint64_t b, uint32_t i, uint32_t j) {
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)(i ^ j));
}

__global__ __launch_bounds__(kChunk) void k_build_chunks(const float4* __restrict__ T0, MeshSource mesh,
                                                         const uint32_t* __restrict__ order, const uint64_t* __restrict__ keys,
                                                         uint32_t* __restrict__ slot_of, float4* __restrict__ TA,
                                                         float4* __restrict__ TB, float4* __restrict__ TC,
                                                         uint32_t n, float* __restrict__ union_box, float* __restrict__ nodes,
                                                         float4* __restrict__ tops, uint32_t* __restrict__ top_counts,
                                                         uint32_t top_cap, uint32_t* __restrict__ overflow, uint32_t* __restrict__ root_done,
                                                         uint32_t* __restrict__ finished_count, float* __restrict__ fine,
                                                         uint32_t* __restrict__ fine_count, uint32_t fine_cap) {
    // per split g (between the chunk's leaves g and g + 1) the two subtrees that meet there, as they arrive: [2 g] the one
    // that ends at g, [2 g + 1] the one that begins at g + 1; (box lo, height) (box hi, -)
    __shared__ float4 s_slot[2][2 * kChunk]; // ([0]: first halves, [1]: second halves)
    // first the keys of the chunk and of one leaf on either side; then, in the same place, per split: s_flag (0, or what
    // the first child to arrive left: its far end + 1 | the prefix beyond that end + 1 << 16, or kMerged) and, per node
    // made here, by name: s_info (1 << 31 | r << 18 | l << 9 | split)
    __shared__ uint64_t s_key[kChunk + 2];
    __shared__ int s_delta[kChunk + 1];   // [p + 1]: common prefix of the keys at p and p + 1, p = -1 .. kChunk - 1; -1: no such leaf
    __shared__ uint32_t s_wcount[kChunk / 64], s_wmade[kChunk / 64];
    uint32_t* const s_flag = (uint32_t*)s_key;
    uint32_t* const s_info = s_flag + kChunk;
    const uint32_t chunk = blockIdx.x, t = threadIdx.x, base = chunk * kChunk, i = base + t;
    const uint32_t count = min(kChunk, n - base); // leaves of this chunk
    const uint32_t src = i < n ? order[i] : 0u;
    s_key[t + 1] = i < n ? keys[i] : 0ull;
    if (t == 0) s_key[0] = base ? keys[base - 1] : 0ull;
    if (t == 1) s_key[kChunk + 1] = base + kChunk < n ? keys[base + kChunk] : 0ull;
    float bx[6] = {0, 0, 0, 0, 0, 0};
    if (i < n) {
        // the gather into sorted order happens here, on the way (see k_refit_chunks)
        float4 a, b, c;
        if (mesh.xyzw) { // (the same for every lane)
            assemble_record(mesh, src, a, b, c);
        } else {
            const float4* rec = T0 + 3 * (size_t)src;
            a = rec[0], b = rec[1], c = rec[2];
        }
        TA[i] = a, TB[i] = b, TC[i] = c;
        if (slot_of) slot_of[src] = i;
        float lo[3], hi[3];
        tri_box(a, b, c, lo, hi);
#pragma unroll
        for (int a = 0; a < 3; a++) bx[a] = lo[a], bx[3 + a] = hi[a];
    }
    __syncthreads();
    {
        // boundary between local positions t and t + 1 (and, by lane 0, between -1 and 0)
        const bool have = i + 1u < n;
        s_delta[t + 1] = have ? key_delta(s_key[t + 1], s_key[t + 2], i, i + 1u) : -1;
        if (t == 0) s_delta[0] = base ? key_delta(s_key[0], s_key[1], base - 1u, base) : -1;
    }
    __syncthreads(); // the keys have been read: their place is the flags' and the names' now
    s_flag[t] = 0u, s_info[t] = 0u;
    __syncthreads();
    uint32_t l = t, r = t;
    v4f lo4 = {bx[0], bx[1], bx[2], 0.0f}, hi4 = {bx[3], bx[4], bx[5], 0.0f}; // the box of [l, r]; lo4.w: the height's bits
    int dl = s_delta[t], dr = s_delta[t + 1]; // the prefixes beyond the two ends of [l, r]
    int exit_gp = -2; // the split this lane's subtree waits at; -1 or kChunk - 1: a chunk boundary; -2: no subtree (or the root)
    {
        // A trip of this loop is one exchange with LDS.  A wave's LDS operations are carried out in the order it issues
        // them, and LDS carries out one at a time: a subtree puts its box in its slot, swaps its far end into the flag and
        // reads the sibling's slot without waiting to learn whether the sibling has been -- if the flag was not empty,
        // the sibling's box went in before the sibling's swap, this swap came after that and the read after this swap.
        // (The compiler must keep that order; a fence of the memory model would also wait for the stores to global
        // memory still in flight.)  The first to arrive stops; the second has the parent's box, leaves, height and, from
        // the flag, the prefix beyond the new end, which tells it the parent's name and where the parent goes next.
        bool alive = t < count;
        while (alive) {
            if (dl < 0 && dr < 0) break; // the root, all in this chunk
            const bool right = dr > dl;
            const int gp = right ? (int)r : (int)l - 1; // the parent's split
            exit_gp = gp;
            if (gp < 0 || gp >= (int)kChunk - 1) break; // the sibling lies in another chunk
            const uint32_t mine = 2u * (uint32_t)gp + (right ? 0u : 1u);
            v4f s0, s1;
            const uint32_t other = lds_put_swap_then_read<(int)(2 * kChunk * 16)>(&s_slot[0][mine], lo4, hi4, &s_flag[gp],
                                                          ((right ? l : r) + 1u) | ((uint32_t)((right ? dl : dr) + 1) << 16),
                                                          &s_slot[0][mine ^ 1u], s0, s1);
            if (other == 0u) break; // the first to arrive
            lo4.x = min_raw(lo4.x, s0.x), lo4.y = min_raw(lo4.y, s0.y), lo4.z = min_raw(lo4.z, s0.z);
            hi4.x = max_raw(hi4.x, s1.x), hi4.y = max_raw(hi4.y, s1.y), hi4.z = max_raw(hi4.z, s1.z);
            const uint32_t h = __float_as_uint(lo4.w), sh = __float_as_uint(s0.w);
            lo4.w = __uint_as_float(1u + (h > sh ? h : sh));
            const uint32_t far = (other & 0xFFFFu) - 1u;
            const int dfar = (int)(other >> 16) - 1;
            if (right) r = far, dr = dfar; else l = far, dl = dfar;
            // the node just made, under its name (a left child is named after its last leaf, a right child after its first)
            const uint32_t me = (dl < 0 && dr < 0) ? 0u : dr > dl ? r : l;
            s_info[me] = 0x80000000u | (r << 18) | (l << 9) | (uint32_t)gp;
            s_flag[gp] = kMerged;
            exit_gp = -2;
        }
    }
    __syncthreads();
    // every node made here, by the lane of its name: its record, its links, the fine cut's records of its small
    // children; the names nobody made go on the list
    const bool valid = i + 1u < n;
    const uint32_t info = s_info[t];
    const bool made = valid && (info >> 31) != 0u;
    uint32_t c0 = 0, c1 = 0;
    uint2 leaves = make_uint2(0u, 0u);
    float b0[6] = {0, 0, 0, 0, 0, 0}, b1[6] = {0, 0, 0, 0, 0, 0};
    if (made) {
        const uint32_t g = info & 511u, l = (info >> 9) & 511u, r = (info >> 18) & 511u;
        const bool leaf0 = l == g, leaf1 = g + 1u == r;
        c0 = (base + g) | (leaf0 ? 0x80000000u : 0u), c1 = (base + g + 1u) | (leaf1 ? 0x80000000u : 0u);
        leaves = make_uint2(base + l, base + r);
        // (the record -- written below -- is all of the node there is: who asks for child or parent arrays gets them made
        // from the records)
        const float4 p0 = s_slot[0][2u * g], p1 = s_slot[1][2u * g], q0 = s_slot[0][2u * g + 1u], q1 = s_slot[1][2u * g + 1u];
        b0[0] = p0.x, b0[1] = p0.y, b0[2] = p0.z, b0[3] = p1.x, b0[4] = p1.y, b0[5] = p1.z;
        b1[0] = q0.x, b1[1] = q0.y, b1[2] = q0.z, b1[3] = q1.x, b1[4] = q1.y, b1[5] = q1.z;
        if (s_delta[l] < 0 && s_delta[r + 1u] < 0) { // the root: one chunk holds it all
            float4* ub = (float4*)union_box; // (its box and height, for the build's summary)
            const uint32_t h0 = __float_as_uint(p0.w), h1 = __float_as_uint(q0.w);
            ub[0] = make_float4(fminf(b0[0], b1[0]), fminf(b0[1], b1[1]), fminf(b0[2], b1[2]), fmaxf(b0[3], b1[3]));
            ub[1] = make_float4(fmaxf(b0[4], b1[4]), fmaxf(b0[5], b1[5]), __uint_as_float(1u + (h0 > h1 ? h0 : h1)), 0.0f);
            *root_done = 1u;
        }
    }
    emit_fine<(int)kChunk>(made, c0, leaves, b0, b1, fine, fine_count, fine_cap);
    // The records of the nodes made here.  A lane that stored its own 64 bytes would make every store instruction of
    // the wave touch 64 different lines, 16 bytes of each; through LDS (the slots' array, which nobody reads any more:
    // emit_fine's barriers lie in between) a wave's records leave as four instructions of 1 KiB of consecutive bytes each.
    {
        // (Piece q of record r lies at 4 r + ((q + r / 4) & 3): a lane's four stores are 64 bytes apart from its
        // neighbours', and without the rotation sixteen lanes would share four of the sixteen 16-byte bank groups; with
        // it both the stores, a record per lane, and the loads, a piece per lane, take each group once.)
        float4* const stage = (float4*)&s_slot[0][0] + 4u * t; // this lane's record
        const uint32_t lane = t & 63u, rot = (t >> 2) & 3u;
        const uint64_t mm = __ballot(made);
        stage[rot] = make_float4(b0[0], b0[1], b0[2], b0[3]);
        stage[(1u + rot) & 3u] = make_float4(b0[4], b0[5], b1[0], b1[1]);
        stage[(2u + rot) & 3u] = make_float4(b1[2], b1[3], b1[4], b1[5]);
        stage[(3u + rot) & 3u] = make_float4(__uint_as_float(c0), __uint_as_float(c1), __uint_as_float(leaves.x), __uint_as_float(leaves.y));
        // (a wave reads only what it wrote itself, and LDS carries a wave's operations out in order)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const float4* const mine = (const float4*)&s_slot[0][0] + 4u * (t - lane); // the wave's 64 records
        float4* const out = (float4*)nodes + 4 * (size_t)(i - lane);
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            const uint32_t piece = j * 64u + lane; // 16-byte piece of the wave's 4 KiB: record piece / 4
            const uint32_t r = piece >> 2; // (the wave's first record is a multiple of 64: r / 4 rotates as t / 4 does)
            if ((mm >> r) & 1ull) out[piece] = mine[4u * r + (((piece & 3u) + (r >> 2)) & 3u)];
        }
    }
    // The tops, in leaf order: the lane that carried a subtree to where it stopped holds it, and the lanes' own leaves
    // lie in their subtrees, so the order of the lanes is the order of the subtrees.
    const bool top = t < count && exit_gp != -2 && (exit_gp < 0 || exit_gp >= (int)kChunk - 1 || s_flag[exit_gp] != kMerged);
    const uint32_t lane = t & 63u, wave = t >> 6;
    const uint64_t m = __ballot(top), mm = __ballot(made);
    if (lane == 0) s_wcount[wave] = (uint32_t)__popcll(m), s_wmade[wave] = (uint32_t)__popcll(mm);
    __syncthreads();
    uint32_t at = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), total = 0, nmade = 0;
    for (uint32_t w = 0; w < kChunk / 64; w++) {
        if (w < wave) at += s_wcount[w];
        total += s_wcount[w], nmade += s_wmade[w];
    }
    if (t == 0) {
        top_counts[chunk] = total;
        if (total > top_cap) *overflow = 1u;
        if (nmade) atomicAdd(finished_count, nmade);
    }
    if (top && at < top_cap) {
        float4* o = tops + 3 * ((size_t)chunk * kTopCap + at);
        o[0] = make_float4(lo4.x, lo4.y, lo4.z, __uint_as_float(base + l));
        o[1] = make_float4(hi4.x, hi4.y, hi4.z, __uint_as_float(base + r));
        o[2] = make_float4(__uint_as_float((uint32_t)(dl + 1) | ((uint32_t)(dr + 1) << 8) | (__float_as_uint(lo4.w) << 16)), 0.0f, 0.0f, 0.0f);
    }
}

// The same merging one level up: a workgroup takes the tops of kLevelUnits consecutive units (chunks, or workgroups of
// the level below) as its items, in leaf order, a lane each, and hands up its own tops.  Nobody waits for anybody (the
// second child to arrive at a split goes on, the first stops).  The time of such a workgroup is the longest chain of
// merges from an item to the top, one lane busy, so a trip of the loop does the least it can: the subtree leaves what
// its parent's record will say of it (box, leaves, height) in its slot at the parent's split, swaps, reads the
// sibling's slot; whoever goes on names the parent.  A lane per split writes the records to memory afterwards, from
// the two slots.  (A loop that also recorded and linked its nodes took three times as long: some 400 instructions a
// trip, issued one wave at a time; one that waited for LDS three times a trip, twice as long as this.)
__global__ __launch_bounds__(kLevelThreads) void k_build_level(const float4* __restrict__ tops_in, const uint32_t* __restrict__ counts_in,
                                                     uint32_t nunits, uint32_t n, uint32_t top_cap, float4* __restrict__ tops_out,
                                                     uint32_t* __restrict__ counts_out, float* __restrict__ union_box, float* __restrict__ nodes,
                                                     uint32_t* __restrict__ overflow, uint32_t* __restrict__ root_done,
                                                     uint32_t* __restrict__ finished_count, float* __restrict__ fine,
                                                     uint32_t* __restrict__ fine_count, uint32_t fine_cap) {
    // per split g (between items g and g + 1) the two subtrees that meet there, as they arrive: [2 g] the one that ends
    // with item g, [2 g + 1] the one that begins with item g + 1; (box lo, first leaf) (box hi, last leaf), and the height
    __shared__ float4 s_slot[2][2 * kLevelItems]; // ([0]: first halves, [1]: second halves)
    __shared__ uint32_t s_sh[2 * kLevelItems];
    __shared__ uint32_t s_nme[kLevelItems];   // per split: the name of the node made there
    // per split: 0, or what the first child to arrive left (its far item + 1 | the prefix beyond that end + 1 << 10), or kMerged
    __shared__ uint32_t s_flag[kLevelItems];
    __shared__ uint32_t s_wcount[kLevelThreads / 64], s_made, s_nfine, s_fbase;
    __shared__ float4 s_fine[2 * kLevelFine]; // the fine cut's records made here, written out together at the end
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t u0 = blockIdx.x * kLevelUnits, nu = min(kLevelUnits, nunits - u0);
    // A unit below that had more tops than it may hand up has handed up some of them only: nothing can be made of
    // those (names and ranges would come out wrong, and records would be written where none belong).  The build is
    // then done again the other way; this launch and those after it do nothing.  (One lane reads the flag for all:
    // another workgroup of this very launch may raise it meanwhile, and the lanes must agree.)
    __shared__ uint32_t s_stop;
    static_assert(kLevelUnits <= 64u && kLevelItems <= kLevelThreads && kLevelItems <= 1023u,
                  "a wave adds up the units' counts; a lane per item; an item's index + 1 fits ten bits of the flag");
    // (every wave adds up the units' counts for itself, asked for together with the flag: one wait for memory, no LDS)
    uint32_t incl = lane < nu ? min(counts_in[u0 + lane], top_cap) : 0u;
    if (tid == 0) s_stop = *overflow, s_made = 0u, s_nfine = 0u;
    if (tid < kLevelItems) s_flag[tid] = 0u;
    __syncthreads();
    if (s_stop != 0u) return;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += v;
    }
    const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); // items in all (lanes past the units add nothing)
    if (M > kLevelItems) { // block-uniform: more tops than fit (a very unbalanced tree)
        if (tid == 0) *overflow = 1u, counts_out[blockIdx.x] = 0u;
        return;
    }
    // Lane k carries item k: the subtree it is part of, as far as this lane takes it, is in registers.
    const uint32_t k = tid;
    uint32_t il = k, ir = k, h = 0;              // items of the subtree, its height
    v4f lo4 = {0, 0, 0, 0}, hi4 = {0, 0, 0, 0};  // its box; lo4.w, hi4.w: the bits of its first and last leaf
    int dl = -1, dr = -1;                        // the prefixes beyond its two ends
    int exit_gp = -2; // the split the subtree waits at; outside [0, M - 2]: the sibling is another workgroup's; -2: no subtree, or the root
    if (k < M) {
        // item k is item k - off of unit u: u units end at or before k, the last of them at off
        uint32_t u = 0, off = 0;
#pragma unroll
        for (int j = 0; j < (int)kLevelUnits; j++) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)incl, j);
            const bool past = e <= k;
            u += past ? 1u : 0u, off = past ? e : off;
        }
        const float4* it = tops_in + 3 * ((size_t)(u0 + u) * kTopCap + (k - off));
        const float4 a = it[0], b = it[1], c = it[2];
        lo4 = (v4f){a.x, a.y, a.z, a.w}, hi4 = (v4f){b.x, b.y, b.z, b.w};
        const uint32_t w = __float_as_uint(c.x);
        h = w >> 16;
        dl = (int)(w & 255u) - 1, dr = (int)((w >> 8) & 255u) - 1;
        // (one exchange with LDS a trip: see k_build_chunks)
        for (;;) {
            if (dl < 0 && dr < 0) break; // the root
            const bool right = dr > dl;
            const int gp = right ? (int)ir : (int)il - 1; // the parent's split, in items
            exit_gp = gp;
            if (gp < 0 || gp >= (int)M - 1) break; // the sibling lies in another workgroup's units
            const uint32_t mine = 2u * (uint32_t)gp + (right ? 0u : 1u);
            v4f s0, s1;
            uint32_t sh;
            const uint32_t other = lds_put_swap_then_read<(int)(2 * kLevelItems * 16)>(&s_slot[0][mine], lo4, hi4, &s_sh[mine], h, &s_flag[gp],
                                                          ((right ? il : ir) + 1u) | ((uint32_t)((right ? dl : dr) + 1) << 10),
                                                          &s_slot[0][mine ^ 1u], &s_sh[mine ^ 1u], s0, s1, sh);
            if (other == 0u) break; // the first to arrive
            lo4.x = min_raw(lo4.x, s0.x), lo4.y = min_raw(lo4.y, s0.y), lo4.z = min_raw(lo4.z, s0.z);
            hi4.x = max_raw(hi4.x, s1.x), hi4.y = max_raw(hi4.y, s1.y), hi4.z = max_raw(hi4.z, s1.z);
            h = 1u + (h > sh ? h : sh);
            const uint32_t far = (other & 1023u) - 1u;
            const int dfar = (int)((other >> 10) & 127u) - 1;
            hi4.w = right ? s1.w : hi4.w, lo4.w = right ? lo4.w : s0.w; // (selects: no branches)
            ir = right ? far : ir, il = right ? il : far;
            dr = right ? dfar : dr, dl = right ? dl : dfar;
            // the node just made gets its name (a left child is named after its last leaf, a right child after its first)
            s_nme[gp] = __float_as_uint((dl < 0 && dr < 0) ? 0.0f : dr > dl ? hi4.w : lo4.w);
            s_flag[gp] = kMerged;
            exit_gp = -2;
        }
    }
    __syncthreads();
    // every node made here, a lane each: its record, its links, the fine cut's records of its small children.  What the
    // two children left in the split's slots is all there is to know about them.
    uint32_t made = 0;
    {
        const uint32_t g = tid;
        if (g + 1u < M && s_flag[g] == kMerged) {
            made = 1u;
            const float4 p0 = s_slot[0][2u * g], p1 = s_slot[1][2u * g], q0 = s_slot[0][2u * g + 1u], q1 = s_slot[1][2u * g + 1u];
            const float b0[6] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z}, b1[6] = {q0.x, q0.y, q0.z, q1.x, q1.y, q1.z};
            const uint32_t me = s_nme[g], l = __float_as_uint(p0.w), r = __float_as_uint(q1.w);
            const uint32_t gamma = __float_as_uint(p1.w); // last leaf of the left child
            const bool leaf0 = l == gamma, leaf1 = gamma + 1u == r;
            const uint32_t c0 = gamma | (leaf0 ? 0x80000000u : 0u), c1 = (gamma + 1u) | (leaf1 ? 0x80000000u : 0u);
            if (me + 1u < n) { // (always, with whole input: a name is a leaf with a neighbour on the named side)
                float4* nd = (float4*)nodes + 4 * (size_t)me;
                nd[0] = make_float4(b0[0], b0[1], b0[2], b0[3]);
                nd[1] = make_float4(b0[4], b0[5], b1[0], b1[1]);
                nd[2] = make_float4(b1[2], b1[3], b1[4], b1[5]);
                nd[3] = make_float4(__uint_as_float(c0), __uint_as_float(c1), __uint_as_float(l), __uint_as_float(r));
                if (l == 0u && r == n - 1u) { // the root: its box and height, for the build's summary
                    float4* ub = (float4*)union_box;
                    const uint32_t h0 = s_sh[2u * g], h1 = s_sh[2u * g + 1u];
                    ub[0] = make_float4(fminf(b0[0], b1[0]), fminf(b0[1], b1[1]), fminf(b0[2], b1[2]), fmaxf(b0[3], b1[3]));
                    ub[1] = make_float4(fmaxf(b0[4], b1[4]), fmaxf(b0[5], b1[5]), __uint_as_float(1u + (h0 > h1 ? h0 : h1)), 0.0f);
                    *root_done = 1u;
                }
                const uint32_t c0n = gamma - l + 1u, c1n = r - gamma;
                const bool big = r - l + 1u > kFineMax;
                const bool f0 = big && c0n <= kFineMax, f1 = big && c1n <= kFineMax;
                if (f0 || f1) {
                    // collected in LDS and written out together: one update of the counter in memory per workgroup
                    const uint32_t cnt = (f0 ? 1u : 0u) + (f1 ? 1u : 0u);
                    const uint32_t at = atomicAdd(&s_nfine, cnt);
                    const uint32_t nlocal = at < kLevelFine ? min(cnt, kLevelFine - at) : 0u; // of this node's records, in LDS
                    const uint32_t gat = nlocal < cnt ? atomicAdd(fine_count, cnt - nlocal) : 0u;
                    uint32_t j = 0;
                    if (f0) {
                        const float4 r0 = make_float4(b0[0], b0[1], b0[2], __uint_as_float(l));
                        const float4 r1 = make_float4(b0[3], b0[4], b0[5], __uint_as_float(c0n));
                        if (j < nlocal)
                            s_fine[2 * (at + j)] = r0, s_fine[2 * (at + j) + 1] = r1;
                        else if (gat + j - nlocal < fine_cap)
                            ((float4*)fine)[2 * (size_t)(gat + j - nlocal)] = r0, ((float4*)fine)[2 * (size_t)(gat + j - nlocal) + 1] = r1;
                        j++;
                    }
                    if (f1) {
                        const float4 r0 = make_float4(b1[0], b1[1], b1[2], __uint_as_float(gamma + 1u));
                        const float4 r1 = make_float4(b1[3], b1[4], b1[5], __uint_as_float(c1n));
                        if (j < nlocal)
                            s_fine[2 * (at + j)] = r0, s_fine[2 * (at + j) + 1] = r1;
                        else if (gat + j - nlocal < fine_cap)
                            ((float4*)fine)[2 * (size_t)(gat + j - nlocal)] = r0, ((float4*)fine)[2 * (size_t)(gat + j - nlocal) + 1] = r1;
                    }
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) made += __shfl_down(made, off);
    if (lane == 0 && made) atomicAdd(&s_made, made);
    // the tops this workgroup hands up, in leaf order: the lane that carried a subtree to where it stopped holds it, and
    // the lanes' own items lie in their subtrees, so the order of the lanes is the order of the subtrees
    uint32_t before = 0;
    {
        const bool top = k < M && exit_gp != -2 && (exit_gp < 0 || exit_gp >= (int)M - 1 || s_flag[exit_gp] != kMerged);
        const uint64_t m = __ballot(top);
        if (lane == 0) s_wcount[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t at = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < kLevelThreads / 64u; w++) {
            if (w < wave) at += s_wcount[w];
            before += s_wcount[w];
        }
        if (top && at < top_cap) {
            float4* o = tops_out + 3 * ((size_t)blockIdx.x * kTopCap + at);
            o[0] = make_float4(lo4.x, lo4.y, lo4.z, lo4.w);
            o[1] = make_float4(hi4.x, hi4.y, hi4.z, hi4.w);
            o[2] = make_float4(__uint_as_float((uint32_t)(dl + 1) | ((uint32_t)(dr + 1) << 8) | (h << 16)), 0.0f, 0.0f, 0.0f);
        }
    }
    __syncthreads(); // s_made and s_nfine are complete
    // (s_nfine also counts the records that did not fit and went to memory at once: the first kLevelFine are in LDS)
    const uint32_t nf = min(s_nfine, kLevelFine);
    if (tid == 0) {
        counts_out[blockIdx.x] = before;
        if (before > top_cap) *overflow = 1u;
        if (s_made) atomicAdd(finished_count, s_made);
        s_fbase = nf ? atomicAdd(fine_count, nf) : 0u;
    }
    __syncthreads();
    for (uint32_t q = tid; q < nf; q += kLevelThreads)
        if (s_fbase + q < fine_cap) {
            ((float4*)fine)[2 * (size_t)(s_fbase + q)] = s_fine[2 * q];
            ((float4*)fine)[2 * (size_t)(s_fbase + q) + 1] = s_fine[2 * q + 1];
        }
}

// The box and height of a child: a leaf's from its triangle, a node's from what the node left in union_box.  The
// lanes of a wave climb chains of their own and meet leaves and nodes in any mix, so both kinds of loads are issued
// for every lane (from element 0 where they do not apply) and the result is picked afterwards: one wait for memory
// per child instead of one per kind.
__device__ __forceinline__ void load_child_box(uint32_t c, const float4* __restrict__ TA, const float4* __restrict__ TB,
                                               const float4* __restrict__ TC, const float* union_box,
                                               float lo[3], float hi[3], uint32_t& height) {
    const bool leaf = (c & 0x80000000u) != 0u;
    const uint32_t k = c & 0x7FFFFFFFu, kt = leaf ? k : 0u, kn = leaf ? 0u : k;
    const float4 ta = TA[kt], tb = TB[kt], tc = TC[kt];
    const float4 a = ((const float4*)union_box)[2 * (size_t)kn], b = ((const float4*)union_box)[2 * (size_t)kn + 1];
    float tlo[3], thi[3];
    tri_box(ta, tb, tc, tlo, thi);
    lo[0] = leaf ? tlo[0] : a.x, lo[1] = leaf ? tlo[1] : a.y, lo[2] = leaf ? tlo[2] : a.z;
    hi[0] = leaf ? thi[0] : a.w, hi[1] = leaf ? thi[1] : b.x, hi[2] = leaf ? thi[2] : b.y;
    height = leaf ? 0u : __float_as_uint(b.z);
}

// Finishes one internal node whose children are both complete: the node's record, and its own box and height for the
// parent.  The caller brings the node's links (its children and the sorted leaves it covers).
__device__ __forceinline__ void finish_node(const float4* __restrict__ TA, const float4* __restrict__ TB,
                                            const float4* __restrict__ TC, uint32_t node, uint32_t c0, uint32_t c1,
                                            uint2 leaves, float* union_box, float* __restrict__ nodes, float b0[6],
                                            float b1[6]) {
    float l0[3], h0[3], l1[3], h1[3];
    uint32_t ht0, ht1;
    load_child_box(c0, TA, TB, TC, union_box, l0, h0, ht0);
    load_child_box(c1, TA, TB, TC, union_box, l1, h1, ht1);
    const uint32_t ht = 1u + (ht0 > ht1 ? ht0 : ht1);
    float4* nd = (float4*)nodes + 4 * (size_t)node;
    nd[0] = make_float4(l0[0], l0[1], l0[2], h0[0]);
    nd[1] = make_float4(h0[1], h0[2], l1[0], l1[1]);
    nd[2] = make_float4(l1[2], h1[0], h1[1], h1[2]);
    nd[3] = make_float4(__uint_as_float(c0), __uint_as_float(c1), __uint_as_float(leaves.x), __uint_as_float(leaves.y));
    float4* u = (float4*)union_box + 2 * (size_t)node;
    u[0] = make_float4(fminf(l0[0], l1[0]), fminf(l0[1], l1[1]), fminf(l0[2], l1[2]), fmaxf(h0[0], h1[0]));
    u[1] = make_float4(fmaxf(h0[1], h1[1]), fmaxf(h0[2], h1[2]), __uint_as_float(ht), 0.0f);
#pragma unroll
    for (int a = 0; a < 3; a++) b0[a] = l0[a], b0[3 + a] = h0[a], b1[a] = l1[a], b1[3 + a] = h1[a];
}

// The nodes that cross chunk boundaries.  A launch per tree level is mostly launch once a level holds a few hundred
// nodes, and the top of an unbalanced tree is thirty-odd such levels.  Workgroups climb instead: a lane takes a node
// that is ready (both children complete), finishes it, counts its arrival at the parent and, if it is the second child
// to arrive there, goes on with the parent; the first to arrive stops.  The first child's box reaches the second
// through memory, and that is safe only inside a workgroup: its lanes sit on one CU behind one L1 and one L2, so
// release and acquire at workgroup scope (the stores have completed before the arrival is counted, the loads are
// issued after it) are all the ordering it takes, while another workgroup may run behind another XCD's L2.  So the
// leaves are cut into spans, a workgroup per span: a parent whose leaves all lie in the span gets both arrivals from
// this workgroup and is climbed into; one that reaches beyond it is handed to the next launch, whose single
// workgroup has everything for its span.  The parent's links are fetched while the arrival is in flight.
//
// Nodes wait in slots: (node + 1, child 0, child 1, parent) (first leaf, last leaf, -, -), node + 1 == 0: empty.  The
// nodes ready after the chunk kernel cover disjoint stretches of leaves, each with its own index inside and a chunk
// boundary as well: at most two with their index in one chunk, one across the chunk's lower end and one across its
// upper end.  Those are the chunk's two slots, and no counter is needed to place them.  The same holds for the nodes
// handed on, with spans for chunks.  A node is climbed into at most once (only the arrival that finds the counter at 1
// goes on), whatever the links hold.
constexpr int kClimbThreads = 1024;
constexpr uint32_t kSpanChunks = 256; // chunks of leaves per workgroup of the first climb

__global__ __launch_bounds__(kClimbThreads) void k_refit_climb(const float4* __restrict__ TA, const float4* __restrict__ TB,
                                                               const float4* __restrict__ TC, const uint32_t* __restrict__ child,
                                                               const uint32_t* __restrict__ parent,
                                                               const uint2* __restrict__ range, const uint4* __restrict__ slots,
                                                               uint32_t nslots, uint32_t slots_per_group, uint32_t span,
                                                               uint4* __restrict__ handed, uint32_t* arrive, float* union_box,
                                                               float* __restrict__ nodes, uint32_t* __restrict__ finished_count,
                                                               uint32_t* __restrict__ root_done,
                                                               float* __restrict__ fine, uint32_t* __restrict__ fine_count,
                                                               uint32_t fine_cap) {
    // span == 0: one workgroup, every leaf is its own
    const uint32_t first = blockIdx.x * slots_per_group, last = min(first + slots_per_group, nslots);
    const uint32_t span_lo = blockIdx.x * span, span_hi = span ? span_lo + (span - 1u) : 0xFFFFFFFFu;
    uint32_t finished = 0;
    for (uint32_t q = first + threadIdx.x; q < last; q += (uint32_t)kClimbThreads) {
        const uint4 e0 = slots[2 * (size_t)q], e1 = slots[2 * (size_t)q + 1];
        if (e0.x == 0u) continue;
        uint32_t node = e0.x - 1u, up = e0.w;
        uint2 c = make_uint2(e0.y, e0.z), leaves = make_uint2(e1.x, e1.y);
        for (;;) {
            float b0[6], b1[6];
            finish_node(TA, TB, TC, node, c.x, c.y, leaves, union_box, nodes, b0, b1);
            finished++;
            // The node's records for the fine cut.  The counter is one address for the whole GPU and its updates take
            // their turns: the lanes of the wave that are still climbing (the others have left the loop) update it
            // once between them.
            {
                const uint32_t gamma = c.x & 0x7FFFFFFFu;
                const uint32_t n0 = gamma - leaves.x + 1u, n1 = leaves.y - gamma;
                const bool big = leaves.y - leaves.x + 1u > kFineMax;
                const bool f0 = big && n0 <= kFineMax, f1 = big && n1 <= kFineMax;
                const uint64_t m0 = __ballot(f0), m1 = __ballot(f1);
                if ((m0 | m1) != 0ull) { // (the same for all of them)
                    const uint32_t lane = threadIdx.x & 63u;
                    const int leader = __ffsll((long long)__ballot(true)) - 1;
                    uint32_t at = 0;
                    if ((int)lane == leader) at = atomicAdd(fine_count, (uint32_t)(__popcll(m0) + __popcll(m1)));
                    at = __shfl(at, leader);
                    const uint64_t below = (1ull << lane) - 1ull;
                    const uint32_t at0 = at + (uint32_t)__popcll(m0 & below);
                    const uint32_t at1 = at + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & below);
                    if (f0 && at0 < fine_cap) {
                        float4* r = (float4*)fine + 2 * (size_t)at0;
                        r[0] = make_float4(b0[0], b0[1], b0[2], __uint_as_float(leaves.x));
                        r[1] = make_float4(b0[3], b0[4], b0[5], __uint_as_float(n0));
                    }
                    if (f1 && at1 < fine_cap) {
                        float4* r = (float4*)fine + 2 * (size_t)at1;
                        r[0] = make_float4(b1[0], b1[1], b1[2], __uint_as_float(gamma + 1u));
                        r[1] = make_float4(b1[3], b1[4], b1[5], __uint_as_float(n1));
                    }
                }
            }
            if (up == 0xFFFFFFFFu) { // the root
                *root_done = 1u;
                break;
            }
            node = up;
            // (these three arrays were written by an earlier launch: nothing here orders them)
            c = ((const uint2*)child)[node];
            leaves = range[node];
            up = parent[node];
            // The counter is shared with other workgroups when the parent reaches beyond the span (device scope, and
            // nothing is read on the strength of it before the next launch); inside the span both arrivals are this
            // workgroup's, and the fences on either side of it order the box behind the first and before the second.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const uint32_t before = __hip_atomic_fetch_add(&arrive[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (before != 1u) break;
            if (leaves.x < span_lo || leaves.y > span_hi) { // (never with span == 0)
                const uint32_t at = boundary_slot(node, leaves, span);
                handed[2 * (size_t)at] = make_uint4(node + 1u, c.x, c.y, up);
                handed[2 * (size_t)at + 1] = make_uint4(leaves.x, leaves.y, 0u, 0u);
                break;
            }
        }
    }
    // the count the host checks the tree by: one update per wave
    for (int off = 32; off > 0; off >>= 1) finished += __shfl_down(finished, off);
    if ((threadIdx.x & 63u) == 0u && finished) atomicAdd(finished_count, finished);
}


// ------------------------------------------------------------------------------------------
// 6. cut: a flat start set for the tile kernel.  The tree is opened breadth-first from the root,
//    level by level, until every open subtree has at most kRangeMax leaves or kCutCap records are in
//    use.  A record is 8 floats: (lo.xyz, ref) (hi.xyz, count).  count <= kRangeMax: ref is the first
//    sorted leaf of a range taken whole; otherwise ref is an internal node that still has to be
//    descended.  One workgroup, a lane per open node (its two sides are two pending records, one after the
//    other), the open nodes' own records in LDS; an order-preserving block scan places the records and the
//    next level's nodes, so the result does not depend on scheduling.
//    The time of this kernel is the number of levels times what a level waits for, so a level waits for as little
//    as it can: a lane asks for the records of its node's children as soon as it knows them, before the scan and
//    its barriers (which wait for LDS only), and whoever is opened puts the record that has arrived into the
//    next level's place.  Nothing is stored to memory on the way (loads and stores are counted together, in order:
//    a level's wait for its loads would wait for the stores of the level before, some 2 us each); the cut's
//    records collect in LDS and leave together at the end.
// ------------------------------------------------------------------------------------------
constexpr int kCutThreads = kCutCap / 2;

// A barrier that waits for the wave's LDS operations only: loads from memory stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The host queues this kernel behind the refit without knowing whether it reached the root: while `root_done` is
// zero the node records are not complete and nothing is read.  summary: [0] number of cut records (kCutUnfinished: refit not finished), [1..6] scene bounds,
// [7] height of the root -- everything the host wants to know at the end of a build, in one place.
constexpr uint32_t kCutUnfinished = 0xFFFFFFFFu;
constexpr uint32_t kSummaryWords = 8;

__global__ __launch_bounds__(kCutThreads) void k_build_cut(const float* __restrict__ nodes, float* __restrict__ cut,
                                                          float* __restrict__ groups,
                                                          const uint32_t* __restrict__ root_done,
                                                          const uint32_t* __restrict__ unsorted,
                                                          const uint32_t* __restrict__ overflow,
                                                          const float* __restrict__ bbox,
                                                          const float* __restrict__ union_box, uint32_t n,
                                                          uint32_t* __restrict__ summary) {
    __shared__ float4 s_rec[4][kCutThreads]; // the records of the open nodes, in breadth-first order, a quarter per row
    __shared__ float4 s_cut[2][kCutCap];     // the cut's records (a half per row): written to memory together at the end
    __shared__ uint32_t s_tot[kCutThreads / 64];
    __shared__ uint32_t s_bad; // a record names a node that does not exist: the tree is not what it should be
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const float4* const nodes4 = (const float4*)nodes;
    // block-uniform (unsorted: the sort gave up; overflow: the bottom-up build gave up: the tree is not to be read)
    if (*root_done == 0u || *unsorted != 0u || *overflow != 0u) {
        if (tid == 0) summary[0] = kCutUnfinished;
        return;
    }
    if (tid < 6) summary[1 + tid] = __float_as_uint(bbox[tid]);
    if (tid == 6) summary[7] = __float_as_uint(union_box[6]);
    if (tid == 0) s_bad = 0u;
    if (tid < 4) s_rec[tid][0] = nodes4[tid]; // the root
    uint32_t nn = 1, nout = 0; // open nodes, records written (block-uniform)
    __syncthreads();
    while (nn != 0) {
        // opening a subtree replaces one pending record by two: at most `budget` more may be opened
        const uint32_t budget = kCutCap - (nout + 2u * nn);
        const bool have = tid < nn;
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
        uint32_t c0 = 0, c1 = 0, first0 = 0, first1 = 0, cnt0 = 0, cnt1 = 0;
        bool big0 = false, big1 = false; // the sides that are opened if the budget allows
        float4 a0 = q0, a1 = q0, a2 = q0, a3 = q0, b0 = q0, b1 = q0, b2 = q0, b3 = q0; // the children's records, on their way
        if (have) {
            q0 = s_rec[0][tid], q1 = s_rec[1][tid], q2 = s_rec[2][tid];
            const float4 q3 = s_rec[3][tid];
            c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            const uint32_t lo = __float_as_uint(q3.z), hi = __float_as_uint(q3.w), gamma = c0 & 0x7FFFFFFFu;
            first0 = lo, cnt0 = gamma - lo + 1u, first1 = gamma + 1u, cnt1 = hi - gamma;
            big0 = !(c0 & 0x80000000u) && cnt0 > kRangeMax, big1 = !(c1 & 0x80000000u) && cnt1 > kRangeMax;
            if (big0 && c0 + 1u >= n) big0 = false, s_bad = 1u; // (never followed)
            if (big1 && c1 + 1u >= n) big1 = false, s_bad = 1u;
            // (with no budget left -- the last level -- nothing will be opened: no loads)
            if (big0 && budget) a0 = nodes4[4 * (size_t)c0], a1 = nodes4[4 * (size_t)c0 + 1], a2 = nodes4[4 * (size_t)c0 + 2], a3 = nodes4[4 * (size_t)c0 + 3];
            if (big1 && budget) b0 = nodes4[4 * (size_t)c1], b1 = nodes4[4 * (size_t)c1 + 1], b2 = nodes4[4 * (size_t)c1 + 2], b3 = nodes4[4 * (size_t)c1 + 3];
        }
        // the big sides before each side, in record order (node p's sides are records 2 p and 2 p + 1)
        const uint64_t m0 = __ballot(big0), m1 = __ballot(big1), below = (1ull << lane) - 1ull;
        if (lane == 0) s_tot[wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
        lds_barrier(); // (every lane has read its node's record)
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < kCutThreads / 64; w++) {
            const uint32_t c = s_tot[w];
            if (w < wave) before += c;
            all += c;
        }
        const uint32_t o0 = before + (uint32_t)(__popcll(m0 & below) + __popcll(m1 & below)), o1 = o0 + (big0 ? 1u : 0u);
        const bool open0 = big0 && o0 < budget, open1 = big1 && o1 < budget;
        const uint32_t opened = min(all, budget);
        // (The wait is written out: the compiler waits for a load where its value is used, which is under a condition
        // here, and then has to wait again, wherever the same registers are written next, for the loads nobody used.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (open0) s_rec[0][o0] = a0, s_rec[1][o0] = a1, s_rec[2][o0] = a2, s_rec[3][o0] = a3;
        if (open1) s_rec[0][o1] = b0, s_rec[1][o1] = b1, s_rec[2][o1] = b2, s_rec[3][o1] = b3;
        if (have) {
            // a record that is not opened is written out: its place is its number less the opened ones before it
            if (!open0) {
                const uint32_t at = nout + 2u * tid - min(o0, budget);
                s_cut[0][at] = make_float4(q0.x, q0.y, q0.z, __uint_as_float(big0 ? c0 : first0));
                s_cut[1][at] = make_float4(q0.w, q1.x, q1.y, __uint_as_float(cnt0));
            }
            if (!open1) {
                const uint32_t at = nout + 2u * tid + 1u - min(o1, budget);
                s_cut[0][at] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(big1 ? c1 : first1));
                s_cut[1][at] = make_float4(q2.y, q2.z, q2.w, __uint_as_float(cnt1));
            }
        }
        nout += 2u * nn - opened;
        nn = opened;
        lds_barrier(); // the next level's records are in place, s_tot has been read
    }
    __syncthreads();
    if (tid == 0) summary[0] = s_bad ? kCutUnfinished : nout;
    // The records go to memory, and with them the boxes of kCutGroup consecutive records (breadth-first order keeps
    // neighbours together), 8 floats per group: a lane per record, then the group's lanes put their boxes together.
    static_assert(kCutGroup == 16, "sixteen consecutive lanes make a group");
    for (uint32_t i0 = 0; i0 < nout; i0 += kCutThreads) { // block-uniform
        const uint32_t i = i0 + tid;
        float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
        if (i < nout) {
            lo = s_cut[0][i], hi = s_cut[1][i];
            ((float4*)cut)[2 * (size_t)i] = lo, ((float4*)cut)[2 * (size_t)i + 1] = hi;
        }
#pragma unroll
        for (int o = 1; o < (int)kCutGroup; o <<= 1) {
            lo.x = fminf(lo.x, __shfl_xor(lo.x, o)), lo.y = fminf(lo.y, __shfl_xor(lo.y, o)), lo.z = fminf(lo.z, __shfl_xor(lo.z, o));
            hi.x = fmaxf(hi.x, __shfl_xor(hi.x, o)), hi.y = fmaxf(hi.y, __shfl_xor(hi.y, o)), hi.z = fmaxf(hi.z, __shfl_xor(hi.z, o));
        }
        if (i < nout && (tid & (kCutGroup - 1u)) == 0u) {
            float4* r = (float4*)groups + 2 * (size_t)(i / kCutGroup);
            r[0] = make_float4(lo.x, lo.y, lo.z, 0.0f), r[1] = make_float4(hi.x, hi.y, hi.z, 0.0f);
        }
    }
}

#define HIP_RET(expr)                     \
    do {                                  \
        hipError_t e_ = (expr);           \
        if (e_ != hipSuccess) return e_;  \
    } while (0)

} // namespace

// Records the fine cut may hold.  A tree over well spread keys has about 1.6 n / kFineMax subtrees of at most kFineMax
// leaves under larger parents; badly unbalanced trees have up to n.  A build that needs more than this leaves the
// scene without a fine cut (its draws then start from the coarse cut, as scenes of a few triangles do).
static uint32_t fine_cap_for(uint32_t n) { return (uint32_t)std::min<uint64_t>(n, (uint64_t)n * 5u / (2u * kFineMax) + 1024u); }
static uint32_t sort_tiles_for(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }
static uint32_t refit_chunks_for(uint32_t n) { return (n + kChunk - 1) / kChunk; }
static uint32_t refit_spans_for(uint32_t n) { return (refit_chunks_for(n) + kSpanChunks - 1) / kSpanChunks; }
// The build's counters, each in a stretch of memory of its own (updates of one address take their turns, and so do
// updates of neighbouring addresses: ten thousand workgroups adding to two counters in one 64-byte line took longer
// than the rest of the chunk kernel), then the summary the cut kernel writes; one copy fetches them all.
constexpr uint32_t kWordStride = 64;
enum : uint32_t { kWFinishedChunks = 0, kWFinishedClimb = kWordStride, kWRootDone = 2 * kWordStride, kWFineCount = 3 * kWordStride,
                  kWUnsorted = 4 * kWordStride, kWSummary = 5 * kWordStride, kWOverflow = 6 * kWordStride, kWords = 7 * kWordStride };
static_assert(kSummaryWords <= 16, "the summary has sixteen words");
static uint32_t level_units_for(uint32_t units) { return (units + kLevelUnits - 1) / kLevelUnits; }
static size_t level_tops_bytes(uint32_t units) { return (size_t)units * kTopCap * 48; }
static size_t build_words_for(uint32_t n) { return kWords + (size_t)refit_spans_for(n) * 2 * 8; }

size_t lbvh_keep_bytes(uint32_t n) {
    if (n < 2) return 0;
    const size_t sizes[] = {(size_t)n * 8, (size_t)n * 4,
                            (size_t)(n - 1) * 64, (size_t)kCutCap * 32, (size_t)(kCutCap / kCutGroup) * 32,
                            (size_t)fine_cap_for(n) * 32, (size_t)n * 4};
    size_t total = 0;
    for (size_t b : sizes) total += Arena::aligned(b);
    return total;
}

size_t lbvh_scratch_bytes(uint32_t n) {
    if (n < 2) return 0;
    const size_t ntiles = sort_tiles_for(n);
    const size_t sizes[] = {(size_t)bounds_blocks_for(n) * 24, 32, (size_t)n * 8, (size_t)n * 4, ntiles * 256 * 4,
                            (size_t)(n - 1) * 4, (ntiles * 256 / kScanBlock + 2) * 4, (size_t)refit_chunks_for(n) * 64,
                            build_words_for(n) * 4, (size_t)(n - 1) * 32, (size_t)(n - 1) * 8,
                            (size_t)(n - 1) * 8, (size_t)(n - 1) * 4, (size_t)n * 4,
                            level_tops_bytes(refit_chunks_for(n)), level_tops_bytes(level_units_for(refit_chunks_for(n))),
                            (size_t)refit_chunks_for(n) * 4, (size_t)level_units_for(refit_chunks_for(n)) * 4};
    size_t total = 0;
    for (size_t b : sizes) total += Arena::aligned(b);
    return total;
}

static hipError_t build_lbvh_with(hipStream_t s, DeviceScene& sc, Arena& keep, Arena& tmp, const float* records,
                                  const float* bounds, const MeshSource& mesh, int sort_shift, bool search_all, bool* overflowed);

float* lbvh_take_bounds(Arena& tmp, uint32_t n) { return n < 2 ? nullptr : tmp.take<float>((size_t)bounds_blocks_for(n) * 24); }

hipError_t build_lbvh(hipStream_t s, DeviceScene& sc, Arena& keep, Arena& tmp, const float* records, const float* bounds,
                      const MeshSource& mesh, int* sort_hint) {
    // Four passes and the runs, five passes and the runs, or all eight: the first that meets no run it cannot order
    // (a build that gives up has cost as much as one that does not, so the caller's next build starts where this one
    // ended).
    static const int shifts[] = {32, 24, 0};
    int at = sort_hint ? *sort_hint : 0;
    at = at < 0 ? 0 : at > 2 ? 2 : at;
    const size_t keep_used = keep.used, tmp_used = tmp.used;
    // INV_BUILD_SEARCH=1 in the environment: every node finds its range by searching (k_karras), the chunks only refit
    // and the nodes across chunk boundaries are climbed through memory, as before the chunks and the levels above them
    // built the hierarchy bottom-up (for measurements and as a cross-check).  A tree so unbalanced that a unit has more
    // tops than it can hand up is built that way too.
    static const bool search_env = getenv("INV_BUILD_SEARCH") && atoi(getenv("INV_BUILD_SEARCH"));
    bool search_all = search_env;
    hipError_t e = hipErrorNotReady;
    for (; at < 3 && e == hipErrorNotReady; at++) {
        keep.used = keep_used, tmp.used = tmp_used;
        bool overflowed = false;
        e = build_lbvh_with(s, sc, keep, tmp, records, bounds, mesh, shifts[at], search_all, &overflowed);
        if (e == hipErrorNotReady && overflowed) search_all = true, at--; // the same sort again, the other hierarchy
    }
    if (sort_hint && e == hipSuccess) *sort_hint = at - 1;
    return e;
}

static hipError_t build_lbvh_with(hipStream_t s, DeviceScene& sc, Arena& keep, Arena& tmp, const float* records,
                                  const float* bounds, const MeshSource& mesh, int sort_shift, bool search_all, bool* overflowed) {
    const uint32_t n = sc.ntri;
    sc.has_bvh = false;
    if (n < 2) return hipSuccess; // nothing to build: such scenes render through the brute-force kernel
    const uint32_t bounds_blocks = bounds_blocks_for(n);
    const uint32_t ntiles = sort_tiles_for(n);
    float* d_bbox = tmp.take<float>(32);
    uint64_t* keys_alt = tmp.take<uint64_t>((size_t)n * 8);
    uint32_t* vals_alt = tmp.take<uint32_t>((size_t)n * 4);
    uint32_t* counts = tmp.take<uint32_t>((size_t)ntiles * 256 * 4);
    uint32_t* arrive = tmp.take<uint32_t>((size_t)(n - 1) * 4);
    uint32_t* scan_totals = tmp.take<uint32_t>(((size_t)ntiles * 256 / kScanBlock + 2) * 4);
    const uint32_t nchunks = refit_chunks_for(n), nspans = refit_spans_for(n);
    uint4* slots = tmp.take<uint4>((size_t)nchunks * 64); // two per chunk (see k_refit_climb)
    // the counters and the summary, and behind them the slots of the nodes handed from the first climb to the second,
    // two per span; all zero to begin with
    uint32_t* words = tmp.take<uint32_t>(build_words_for(n) * 4);
    uint4* const handed = (uint4*)(words + kWords);
    float* union_box = tmp.take<float>((size_t)(n - 1) * 32);
    uint2* range = tmp.take<uint2>((size_t)(n - 1) * 8);
    // the tops the chunks hand up, and those of the levels above them in turn (two sets, used alternately)
    const uint32_t nunits2 = level_units_for(nchunks);
    float4* tops[2] = {tmp.take<float4>(level_tops_bytes(nchunks)), tmp.take<float4>(level_tops_bytes(nunits2))};
    uint32_t* top_counts[2] = {tmp.take<uint32_t>((size_t)nchunks * 4), tmp.take<uint32_t>((size_t)nunits2 * 4)};
    // what the scene keeps
    sc.morton = keep.take<uint64_t>((size_t)n * 8);
    sc.order = keep.take<uint32_t>((size_t)n * 4);
    // (links of the build that searches; the scene keeps the node records only: inv_scene_read_lbvh makes its child and
    // parent arrays from those)
    sc.child = sc.parent = sc.leaf_parent = nullptr;
    uint32_t* const b_child = tmp.take<uint32_t>((size_t)(n - 1) * 8);
    uint32_t* const b_parent = tmp.take<uint32_t>((size_t)(n - 1) * 4);
    uint32_t* const b_leaf_parent = tmp.take<uint32_t>((size_t)n * 4);
    sc.nodes = keep.take<float>((size_t)(n - 1) * 64);
    sc.cut = keep.take<float>((size_t)kCutCap * 32);
    sc.cut_groups = keep.take<float>((size_t)(kCutCap / kCutGroup) * 32);
    const uint32_t fine_cap = fine_cap_for(n);
    sc.fine = keep.take<float>((size_t)fine_cap * 32);
    sc.nfine = 0;
    sc.slot_of = keep.take<uint32_t>((size_t)n * 4);
    if (!range || !top_counts[1] || !b_leaf_parent || !sc.cut_groups || !sc.fine || !sc.slot_of) return hipErrorOutOfMemory; // the arenas were sized for something else
    float4 *A1 = (float4*)sc.triA, *B1 = (float4*)sc.triB, *C1 = (float4*)sc.triC;

    const float4* T0 = (const float4*)records;
    const dim3 blk(256), grid_n((n + 255) / 256);
    uint32_t* const too_long = words + kWUnsorted;
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(1024), 0, s, bounds, bounds_blocks, d_bbox, words, (uint32_t)build_words_for(n));
    // 63 significant key bits -> 8 passes of 8 bits, or the upper digits only and k_sort_runs for the rest.  The
    // passes and the run kernel go from one pair of arrays to the other; the keys start in the pair that makes them
    // end in sc.morton / sc.order.
    const int steps = (8 - sort_shift / 8) + (sort_shift ? 1 : 0);
    uint64_t* kin = (steps & 1) ? keys_alt : sc.morton;
    uint32_t* vin = (steps & 1) ? vals_alt : sc.order;
    uint64_t* kout = (steps & 1) ? sc.morton : keys_alt;
    uint32_t* vout = (steps & 1) ? sc.order : vals_alt;
    static_assert(kSortThreads == 256, "k_morton counts the digits of one sort tile per workgroup");
    hipLaunchKernelGGL(k_morton, dim3(xcd_grid(ntiles)), dim3(kMortonThreads), 0, s, T0, mesh, n, d_bbox, kin, vin, sort_shift, ntiles, counts);
    for (int pass = sort_shift / 8; pass < 8; pass++) {
        const int shift = 8 * pass;
        if (pass != sort_shift / 8) // (the first pass's digits were counted with the keys)
            hipLaunchKernelGGL(k_sort_hist, dim3(xcd_grid(ntiles)), dim3(kSortThreads), 0, s, kin, n, shift, ntiles, counts);
        const uint32_t ncount = ntiles * 256u;
        uint32_t raw_blocks = 0;
        if (ncount <= 16384u) {
            hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, s, counts, ncount);
        } else {
            const uint32_t nb = (ncount + kScanBlock - 1) / kScanBlock;
            hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(1024), 0, s, counts, ncount, scan_totals);
            static const bool scan_kernel = getenv("INV_SORT_SCAN_KERNEL") && atoi(getenv("INV_SORT_SCAN_KERNEL"));
            if (nb <= (uint32_t)kSortThreads && !scan_kernel)
                raw_blocks = nb; // the scatter adds them up itself
            else
                hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, s, scan_totals, nb);
        }
        hipLaunchKernelGGL(k_sort_scatter, dim3(xcd_grid(ntiles)), dim3(kSortThreads), 0, s, kin, vin, n, shift, ntiles, counts,
                           (const uint32_t*)(ncount <= 16384u ? nullptr : scan_totals), raw_blocks, kout, vout);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    if (sort_shift)
        hipLaunchKernelGGL(k_sort_runs, grid_n, blk, 0, s, (const uint64_t*)kin, (const uint32_t*)vin, n, sort_shift, kout, vout,
                           too_long);
    if (search_all)
        hipLaunchKernelGGL(k_karras, dim3(xcd_grid((n - 1 + 255) / 256)), blk, 0, s, sc.morton, n, b_child, b_parent, b_leaf_parent,
                           range);
    uint32_t* fine_count = words + kWFineCount;
    // (the inverse order is a scatter of four bytes per triangle all over memory, and only the resolve pass of frames
    // drawn one lane per triangle reads it -- unless it assembles its winners from the cached mesh)
    if (mesh.xyzw && mesh_draw_enabled()) sc.slot_of = nullptr;
    if (search_all) {
        hipLaunchKernelGGL(k_refit_chunks, dim3(nchunks), dim3(kChunk), 0, s, T0, mesh, (const uint32_t*)sc.order, sc.slot_of, A1, B1,
                           C1, b_child, b_parent, range, n, union_box, sc.nodes, arrive, slots, words + kWRootDone,
                           words + kWFinishedChunks, sc.fine, fine_count, fine_cap);
    } else {
        // (INV_BUILD_TOP_CAP in the environment: fewer tops per unit than the arrays hold, so that ordinary scenes take the
        // way of trees too unbalanced for them; for tests)
        static const uint32_t top_cap = getenv("INV_BUILD_TOP_CAP") ? std::min<uint32_t>(kTopCap, (uint32_t)atoi(getenv("INV_BUILD_TOP_CAP"))) : kTopCap;
        hipLaunchKernelGGL(k_build_chunks, dim3(nchunks), dim3(kChunk), 0, s, T0, mesh, (const uint32_t*)sc.order,
                           (const uint64_t*)sc.morton, sc.slot_of, A1, B1, C1, n, union_box, sc.nodes, tops[0], top_counts[0], top_cap, words + kWOverflow, words + kWRootDone,
                           words + kWFinishedChunks, sc.fine, fine_count, fine_cap);
        int in = 0;
        for (uint32_t units = nchunks; units > 1u; in ^= 1) {
            const uint32_t wgs = level_units_for(units);
            hipLaunchKernelGGL(k_build_level, dim3(wgs), dim3(kLevelThreads), 0, s, (const float4*)tops[in], (const uint32_t*)top_counts[in],
                               units, n, top_cap, tops[in ^ 1], top_counts[in ^ 1], union_box, sc.nodes,
                               words + kWOverflow, words + kWRootDone, words + kWFinishedClimb, sc.fine, fine_count, fine_cap);
            units = wgs;
        }
    }
    if (search_all) {
        // the nodes that cross chunk boundaries: a workgroup per span of chunks as far as the spans go, one for the rest
        if (nspans > 1)
            hipLaunchKernelGGL(k_refit_climb, dim3(nspans), dim3(kClimbThreads), 0, s, (const float4*)A1, (const float4*)B1,
                               (const float4*)C1, b_child, b_parent, range, (const uint4*)slots, 2u * nchunks, 2u * kSpanChunks,
                               kSpanChunks * kChunk, handed, arrive, union_box, sc.nodes, words + kWFinishedClimb, words + kWRootDone,
                               sc.fine, fine_count, fine_cap);
        if (nchunks > 1)
            hipLaunchKernelGGL(k_refit_climb, dim3(1), dim3(kClimbThreads), 0, s, (const float4*)A1, (const float4*)B1,
                               (const float4*)C1, b_child, b_parent, range, (const uint4*)(nspans > 1 ? handed : slots),
                               2u * (nspans > 1 ? nspans : nchunks), 2u * (nspans > 1 ? nspans : nchunks), 0u, (uint4*)nullptr, arrive,
                               union_box, sc.nodes, words + kWFinishedClimb, words + kWRootDone, sc.fine, fine_count, fine_cap);
    }
    hipLaunchKernelGGL(k_build_cut, dim3(1), dim3(kCutThreads), 0, s, (const float*)sc.nodes, sc.cut, sc.cut_groups,
                       (const uint32_t*)(words + kWRootDone), (const uint32_t*)too_long, (const uint32_t*)(words + kWOverflow),
                       (const float*)d_bbox,
                       (const float*)union_box, n, words + kWSummary);
    HIP_RET(hipGetLastError());
    uint32_t host_words[kWords] = {0};
    const uint32_t* summary = host_words + kWSummary;
    HIP_RET(hipMemcpyAsync(host_words, words, sizeof host_words, hipMemcpyDeviceToHost, s));
    HIP_RET(hipStreamSynchronize(s)); // the caller frees the scratch arena after this returns
    if (sort_shift && host_words[kWUnsorted] != 0u) return hipErrorNotReady; // the caller sorts again with more passes
    if (host_words[kWOverflow] != 0u) { // a unit with more tops than it can hand up: the caller builds the other way
        *overflowed = true;
        return hipErrorNotReady;
    }
    const uint64_t finished = (uint64_t)host_words[kWFinishedChunks] + host_words[kWFinishedClimb];
    if (finished != n - 1 || host_words[kWRootDone] == 0u || summary[0] == kCutUnfinished) return hipErrorUnknown; // every internal node must have been refitted
    sc.ncut = summary[0];
    const uint32_t nfine = host_words[kWFineCount];
    sc.nfine = nfine <= fine_cap ? nfine : 0u; // more than the array holds: no fine cut
    std::memcpy(sc.bbox, summary + 1, 24);
    sc.height = summary[7];
    sc.has_bvh = true;
    return hipSuccess;
}

} // namespace inv
