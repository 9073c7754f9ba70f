/*
 * invert.h -- C ABI of the MI355X (gfx950) per-pixel hot path.
 *
 * The reference (ins2cvt/renderer) has no plugin or FFI seam; its hot path is the
 * argument set of one indexed draw (src/invert.cpp:815-828) plus the attachments
 * that draw writes (src/invert.cpp:743-750, 795-802).  Each entry point below
 * names the reference code it stands in for.  All functions return 0 (INV_OK) or
 * an inv_status code; inv_last_error() gives the message for the calling thread.
 * No function aborts, throws across the boundary or frees caller memory.
 *
 * Threading: a context, and every scene built on it, is used by one thread at a
 * time (the reference issues all GPU work from one renderer thread,
 * src/invert.cpp:1865-1874).  One context per GPU / rank.
 */
#ifndef INVERT_H
#define INVERT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum inv_status {
    INV_OK = 0,
    INV_ERR_ARG = 1,         /* null pointer, zero extent, index out of range ... */
    INV_ERR_IO = 2,          /* file cannot be opened or read */
    INV_ERR_FORMAT = 3,      /* OBJ text is not in the fixed format the reference loader reads */
    INV_ERR_NOMEM = 4,
    INV_ERR_HIP = 5,         /* a HIP runtime call failed; message carries hipGetErrorString */
    INV_ERR_UNSUPPORTED = 6, /* e.g. a projection matrix that is not a GLM-style perspective */
    INV_ERR_NO_DEVICE = 7,   /* no usable gfx950 device: there is no CPU fallback */
} inv_status;

typedef struct inv_ctx inv_ctx;     /* device + stream; replaces Renderer's VkDevice/VkQueue (src/invert.cpp:1483-1697) */
typedef struct inv_mesh inv_mesh;   /* host mesh; replaces struct Obj (src/invert.cpp:25-39) */
typedef struct inv_scene inv_scene; /* HBM-resident triangles + LBVH; replaces vertexBuffer/indexBuffer (src/invert.cpp:1113-1164) */

/* UniformBufferObject (src/invert.cpp:266-271) + the vec3 push constant (src/invert.cpp:633, 817).
 * Matrices are column-major exactly as GLM lays them out for the memcpy at src/invert.cpp:641. */
typedef struct inv_camera {
    float model[16];
    float view[16];
    float proj[16];
    float cam_angle[3];
} inv_camera;

enum {
    INV_LAMBERT = 1u,     /* colour *= max(0, n_geom . -cam_angle): the use of the push constant that src/shader.slang:26 leaves as a TODO */
    INV_INVERT = 2u,      /* post-pass c' = 255 - c on R,G,B of every pixel after the sRGB encode */
    INV_RGBA = 4u,        /* colour bytes R,G,B,A; default is B,G,R,A = VK_FORMAT_B8G8R8A8_SRGB memory order (src/invert.cpp:519) */
    INV_BRUTE_FORCE = 8u, /* test every triangle for every ray instead of traversing the LBVH */
    INV_LANE_TRAVERSAL = 16u, /* one lane per ray with a stack in LDS instead of the tile kernel's candidate lists */
    INV_SPARSE = 32u,     /* with nranks > 1: deal out only the tiles that can contain hits (see inv_sparse_layout) */
    INV_ROOT_IDLE = 128u, /* with nranks >= 3: rank 0 takes no tiles (it alone writes the background of a sparse composite and receives everyone's tiles); tile id belongs to rank 1 + id % (nranks - 1).  Every rank of a frame passes the same flag */
    INV_TILE_TRAVERSAL = 256u, /* draw through the LBVH with the tile kernel even where triangles are so much smaller than a pixel that the frame would be drawn one lane per triangle (which needs 8 bytes per pixel of visibility buffer and never touches the tree): same pixels, for callers that want the tree walked and for measuring one path against the other */
    INV_BLOCKS = 64u,     /* trace 16x16 pixel blocks with a wave per quadrant, as inv_render_batch does, instead of 8x8 tiles whose waves share the candidates: less arithmetic, longer for a draw on its own, and slow where a block meets hundreds of triangles (its pixels then go through the per-lane traversal) */
};

/* What the reference fixes as literals or takes from the window: extent (src/invert.cpp:779-791),
 * one sample per pixel (src/invert.cpp:1375-1378).  spp > 1, ao_rays and the tile/rank fields extend it. */
typedef struct inv_params {
    uint32_t width, height;
    uint32_t spp;     /* samples per pixel; sample 0 is the pixel centre; 0 is read as 1 */
    uint32_t ao_rays; /* ambient-occlusion rays per hit pixel; 0 = off */
    uint32_t flags;   /* INV_* */
    uint32_t seed;    /* jitter / AO hash seed */
    uint32_t tile;    /* tile edge in pixels (a multiple of 32) used to shard a frame over ranks; 0 is read as 64 */
    uint32_t rank;    /* this context renders the tiles with tile_id % nranks == rank (INV_ROOT_IDLE: see there) */
    uint32_t nranks;  /* 0 is read as 1 */
    float ao_tmax;    /* AO ray length; 0 = scene bounding-box diagonal */
} inv_params;

/* Colour / depth attachments (src/invert.cpp:743-750, 795-802) plus a primitive-ID buffer the reference does
 * not have.  Any pointer may be NULL to skip that output.  Row-major, width*height elements, no row padding.
 * With nranks > 1 the buffers hold only this rank's tiles, tile after tile (see inv_tile_layout). */
typedef struct inv_outputs {
    void* color;    /* uint8_t[4 * n] */
    void* depth;    /* float[n]    : z_clip / w_clip in [0,1), 1.0f where nothing was hit (clear value, src/invert.cpp:793) */
    void* primid;   /* uint32_t[n] : index of the OBJ face, 0xFFFFFFFF where nothing was hit */
    int on_device;  /* 0: host pointers (the call copies back and synchronises); 1: device pointers on the context's GPU (the call only enqueues) */
} inv_outputs;

/* ---- context --------------------------------------------------------------------------------------------- */
/* Picks GPU `device` and a stream.  `hip_stream` may be an existing hipStream_t (e.g. the caller's current
 * stream) or NULL to create one.  Fails with INV_ERR_NO_DEVICE when no HIP device is present. */
int inv_ctx_create(int device, void* hip_stream, inv_ctx** out);
/* Waits for the context's stream and frees the context.  Scenes hold a pointer to their context: destroy every
 * scene first (INV_ERR_ARG while one is alive; nothing is freed then). */
int inv_ctx_destroy(inv_ctx* ctx);
int inv_ctx_synchronize(inv_ctx* ctx);
/* The hipStream_t the context enqueues on (the one given to inv_ctx_create, or the one it created): callers that
 * follow a draw with work of their own (a collective, a copy) enqueue it there or order their stream after it. */
int inv_ctx_stream(inv_ctx* ctx, void** hip_stream);
/* A context keeps the device memory its last scene build needed only while it ran (about 120 bytes per triangle)
 * and the allocation of the largest scene destroyed since (for the next build that fits it),
 * so that the next build does not allocate them again: the counterpart of the staging buffer transferData creates and
 * destroys on every call, with the TODO to keep it (src/invert.cpp:991-992, 1000-1032, 1078-1079).  This call
 * releases both; inv_ctx_destroy does too. */
int inv_ctx_trim(inv_ctx* ctx);

/* ---- mesh: struct Obj ------------------------------------------------------------------------------------ */
/* Obj::Obj(const char*) (src/invert.cpp:42-209): same fixed text format, same float32 arithmetic, so the
 * coordinates are bit-identical to the reference loader's.  Malformed text is INV_ERR_FORMAT (the reference
 * has undefined behaviour there). */
int inv_mesh_load_obj(const char* path, inv_mesh** out);
int inv_mesh_parse_obj(const char* text, uint64_t size, inv_mesh** out);
/* A second, tolerant reader for meshes that are not in that format (the TODOs at src/invert.cpp:24, 41): free-form
 * numbers (correctly rounded, so NOT bit-identical to the loader above on the same text), `f` corners i, i/t, i//n,
 * i/t/n, negative indices, polygons (fanned), CRLF, missing final newline; other line kinds are ignored. */
int inv_mesh_load_obj_general(const char* path, inv_mesh** out);
/* The arrays the draw binds (src/invert.cpp:826-827): float4-stride vertices (lane 3 ignored), u32 indices. */
int inv_mesh_from_arrays(const float* xyzw, uint32_t nv, const uint32_t* idx, uint32_t nidx, inv_mesh** out);
/* Obj::vertexDataSize / numIndices (src/invert.cpp:27-31, 77-79), in elements. */
int inv_mesh_counts(const inv_mesh* mesh, uint32_t* nv, uint32_t* nidx);
const float* inv_mesh_vertices(const inv_mesh* mesh);  /* Obj::vertexData, nv*4 floats */
const uint32_t* inv_mesh_indices(const inv_mesh* mesh); /* Obj::indexData, nidx values, 0-based */
/* Moves the mesh's arrays into page-locked host memory, so that the uploads of later scene builds are DMA straight
 * out of them (the staging buffer transferData keeps host-visible for its transfer queue, src/invert.cpp:993-1082, and
 * its TODO about keeping staging buffers mapped, src/invert.cpp:992).  Optional: builds from an unpinned mesh copy
 * through the runtime's own staging.  Needs a HIP device (INV_ERR_NO_DEVICE otherwise); pointers handed out by
 * inv_mesh_vertices / inv_mesh_indices before the call are invalid after it. */
int inv_mesh_pin(inv_mesh* mesh);
int inv_mesh_destroy(inv_mesh* mesh);                   /* Obj::~Obj (src/invert.cpp:33-39) */

/* ---- camera: updateUniformBuffer ------------------------------------------------------------------------- */
/* The camera the reference hard-codes (src/invert.cpp:630-639): eye (0,0,1) looking at the origin, a 180 degree
 * roll, 45 degree vertical field of view, near 0.1, far 10, aspect width/height, zero-to-one depth. */
int inv_camera_reference(uint32_t width, uint32_t height, inv_camera* cam);
/* The same camera with the model matrix turned about the up axis by elapsed_seconds * 90 degrees: a use for the
 * `elapsed` the reference measures every frame and ignores (src/invert.cpp:623-626).  The scene's LBVH is built in
 * object space, so a sequence of such frames (or of any cameras and extents) reuses it unchanged. */
int inv_camera_turntable(uint32_t width, uint32_t height, float elapsed_seconds, inv_camera* cam);

/* ---- scene: vertex/index buffers + acceleration structure -------------------------------------------------- */
/* Uploads the mesh (src/invert.cpp:1136-1140, 1161-1164), assembles triangle k from indices 3k..3k+2
 * (VK_PRIMITIVE_TOPOLOGY_TRIANGLE_LIST, src/invert.cpp:1307-1310) into the HBM triangle arrays and builds the
 * LBVH over them on the GPU.  The mesh may be destroyed afterwards. */
int inv_scene_build(inv_ctx* ctx, const inv_mesh* mesh, inv_scene** out);
/* Same, with `ninst` translated copies of the mesh flattened into one triangle array; primitive ID =
 * instance * (nidx/3) + face. */
int inv_scene_build_instanced(inv_ctx* ctx, const inv_mesh* mesh, const float* translations_xyz, uint32_t ninst,
                              inv_scene** out);
int inv_scene_counts(const inv_scene* scene, uint32_t* ntri, uint32_t* nnodes, uint64_t* scene_bytes);
int inv_scene_destroy(inv_scene* scene);
/* Copies the LBVH to host arrays (any may be NULL): morton[ntri], order[ntri] (sorted position -> triangle),
 * child[2*(ntri-1)] (bit 31 = leaf), parent[ntri-1], leaf_parent[ntri], box[12*(ntri-1)]. */
int inv_scene_read_lbvh(inv_scene* scene, uint64_t* morton, uint32_t* order, uint32_t* child, uint32_t* parent,
                        uint32_t* leaf_parent, float* box);

/* The scene's fine cut, which the per-frame binning pass reads instead of walking the tree: every subtree (or single
 * leaf) of at most INV_FINE_CUT_LEAVES sorted leaves whose parent covers more, as 8 floats per record: box lo.xyz, first sorted leaf
 * (uint32 bits), box hi.xyz, leaf count (uint32 bits); in no particular order.  *nrecords is the number of records
 * the scene holds (0: none -- no more than INV_FINE_CUT_LEAVES triangles, or a tree too unbalanced for the array); at most `cap` are
 * copied.  `records` may be NULL. */
#define INV_FINE_CUT_LEAVES 16u
int inv_scene_read_fine_cut(inv_scene* scene, float* records, uint32_t cap, uint32_t* nrecords);

/* ---- the vertex stage on its own ---------------------------------------------------------------------------- */
/* vertexShader (src/shader.slang:22-29) over every vertex of the scene's mesh: clip = P*(V*(M*(pos,1))) as three
 * successive products, colour = (1, 1, 10*pos.z).  clip4: nv*4 floats, color3: nv*3 floats (either may be NULL). */
int inv_vertex_shader(inv_scene* scene, const inv_camera* cam, float* clip4, float* color3, int on_device);

/* ---- the draw ------------------------------------------------------------------------------------------------ */
/* vkCmdDrawIndexed and its attachments (src/invert.cpp:743-750, 795-802, 815-828): for every pixel the visible
 * triangle (cull BACK / front CLOCKWISE, depth LESS against a 1.0 clear, lower primitive wins ties:
 * src/invert.cpp:1331-1337, 1380-1385), its depth and its colour (src/shader.slang:27, 31-36) stored as sRGB8. */
int inv_render(inv_scene* scene, const inv_camera* cam, const inv_params* params, const inv_outputs* out);
/* inv_render for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same draw, same results, on the context's
 * stream; it calls no hipMalloc / hipFree: the occlusion queue, which inv_render allocates on first use, lives in a
 * static device array of the library (INV_NOALLOC_QUEUE_PIXELS records, 13.6 MB per GPU, held until the process
 * ends).  Limits: out->on_device must be 1 (INV_ERR_ARG otherwise); with ao_rays > 0 the hit rectangle of the frame
 * (inv_scene_hit_rect) must hold at most INV_NOALLOC_QUEUE_PIXELS pixels (INV_ERR_ARG otherwise, nothing is drawn).
 * Every call on a GPU shares that GPU's array: calls with ao_rays > 0 that overlap on one GPU (other contexts,
 * other threads, other streams) race; calls on one context's stream are ordered and safe. */
#define INV_NOALLOC_QUEUE_PIXELS 262144u
int inv_render_noalloc(inv_scene* scene, const inv_camera* cam, const inv_params* params, const inv_outputs* out);

/* Which kernels inv_render would draw this frame with (device attachments assumed): one of the INV_PATH_* values,
 * plus INV_PATH_BINNED when a binning pass runs first.  For tests, benchmarks and reports: the pixels are the same
 * whichever path draws them. */
enum {
    INV_PATH_BRUTE = 0,   /* every triangle for every ray (INV_BRUTE_FORCE, scenes without an LBVH, very deep trees) */
    INV_PATH_LANES = 1,   /* one lane per ray with a stack (INV_LANE_TRAVERSAL) */
    INV_PATH_BLOCKS = 2,  /* 16x16 pixel blocks, a wave per quadrant */
    INV_PATH_TILES = 3,   /* 8x8 pixel tiles whose waves share the candidates or the samples */
    INV_PATH_TILES_BATCHED = 4, /* 8x8 pixel tiles that work through their candidates in batches */
    INV_PATH_SCATTER = 5, /* one lane per triangle into a visibility buffer, then a resolve pass */
    INV_PATH_BINNED = 16u,
};
int inv_draw_path(const inv_scene* scene, const inv_camera* cam, const inv_params* params, uint32_t* path);

/* ---- frames in flight ------------------------------------------------------------------------------------------ */
/* The reference keeps MAX_FRAMES_IN_FLIGHT = 2 frames in flight (src/invert.cpp:241): drawFrame waits for the fence
 * of the frame slot it is about to reuse (src/invert.cpp:646), records and submits that slot's command buffer with
 * the fence (src/invert.cpp:655-672), moves on to the next slot (src/invert.cpp:698) and so returns while the GPU
 * still draws.  Here a frame slot is a HIP stream of the context's own plus whatever the draw needs beside the
 * caller's attachments (host staging, the occlusion queue), so draws enqueued on different slots run on the GPU
 * at the same time: the dispatch gap and the tail of one frame's slowest tiles are filled by the next frame.
 *
 * inv_render_in_flight is inv_render on slot `slot` (< INV_MAX_FRAMES_IN_FLIGHT) instead of the context's stream.
 * Slots are independent of each other and of the context's stream: give every slot attachments of its own, and
 * before anything else touches a slot's attachments either wait on the host (inv_frame_wait) or order the context's
 * stream after the slots (inv_frames_join).  Streams are created on first use and live as long as the context. */
#define INV_MAX_FRAMES_IN_FLIGHT 4
int inv_render_in_flight(inv_scene* scene, uint32_t slot, const inv_camera* cam, const inv_params* params,
                         const inv_outputs* out);
/* vkWaitForFences on the slot's fence (src/invert.cpp:646): blocks the host until every draw enqueued on the slot
 * has finished.  A slot that was never used returns at once. */
int inv_frame_wait(inv_ctx* ctx, uint32_t slot);
/* Device-side join: work enqueued on the context's stream after this call starts only when everything enqueued on
 * the slots so far has finished (the host does not wait).  The counterpart of the semaphore the reference's submit
 * signals for its present (src/invert.cpp:660-683). */
int inv_frames_join(inv_ctx* ctx);
/* The slot's hipStream_t, for callers that record their own events on it or order their own streams after it. */
int inv_frame_stream(inv_ctx* ctx, uint32_t slot, void** hip_stream);

/* ---- several draws in one submission --------------------------------------------------------------------------- */
/* vkQueueSubmit takes commandBufferCount command buffers (src/invert.cpp:661-672); the reference always hands it one.
 * inv_render_batch enqueues `nframes` draws of one scene at once: frame i is drawn with cams[i] and params[i] into
 * outs[i], exactly as inv_render would draw it (extents, flags and attachments may differ from frame to frame; the
 * attachments of different frames must not overlap).  Frames the tile kernel can draw directly into device
 * attachments (LBVH traversal, no occlusion rays; shards of several ranks included) share launches, up to eight frames each, so that the
 * dispatch gap and the wait for a frame's slowest tiles are paid once per launch instead of once per frame;
 * a frame that cannot share (occlusion rays, host attachments, four or more samples per pixel, triangles far smaller
 * than a pixel, another traversal) is drawn on its own between the shared launches, and so is a frame that would be
 * alone in its launch (a draw on its own picks the kernel variant that finishes one frame soonest).  `slot` is a frame slot (see inv_render_in_flight) or INV_CONTEXT_STREAM
 * for the context's own stream. */
#define INV_CONTEXT_STREAM 0xFFFFFFFFu
int inv_render_batch(inv_scene* scene, uint32_t slot, uint32_t nframes, const inv_camera* cams, const inv_params* params,
                     const inv_outputs* outs);

/* Number of pixels rank `rank` owns and, optionally, the tile ids in the order they are stored. */
int inv_tile_layout(const inv_params* params, uint64_t* npixels, uint32_t* tile_ids, uint32_t tile_ids_cap,
                    uint32_t* ntiles);
/* Rank 0 after the gather: scatter `nranks` tile-major buffers into one row-major frame.  Rank r's buffer starts
 * `rank_stride` elements after rank r-1's (0 = back to back).  Device pointers on the context's GPU; elem_size is
 * 4 for every attachment.  Only enqueues. */
int inv_deinterleave(inv_ctx* ctx, const inv_params* params, const void* gathered, uint64_t rank_stride, void* frame,
                     uint32_t elem_size);

/* Sparse composite (INV_SPARSE, nranks > 1).  No counterpart in the reference, which draws on one GPU.
 * Only the tiles touched by the hit rectangle -- the pixel rectangle [x0, x1) x [y0, y1) outside which every pixel
 * of this scene and camera is background -- are dealt out: tile j of that grid (row-major) goes to rank
 * j % nranks, slot j / nranks.  Rank 0's outputs are the row-major frame: it renders its own tiles in place and
 * clears every pixel outside the dealt tiles; ranks 1.. hold only their slots.  After gathering the slot buffers,
 * rank 0 copies the other ranks' tiles into its frame with inv_deinterleave_sparse.  inv_render computes the same
 * rectangle from the scene and camera on every rank. */
int inv_scene_hit_rect(const inv_scene* scene, const inv_camera* cam, uint32_t width, uint32_t height,
                       uint32_t rect[4]);
/* Layout for params->rank given the hit rectangle (host arithmetic only): grid = {first tile x, first tile y,
 * tiles across, tiles down}, npixels = elements per attachment of this rank's buffers, tile_ids = the frame tile
 * ids (row-major over the whole frame) of its tiles in slot order. */
int inv_sparse_layout(const inv_params* params, const uint32_t rect[4], uint32_t grid[4], uint64_t* npixels,
                      uint32_t* tile_ids, uint32_t tile_ids_cap, uint32_t* ntiles);
/* Rank 0 after the gather: `nplanes` attachments at once.  Attachment k of rank r starts at element
 * r * rank_stride + k * plane_stride of `gathered` and at k * frame_plane_stride of `frame` (4-byte elements,
 * device pointers).  Rank 0's own part of `gathered` is not read.  Only enqueues. */
int inv_deinterleave_sparse(inv_ctx* ctx, const inv_params* params, const uint32_t rect[4], const void* gathered,
                            uint64_t rank_stride, uint64_t plane_stride, uint32_t nplanes, void* frame,
                            uint64_t frame_plane_stride);

/* ---- the exchange of a multi-GPU composite, from C ------------------------------------------------------------ */
/* No counterpart in the reference (one GPU).  One process per GPU; each renders its shard with inv_render (params
 * rank / nranks), then every rank calls inv_comm_gather and rank 0 puts the tiles into its frame with inv_deinterleave
 * or inv_deinterleave_sparse.  Two transports:
 *   inv_comm_create_rccl  grouped ncclSend / ncclRecv to rank 0 on the context's stream (only enqueues: the call
 *                         returns before the data has moved; later work on the context's stream is ordered behind it).
 *                         Rank 0 makes an id with inv_comm_unique_id and hands its INV_COMM_ID_BYTES bytes to the other
 *                         ranks by whatever means the launcher has (a file, MPI, a socket); creating the communicator
 *                         is collective: every rank must call it.  librccl is opened at run time (INV_ERR_UNSUPPORTED
 *                         if it cannot be).
 *   inv_comm_create_host  staging through files in directory `dir`, which every rank must reach (/dev/shm on one
 *                         node): for rehearsing the several-rank path where RCCL cannot run (it refuses two ranks on one
 *                         GPU), and as a fallback.  Blocking: the call returns when the data has arrived.  Give every
 *                         communicator a fresh directory of its own (mkdtemp): files are named by exchange number and
 *                         rank, and each carries a header (exchange, rank, rank count, element count) that rank 0
 *                         checks, so a file left by another run is refused (INV_ERR_IO), not taken for tiles. */
typedef struct inv_comm inv_comm;
#define INV_COMM_ID_BYTES 128
enum { INV_COMM_RCCL = 0, INV_COMM_HOST = 1 };
int inv_comm_unique_id(void* id);
int inv_comm_create_rccl(inv_ctx* ctx, const void* id, uint32_t rank, uint32_t nranks, inv_comm** out);
int inv_comm_create_host(inv_ctx* ctx, const char* dir, uint32_t rank, uint32_t nranks, inv_comm** out);
/* Every rank > 0 sends `count` 4-byte elements from `send`; rank 0 receives rank r's at element r * rank_stride of `recv`
 * (its own part is not touched, `send` may be NULL there).  Device pointers on the communicator's GPU.  All ranks make
 * the same sequence of calls with the same `count`. */
int inv_comm_gather(inv_comm* comm, const void* send, uint64_t count, void* recv, uint64_t rank_stride);
int inv_comm_destroy(inv_comm* comm);
/* The GPU a context was created on. */
int inv_ctx_device(inv_ctx* ctx, int* device);

/* ---- making the attachments inspectable (host buffers) ----------------------------------------------------------- */
/* The reference presents its colour image (src/invert.cpp:674-683) and stores nothing.  Binary PPM (P6) of a colour
 * buffer (pass INV_RGBA in `flags` if it was rendered with it), little-endian PFM of a depth buffer, and a raw
 * uint32 dump of primitive ids behind a 16-byte header {"PID1", width, height, 0}. */
int inv_write_ppm(const char* path, const void* color, uint32_t width, uint32_t height, uint32_t flags);
/* The same buffer as an 8-bit RGBA PNG with an sRGB chunk (the attachment holds sRGB-encoded values); the image data
 * is stored uncompressed (about 4 bytes per pixel). */
int inv_write_png(const char* path, const void* color, uint32_t width, uint32_t height, uint32_t flags);
int inv_write_pfm(const char* path, const float* depth, uint32_t width, uint32_t height);
int inv_write_primid(const char* path, const uint32_t* primid, uint32_t width, uint32_t height);

const char* inv_last_error(void);
const char* inv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* INVERT_H */
