// The exchange step of a multi-GPU composite, from C: every rank's tile slots to rank 0.
//
// Two transports behind one call:
//   RCCL   grouped ncclSend / ncclRecv on the context's stream (one process per GPU, xGMI between them).  The
//          library is opened at run time (dlopen): a process that never composites does not need it, and a process
//          that has loaded a RCCL of its own (PyTorch bundles one) gets that one.
//   host   staging through files in a directory that every rank can reach (/dev/shm on one node): device -> host ->
//          file -> host -> device.  For rehearsals of the several-rank code path on one GPU, where RCCL refuses two
//          ranks on one device, and for machines without a working RCCL.  Blocking.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

namespace {

// what is used of rccl.h, declared here so that the library builds and loads without it
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == INV_COMM_ID_BYTES, "the id travels as INV_COMM_ID_BYTES bytes");
constexpr int kNcclSuccess = 0, kNcclUint32 = 3; // ncclResult_t ncclSuccess, ncclDataType_t ncclUint32

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl* r = [] {
        Rccl* x = new Rccl;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x->handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x->handle) break;
        }
        if (!x->handle) {
            const char* why = dlerror(); // (a second call would return null: the first one clears the message)
            x->error = std::string("cannot open librccl.so.1: ") + (why ? why : "?");
            return x;
        }
#define INV_SYM(field, sym)                                                                  \
    if (!(*(void**)(&x->field) = dlsym(x->handle, sym)) && x->error.empty()) x->error = std::string("librccl lacks ") + sym
        INV_SYM(GetUniqueId, "ncclGetUniqueId");
        INV_SYM(CommInitRank, "ncclCommInitRank");
        INV_SYM(CommDestroy, "ncclCommDestroy");
        INV_SYM(GroupStart, "ncclGroupStart");
        INV_SYM(GroupEnd, "ncclGroupEnd");
        INV_SYM(Send, "ncclSend");
        INV_SYM(Recv, "ncclRecv");
        INV_SYM(GetErrorString, "ncclGetErrorString");
#undef INV_SYM
        return x;
    }();
    return r;
}

int nccl_fail(int rc, const char* what) {
    Rccl* r = rccl();
    return inv::fail(INV_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}

bool file_exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

} // namespace

struct inv_comm {
    int kind = INV_COMM_RCCL;
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t rank = 0, nranks = 1;
    ncclComm_t nccl = nullptr;
    std::string dir;      // host transport: where the ranks meet
    uint64_t seq = 0;     // host transport: exchanges done so far
    std::vector<uint32_t> host; // host transport: staging
};

using namespace inv;

extern "C" {

int inv_comm_unique_id(void* id) {
    if (!id) return fail(INV_ERR_ARG, "inv_comm_unique_id: id is null");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(INV_ERR_UNSUPPORTED, "inv_comm_unique_id: " + r->error);
    ncclUniqueId u;
    const int rc = r->GetUniqueId(&u);
    if (rc != kNcclSuccess) return nccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof u);
    return INV_OK;
}

int inv_comm_create_rccl(inv_ctx* ctx, const void* id, uint32_t rank, uint32_t nranks, inv_comm** out) {
    if (!out) return fail(INV_ERR_ARG, "inv_comm_create_rccl: out is null");
    *out = nullptr;
    if (!ctx || !id) return fail(INV_ERR_ARG, "inv_comm_create_rccl: null argument");
    if (nranks == 0 || rank >= nranks || nranks > 65536u) return fail(INV_ERR_ARG, "inv_comm_create_rccl: rank must be below nranks");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(INV_ERR_UNSUPPORTED, "inv_comm_create_rccl: " + r->error);
    void* stream = nullptr;
    int device = 0;
    int rc = inv_ctx_stream(ctx, &stream);
    if (rc != INV_OK) return rc;
    if ((rc = inv_ctx_device(ctx, &device)) != INV_OK) return rc;
    if (hipSetDevice(device) != hipSuccess) return fail(INV_ERR_HIP, "inv_comm_create_rccl: hipSetDevice");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const int nrc = r->CommInitRank(&comm, (int)nranks, u, (int)rank);
    if (nrc != kNcclSuccess) return nccl_fail(nrc, "ncclCommInitRank");
    inv_comm* c = new (std::nothrow) inv_comm;
    if (!c) {
        (void)r->CommDestroy(comm);
        return fail(INV_ERR_NOMEM, "inv_comm_create_rccl: out of memory");
    }
    c->kind = INV_COMM_RCCL, c->device = device, c->stream = (hipStream_t)stream, c->rank = rank, c->nranks = nranks, c->nccl = comm;
    *out = c;
    return INV_OK;
}

int inv_comm_create_host(inv_ctx* ctx, const char* dir, uint32_t rank, uint32_t nranks, inv_comm** out) try {
    if (!out) return fail(INV_ERR_ARG, "inv_comm_create_host: out is null");
    *out = nullptr;
    if (!ctx || !dir || !*dir) return fail(INV_ERR_ARG, "inv_comm_create_host: null argument");
    if (nranks == 0 || rank >= nranks || nranks > 65536u) return fail(INV_ERR_ARG, "inv_comm_create_host: rank must be below nranks");
    struct stat st;
    if (::stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return fail(INV_ERR_IO, std::string("inv_comm_create_host: not a directory: ") + dir);
    void* stream = nullptr;
    int device = 0;
    int rc = inv_ctx_stream(ctx, &stream);
    if (rc != INV_OK) return rc;
    if ((rc = inv_ctx_device(ctx, &device)) != INV_OK) return rc;
    inv_comm* c = new inv_comm;
    c->kind = INV_COMM_HOST, c->device = device, c->stream = (hipStream_t)stream, c->rank = rank, c->nranks = nranks, c->dir = dir;
    *out = c;
    return INV_OK;
} catch (...) {
    return fail(INV_ERR_NOMEM, "inv_comm_create_host: out of host memory");
}

int inv_comm_destroy(inv_comm* comm) {
    if (!comm) return INV_OK;
    if (comm->nccl) {
        (void)hipSetDevice(comm->device);
        (void)hipStreamSynchronize(comm->stream);
        (void)rccl()->CommDestroy(comm->nccl);
    }
    delete comm;
    return INV_OK;
}

int inv_comm_gather(inv_comm* comm, const void* send, uint64_t count, void* recv, uint64_t rank_stride) try {
    if (!comm) return fail(INV_ERR_ARG, "inv_comm_gather: comm is null");
    if (comm->nranks == 1 || count == 0) return INV_OK; // nothing to move
    if (comm->rank == 0 && !recv) return fail(INV_ERR_ARG, "inv_comm_gather: rank 0 needs a receive buffer");
    if (comm->rank != 0 && !send) return fail(INV_ERR_ARG, "inv_comm_gather: send is null");
    if (comm->rank == 0 && rank_stride < count) return fail(INV_ERR_ARG, "inv_comm_gather: rank_stride is smaller than count");
    if (hipSetDevice(comm->device) != hipSuccess) return fail(INV_ERR_HIP, "inv_comm_gather: hipSetDevice");
    if (comm->kind == INV_COMM_RCCL) {
        Rccl* r = rccl();
        int rc = r->GroupStart();
        if (rc != kNcclSuccess) return nccl_fail(rc, "ncclGroupStart");
        if (comm->rank == 0) {
            for (uint32_t p = 1; p < comm->nranks && rc == kNcclSuccess; p++)
                rc = r->Recv((uint32_t*)recv + (size_t)p * rank_stride, (size_t)count, kNcclUint32, (int)p, comm->nccl, comm->stream);
        } else {
            rc = r->Send(send, (size_t)count, kNcclUint32, 0, comm->nccl, comm->stream);
        }
        const int rc_end = r->GroupEnd();
        if (rc != kNcclSuccess) return nccl_fail(rc, comm->rank == 0 ? "ncclRecv" : "ncclSend");
        if (rc_end != kNcclSuccess) return nccl_fail(rc_end, "ncclGroupEnd");
        return INV_OK;
    }
    // host transport: one file per sending rank and exchange
    const uint64_t seq = comm->seq++;
    const auto path = [&](uint32_t rank, const char* ext) {
        char name[96];
        std::snprintf(name, sizeof name, "/inv_gather_%llu_%u.%s", (unsigned long long)seq, rank, ext);
        return comm->dir + name;
    };
    comm->host.resize((size_t)count);
    // every file starts with what it is: a file of another exchange, another communicator's size or an older run
    // left in the directory is refused instead of being taken for this exchange's tiles
    const auto header = [&](uint32_t rank, uint64_t out[4]) {
        out[0] = 0x31564e49484d4f43ull /* "COMHINV1" */, out[1] = seq, out[2] = ((uint64_t)comm->nranks << 32) | rank, out[3] = count;
    };
    if (comm->rank != 0) {
        if (hipMemcpyAsync(comm->host.data(), send, (size_t)count * 4, hipMemcpyDeviceToHost, comm->stream) != hipSuccess ||
            hipStreamSynchronize(comm->stream) != hipSuccess)
            return fail(INV_ERR_HIP, "inv_comm_gather: copy to the host failed");
        const std::string tmp = path(comm->rank, "tmp"), fin = path(comm->rank, "bin");
        FILE* fp = std::fopen(tmp.c_str(), "wb");
        if (!fp) return fail(INV_ERR_IO, "inv_comm_gather: cannot create " + tmp);
        uint64_t hd[4];
        header(comm->rank, hd);
        const size_t wrote = std::fwrite(hd, sizeof hd, 1, fp) == 1 ? std::fwrite(comm->host.data(), 4, (size_t)count, fp) : 0;
        if (std::fclose(fp) != 0 || wrote != (size_t)count) return fail(INV_ERR_IO, "inv_comm_gather: cannot write " + tmp);
        if (std::rename(tmp.c_str(), fin.c_str()) != 0) return fail(INV_ERR_IO, "inv_comm_gather: cannot rename " + tmp);
        return INV_OK;
    }
    for (uint32_t p = 1; p < comm->nranks; p++) {
        const std::string fin = path(p, "bin");
        const auto t0 = std::chrono::steady_clock::now();
        while (!file_exists(fin)) { // a whole file appears at once (rename)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return fail(INV_ERR_IO, "inv_comm_gather: rank " + std::to_string(p) + " did not deliver " + fin);
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        FILE* fp = std::fopen(fin.c_str(), "rb");
        if (!fp) return fail(INV_ERR_IO, "inv_comm_gather: cannot open " + fin);
        uint64_t hd[4], want[4];
        header(p, want);
        const bool hd_ok = std::fread(hd, sizeof hd, 1, fp) == 1 && std::memcmp(hd, want, sizeof hd) == 0;
        const size_t got = hd_ok ? std::fread(comm->host.data(), 4, (size_t)count, fp) : 0;
        const bool at_end = hd_ok && std::fgetc(fp) == EOF;
        std::fclose(fp);
        (void)::unlink(fin.c_str());
        if (!hd_ok) return fail(INV_ERR_IO, "inv_comm_gather: " + fin + " is not this exchange's file (left by another run or communicator?)");
        if (got != (size_t)count || !at_end) return fail(INV_ERR_IO, "inv_comm_gather: " + fin + " has the wrong length");
        if (hipMemcpyAsync((uint32_t*)recv + (size_t)p * rank_stride, comm->host.data(), (size_t)count * 4, hipMemcpyHostToDevice,
                           comm->stream) != hipSuccess ||
            hipStreamSynchronize(comm->stream) != hipSuccess) // the staging vector is reused for the next rank
            return fail(INV_ERR_HIP, "inv_comm_gather: copy to the device failed");
    }
    return INV_OK;
} catch (...) {
    return fail(INV_ERR_NOMEM, "inv_comm_gather: out of host memory");
}

} // extern "C"
