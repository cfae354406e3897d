// libhalo_rccl.so: halo_allgather_fn over RCCL (include/halo_rccl.h).  Optional; the core library neither links nor loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/halo_rccl.h"

static_assert(sizeof(ncclUniqueId) == HALO_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");

struct halo_rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    bool owns = false;      // communicator and stream are ours to destroy
    int world = 1, device = 0;
    size_t cap_words = 0;   // per rank
    uint64_t *h_send = nullptr, *h_recv = nullptr;  // pinned
    uint64_t *d_send = nullptr, *d_recv = nullptr;
    size_t calls = 0;
};

namespace {
thread_local std::string g_err;
int fail(const char *what, const char *why) {
    g_err = std::string(what) + ": " + why;
    return -1;
}
#define RC_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, hipGetErrorString(e_)); } while (0)
#define RC_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(#x, ncclGetErrorString(r_)); } while (0)

void release_buffers(halo_rccl *g) {
    if (g->h_send) (void)hipHostFree(g->h_send);
    if (g->h_recv) (void)hipHostFree(g->h_recv);
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    g->h_send = g->h_recv = g->d_send = g->d_recv = nullptr;
    g->cap_words = 0;
}
int ensure_buffers(halo_rccl *g, size_t words) {
    if (words <= g->cap_words) return 0;
    release_buffers(g);
    size_t cap = words < 64 ? 64 : words;  // the library's records are at most 33 words
    RC_HIP(hipHostMalloc(&g->h_send, cap * 8));
    RC_HIP(hipHostMalloc(&g->h_recv, cap * 8 * (size_t)g->world));
    RC_HIP(hipMalloc(&g->d_send, cap * 8));
    RC_HIP(hipMalloc(&g->d_recv, cap * 8 * (size_t)g->world));
    g->cap_words = cap;
    return 0;
}
}  // namespace

extern "C" {

const char *halo_rccl_last_error(void) { return g_err.c_str(); }

int halo_rccl_unique_id(uint8_t id[HALO_RCCL_ID_BYTES]) {
    if (!id) return fail("halo_rccl_unique_id", "null pointer");
    ncclUniqueId u;
    RC_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, HALO_RCCL_ID_BYTES);
    return 0;
}

int halo_rccl_create(const uint8_t id[HALO_RCCL_ID_BYTES], int rank, int world, int device, halo_rccl **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail("halo_rccl_create", "bad argument");
    RC_HIP(hipSetDevice(device));
    halo_rccl *g = new halo_rccl();
    g->world = world;
    g->device = device;
    g->owns = true;
    ncclUniqueId u;
    std::memcpy(&u, id, HALO_RCCL_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&g->comm, world, u, rank);
    if (r != ncclSuccess) { delete g; return fail("ncclCommInitRank", ncclGetErrorString(r)); }
    hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)ncclCommDestroy(g->comm); delete g; return fail("hipStreamCreateWithFlags", hipGetErrorString(e)); }
    if (ensure_buffers(g, 64)) { halo_rccl_destroy(g); return -1; }
    *out = g;
    return 0;
}

int halo_rccl_wrap(void *nccl_comm, void *hip_stream, int world, int device, halo_rccl **out) {
    if (!nccl_comm || !out || world < 1) return fail("halo_rccl_wrap", "bad argument");
    RC_HIP(hipSetDevice(device));
    halo_rccl *g = new halo_rccl();
    g->comm = static_cast<ncclComm_t>(nccl_comm);
    g->stream = static_cast<hipStream_t>(hip_stream);
    g->world = world;
    g->device = device;
    g->owns = false;
    if (ensure_buffers(g, 64)) { halo_rccl_destroy(g); return -1; }
    *out = g;
    return 0;
}

void halo_rccl_destroy(halo_rccl *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    release_buffers(g);
    if (g->owns) {
        if (g->comm) (void)ncclCommDestroy(g->comm);
        if (g->stream) (void)hipStreamDestroy(g->stream);
    }
    delete g;
}

int halo_allgather_rccl(void *user, const uint64_t *send, size_t words, uint64_t *recv) {
    halo_rccl *g = static_cast<halo_rccl *>(user);
    if (!g || !send || !recv || words == 0) return fail("halo_allgather_rccl", "bad argument");
    RC_HIP(hipSetDevice(g->device));
    if (ensure_buffers(g, words)) return -1;
    std::memcpy(g->h_send, send, words * 8);
    RC_HIP(hipMemcpyAsync(g->d_send, g->h_send, words * 8, hipMemcpyHostToDevice, g->stream));
    RC_NCCL(ncclAllGather(g->d_send, g->d_recv, words, ncclUint64, g->comm, g->stream));
    RC_HIP(hipMemcpyAsync(g->h_recv, g->d_recv, words * 8 * (size_t)g->world, hipMemcpyDeviceToHost, g->stream));
    RC_HIP(hipStreamSynchronize(g->stream));
    std::memcpy(recv, g->h_recv, words * 8 * (size_t)g->world);
    g->calls++;
    return 0;
}

size_t halo_rccl_calls(const halo_rccl *g) { return g ? g->calls : 0; }
int halo_rccl_world(const halo_rccl *g) { return g ? g->world : 0; }

}  // extern "C"
