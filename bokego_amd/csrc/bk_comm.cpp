// bk_comm.cpp -- libbkcomm.so: the end-of-generation all-reduce of self-play statistics on RCCL (see
// include/bokego_comm.h).  Kept out of libbokego_amd.so so that the engine has no RCCL dependency.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/bokego_comm.h"

static_assert(BK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

struct bk_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* d_buf = nullptr;
    float* d_bc = nullptr;       // broadcast staging, grown on demand
    size_t bc_bytes = 0;
    int rank = 0, world = 1, device = 0;
};

namespace {
thread_local std::string g_err;
constexpr int kMaxN = 4096;

int fail(const std::string& m) {
    g_err = m;
    return -1;
}
}  // namespace

#define TRY_HIP(call)                                                                           \
    do {                                                                                        \
        hipError_t _s = (call);                                                                 \
        if (_s != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(_s));   \
    } while (0)
#define TRY_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t _s = (call);                                                               \
        if (_s != ncclSuccess) return fail(std::string(#call) + ": " + ncclGetErrorString(_s)); \
    } while (0)

extern "C" {

int bk_comm_abi_version(void) { return 2; }

const char* bk_comm_last_error(void) { return g_err.c_str(); }

int bk_comm_unique_id(uint8_t id[BK_COMM_ID_BYTES]) {
    if (!id) return fail("id is NULL");
    ncclUniqueId u;
    TRY_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, BK_COMM_ID_BYTES);
    return 0;
}

int bk_comm_init(int rank, int world, const uint8_t id[BK_COMM_ID_BYTES], int device_id, bk_comm** out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail("bad rank/world/id");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return fail("device_id out of range (no GPU?)");
    bk_comm* c = new (std::nothrow) bk_comm();
    if (!c) return fail("host allocation failed");
    c->rank = rank;
    c->world = world;
    c->device = device_id;
    auto bail = [&]() { bk_comm_destroy(c); return -1; };
    if (hipSetDevice(device_id) != hipSuccess) { g_err = "hipSetDevice failed"; return bail(); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; return bail(); }
    if (hipMalloc((void**)&c->d_buf, kMaxN * sizeof(double)) != hipSuccess) { g_err = "hipMalloc failed"; return bail(); }
    ncclUniqueId u;
    std::memcpy(u.internal, id, BK_COMM_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { g_err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return bail(); }
    *out = c;
    return 0;
}

int bk_comm_allreduce_sum_f64(bk_comm* c, double* buf, int n) {
    if (!c || !buf) return fail("comm or buf is NULL");
    if (n < 0 || n > kMaxN) return fail("n must be in [0, 4096]");
    if (n == 0) return 0;
    TRY_HIP(hipSetDevice(c->device));
    TRY_HIP(hipMemcpyAsync(c->d_buf, buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    TRY_NCCL(ncclAllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
    TRY_HIP(hipMemcpyAsync(buf, c->d_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int bk_comm_broadcast_f32(bk_comm* c, float* buf, int64_t n, int root) {
    if (!c || !buf) return fail("comm or buf is NULL");
    if (n < 0 || root < 0 || root >= c->world) return fail("bad n or root");
    if (n == 0) return 0;
    TRY_HIP(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * sizeof(float);
    if (bytes > c->bc_bytes) {
        if (c->d_bc) (void)hipFree(c->d_bc);
        c->d_bc = nullptr;
        c->bc_bytes = 0;
        TRY_HIP(hipMalloc((void**)&c->d_bc, bytes));
        c->bc_bytes = bytes;
    }
    if (c->rank == root) TRY_HIP(hipMemcpyAsync(c->d_bc, buf, bytes, hipMemcpyHostToDevice, c->stream));
    TRY_NCCL(ncclBroadcast(c->d_bc, c->d_bc, (size_t)n, ncclFloat, root, c->comm, c->stream));
    if (c->rank != root) TRY_HIP(hipMemcpyAsync(buf, c->d_bc, bytes, hipMemcpyDeviceToHost, c->stream));
    TRY_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int bk_comm_barrier(bk_comm* c) {
    if (!c) return fail("comm is NULL");
    TRY_HIP(hipSetDevice(c->device));
    TRY_HIP(hipMemsetAsync(c->d_buf, 0, sizeof(double), c->stream));
    TRY_NCCL(ncclAllReduce(c->d_buf, c->d_buf, 1, ncclDouble, ncclSum, c->comm, c->stream));
    TRY_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int bk_comm_rccl_version(void) {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

int bk_comm_device_pci(const bk_comm* c, char* out, int cap) {
    if (!c || !out || cap < 16) return fail("comm or out is NULL, or cap < 16");
    TRY_HIP(hipDeviceGetPCIBusId(out, cap, c->device));
    return 0;
}

int bk_comm_rank(const bk_comm* c) { return c ? c->rank : -1; }
int bk_comm_world(const bk_comm* c) { return c ? c->world : -1; }

int bk_comm_destroy(bk_comm* c) {
    if (!c) return -1;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->d_bc) (void)hipFree(c->d_bc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

}  // extern "C"
