// Data-parallel gradient exchange through RCCL, called from the C ABI on the caller's HIP streams.
//
// The reference reduces gradients inside nn.DataParallel (main.py:79: ReduceAddCoalesced to GPU 0 + a parameter
// broadcast per step).  Here every rank owns its shard and the only exchange is a SUM all-reduce of the flat
// live-gradient prefix.  This file binds RCCL directly (ncclAllReduce on a hipStream_t) so the collective is
// one more node on the step's own stream(s): no framework enqueue, no framework-side cross-stream events, capturable in
// the same hipGraph as the kernels.  RCCL is resolved with dlopen at first use - the copy that is already in the process
// (PyTorch-ROCm ships one next to its HIP runtime) or the ROCm one - so the library has no link-time dependency on it
// and a single-GPU process never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/ta3n_hip.h"
#include "ta3n_kernels.h"
#include "ta3n_plan.h"

using namespace ta3n;

namespace {

typedef struct { char internal[128]; } NcclUniqueId;        // rccl.h: NCCL_UNIQUE_ID_BYTES 128
typedef void *NcclComm;
constexpr int kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclSum = 0;   // rccl.h: ncclDataType_t, ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;      // (send, recv, recvcount, type, op, ...)
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;               // (send, recv, sendcount, type, ...)
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {            // first a copy that is already mapped (one HIP runtime, one RCCL per process)
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle) break;
    }
    for (const char *n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) { r.error = std::string("librccl.so not found: ") + dlerror(); return r; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(dlsym(r.handle, "ncclReduceScatter"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) r.error = "librccl.so lacks the ncclAllReduce entry points";
    return r;
}

int fail(int code, const std::string &msg) {
    ta3n::set_error(msg);
    return code;
}

__global__ void from_bf16_kernel(const uint2 *__restrict__ src, float4 *__restrict__ dst, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 v = src[i];
        dst[i] = make_float4(__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u),
                             __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u));
    }
}

}  // namespace

struct ta3n_comm {
    ta3n_peer *peer = nullptr;       // ta3n_comm_attach_peer: the exchange goes over peer-mapped buffers instead of ncclAllReduce
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t fork2 = nullptr, join2 = nullptr;      // sharded update: the parameter all-gather that overlaps the next step's first launch
};

// ---- pieces of the sharded update (ta3n_api.hip: ta3n_train_steps_sharded composes them with the step's launches) ----
namespace ta3n {
int comm_rank(const ta3n_comm *c) { return c->rank; }
int comm_world_size(const ta3n_comm *c) { return c->world; }
hipEvent_t comm_event(ta3n_comm *c, int which) {
    hipEvent_t *e[4] = {&c->fork, &c->join, &c->fork2, &c->join2};
    if (!*e[which] && hipEventCreateWithFlags(e[which], hipEventDisableTiming) != hipSuccess) return nullptr;
    return *e[which];
}

// In place: buf[begin + r chunk, + chunk) of rank r receives the SUM over ranks of that range; the other ranges of buf hold junk
// afterwards.  chunk % 4 == 0.  scratch_bf16 (covering the same element offsets, 2 bytes each): bf16 transport.
int comm_reduce_scatter_sum(ta3n_comm *c, float *buf, int64_t begin, int64_t chunk, void *scratch_bf16, hipStream_t s) {
    if (chunk <= 0) return TA3N_OK;
    if (c->peer) return fail(TA3N_ERR_INVALID, "the sharded update runs on RCCL (no peer transport attached)");
    Rccl &r = rccl();
    if (!r.ReduceScatter || !r.AllGather) return fail(TA3N_ERR_HIP, "librccl.so lacks ncclReduceScatter / ncclAllGather");
    int rc;
    if (scratch_bf16) {
        char *s16 = static_cast<char *>(scratch_bf16) + 2 * begin;
        if (launch_to_bf16(buf + begin, reinterpret_cast<float *>(s16), chunk * c->world, s) != 0) return fail(TA3N_ERR_HIP, "bf16 pack launch failed");
        rc = r.ReduceScatter(s16, s16 + 2 * chunk * c->rank, (size_t)chunk, kNcclBfloat16, kNcclSum, c->comm, s);
        if (rc == 0) {
            const int64_t n4 = chunk / 4;
            const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
            hipLaunchKernelGGL(from_bf16_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint2 *>(s16 + 2 * chunk * c->rank),
                               reinterpret_cast<float4 *>(buf + begin + chunk * c->rank), n4);
            if (hipGetLastError() != hipSuccess) return fail(TA3N_ERR_HIP, "bf16 unpack launch failed");
        }
    } else {
        rc = r.ReduceScatter(buf + begin, buf + begin + chunk * c->rank, (size_t)chunk, kNcclFloat32, kNcclSum, c->comm, s);
    }
    if (rc != 0) return fail(TA3N_ERR_HIP, std::string("ncclReduceScatter: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
    return TA3N_OK;
}

// In place: every rank's buf[begin + r chunk, + chunk) to everybody.
int comm_all_gather(ta3n_comm *c, float *buf, int64_t begin, int64_t chunk, hipStream_t s) {
    if (chunk <= 0) return TA3N_OK;
    Rccl &r = rccl();
    if (!r.AllGather) return fail(TA3N_ERR_HIP, "librccl.so lacks ncclAllGather");
    const int rc = r.AllGather(buf + begin + chunk * c->rank, buf + begin, (size_t)chunk, kNcclFloat32, c->comm, s);
    if (rc != 0) return fail(TA3N_ERR_HIP, std::string("ncclAllGather: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
    return TA3N_OK;
}
}  // namespace ta3n

extern "C" {

int ta3n_comm_unique_id(char *id128) {
    if (!id128) return fail(TA3N_ERR_INVALID, "null argument");
    Rccl &r = rccl();
    if (!r.error.empty()) return fail(TA3N_ERR_HIP, r.error);
    NcclUniqueId id;
    const int rc = r.GetUniqueId(&id);
    if (rc != 0) return fail(TA3N_ERR_HIP, std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
    std::memcpy(id128, id.internal, 128);
    return TA3N_OK;
}

int ta3n_comm_create(const char *id128, int rank, int world, ta3n_comm **out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(TA3N_ERR_INVALID, "bad communicator arguments");
    Rccl &r = rccl();
    if (!r.error.empty()) return fail(TA3N_ERR_HIP, r.error);
    ta3n_comm *c = new ta3n_comm();
    c->rank = rank; c->world = world;
    NcclUniqueId id;
    std::memcpy(id.internal, id128, 128);
    const int rc = r.CommInitRank(&c->comm, world, id, rank);      // collective over the `world` ranks, on the current device
    if (rc != 0) {
        delete c;
        return fail(TA3N_ERR_HIP, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
    }
    if (hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail(TA3N_ERR_HIP, "hipEventCreate failed");
    }
    *out = c;
    return TA3N_OK;
}

void ta3n_comm_destroy(ta3n_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->fork2) (void)hipEventDestroy(c->fork2);
    if (c->join2) (void)hipEventDestroy(c->join2);
    delete c;
}

int ta3n_comm_world(const ta3n_comm *c) { return c ? c->world : TA3N_ERR_INVALID; }

int ta3n_comm_attach_peer(ta3n_comm *c, ta3n_peer *peer) {
    if (!c) return fail(TA3N_ERR_INVALID, "null communicator");
    c->peer = peer;      // NULL detaches; the communicator does not own the peer object
    return TA3N_OK;
}

int ta3n_all_reduce_sum(ta3n_comm *c, float *buf, int64_t count, void *scratch_bf16, void *stream) {
    if (!c || !buf || count < 0) return fail(TA3N_ERR_INVALID, "bad all-reduce arguments");
    if (count == 0) return TA3N_OK;
    if (c->peer) return ta3n_peer_all_reduce_sum(c->peer, buf, count, stream);      // (transport type chosen at ta3n_peer_create)
    Rccl &r = rccl();
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    if (scratch_bf16) {     // bf16 transport: round to nearest even, sum in bf16, widen back
        if (count % 4) return fail(TA3N_ERR_INVALID, "bf16 transport needs a multiple of 4 elements");
        if (launch_to_bf16(buf, static_cast<float *>(scratch_bf16), count, s) != 0) return fail(TA3N_ERR_HIP, "bf16 pack launch failed");
        rc = r.AllReduce(scratch_bf16, scratch_bf16, (size_t)count, kNcclBfloat16, kNcclSum, c->comm, s);
        if (rc == 0) {
            const int64_t n4 = count / 4;
            const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
            hipLaunchKernelGGL(from_bf16_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint2 *>(scratch_bf16),
                               reinterpret_cast<float4 *>(buf), n4);
            if (hipGetLastError() != hipSuccess) return fail(TA3N_ERR_HIP, "bf16 unpack launch failed");
        }
    } else {
        rc = r.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm, s);
    }
    if (rc != 0) return fail(TA3N_ERR_HIP, std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
    return TA3N_OK;
}

int ta3n_train_step_ddp(ta3n_plan *p, ta3n_comm *c, const float *x, const float *params, float *grads, float *ws,
                        void *scratch_bf16, void *stream, void *comm_stream) {
    if (!p || !c || !x || !params || !grads || !ws) return fail(TA3N_ERR_INVALID, "null argument");
    const int n = ta3n_num_phases(p, 4);
    if (n < 2) return fail(TA3N_ERR_INVALID, "no fused step for this configuration");
    const int64_t n1 = p->first_floats, live = p->live_floats;
    char *s16 = static_cast<char *>(scratch_bf16);
    hipStream_t s = static_cast<hipStream_t>(stream), cs = static_cast<hipStream_t>(comm_stream);
    if (!cs || cs == s) {       // one stream: the step, then ONE collective over the whole live prefix
        int rc = ta3n_train_step(p, x, params, grads, ws, stream);
        if (rc != TA3N_OK) return rc;
        return ta3n_all_reduce_sum(c, grads, live, scratch_bf16, stream);
    }
    // two streams: everything but the shared frame FC's gradient (the last launch's only output, first in the flat layout)
    // is reduced on `comm_stream` while that launch runs; the rest follows; `stream` continues after both.
    int rc = ta3n_train_step_range(p, x, params, grads, ws, 0, n - 1, stream);
    if (rc != TA3N_OK) return rc;
    if (hipEventRecord(c->fork, s) != hipSuccess || hipStreamWaitEvent(cs, c->fork, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "event fork failed");
    rc = ta3n_all_reduce_sum(c, grads + n1, live - n1, s16 ? s16 + 2 * n1 : nullptr, comm_stream);
    if (rc != TA3N_OK) return rc;
    rc = ta3n_train_step_range(p, x, params, grads, ws, n - 1, 1, stream);
    if (rc != TA3N_OK) return rc;
    if (hipEventRecord(c->join, s) != hipSuccess || hipStreamWaitEvent(cs, c->join, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "event join failed");
    rc = ta3n_all_reduce_sum(c, grads, n1, scratch_bf16, comm_stream);
    if (rc != TA3N_OK) return rc;
    if (hipEventRecord(c->fork, cs) != hipSuccess || hipStreamWaitEvent(s, c->fork, 0) != hipSuccess) return fail(TA3N_ERR_HIP, "event join failed");
    return TA3N_OK;
}

}  // extern "C"
