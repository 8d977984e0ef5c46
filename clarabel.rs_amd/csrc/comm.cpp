// comm.cpp -- the exchange step of the sharded path: native RCCL over xGMI, called
// from C++ on HIP streams; no PyTorch anywhere in the product.
//
// The KKT path shards across connected components of the elimination forest (BASELINE config 4:
// 1024 independent SOCPs, 128 per GPU): factorisation, substitutions and refinement of a rank's
// blocks need no exchange.  What couples the blocks in the interior-point loop is
//   * the step direction, which every rank (and the host driver) must see in full:
//     ONE all-gather of N fp64 values per KKT solve (chip_kkt_allgather_step), and
//   * a handful of scalars (||e||inf, the dot products of kktsystem.rs:175-186, the minimum step
//     length of compositecone.rs:300-340): chip_comm_allreduce.
// The all-gather runs on the communicator's own stream, ordered behind the solve by an event on the
// engine's stream -- never by a host synchronisation -- so it overlaps the next solve; the engine's
// stream is made to wait for it (chip_kkt_wait_comm) only before the gathered buffer or the send
// buffer is reused.
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "engine.hpp"

using namespace chip;

struct chip_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_solve = nullptr, ev_done = nullptr;
    int world = 1, rank = 0, device = 0;
    double *scal_dev = nullptr;   // small device buffer of the scalar reductions
    double *scal_host = nullptr;  // pinned
};

namespace {
constexpr int SCAL_MAX = 64;
int fail_nccl(ncclResult_t r, const char *what) {
    set_error(std::string(what) + ": " + ncclGetErrorString(r));
    return CHIP_ERR_HIP;
}
#define CHIP_NCCL(expr)                                   \
    do {                                                  \
        ncclResult_t _r = (expr);                         \
        if (_r != ncclSuccess) return fail_nccl(_r, #expr); \
    } while (0)
} // namespace

namespace chip {
hipStream_t kkt_stream(::chip_kkt *h);      // capi.cpp
void kkt_set_world(::chip_kkt *h, int world);
int kkt_note_exchange(::chip_kkt *h, hipStream_t comm_stream);
} // namespace chip

extern "C" {

int32_t chip_comm_get_unique_id(uint8_t id[CHIP_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= CHIP_COMM_ID_BYTES, "ncclUniqueId does not fit the ABI's id buffer");
    if (!id) return CHIP_ERR_ARG;
    ncclUniqueId u;
    CHIP_NCCL(ncclGetUniqueId(&u));
    std::memset(id, 0, CHIP_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return CHIP_OK;
}

int32_t chip_comm_create(chip_comm **out, const uint8_t id[CHIP_COMM_ID_BYTES], int32_t world, int32_t rank,
                         int32_t device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return CHIP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (the product has no CPU fallback)");
        return CHIP_ERR_NO_DEVICE;
    }
    chip_comm *c = new chip_comm();
    c->world = world;
    c->rank = rank;
    if (device < 0) (void)hipGetDevice(&device);
    c->device = device;
    auto bail = [&](int rc) {
        chip_comm_destroy(c);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(CHIP_ERR_HIP);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) return bail(fail_nccl(r, "ncclCommInitRank"));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(CHIP_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_solve, hipEventDisableTiming) != hipSuccess) return bail(CHIP_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) return bail(CHIP_ERR_HIP);
    if (hipMalloc((void **)&c->scal_dev, SCAL_MAX * sizeof(double)) != hipSuccess) return bail(CHIP_ERR_HIP);
    if (hipHostMalloc((void **)&c->scal_host, SCAL_MAX * sizeof(double), hipHostMallocDefault) != hipSuccess)
        return bail(CHIP_ERR_HIP);
    *out = c;
    return CHIP_OK;
}

void chip_comm_destroy(chip_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ev_solve) (void)hipEventDestroy(c->ev_solve);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->scal_dev) (void)hipFree(c->scal_dev);
    if (c->scal_host) (void)hipHostFree(c->scal_host);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t chip_comm_info(const chip_comm *c, int32_t *world, int32_t *rank) {
    if (!c) return CHIP_ERR_ARG;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return CHIP_OK;
}

int32_t chip_kkt_attach_comm(chip_kkt *h, chip_comm *c) {
    if (!h || !c) return CHIP_ERR_ARG;
    chip::kkt_set_world(h, c->world);
    return CHIP_OK;
}

// recv[offset_r .. offset_r + counts[r]) <- rank r's send[0 .. counts[r]); equal counts take the ring
// all-gather, ragged ones a group of broadcasts (the usual all-gather-v over NCCL primitives)
int32_t chip_kkt_allgather_step(chip_kkt *h, chip_comm *c, const double *send_dev, double *recv_dev,
                                const int64_t *counts) {
    if (!h || !c || !send_dev || !recv_dev || !counts) return CHIP_ERR_ARG;
    hipStream_t ks = chip::kkt_stream(h);
    if (!ks) return CHIP_ERR_NO_DEVICE;
    CHIP_HIP(hipSetDevice(c->device));
    // the collective starts when the solve that produced send_dev has finished ON THE DEVICE
    CHIP_HIP(hipEventRecord(c->ev_solve, ks));
    CHIP_HIP(hipStreamWaitEvent(c->stream, c->ev_solve, 0));
    bool equal = true;
    for (int r = 1; r < c->world; r++) equal = equal && counts[r] == counts[0];
    if (equal) {
        CHIP_NCCL(ncclAllGather(send_dev, recv_dev, (size_t)counts[0], ncclDouble, c->comm, c->stream));
    } else {
        int64_t off = 0;
        CHIP_NCCL(ncclGroupStart());
        for (int r = 0; r < c->world; r++) {
            CHIP_NCCL(ncclBroadcast(r == c->rank ? (const void *)send_dev : (const void *)(recv_dev + off),
                                    recv_dev + off, (size_t)counts[r], ncclDouble, r, c->comm, c->stream));
            off += counts[r];
        }
        CHIP_NCCL(ncclGroupEnd());
    }
    CHIP_HIP(hipEventRecord(c->ev_done, c->stream));
    // the handle's next PERSISTENT solve launch waits for the exchange on the device (capi.cpp: chip_kkt::exch_event); the
    // cone update and the factorisation enqueued before it run beside the collective's kernels
    return chip::kkt_note_exchange(h, c->stream);
}

#ifdef CHIP_TESTING
// test hook (include/clarabel_hip_testing.h): a kernel that only holds `blocks` workgroups for `usec` microseconds ON THE
// COMMUNICATOR'S STREAM, behind the collective enqueued last -- a stand-in, on one GPU, for the time RCCL's ring kernel
// occupies CUs when eight ranks exchange over xGMI; the completion event moves behind it
int32_t chip_comm_debug_spin(chip_comm *c, int32_t blocks, int32_t threads, double usec) {
    if (!c || blocks <= 0 || threads <= 0 || threads > 1024) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(c->device));
    chip::dev::debug_spin(c->stream, blocks, threads, 0, usec);
    CHIP_HIP(hipEventRecord(c->ev_done, c->stream));
    return CHIP_OK;
}
#endif

int32_t chip_kkt_wait_comm(chip_kkt *h, chip_comm *c) {
    if (!h || !c) return CHIP_ERR_ARG;
    hipStream_t ks = chip::kkt_stream(h);
    if (!ks) return CHIP_ERR_NO_DEVICE;
    CHIP_HIP(hipStreamWaitEvent(ks, c->ev_done, 0));
    return CHIP_OK;
}

int32_t chip_comm_synchronize(chip_comm *c) {
    if (!c) return CHIP_ERR_ARG;
    CHIP_HIP(hipSetDevice(c->device));
    CHIP_HIP(hipStreamSynchronize(c->stream));
    return CHIP_OK;
}

int32_t chip_comm_allreduce(chip_comm *c, double *vals, int32_t count, int32_t op) {
    if (!c || !vals || count < 0 || count > SCAL_MAX || op < 0 || op > 2) return CHIP_ERR_ARG;
    if (count == 0) return CHIP_OK;
    CHIP_HIP(hipSetDevice(c->device));
    std::memcpy(c->scal_host, vals, (size_t)count * sizeof(double));
    CHIP_HIP(hipMemcpyAsync(c->scal_dev, c->scal_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
    CHIP_NCCL(ncclAllReduce(c->scal_dev, c->scal_dev, (size_t)count, ncclDouble, ops[op], c->comm, c->stream));
    CHIP_HIP(hipMemcpyAsync(c->scal_host, c->scal_dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CHIP_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(vals, c->scal_host, (size_t)count * sizeof(double));
    return CHIP_OK;
}

} // extern "C"
