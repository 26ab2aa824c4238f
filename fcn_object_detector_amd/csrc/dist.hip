// Data-parallel exchange step: RCCL all-reduce of the flat gradient buffer over xGMI.
//
// The reference trains on one GPU (`--gpu=0`, train/train.sh:26); Caffe's multi-GPU semantics for `--gpu=a,b,...` are
// "every GPU runs the prototxt batch, gradients are summed and scaled by 1/G" — this file provides the sum.  One process
// per GPU; the communicator is created from an ncclUniqueId that rank 0 hands to the other ranks over the Python
// control plane (fcn_object_detector_amd/dp.py).  The collective is enqueued on the caller's stream, so the engine can
// run it on a side stream and overlap it with the rest of the backward pass.
#include <rccl/rccl.h>

#include "common.h"

using namespace fcn;

#define FCN_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r__ = (call);                                                                  \
        if (r__ != ncclSuccess) return ::fcn::set_err(FCN_E_STATE, "%s failed: %s", #call, ncclGetErrorString(r__)); \
    } while (0)

extern "C" {

int fcn_comm_unique_id(char* h_id128) {
    FCN_REQUIRE(h_id128, FCN_E_ARG, "fcn_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    FCN_NCCL(ncclGetUniqueId(&id));
    memcpy(h_id128, &id, sizeof(id));
    return 0;
}

int fcn_comm_init(fcn_comm_t* comm, const char* h_id128, int world, int rank) {
    FCN_REQUIRE(comm && h_id128 && world > 0 && rank >= 0 && rank < world, FCN_E_ARG, "fcn_comm_init: bad args");
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    ncclComm_t c = nullptr;
    FCN_NCCL(ncclCommInitRank(&c, world, id, rank));
    *comm = c;
    return 0;
}

int fcn_comm_allreduce_sum_f32(fcn_comm_t comm, float* buf, size_t count, fcn_stream_t s) {
    FCN_REQUIRE(comm && buf, FCN_E_ARG, "fcn_comm_allreduce_sum_f32: null");
    if (!count) return 0;
    FCN_NCCL(ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)comm, as_stream(s)));
    return 0;
}

int fcn_comm_destroy(fcn_comm_t comm) {
    if (comm) FCN_NCCL(ncclCommDestroy((ncclComm_t)comm));
    return 0;
}

}  // extern "C"
