// RCCL, bound at run time (dlopen): the library itself has no link-time dependency on librccl, and a process that already
// holds a copy (PyTorch bundles one) shares it.  Only what the one exchange step of the path needs: the sum of the partial
// output blocks of an input-split matrix (NToMonoConvolve.cpp:39-42 across GPUs) as ncclAllReduce(ncclFloat, ncclSum) on the
// engine's own stream.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace hcv
{
    constexpr size_t kRcclIdBytes = 128;                // sizeof(ncclUniqueId)

    struct RcclComm;                                    // one communicator (one rank of one row group)

    bool rccl_available(std::string *err);
    bool rccl_unique_id(void *out128, std::string *err);
    RcclComm *rccl_comm_create(const void *id128, int rank, int nranks, int device, std::string *err);
    void rccl_comm_destroy(RcclComm *c);
    int rccl_comm_size(const RcclComm *c);
    // in place over `rows` rows of `n` floats, `stride` floats apart (one call when the block is contiguous, a group otherwise)
    bool rccl_all_reduce_sum(RcclComm *c, float *buf, size_t rows, size_t n, size_t stride, hipStream_t stream, std::string *err);
}
