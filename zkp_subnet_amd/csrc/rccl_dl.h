// rccl_dl.h -- RCCL as the library's own collective back end, bound at run time.
//
// The multi-GPU step of the SRS-sharded MSM is ONE ncclAllGather of 192 bytes per rank, enqueued by the
// library on the lane's own stream (kzg_msm_sharded, comm.hip).  RCCL is resolved with dlopen at the first kzg_comm_*
// call instead of a link-time dependency:
//   * a single-GPU miner (the reference's own deployment: one prover per process, base/miner.py:73-84) never pays for
//     mapping a ~570-MB collective library at start;
//   * in a process that already holds an RCCL (torch ships its own copy with the same soname, librccl.so.1) the
//     dynamic loader hands back THAT copy -- one RCCL per process, never two with interposed symbols.
// The types come from <rccl/rccl.h> (header only); KZG_RCCL_LIB names another file.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <string>

namespace kzg_rccl {

struct Api {
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;      // what dlopen was given
    int version = 0;       // ncclGetVersion (e.g. 22707)
};

// the process-wide binding; null with *err filled in when no RCCL can be loaded or a symbol is missing
const Api* api(std::string* err);

}  // namespace kzg_rccl
