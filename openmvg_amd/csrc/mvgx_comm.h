// RCCL binding used by the BA solver (see mvgx_comm.hip).
#pragma once
#include "mvgx_common.h"

namespace mvgx {
struct RcclComm;
int rccl_unique_id(void* out128);
int rccl_init(RcclComm** out, int world, int rank, const void* unique_id);
void rccl_destroy(RcclComm* c);
void rccl_abort(RcclComm* c);   // from any thread: fails the collectives in flight of this communicator (a peer rank failed)
int rccl_allreduce_f64(RcclComm* c, double* device_buffer, uint64_t count, int op, hipStream_t stream);
int rccl_self_check(RcclComm* c, hipStream_t stream);   // known-answer all-reduce (sum, max): guards the restated rccl.h codes
}  // namespace mvgx
