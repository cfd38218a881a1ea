// The v3 decode GEMV's instantiations WITHOUT the run-time flag paths (FL = false): the decode engine's plain launches.
// A translation unit of its own so that the two halves of the instantiation set compile in parallel (gemv_v3_dispatch.h).
#include "gemv_v3_dispatch.h"

namespace qeft {

hipError_t gemv_v3_dispatch_plain(const V3Args& a, int mode, size_t smem, int depth, hipStream_t st) {
    return a.bits == 3 ? launch_b<3, false>(a, mode, a.nblk, smem, depth, st) : launch_b<4, false>(a, mode, a.nblk, smem, depth, st);
}

}  // namespace qeft
