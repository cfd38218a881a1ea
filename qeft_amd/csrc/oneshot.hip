// One-shot all-reduce for the tensor-parallel decode path ("decode m = 1 -> latency-bound on xGMI, use a
// single-kernel direct peer-write / one-shot all-gather over the full mesh rather than ring").  No reference counterpart
// (qeft/utils/modelutils.py:21-57 is layer placement only).
//
// A decode token makes two all-reduces per layer of an fp32 [hidden] partial (16 KB at hidden 4096): a ring collective pays
// 2 (P - 1) dependent hops plus its launch machinery for a payload one xGMI link moves in ~0.1 us.  Here each rank owns a
// MAILBOX in its own HBM, mapped into every peer through hipIpcMemHandle, and ONE kernel per rank does the whole exchange:
//   1. every thread reads its elements of the partial and stores them as 8-byte {value, tag} granules -- one system-scope store
//      each, the data is its own flag (MI355X guide, Guideline 16 recipe R2) -- into slot `rank` of EVERY rank's mailbox (its
//      own included): P direct peer writes, all links of the mesh busy at once;
//   2. it polls the P slots of its OWN mailbox until every granule carries this call's tag, and sums them IN RANK ORDER
//      0 .. P - 1 -- the same order on every rank, so the result is bit-identical everywhere (fp32, deterministic);
//   3. the sum replaces the partial in place.
// Tags: a device-resident sequence word per rank (all ranks call in lock step, so they agree); mailboxes are double-buffered
// by the tag's parity: a rank can write call n + 2 only after it has finished call n + 1, which needed every peer's
// contribution to n + 1, which a peer sends only after it has finished reading call n -- so the slot being overwritten is dead.
// Every spin is bounded (status word, give-up code); nothing is allocated, synchronised or copied here, so the call is
// hipGraph-capturable (the sequence word lives in device memory, not in a kernel argument).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/qeft_hip.h"

namespace qeft {

constexpr int OS_MAX_WORLD = 8;
constexpr uint32_t OS_ST_TIMEOUT = 0x7200;

struct OsArgs {
    float* t;                                   // [n] fp32: partial in, sum out
    unsigned long long* box[OS_MAX_WORLD];      // every rank's mailbox (box[rank] = this rank's own), [2 parities][world][n] granules
    uint32_t* seq;                              // this rank's sequence word
    uint32_t* status;                           // [0]: first give-up code
    int n, rank, world;
    uint32_t timeout_ticks;                     // 100 MHz ticks
};

typedef __attribute__((address_space(1))) unsigned long long os_gu64;

__global__ __launch_bounds__(256) void oneshot_allreduce_kernel(OsArgs a) {
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 2;        // two elements per thread
    uint32_t seq;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seq) : "s"(a.seq) : "memory");
    const uint32_t tag = seq + 1u;
    const size_t slot_n = (size_t)a.n, par = (size_t)(tag & 1u) * a.world * slot_n;
    if (i0 < a.n) {
        const float v0 = a.t[i0], v1 = i0 + 1 < a.n ? a.t[i0 + 1] : 0.f;
        const unsigned long long g0 = ((unsigned long long)tag << 32) | __builtin_bit_cast(uint32_t, v0);
        const unsigned long long g1 = ((unsigned long long)tag << 32) | __builtin_bit_cast(uint32_t, v1);
        // 1. my partial into slot `rank` of every mailbox (peers first: their links carry the latency)
        for (int d = 1; d <= a.world; ++d) {
            const int p = (a.rank + d) % a.world;
            os_gu64* dst = (os_gu64*)a.box[p] + par + (size_t)a.rank * slot_n + i0;
            __hip_atomic_store(dst, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (i0 + 1 < a.n) __hip_atomic_store(dst + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // 2. the P slots of my own mailbox, in rank order
        const os_gu64* mine = (const os_gu64*)a.box[a.rank] + par + i0;
        float s0 = 0.f, s1 = 0.f;
        const long long t0 = wall_clock64();
        // a wait that gave up in an EARLIER call leaves its code in status[0]: this group is out of step for good, so later calls do
        // not wait at all (they would each sit out the whole timeout) -- the caller reads the status word and falls back
        bool dead = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        for (int s = 0; s < a.world; ++s) {
            unsigned long long q0, q1;
            for (unsigned spins = 0;; ++spins) {
                q0 = __hip_atomic_load(mine + (size_t)s * slot_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                q1 = i0 + 1 < a.n ? __hip_atomic_load(mine + (size_t)s * slot_n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : q0;
                if (((uint32_t)(q0 >> 32) == tag && (uint32_t)(q1 >> 32) == tag) || dead) break;
                if ((spins & 63u) == 63u && wall_clock64() - t0 > (long long)a.timeout_ticks) {
                    dead = true;
                    atomicCAS(a.status, 0u, OS_ST_TIMEOUT | (uint32_t)s);
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const uint32_t u0 = (uint32_t)q0, u1 = (uint32_t)q1;
            s0 += __builtin_bit_cast(float, u0);
            s1 += __builtin_bit_cast(float, u1);
        }
        // 3. in place
        a.t[i0] = s0;
        if (i0 + 1 < a.n) a.t[i0 + 1] = s1;
    }
    // the sequence word: the last block to finish bumps it (every block has read it at entry; a counter in the status block)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(a.status + 1, 1u);
        if (done == gridDim.x - 1) {
            a.status[1] = 0;
            __hip_atomic_store(a.seq, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace qeft

extern "C" {

long long qeft_oneshot_mailbox_bytes(int world, int n) {
    if (world < 1 || world > qeft::OS_MAX_WORLD || n < 1) return 0;
    return (long long)2 * world * n * 8;
}
int qeft_oneshot_max_world(void) { return qeft::OS_MAX_WORLD; }

/* Set-up helpers (the only entry points of the library that allocate / synchronise): the mailbox must be a hipMalloc block of
 * its own -- an IPC handle exports the whole allocation a pointer lives in, and a pointer carved out of a caching allocator's
 * segment would arrive in the peer at an offset nobody knows. */
int qeft_oneshot_mailbox_alloc(int world, int n, void** dev_ptr_out) {
    if (!dev_ptr_out) return QEFT_ERR_NULL;
    const long long bytes = qeft_oneshot_mailbox_bytes(world, n);
    if (bytes <= 0) return QEFT_ERR_SHAPE;
    // FINE-GRAINED device memory (what RCCL uses for its peer-visible buffers): remote stores over xGMI and the owner's polling
    // loads stay coherent without a kernel boundary; coarse-grained memory is only guaranteed coherent at kernel boundaries
    if (hipExtMallocWithFlags(dev_ptr_out, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(dev_ptr_out, (size_t)bytes) != hipSuccess) return QEFT_ERR_LAUNCH;
    }
    if (hipMemset(*dev_ptr_out, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return QEFT_ERR_LAUNCH;
    return QEFT_OK;
}
int qeft_oneshot_mailbox_free(void* dev_ptr) {
    if (!dev_ptr) return QEFT_ERR_NULL;
    return hipFree(dev_ptr) == hipSuccess ? QEFT_OK : QEFT_ERR_LAUNCH;
}

int qeft_oneshot_ipc_export(void* dev_ptr, void* handle_out) {
    if (!dev_ptr || !handle_out) return QEFT_ERR_NULL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 opaque bytes");
    return hipIpcGetMemHandle((hipIpcMemHandle_t*)handle_out, dev_ptr) == hipSuccess ? QEFT_OK : QEFT_ERR_LAUNCH;
}
int qeft_oneshot_ipc_open(const void* handle, void** dev_ptr_out) {
    if (!handle || !dev_ptr_out) return QEFT_ERR_NULL;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle, sizeof h);
    return hipIpcOpenMemHandle(dev_ptr_out, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? QEFT_OK : QEFT_ERR_LAUNCH;
}
int qeft_oneshot_ipc_close(void* dev_ptr) {
    if (!dev_ptr) return QEFT_ERR_NULL;
    return hipIpcCloseMemHandle(dev_ptr) == hipSuccess ? QEFT_OK : QEFT_ERR_LAUNCH;
}

/* In-place sum of t[n] (fp32) over `world` ranks.  boxes: HOST array of `world` device pointers -- box[r] = rank r's mailbox as
 * mapped into THIS process (box[rank] = the local allocation), each qeft_oneshot_mailbox_bytes(world, n) bytes, zeroed once;
 * seq: one zeroed uint32 in device memory; status: two zeroed uint32 (status[0] != 0 afterwards: a wait gave up). */
int qeft_oneshot_allreduce_f32(void* t, int n, void* const* boxes, int rank, int world, void* seq, void* status,
                               qeft_stream_t stream) {
    if (!t || !boxes || !seq || !status) return QEFT_ERR_NULL;
    if (world < 1 || world > qeft::OS_MAX_WORLD || rank < 0 || rank >= world || n < 1) return QEFT_ERR_SHAPE;
    qeft::OsArgs a{};
    a.t = (float*)t;
    for (int r = 0; r < world; ++r) {
        if (!boxes[r] || (reinterpret_cast<uintptr_t>(boxes[r]) & 7u)) return QEFT_ERR_ALIGN;
        a.box[r] = (unsigned long long*)boxes[r];
    }
    a.seq = (uint32_t*)seq;
    a.status = (uint32_t*)status;
    a.n = n; a.rank = rank; a.world = world;
    a.timeout_ticks = 100u * 1000u * 2000u;     // 2 s: a peer that has not even been launched yet is not an error
    const int blocks = (n + 511) / 512;
    hipLaunchKernelGGL(qeft::oneshot_allreduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QEFT_OK : QEFT_ERR_LAUNCH;
}

}  // extern "C"
