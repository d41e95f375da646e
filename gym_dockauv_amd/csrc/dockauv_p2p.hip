// dockauv_p2p.hip -- peer-to-peer gather of the packed [obs | reward | done] rows over xGMI (include/dockauv.h,
// "multi-GPU" block).  One process per GPU; every rank owns a gather buffer [world][n_local][row] that its peers
// map through HIP IPC.  After its step kernel a rank PUSHES its slice into the same slice of every peer's buffer
// (one copy kernel, 16-byte stores, blockIdx.y = destination) and then raises a step stamp in every peer's flag
// array; the peers' wait kernel spins -- bounded -- on those stamps.  No collective library call sits in the loop:
// at 4 096 envs per GPU the slice is 330 KB and the step is 6 us, an all-gather's launch latency alone is several
// times that.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/dockauv.h"

namespace dockauv {
extern thread_local std::string g_create_error;   // dockauv_last_error(NULL)
}

namespace {

int p2p_fail(int code, const char* what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed: %s", what, hipGetErrorString(e));
    dockauv::g_create_error = buf;
    return code;
}
int p2p_invalid(const char* msg) {
    dockauv::g_create_error = msg;
    return DOCKAUV_E_INVALID;
}
#define P2P_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return p2p_fail(DOCKAUV_E_HIP, #expr, e_); \
    } while (0)

static_assert(sizeof(hipIpcMemHandle_t) == DOCKAUV_P2P_HANDLE_BYTES, "IPC handle size");

struct PushArgs {
    const uint4* src;
    unsigned long n16;      // whole 16-byte chunks
    unsigned long bytes;    // total (tail bytes copied by the first lanes of block 0)
    uint4* dst[DOCKAUV_P2P_MAX_PEERS + 1];   // own slice included
};

// Rows are copied with SYSTEM-SCOPE (write-through, `sc0 sc1`) 16-byte stores: such a store is acknowledged only once
// it has left this GPU's caches, so "every store of the block has been acknowledged" (s_waitcnt vmcnt(0)) is all a block
// needs before it reports itself done.  A system-scope release FENCE per block instead costs an L2 write-back walk per
// block, and those serialise: measured 13 us for the 152 blocks of a 622 KB slice, 119 us for 1 216 blocks (5 MB).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_sys(u32x4* p, u32x4 v) {
    // (a volatile store gives the same instruction but the compiler then waits for each acknowledgement in turn)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void copy_rows(const PushArgs& a, int d) {
    u32x4* dst = reinterpret_cast<u32x4*>(a.dst[d]);
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(a.src);
    const unsigned long stride = (unsigned long)gridDim.x * 256;
    unsigned long i = (unsigned long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < a.n16; i += 4 * stride) {
        const u32x4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        store_sys(dst + i, v0);
        store_sys(dst + i + stride, v1);
        store_sys(dst + i + 2 * stride, v2);
        store_sys(dst + i + 3 * stride, v3);
    }
    for (; i < a.n16; i += stride) store_sys(dst + i, src[i]);
    if (blockIdx.x == 0) {
        const unsigned long tail0 = a.n16 * 16;
        const unsigned char* s = reinterpret_cast<const unsigned char*>(a.src);
        volatile unsigned char* t = reinterpret_cast<volatile unsigned char*>(a.dst[d]);
        for (unsigned long k = tail0 + threadIdx.x; k < a.bytes; k += 256) t[k] = s[k];
    }
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0): all of this wave's stores acknowledged
}

// grid (x = chunk blocks, y = destination); the stamp is raised by the NEXT kernel on the stream.
__global__ __launch_bounds__(256) void push_rows_kernel(PushArgs a) { copy_rows(a, blockIdx.y); }

struct SignalArgs {
    uint32_t* slot[DOCKAUV_P2P_MAX_PEERS];   // &flags_of_peer[my_rank]
    const uint32_t* my_flags;                // [world], written by the peers
    uint32_t* status;                        // [2]: sticky time-out bit, stamp it happened at
    unsigned long max_spins;
    uint32_t stamp;        // raised at the peers (0 = raise nothing)
    uint32_t wait_stamp;   // waited for here (0 = wait for nothing)
    int n_peers, world, my_rank;
};

// One wave.  Lane p < n_peers raises `stamp` at peer p; lane r < world (r != my_rank) then waits until rank r's
// stamp has reached `wait_stamp` here (wait_stamp = stamp: closed loop; wait_stamp = stamp - 1: the gather of step t
// overlaps the kernel of step t + 1).  The spin is bounded and a time-out is sticky: once `status[0]` is set every later wait
// returns at once, so a dead peer costs one time-out, not one per step, and the grid always drains.
__global__ __launch_bounds__(64) void signal_wait_kernel(SignalArgs a) {
    const int t = threadIdx.x;
    __threadfence_system();
    if (a.stamp != 0 && t < a.n_peers) __hip_atomic_store(a.slot[t], a.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.wait_stamp != 0 && t < a.world && t != a.my_rank) {
        const bool dead = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        unsigned long k = dead ? a.max_spins : 0;
        bool seen = false;
        for (; k < a.max_spins; ++k) {
            const uint32_t v = __hip_atomic_load(a.my_flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)(v - a.wait_stamp) >= 0) { seen = true; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!seen && !dead) {
            __hip_atomic_fetch_or(a.status, 1u << (t & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.status[1] = a.wait_stamp;
        }
    }
    __threadfence_system();
}

// Fused gather: copy blocks as in push_rows_kernel; each block counts itself in once its stores are acknowledged; the
// block that counts in last has, by then, every other block's rows behind it, so its wave 0 may raise the stamp -- and
// then waits for the peers' (bounded, as above).
struct GatherArgs {
    PushArgs push;
    SignalArgs sig;
    uint32_t* counter;
};

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
    copy_rows(a.push, blockIdx.y);
    __shared__ uint32_t last;
    __syncthreads();            // every wave of the block has its stores acknowledged
    if (threadIdx.x == 0) {
        const uint32_t total = gridDim.x * gridDim.y;
        const uint32_t seen = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (seen + 1 == total) ? 1u : 0u;
        if (last) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    const SignalArgs& g = a.sig;
    const int t = threadIdx.x;
    __threadfence_system();
    if (g.stamp != 0 && t < g.n_peers) __hip_atomic_store(g.slot[t], g.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (g.wait_stamp != 0 && t < g.world && t != g.my_rank) {
        const bool dead = __hip_atomic_load(g.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        unsigned long k = dead ? g.max_spins : 0;
        bool seen = false;
        for (; k < g.max_spins; ++k) {
            const uint32_t v = __hip_atomic_load(g.my_flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)(v - g.wait_stamp) >= 0) { seen = true; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!seen && !dead) {
            __hip_atomic_fetch_or(g.status, 1u << (t & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g.status[1] = g.wait_stamp;
        }
    }
    __threadfence_system();
}

}  // namespace

namespace dockauv {
// also called by dockauv_step_gather_sequence (dockauv_capi.hip); returns a DOCKAUV_* code, message in g_create_error
int launch_gather(const dockauv_p2p_plan* pl, const void* src, uint32_t stamp, uint32_t wait_stamp, hipStream_t stream) {
    if (!pl || !src) return p2p_invalid("dockauv_p2p_gather: null argument");
    if (pl->n_dsts < 1 || pl->n_dsts > DOCKAUV_P2P_MAX_PEERS + 1 || pl->n_peers < 0 || pl->n_peers > DOCKAUV_P2P_MAX_PEERS ||
        pl->world < 1 || pl->world > DOCKAUV_P2P_MAX_PEERS + 1 || pl->my_rank < 0 || pl->my_rank >= pl->world ||
        !pl->my_flags || !pl->status || !pl->counter || pl->bytes == 0)
        return p2p_invalid("dockauv_p2p_gather: bad plan");
    if ((reinterpret_cast<uintptr_t>(src) & 15) != 0) return p2p_invalid("dockauv_p2p_gather: src must be 16-byte aligned");
    GatherArgs a{};
    a.push.src = static_cast<const uint4*>(src);
    a.push.n16 = pl->bytes / 16;
    a.push.bytes = pl->bytes;
    for (int d = 0; d < pl->n_dsts; ++d) {
        if (!pl->dsts[d] || (reinterpret_cast<uintptr_t>(pl->dsts[d]) & 15) != 0)
            return p2p_invalid("dockauv_p2p_gather: destinations must be non-null and 16-byte aligned");
        a.push.dst[d] = static_cast<uint4*>(pl->dsts[d]);
    }
    for (int p = 0; p < pl->n_peers; ++p) {
        if (!pl->peer_slots[p]) return p2p_invalid("dockauv_p2p_gather: null peer slot");
        a.sig.slot[p] = pl->peer_slots[p];
    }
    a.sig.my_flags = pl->my_flags;
    a.sig.status = pl->status;
    a.sig.max_spins = pl->max_spins;
    a.sig.stamp = stamp;
    a.sig.wait_stamp = wait_stamp;
    a.sig.n_peers = pl->n_peers;
    a.sig.world = pl->world;
    a.sig.my_rank = pl->my_rank;
    a.counter = pl->counter;
    unsigned long bx = (a.push.n16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)bx, (unsigned)pl->n_dsts), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return p2p_fail(DOCKAUV_E_HIP, "gather_kernel launch", e);
    return DOCKAUV_OK;
}
}  // namespace dockauv

extern "C" {

int dockauv_p2p_gather(const dockauv_p2p_plan* plan, const void* src, uint32_t stamp, uint32_t wait_stamp,
                       void* hip_stream) {
    return dockauv::launch_gather(plan, src, stamp, wait_stamp, static_cast<hipStream_t>(hip_stream));
}

int dockauv_p2p_alloc(int device, size_t bytes, int uncached, void** dev_ptr, unsigned char* handle) {
    if (!dev_ptr || bytes == 0) return p2p_invalid("dockauv_p2p_alloc: null pointer or zero size");
    P2P_TRY(hipSetDevice(device));
    void* p = nullptr;
    if (uncached) {
        P2P_TRY(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
    } else {
        P2P_TRY(hipMalloc(&p, bytes));
    }
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && handle) {
        hipIpcMemHandle_t h;
        e = hipIpcGetMemHandle(&h, p);
        if (e == hipSuccess) memcpy(handle, &h, sizeof h);
    }
    if (e != hipSuccess) {
        (void)hipFree(p);
        return p2p_fail(DOCKAUV_E_HIP, "dockauv_p2p_alloc (memset / hipIpcGetMemHandle)", e);
    }
    *dev_ptr = p;
    return DOCKAUV_OK;
}

int dockauv_p2p_free(void* dev_ptr) {
    if (dev_ptr) P2P_TRY(hipFree(dev_ptr));
    return DOCKAUV_OK;
}

int dockauv_p2p_open(int device, const unsigned char* handle, void** dev_ptr) {
    if (!handle || !dev_ptr) return p2p_invalid("dockauv_p2p_open: null argument");
    P2P_TRY(hipSetDevice(device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    void* p = nullptr;
    P2P_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *dev_ptr = p;
    return DOCKAUV_OK;
}

int dockauv_p2p_close(void* dev_ptr) {
    if (dev_ptr) P2P_TRY(hipIpcCloseMemHandle(dev_ptr));
    return DOCKAUV_OK;
}

int dockauv_p2p_push(const void* src, size_t bytes, void* const* dsts, int n_dsts, void* hip_stream) {
    if (n_dsts == 0 || bytes == 0) return DOCKAUV_OK;
    if (!src || !dsts || n_dsts < 0 || n_dsts > DOCKAUV_P2P_MAX_PEERS + 1) return p2p_invalid("dockauv_p2p_push: bad argument");
    if ((reinterpret_cast<uintptr_t>(src) & 15) != 0) return p2p_invalid("dockauv_p2p_push: src must be 16-byte aligned");
    PushArgs a{};
    a.src = static_cast<const uint4*>(src);
    a.n16 = bytes / 16;
    a.bytes = bytes;
    for (int d = 0; d < n_dsts; ++d) {
        if (!dsts[d] || (reinterpret_cast<uintptr_t>(dsts[d]) & 15) != 0) return p2p_invalid("dockauv_p2p_push: destinations must be non-null and 16-byte aligned");
        a.dst[d] = static_cast<uint4*>(dsts[d]);
    }
    unsigned long bx = (a.n16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(push_rows_kernel, dim3((unsigned)bx, (unsigned)n_dsts), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    P2P_TRY(hipGetLastError());
    return DOCKAUV_OK;
}

int dockauv_p2p_signal_wait(uint32_t* const* peer_slots, int n_peers, const uint32_t* my_flags, int world, int my_rank,
                            uint32_t stamp, uint32_t wait_stamp, uint64_t max_spins, uint32_t* status, void* hip_stream) {
    if (n_peers < 0 || n_peers > DOCKAUV_P2P_MAX_PEERS || world < 1 || world > DOCKAUV_P2P_MAX_PEERS + 1 ||
        my_rank < 0 || my_rank >= world || !my_flags || !status || (n_peers > 0 && !peer_slots))
        return p2p_invalid("dockauv_p2p_signal_wait: bad argument");
    SignalArgs a{};
    for (int p = 0; p < n_peers; ++p) {
        if (!peer_slots[p]) return p2p_invalid("dockauv_p2p_signal_wait: null peer slot");
        a.slot[p] = peer_slots[p];
    }
    a.my_flags = my_flags;
    a.status = status;
    a.max_spins = max_spins;
    a.stamp = stamp;
    a.wait_stamp = wait_stamp;
    a.n_peers = n_peers;
    a.world = world;
    a.my_rank = my_rank;
    hipLaunchKernelGGL(signal_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(hip_stream), a);
    P2P_TRY(hipGetLastError());
    return DOCKAUV_OK;
}

}  // extern "C"
