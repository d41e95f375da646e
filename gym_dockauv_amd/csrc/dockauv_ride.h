// dockauv_ride.h -- copy groups that ride in a step kernel's grid (include/dockauv.h, dockauv_step_gather_sequence
// with lag 1): the workgroups behind the step groups push the PREVIOUS step's packed rows to every rank's gather
// buffer while the step groups integrate, so the fabric transfer of step t runs beside the arithmetic of step t + 1
// inside one launch -- no second stream, no extra kernel.  Same protocol as gather_kernel (dockauv_p2p.hip):
// system-scope write-through stores, a device counter, the group that counts in last raises the stamp at the peers
// and waits -- bounded -- for theirs.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dockauv.h"

namespace dockauv {

typedef unsigned ride_u32x4 __attribute__((ext_vector_type(4)));

// kernarg part of a ride (24 bytes); the plan itself lives in DEVICE memory (uploaded once per handle)
struct RideArgs {
    const dockauv_p2p_plan* plan;   // device copy
    const void* src;                // previous step's rows (plan->bytes, a multiple of 16)
    uint32_t stamp, wait_stamp;
};

__device__ __forceinline__ void ride_store_sys(ride_u32x4* p, ride_u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// copy group g of n_groups, NT threads; `lds_flag`: one uint32 of this group's LDS
template <int NT>
__device__ __forceinline__ void ride_body(const RideArgs& r, unsigned g, unsigned n_groups, uint32_t* lds_flag) {
    const dockauv_p2p_plan* __restrict__ pl = r.plan;
    const unsigned long n16 = pl->bytes / 16;
    const int nd = pl->n_dsts;
    const ride_u32x4* __restrict__ src = static_cast<const ride_u32x4*>(r.src);
    const unsigned long stride = (unsigned long)n_groups * NT;
    for (unsigned long i = (unsigned long)g * NT + threadIdx.x; i < n16; i += stride) {
        const ride_u32x4 v = src[i];
        for (int d = 0; d < nd; ++d) ride_store_sys(static_cast<ride_u32x4*>(pl->dsts[d]) + i, v);
    }
    __builtin_amdgcn_s_waitcnt(0);   // every store of this wave acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t seen = __hip_atomic_fetch_add(pl->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (seen + 1 == n_groups) ? 1u : 0u;
        if (last) __hip_atomic_store(pl->counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = last;
    }
    __syncthreads();
    if (*lds_flag == 0 || threadIdx.x >= 64) return;
    const int t = threadIdx.x;
    __threadfence_system();
    if (r.stamp != 0 && t < pl->n_peers)
        __hip_atomic_store(pl->peer_slots[t], r.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (r.wait_stamp != 0 && t < pl->world && t != pl->my_rank) {
        const bool dead = __hip_atomic_load(pl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const unsigned long max_spins = pl->max_spins;
        unsigned long k = dead ? max_spins : 0;
        bool seen = false;
        for (; k < max_spins; ++k) {
            const uint32_t v = __hip_atomic_load(pl->my_flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)(v - r.wait_stamp) >= 0) { seen = true; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!seen && !dead) {
            __hip_atomic_fetch_or(pl->status, 1u << (t & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pl->status[1] = r.wait_stamp;
        }
    }
    __threadfence_system();
}

}  // namespace dockauv
