// Primitives of the small persistent cooperative kernels (BiLSTM, GRU): write-through stores / loads that bypass
// the non-coherent per-XCD L2, and a bounded group barrier (arrival counter + relaxed polling).  MI355X guide,
// Guideline 16 "R1" form.  Every spin is bounded: on a timeout the error word is set and all workgroups leave.
#pragma once
#include "t2v_common.h"

#define BL_SPIN_LIMIT 4000000

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all workgroups of one direction arrive; returns false on timeout (error word set)
__device__ __forceinline__ bool group_barrier(unsigned* counter, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores are out
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > BL_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

