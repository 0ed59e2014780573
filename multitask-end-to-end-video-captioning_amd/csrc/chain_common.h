// chain_common.h -- what the persistent-recurrence translation units (chain.hip: forward; chain_bwd.hip: backward) share:
// the device-side hand-off helpers, and the host-side launch state that chain.hip owns (per-device CU count and fault word,
// the process-wide "one persistent grid at a time" ordering, the fault acknowledgement).
#pragma once
#include <hip/hip_runtime.h>

#include "detmath.h"
#include "internal.h"

namespace s2vt {

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;

// A-fragment load: 16 bytes per lane, sc1 (served by L2, never by this CU's L1 -- the hand-off's load form).  A builtin,
// not asm: hipcc then knows when the data lands, keeps its own vmcnt count and may place the registers anywhere (with
// asm-issued loads it copied ring registers to AGPRs right behind the load, before the data had arrived); the loads are
// kept where they are written by sched_barrier fences.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload16_sc1(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16);      // aux 16 = sc1
    return __builtin_bit_cast(f32x4, v);
}

// 16-byte write-through store (sc1: the bytes leave this XCD's L2 -- the store form of the hand-off)
__device__ __forceinline__ void bstore16_sc1(__amdgpu_buffer_rsrc_t rsrc, u32x4v v, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, 16);                    // aux 16 = sc1
}

typedef __attribute__((address_space(3))) void* lds_ptr4;

constexpr int kShards = 8;             // counter shards, one 128-byte line each; word kShards * 32 = timeout flag
constexpr unsigned kSpinLimitDefault = 1u << 21;   // polls (each >= ~0.3 us: s_sleep + an L2 round trip) before a wait gives up: ~1-2 s

// The grid-wide hand-off of the persistent kernels (MI355X_MICROARCH.md "Valid forms", row 1): every storing wave drains its
// write-through stores / atomics, the workgroup's barrier, ONE lane adds to its shard of the arrival counter; one wave per
// workgroup polls every shard with sc1 loads (bounded: a timeout raises the status words and stops further waiting), the
// other waves continue behind the workgroup barrier.  Arrival numbers count from 0 within the launch; the counters are
// zeroed by a memset node ahead of every launch.
struct GridSync {
    gu32* sync;
    unsigned* status;
    unsigned* fault;
    unsigned spin_limit;
    int nwg;                               // arrivers per step = mult x (nwg units dealt round-robin over the shards by id & 7)
    bool dead;
    unsigned mult = 1u;
    __device__ __forceinline__ void raise(int lane)
    {
        if (lane == 0) {
            __hip_atomic_store(sync + kShards * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fault) __hip_atomic_store((gu32*)fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // device-resident: later update kernels skip
            if (status) __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible
        }
        dead = true;
    }
    // one counter, `expected` arrivals (a small group of workgroups handing tiles to each other)
    __device__ __forceinline__ void arrive_one(gu32* counter, int tid)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void wait_one(gu32* counter, unsigned expected, int pwave, int lane)
    {
        if (pwave == 0 && !dead) {
            unsigned spins = 0;
            for (;;) {
                const unsigned v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= expected) break;
                if (++spins > spin_limit) { raise(lane); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ void arrive(int tid)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(sync + (blockIdx.x & (kShards - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the same for a hand-off whose arrivers are a SUBSET of the grid numbered 0 .. nwg-1 by `id` (not by blockIdx)
    __device__ __forceinline__ void arrive_as(int id, int tid)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(sync + (id & (kShards - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void wait_all(unsigned arrival, int pwave, int lane)
    {
        if (pwave == 0 && !dead) {
            const unsigned mine = lane < kShards ? mult * (unsigned)((nwg + kShards - 1 - lane) / kShards) * (arrival + 1u) : 0u;
            unsigned spins = 0;
            for (;;) {
                const unsigned v = lane < kShards ? __hip_atomic_load(sync + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                if (__all(lane >= kShards || v >= mine)) break;
                if (++spins > spin_limit) { raise(lane); break; }   // never hang: flag it and go on (results are then garbage; the fault words say so)
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
};

}  // namespace

// ---- host side, owned by chain.hip
struct ChainHost {
    int num_cus;            // 0: the persistent forms are unavailable on the current device
    int device;
    unsigned* status_dev;   // host-mapped count of timed-out waits (device pointer)
    unsigned* fault;        // device-resident fault word of the current device
    unsigned spin_limit;
};
bool chain_host(ChainHost* out);                 // false: unavailable (no device state) or switched off (S2VT_CHAIN=0 handled by the callers' own knobs; chain_ack(disable))
bool chain_persistent_disabled();                // chain_ack(disable) has switched every persistent form off, or a hold is in force
void chain_hold(bool on);                        // nestable: while held, every auto-selected recurrence takes its per-step form (same bits)
// One persistent grid at a time per process: lock, order `st` behind the previous persistent launch (other stream / device), and
// after the launch record the event the next one will wait for.
struct ChainLaunchOrder {
    ChainLaunchOrder();
    ~ChainLaunchOrder();
    hipError_t before(hipStream_t st, int dev);
    hipError_t after(hipStream_t st, int dev);
};

}  // namespace s2vt
