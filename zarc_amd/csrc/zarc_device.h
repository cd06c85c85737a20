// zarc_amd/csrc/zarc_device.h -- device-side primitives shared by the gfx950 kernels.
//
// wave64 throughout (CDNA4): ballots are 64-bit, shuffles span 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zd {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// Make LDS / global writes of this wave's lanes visible to the other lanes of the same wave.
// Hardware executes a wave's memory instructions in order, so only the compiler has to be fenced;
// the test emulator runs lanes one after another and needs a real rendezvous here.
__device__ __forceinline__ void wave_sync()
{
#ifdef ZARC_HIPEMU
    hipemu_wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Ordering point between LDS instructions of ONE wave (e.g. table lookups that must precede the same wave's
// inserts).  The LDS executes a wave's instructions in program order, so on hardware this only stops the
// compiler from reordering; it costs no wait.  The emulator needs a rendezvous.
__device__ __forceinline__ void wave_lds_order()
{
#ifdef ZARC_HIPEMU
    hipemu_wave_sync();
#else
    asm volatile("" ::: "memory");
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains this wave's outstanding
// global stores (s_waitcnt vmcnt(0)) -- a ~1-2 us stall per tile in kernels that stream results to HBM which
// nobody in the workgroup reads back.  Use this one between LDS-only phases.
__device__ __forceinline__ void lds_barrier()
{
#ifdef ZARC_HIPEMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Issue priority of this wave (s_setprio, 0..3).  A serial section that the rest of the workgroup waits for (a few waves
// working while the others sit at a barrier) competes for issue slots with the co-resident workgroup: raise it there.
template <int PRIO> __device__ __forceinline__ void wave_priority()
{
#ifndef ZARC_HIPEMU
    __builtin_amdgcn_s_setprio(PRIO);
#endif
}

// Same for global-memory hand-offs between lanes of one wave (store -> fence -> load by another lane):
// the release/acquire pair at workgroup scope waits for the stores (s_waitcnt vmcnt(0)); all waves of
// a workgroup share the CU's vector L1, so no cache maintenance is involved.
__device__ __forceinline__ void wave_sync_global()
{
#ifdef ZARC_HIPEMU
    hipemu_wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

// value of the first active lane, as a wave-uniform (scalar) value
__device__ __forceinline__ uint32_t uniform(uint32_t v)
{
#ifdef ZARC_HIPEMU
    return hipemu_readfirstlane(v);
#else
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// v from lane `idx`, where idx is the same in every lane (v_readlane_b32: no LDS round trip)
__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t idx)
{
#ifdef ZARC_HIPEMU
    return hipemu_readfirstlane((uint32_t)__shfl((int)v, (int)idx, 64));
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)idx));
#endif
}

__device__ __forceinline__ uint64_t ballot(bool p) { return (uint64_t)__ballot(p ? 1 : 0); }
__device__ __forceinline__ uint32_t shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ uint32_t shfl_up(uint32_t v, unsigned d) { return (uint32_t)__shfl_up((int)v, d, 64); }
__device__ __forceinline__ uint32_t shfl_down(uint32_t v, unsigned d) { return (uint32_t)__shfl_down((int)v, d, 64); }
__device__ __forceinline__ uint32_t shfl_xor(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }

// Global-memory words that one workgroup updates with atomics and reads back later (the match finder's far tables).  Atomics
// execute in L2; a plain load may be served by the CU's vector L1 and miss them, so the read-back is an agent-scope atomic load
// (global_load ... sc1: served by L2).  The atomic has no return value (fire and forget).
__device__ __forceinline__ uint32_t load_l2_u32(const uint32_t *p)
{
#ifdef ZARC_HIPEMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// two adjacent entries (8-byte aligned) in one request
__device__ __forceinline__ uint64_t load_l2_u64(const uint32_t *p)
{
#ifdef ZARC_HIPEMU
    return (uint64_t)p[0] | ((uint64_t)p[1] << 32);
#else
    return __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void atomic_max_l2(uint32_t *p, uint32_t v)
{
#ifdef ZARC_HIPEMU
    if (*p < v) *p = v;
#else
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// wait until every vector-memory operation this wave has issued (loads, stores, atomics) is complete
__device__ __forceinline__ void wait_vmem()
{
#ifndef ZARC_HIPEMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// a store whose data is not read again by this kernel (non-temporal: does not displace what the caches should keep)
__device__ __forceinline__ void store_streaming(uint64_t *p, uint64_t v)
{
#ifdef ZARC_HIPEMU
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
__device__ __forceinline__ int hb32(uint32_t v) { return 31 - __clz((int)v); }          // floor(log2 v), v > 0
__device__ __forceinline__ int ctz64(uint64_t v) { return __ffsll((long long)v) - 1; }  // v != 0
__device__ __forceinline__ int ctz32(uint32_t v) { return __ffs((int)v) - 1; }          // v != 0

// inclusive prefix sum over the wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
#ifdef ZARC_HIPEMU
    int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = shfl_up(v, (unsigned)d);
        if (l >= d) v += t;
    }
    return v;
#else
    // Six data-parallel-primitive adds instead of six ds_bpermute round trips: shifts by 1, 2, 4, 8 inside the 16-lane rows (a lane
    // without a source adds the 0 it is given), then lane 15 of rows 0 / 2 to rows 1 / 3 and lane 31 to rows 2 and 3.  Every caller
    // runs it with the whole wave active.
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
#endif
}
// value of the lane below (lane 0 keeps its own): a whole-wave shift by one as a data-parallel primitive, no LDS round trip
__device__ __forceinline__ uint32_t shfl_up1(uint32_t v)
{
#ifdef ZARC_HIPEMU
    return shfl_up(v, 1u);
#else
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false); // wave_shr:1
#endif
}
// inclusive prefix sum inside every 16-lane row (the first four steps of the scan above)
__device__ __forceinline__ uint32_t row_scan_incl(uint32_t v)
{
#ifdef ZARC_HIPEMU
    int l = lane_id() & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        uint32_t t = shfl_up(v, (unsigned)d);
        if (l >= d) v += t;
    }
    return v;
#else
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    return v;
#endif
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { uint32_t t = shfl_xor(v, d); v = t > v ? t : v; }
    return v;
}

// little-endian loads from byte pointers with no alignment requirement.  gfx950 executes unaligned
// global_load_dwordx2 / ds_read_b64 natively (the HSA ABI runs compute queues in unaligned access mode, and
// hipcc lowers a byte-pointer memcpy to exactly one such instruction), so these are single loads.
__device__ __forceinline__ uint32_t load_u32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t load_u64(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

} // namespace zd
