// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded emulator of the subset of the HIP programming model the engine's kernels
// use, so that the *actual kernel sources* under zarc_amd/csrc can be compiled with g++ and executed
// in this GPU-less container (logic bugs, out-of-bounds accesses under ASan/valgrind, barrier
// divergence).  It is put on the include path *instead of* ROCm's <hip/hip_runtime.h> only by
// tests/emu/Makefile; the product library is never built against it and has no CPU fallback.
//
// Model: workgroups run one after another; every work-item is a ucontext coroutine; __syncthreads
// and the wave64 collectives (__shfl, __ballot, ...) are rendezvous points.  Lanes of a wave run
// sequentially between rendezvous points, which is *stricter* than hardware lockstep: code relying
// on implicit lockstep for LDS hand-offs fails here and must use zd::wave_sync().
#ifndef ZARC_HIPEMU_RUNTIME_H
#define ZARC_HIPEMU_RUNTIME_H
#define ZARC_HIPEMU 1

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)hipemu::dyn_smem();

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 v = {a, b, c, d}; return v; }
static inline uint2 make_uint2(unsigned a, unsigned b) { uint2 v = {a, b}; return v; }

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct hipemu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

namespace hipemu {
struct ThreadCtx;
ThreadCtx *current();
dim3 &tidx();
dim3 &bidx();
dim3 &bdim();
dim3 &gdim();
void *dyn_smem();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void block_barrier();
// wave rendezvous: deposits v, returns every live lane's value in out[64] and the live-lane mask
uint64_t wave_gather(uint64_t v, uint64_t out[64]);
int lane();
} // namespace hipemu

#define threadIdx (hipemu::tidx())
#define blockIdx (hipemu::bidx())
#define blockDim (hipemu::bdim())
#define gridDim (hipemu::gdim())
#define warpSize 64

template <typename F, typename... A>
static inline void hipLaunchKernelGGL(F f, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... a)
{
    hipemu::launch(grid, block, shmem, [&]() { f(a...); });
}

// ---------------------------------------------------------------- host API (trivial) ------------
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { if (n) memcpy(d, s, n); return hipSuccess; }
// pointer attributes: the emulator has one kind of memory.  HIPEMU_ALL_PINNED=1 reports every address as page-locked host memory, so
// that the engine's zero-copy path for pinned caller buffers runs in the CPU tests; otherwise every query fails as for ordinary memory.
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer; void *hostPointer; int isManaged; unsigned allocationFlags; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
    const char *e = getenv("HIPEMU_ALL_PINNED");
    if (!(e && atoi(e) > 0)) return hipErrorInvalidValue;
    memset(a, 0, sizeof *a);
    a->type = hipMemoryTypeHost;
    a->hostPointer = (void *)p;
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamDefault = 0 };
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return hipSuccess; }
template <typename K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 2; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; } // launches are synchronous here
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
// HIPEMU_DEVICES=N makes the emulator report N devices (all the same host memory), so that the multi-device host paths
// (`zarc pack|unpack --gpus N`, Encoder / FrameReader over several handles) open distinct ordinals in the CPU tests
static inline hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("HIPEMU_DEVICES"); *n = e && atoi(e) > 0 ? atoi(e) : 1; return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)2 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof *p);
    strcpy(p->name, "hipemu");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 2;
    p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }

// ---------------------------------------------------------------- device API --------------------
static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline int __lane_id() { return hipemu::lane(); }

static inline unsigned long long __ballot(int pred)
{
    uint64_t v[64], live = hipemu::wave_gather(pred ? 1 : 0, v), m = 0;
    for (int i = 0; i < 64; i++) if (((live >> i) & 1) && v[i]) m |= 1ull << i;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred)
{
    uint64_t v[64], live = hipemu::wave_gather(pred ? 1 : 0, v);
    for (int i = 0; i < 64; i++) if (((live >> i) & 1) && !v[i]) return 0;
    return 1;
}
template <typename T> static inline T __shfl(T val, int src, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shfl width");
    uint64_t raw = 0, v[64];
    memcpy(&raw, &val, sizeof(T));
    hipemu::wave_gather(raw, v);
    int l = hipemu::lane(), base = l & ~(width - 1);
    T out;
    memcpy(&out, &v[base + (src & (width - 1))], sizeof(T));
    return out;
}
template <typename T> static inline T __shfl_up(T val, unsigned d, int width = 64)
{
    uint64_t raw = 0, v[64];
    memcpy(&raw, &val, sizeof(T));
    hipemu::wave_gather(raw, v);
    int l = hipemu::lane(), rel = l & (width - 1);
    T out = val;
    if (rel >= (int)d) memcpy(&out, &v[l - d], sizeof(T));
    return out;
}
template <typename T> static inline T __shfl_down(T val, unsigned d, int width = 64)
{
    uint64_t raw = 0, v[64];
    memcpy(&raw, &val, sizeof(T));
    hipemu::wave_gather(raw, v);
    int l = hipemu::lane(), rel = l & (width - 1);
    T out = val;
    if (rel + (int)d < width) memcpy(&out, &v[l + d], sizeof(T));
    return out;
}
template <typename T> static inline T __shfl_xor(T val, int mask, int width = 64)
{
    uint64_t raw = 0, v[64];
    memcpy(&raw, &val, sizeof(T));
    hipemu::wave_gather(raw, v);
    int l = hipemu::lane();
    T out;
    memcpy(&out, &v[(l ^ mask) & 63], sizeof(T));
    (void)width;
    return out;
}
// first live lane's value (hardware: first active lane)
static inline uint32_t hipemu_readfirstlane(uint32_t x)
{
    uint64_t v[64], live = hipemu::wave_gather(x, v);
    return (uint32_t)v[__builtin_ctzll(live)];
}
static inline void hipemu_wave_sync() { uint64_t v[64]; hipemu::wave_gather(0, v); }

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

#define HIPEMU_ATOMIC(name, T, expr) \
    static inline T name(T *p, T v) { T old = *p; *p = (expr); return old; }
HIPEMU_ATOMIC(atomicAdd, unsigned, old + v)
HIPEMU_ATOMIC(atomicAdd, int, old + v)
HIPEMU_ATOMIC(atomicAdd, unsigned long long, old + v)
HIPEMU_ATOMIC(atomicSub, unsigned, old - v)
HIPEMU_ATOMIC(atomicMax, unsigned, old > v ? old : v)
HIPEMU_ATOMIC(atomicMax, int, old > v ? old : v)
HIPEMU_ATOMIC(atomicMin, unsigned, old < v ? old : v)
HIPEMU_ATOMIC(atomicOr, unsigned, old | v)
HIPEMU_ATOMIC(atomicOr, unsigned long long, old | v)
HIPEMU_ATOMIC(atomicAnd, unsigned, old & v)
HIPEMU_ATOMIC(atomicExch, unsigned, v)
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v) { unsigned old = *p; if (old == cmp) *p = v; return old; }

#endif
