// Which lane's value stays when several lanes of one wave store to the same LDS address in one instruction?
// (zge_match.hip S2: the 16-bit near table has no atomic max; the kernel settles contested stores exactly, this only tells how often
// it has to.)  Build: hipcc --offload-arch=gfx950 -O2 -o lds_write_order lds_write_order.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// ... and in which order do same-address LDS atomics WITH RETURN of one instruction execute?  (zge_entropy.hip, FSE state tables: every
// cell takes the next slot of its symbol; 64 cells per instruction need increasing slots in lane order.)
__global__ void k_add(const uint32_t *slots, uint32_t *out, int rounds)
{
    __shared__ uint32_t cnt[4096];
    const int lane = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        const uint32_t s = slots[r * 64 + lane] & 4095;
        cnt[s] = 0;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        out[r * 64 + lane] = atomicAdd(&cnt[s], 1u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
__global__ void k(const uint32_t *slots, uint32_t *out16, uint32_t *out32, int rounds)
{
    __shared__ uint16_t t16[4096];
    __shared__ uint32_t t32[4096];
    const int lane = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        const uint32_t s = slots[r * 64 + lane] & 4095;
        t16[s] = 0xFFFF; t32[s] = 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t16[s] = (uint16_t)lane;
        t32[s] = (uint32_t)lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        out16[r * 64 + lane] = t16[s];
        out32[r * 64 + lane] = t32[s];
    }
}
int main()
{
    const int R = 20000;
    uint32_t *h = (uint32_t *)malloc(R * 64 * 4), *d, *o16, *o32, *r16 = (uint32_t *)malloc(R * 64 * 4), *r32 = (uint32_t *)malloc(R * 64 * 4);
    srand(7);
    for (int r = 0; r < R; r++) {
        const int mode = r % 5; // 0: random in 64 slots (many conflicts), 1: period 2..17, 2: random in 4096, 3: all same, 4: halves of one dword (slots 2k, 2k+1)
        for (int l = 0; l < 64; l++) {
            uint32_t s;
            if (mode == 0) s = rand() % 64; else if (mode == 1) s = 100 + l % (2 + r % 16); else if (mode == 2) s = rand() % 4096; else if (mode == 3) s = 77; else s = 200 + (rand() % 8);
            h[r * 64 + l] = s;
        }
    }
    hipMalloc(&d, R * 64 * 4); hipMalloc(&o16, R * 64 * 4); hipMalloc(&o32, R * 64 * 4);
    hipMemcpy(d, h, R * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o16, o32, R);
    hipMemcpy(r16, o16, R * 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(r32, o32, R * 64 * 4, hipMemcpyDeviceToHost);
    long groups = 0, hi16 = 0, lo16 = 0, other16 = 0, hi32 = 0, lo32 = 0, other32 = 0;
    for (int r = 0; r < R; r++)
        for (int l = 0; l < 64; l++) {
            int maxl = l, minl = l, cnt = 0;
            for (int j = 0; j < 64; j++) if ((h[r * 64 + j] & 4095) == (h[r * 64 + l] & 4095)) { cnt++; if (j > maxl) maxl = j; if (j < minl) minl = j; }
            if (cnt < 2 || l != minl) continue; // once per contested slot
            groups++;
            const int w16 = (int)r16[r * 64 + l], w32 = (int)r32[r * 64 + l];
            if (w16 == maxl) hi16++; else if (w16 == minl) lo16++; else other16++;
            if (w32 == maxl) hi32++; else if (w32 == minl) lo32++; else other32++;
        }
    { // atomics with return: the value a lane gets must be the number of LOWER lanes with the same slot
        uint32_t *oa, *ra = (uint32_t *)malloc(R * 64 * 4);
        hipMalloc(&oa, R * 64 * 4);
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, 0, d, oa, R);
        hipMemcpy(ra, oa, R * 64 * 4, hipMemcpyDeviceToHost);
        long ok = 0, bad = 0;
        for (int r = 0; r < R; r++)
            for (int l = 0; l < 64; l++) {
                uint32_t below = 0;
                for (int j = 0; j < l; j++) below += (h[r * 64 + j] & 4095) == (h[r * 64 + l] & 4095);
                if (ra[r * 64 + l] == below) ok++; else bad++;
            }
        printf("atomic add with return: lane order %ld, other %ld\n", ok, bad);
    }
    printf("contested slots %ld | b16: highest lane wins %ld, lowest %ld, other %ld | b32: highest %ld, lowest %ld, other %ld\n", groups, hi16, lo16, other16, hi32, lo32, other32);
    return 0;
}
