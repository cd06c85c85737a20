// tests/emu/hipemu.cpp -- TEST INFRASTRUCTURE ONLY: runtime of the HIP emulator (see hip/hip_runtime.h).
#include "hip/hip_runtime.h"
#include <chrono>
#include <csetjmp>
#include <mutex>
#include <sys/mman.h>
#include <ucontext.h>
#include <vector>

namespace hipemu {

enum State { RUNNABLE = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3 };

struct ThreadCtx {
    ucontext_t ctx;
    jmp_buf jb;          // after the first entry, switches use _setjmp/_longjmp (no sigprocmask syscalls)
    bool started = false;
    char *stack = nullptr;
    dim3 tidx;
    int linear = 0, wave = 0, lane = 0;
    State state = RUNNABLE;
    uint64_t xchg[2] = {0, 0};
    uint32_t wave_seq = 0; // number of wave rendezvous this lane has entered
};

struct WaveCtx {
    uint64_t live[2] = {0, 0};
    uint32_t released = 0; // number of rendezvous released
};

static const size_t STACK_BYTES = 256 * 1024;
static ucontext_t g_sched;
static jmp_buf g_sched_jb;
static ThreadCtx *g_cur = nullptr;
static std::vector<ThreadCtx> g_threads;
static std::vector<WaveCtx> g_waves;
static dim3 g_bidx, g_bdim, g_gdim;
static void *g_dyn = nullptr;
static const std::function<void()> *g_body = nullptr;
static std::vector<char *> g_stack_pool;

ThreadCtx *current() { return g_cur; }
dim3 &tidx() { return g_cur->tidx; }
dim3 &bidx() { return g_bidx; }
dim3 &bdim() { return g_bdim; }
dim3 &gdim() { return g_gdim; }
void *dyn_smem() { return g_dyn; }
int lane() { return g_cur->lane; }

static void yield_to_sched()
{
    if (!_setjmp(g_cur->jb)) _longjmp(g_sched_jb, 1);
}
static void resume(ThreadCtx &t)
{
    g_cur = &t;
    if (!_setjmp(g_sched_jb)) {
        if (!t.started) { t.started = true; swapcontext(&g_sched, &t.ctx); }
        else _longjmp(t.jb, 1);
    }
}

void block_barrier()
{
    g_cur->state = AT_BARRIER;
    yield_to_sched();
}

uint64_t wave_gather(uint64_t v, uint64_t out[64])
{
    ThreadCtx *t = g_cur;
    uint32_t seq = t->wave_seq++;
    t->xchg[seq & 1] = v;
    t->state = AT_WAVE;
    yield_to_sched();
    // released: every live lane of this wave has deposited its value for `seq`
    ThreadCtx *base = &g_threads[(size_t)t->wave * 64];
    size_t nthreads = g_threads.size();
    for (int i = 0; i < 64; i++) {
        size_t idx = (size_t)t->wave * 64 + (size_t)i;
        out[i] = idx < nthreads ? base[i].xchg[seq & 1] : 0;
    }
    return g_waves[(size_t)t->wave].live[seq & 1];
}

static void trampoline()
{
    (*g_body)();
    g_cur->state = DONE;
    _longjmp(g_sched_jb, 1);
}

static char *get_stack(size_t i)
{
    while (g_stack_pool.size() <= i) {
        void *p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("hipemu mmap"); abort(); }
        g_stack_pool.push_back((char *)p);
    }
    return g_stack_pool[i];
}

static void run_block(dim3 block)
{
    size_t n = (size_t)block.x * block.y * block.z;
    g_threads.assign(n, ThreadCtx());
    g_waves.assign((n + 63) / 64, WaveCtx());
    for (size_t i = 0; i < n; i++) {
        ThreadCtx &t = g_threads[i];
        t.linear = (int)i;
        t.wave = (int)(i / 64);
        t.lane = (int)(i % 64);
        t.tidx = dim3((unsigned)(i % block.x), (unsigned)((i / block.x) % block.y), (unsigned)(i / ((size_t)block.x * block.y)));
        t.stack = get_stack(i);
        getcontext(&t.ctx);
        t.ctx.uc_stack.ss_sp = t.stack;
        t.ctx.uc_stack.ss_size = STACK_BYTES;
        t.ctx.uc_link = nullptr;
        makecontext(&t.ctx, trampoline, 0);
    }
    size_t live = n;
    while (live > 0) {
        bool progressed = false;
        for (size_t i = 0; i < n; i++) {
            ThreadCtx &t = g_threads[i];
            if (t.state != RUNNABLE) continue;
            resume(t);
            progressed = true;
            if (t.state == DONE) live--;
        }
        // wave rendezvous: release a wave once all of its live lanes wait on the same sequence number
        for (size_t w = 0; w < g_waves.size(); w++) {
            size_t lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
            uint64_t livemask = 0;
            bool all = true, any = false;
            uint32_t seq = 0;
            for (size_t i = lo; i < hi; i++) {
                ThreadCtx &t = g_threads[i];
                if (t.state == DONE) continue;
                livemask |= 1ull << (i - lo);
                if (t.state != AT_WAVE) { all = false; continue; }
                if (!any) { seq = t.wave_seq; any = true; }
                else if (t.wave_seq != seq) all = false;
            }
            if (any && all) {
                g_waves[w].live[(seq - 1) & 1] = livemask;
                for (size_t i = lo; i < hi; i++)
                    if (g_threads[i].state == AT_WAVE) g_threads[i].state = RUNNABLE;
                progressed = true;
            }
        }
        // block barrier: release when every live thread waits on it
        {
            bool all = live > 0;
            for (size_t i = 0; i < n && all; i++)
                if (g_threads[i].state != DONE && g_threads[i].state != AT_BARRIER) all = false;
            if (all) {
                for (size_t i = 0; i < n; i++)
                    if (g_threads[i].state == AT_BARRIER) g_threads[i].state = RUNNABLE;
                progressed = true;
            }
        }
        if (!progressed && live > 0) {
            fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): divergent barrier / wave collective\n", g_bidx.x, g_bidx.y, g_bidx.z);
            for (size_t i = 0; i < n; i++)
                if (g_threads[i].state != DONE)
                    fprintf(stderr, "  thread %zu state=%d wave_seq=%u\n", i, (int)g_threads[i].state, g_threads[i].wave_seq);
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
{
    // one kernel at a time, whichever host thread launches it (the engine's staging helper thread launches a gather kernel)
    static std::mutex launch_mutex;
    std::lock_guard<std::mutex> guard(launch_mutex);
    g_body = &body;
    g_gdim = grid;
    g_bdim = block;
    std::vector<uint64_t> dyn((shmem + 7) / 8 + 1);
    g_dyn = dyn.data();
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                g_bidx = dim3(x, y, z);
                memset(dyn.data(), 0xA5, dyn.size() * 8); // LDS is not zero-initialised on hardware
                run_block(block);
            }
    g_cur = nullptr;
    g_dyn = nullptr;
}

} // namespace hipemu

struct hipemu_event { std::chrono::steady_clock::time_point t; };
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCD, n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
