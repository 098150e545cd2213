// dw_lane_wave.h -- execution-model shim of the LANE kernels (dw_lane*.h): one lane = one environment, one wavefront = one
// limb of 64 environments, a workgroup = 4 wavefronts = the whole robot of 64 environments.
//
// Why.  The earlier generations spread ONE env over the lanes of a wave (64, then 4, then 8 lanes per env): limbs of different
// length, mirrored halves and idle lanes left a third of the vector lanes doing useful arithmetic on a kernel that is bound by
// vector-instruction issue.  Here every vector instruction works for 64 envs at once -- the body a wave is
// at is wave-uniform, so the model's constants are scalar operands and the branches are scalar branches -- and the
// parallelism inside an env is spread over the four SIMDs of the CU instead of over lanes: wave 0 the left leg, 1 the right
// leg, 2 waist + left arm, 3 right arm (+ the head on wave 0).  Where limbs meet, 28-word records cross through LDS behind a
// workgroup barrier.
//
// Vocabulary (all of it in workgroup- or wave-uniform control flow):
//   tid() / lane() / wave()   thread in the workgroup (0..255), lane = env within the workgroup (0..63), wave (0..3, scalar)
//   wg_barrier()              LDS written before it by any wave is visible to every wave after it (s_waitcnt lgkmcnt(0) +
//                             s_barrier: outstanding GLOBAL loads stay in flight across it)
//   wg_barrier_global()       the same for global memory too (workgroup-scope release / acquire: waits for the wave's
//                             outstanding global accesses)
//   wave_any(p)               true in every lane of the wave iff p holds in some lane
//
// The second half is NOT a product path: tests/emul/ compiles the kernel source with g++ and runs a workgroup as 256 fibers
// (one per thread), switching at every barrier / vote, so that indexing and synchronisation mistakes show up in the CPU suite
// and under ASan / UBSan before a GPU run that could fault.  Host fibers do not run in lock step between those points.
#pragma once

#include "dw_quad_wave.h"          // DQ_HD, DQ_OPAQUE, the fiber context switch of the host emulation

// member functions (DQ_HD is `static inline` on the host)
#if defined(__HIPCC__)
#define DL_MEM __device__ __forceinline__
#else
#define DL_MEM inline
#endif

#if defined(__HIPCC__)
namespace dwl {

DQ_HD int tid() { return (int)threadIdx.x; }
DQ_HD int lane() { return (int)(threadIdx.x & 63u); }
DQ_HD int wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
DQ_HD void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
DQ_HD void wg_barrier_global() { __syncthreads(); }
DQ_HD bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
DQ_HD int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }          // a value every lane of the wave holds, into a scalar register
// Hand-off between TWO waves of the workgroup without stopping the others at a barrier: the producer's LDS stores, then
// flag_post(mine, n); the consumer's flag_wait(theirs, n), then its LDS loads.  A wave's LDS operations execute in order, so
// whoever sees the flag sees the data; n counts up for the life of the kernel (the flags are zeroed before its first barrier).
DQ_HD void flag_post(int *flag, int n) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63u) == 0) *reinterpret_cast<volatile int *>(flag) = n;
}
DQ_HD void flag_wait(const int *flag, int n) {
    for (;;) {
        const int v = *reinterpret_cast<const volatile int *>(flag);
        if (__builtin_amdgcn_readfirstlane(v) >= n) break;
    }
    asm volatile("" ::: "memory");
}

}  // namespace dwl
#else
namespace dwl {

struct WgEmu {
    static constexpr int NT = 256;
    static constexpr size_t STACK = 768 * 1024;
    void *sp[NT];
    void *main_sp;
    char *stacks;
    int   cur;
    bool  done[NT];
    long  nbar[NT];               // workgroup barriers passed by each thread
    long  nvote[NT];              // wave votes passed by each thread
    int   xi[2][NT];
    long  idle_switches;          // switches since the last thread made progress (deadlock detection)
    void (*body)(void *, int);
    void *arg;
#if defined(__SANITIZE_ADDRESS__)
    void *fake[NT + 1];
    const void *main_bottom; size_t main_size;
#endif
};
extern thread_local WgEmu *g_wg;

#if defined(DWQ_EMUL_IMPLEMENTATION)
thread_local WgEmu *g_wg = nullptr;
static void wg_switch_to(WgEmu *e, int from, int to) {
    void **fsp = from < 0 ? &e->main_sp : &e->sp[from];
    void *tsp = to < 0 ? e->main_sp : e->sp[to];
    e->cur = to;
#if defined(__SANITIZE_ADDRESS__)
    if (to < 0) __sanitizer_start_switch_fiber(&e->fake[from], e->main_bottom, e->main_size);
    else __sanitizer_start_switch_fiber(from < 0 ? &e->fake[WgEmu::NT] : &e->fake[from], e->stacks + (size_t)to * WgEmu::STACK, WgEmu::STACK);
#endif
    dwq::dwq_ctx_switch(fsp, tsp);
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(from < 0 ? e->fake[WgEmu::NT] : e->fake[from], nullptr, nullptr);
#endif
}
static int wg_next_live(WgEmu *e, int l) {
    for (int k = 1; k <= WgEmu::NT; ++k) { const int c = (l + k) % WgEmu::NT; if (!e->done[c]) return c; }
    return -1;
}
static void wg_entry() {
    WgEmu *e = g_wg;
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(nullptr, &e->main_bottom, &e->main_size);
#endif
    const int l = e->cur;
    e->body(e->arg, l);
    e->done[l] = true;
    e->idle_switches = 0;
    for (;;) {
        const int nxt = wg_next_live(e, l);
#if defined(__SANITIZE_ADDRESS__)
        if (nxt < 0) __sanitizer_start_switch_fiber(nullptr, e->main_bottom, e->main_size);
        else __sanitizer_start_switch_fiber(nullptr, e->stacks + (size_t)nxt * WgEmu::STACK, WgEmu::STACK);
        e->cur = nxt;
        dwq::dwq_ctx_switch(&e->sp[l], nxt < 0 ? e->main_sp : e->sp[nxt]);
#else
        wg_switch_to(e, l, nxt);
#endif
        fprintf(stderr, "dwl emulation: finished thread %d resumed\n", l);
        abort();
    }
}
// Runs body(arg, tid) for the 256 threads of one workgroup as fibers.  Returns false if the threads disagreed on the number
// of barriers (a barrier in divergent control flow).
bool run_workgroup(void (*body)(void *, int), void *arg) {
    WgEmu *e = (WgEmu *)calloc(1, sizeof(WgEmu));
    e->stacks = (char *)aligned_alloc(64, WgEmu::STACK * WgEmu::NT);
    e->body = body; e->arg = arg;
    for (int l = 0; l < WgEmu::NT; ++l) {
        char *top = e->stacks + (size_t)(l + 1) * WgEmu::STACK;
        void **s = (void **)(((uintptr_t)top - 64) & ~(uintptr_t)15);
        s -= 1; *s = nullptr;
        s -= 1; *s = (void *)&wg_entry;
        for (int i = 0; i < 6; ++i) { s -= 1; *s = nullptr; }
        e->sp[l] = (void *)s;
    }
    WgEmu *prev = g_wg;
    g_wg = e;
    wg_switch_to(e, -1, 0);
    g_wg = prev;
    bool ok = true;
    for (int l = 0; l < WgEmu::NT; ++l) ok = ok && e->done[l] && e->nbar[l] == e->nbar[0];
    free(e->stacks);
    free(e);
    return ok;
}
void wg_yield_public() {
    WgEmu *e = g_wg;
    const int l = e->cur, nxt = wg_next_live(e, l);
    if (++e->idle_switches > 8L * WgEmu::NT) {
        fprintf(stderr, "dwl emulation: deadlock -- thread %d (wave %d) waits at barrier %ld / vote %ld while others never arrive\n", l, l >> 6, e->nbar[l], e->nvote[l]);
        abort();
    }
    if (nxt >= 0 && nxt != l) wg_switch_to(e, l, nxt);
}
#else
bool run_workgroup(void (*body)(void *, int), void *arg);
void wg_yield_public();
#endif

DQ_HD int tid() { return g_wg->cur; }
DQ_HD int lane() { return g_wg->cur & 63; }
DQ_HD int wave() { return g_wg->cur >> 6; }
static inline void wg_barrier() {
    WgEmu *e = g_wg;
    const int l = e->cur;
    const long mine = ++e->nbar[l];
    e->idle_switches = 0;
    for (;;) {
        bool all = true;
        for (int k = 0; k < WgEmu::NT; ++k) if (!e->done[k] && e->nbar[k] < mine) { all = false; break; }
        if (all) return;
        wg_yield_public();
    }
}
static inline void wg_barrier_global() { wg_barrier(); }
static inline int uniform(int x) { return x; }
static inline bool wave_any(bool p);
// (host lanes do not run in lock step: the wave's lanes meet before lane 0 posts, as the device's do by construction)
static inline void flag_post(int *flag, int n) { (void)wave_any(true); if ((g_wg->cur & 63) == 0) *flag = n; }
static inline void flag_wait(const int *flag, int n) {
    g_wg->idle_switches = 0;
    while (*reinterpret_cast<const volatile int *>(flag) < n) wg_yield_public();
}
static inline bool wave_any(bool p) {
    WgEmu *e = g_wg;
    const int l = e->cur, w0 = l & ~63;
    const int par = (int)(e->nvote[l] & 1);
    e->xi[par][l] = p ? 1 : 0;
    const long mine = ++e->nvote[l];
    e->idle_switches = 0;
    for (;;) {
        bool all = true;
        for (int k = w0; k < w0 + 64; ++k) if (!e->done[k] && e->nvote[k] < mine) { all = false; break; }
        if (all) break;
        wg_yield_public();
    }
    int any = 0;
    for (int k = w0; k < w0 + 64; ++k) if (!e->done[k]) any |= e->xi[par][k];
    return any != 0;
}

}  // namespace dwl
#endif
