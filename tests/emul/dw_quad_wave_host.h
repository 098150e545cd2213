// dw_quad_wave_host.h -- TEST INFRASTRUCTURE: the host emulation of the cross-lane vocabulary of isaacgymdyros_amd/csrc/dw_quad_wave.h (which
// lists and documents it).  The kernel sources are compiled by g++ against this file and run with one FIBER per lane -- a hand-written
// x86-64 context switch, a round-robin switch at every cross-lane operation -- so that indexing and synchronisation mistakes are found by
// the CPU test-suite (and by ASan/UBSan) before a GPU run that could fault.  Unlike the device, host lanes do NOT run in lock step between
// cross-lane operations, so a missing wave_sync() is an error here even where the device would forgive it.  Nothing in the product
// includes this file: dw_quad_wave.h pulls it in only when a host compiler names it (-DDWQ_HOST_SHIM_HEADER, tests/emul/Makefile).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define DQ_HD static inline
#define DQ_OPAQUE(i) ((void)0)
#define DQ_SGPR_KEEP(x) ((void)0)
#define DQ_SCHED_FENCE() ((void)0)
#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
namespace dwq {

// Minimal cooperative context switch (System V x86-64): callee-saved registers and the stack pointer.
extern "C" void dwq_ctx_switch(void **from_sp, void *to_sp);
#if defined(DWQ_EMUL_IMPLEMENTATION)
asm(".text\n.globl dwq_ctx_switch\n.type dwq_ctx_switch,@function\ndwq_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size dwq_ctx_switch,.-dwq_ctx_switch\n");
#endif

struct WaveEmu {
    static constexpr int NL = 64;
    static constexpr size_t STACK = 512 * 1024;
    void *sp[NL];                 // saved stack pointers of the lane fibers
    void *main_sp;
    char *stacks;
    int   cur;                    // running lane, -1 = scheduler
    bool  done[NL];
    long  nsync[NL];              // cross-lane operations executed by each lane (must agree at the end)
    float xf[2][NL];              // exchange slots, double-buffered by operation parity
    int   xi[2][NL];
    void (*body)(void *, int);
    void *arg;
#if defined(__SANITIZE_ADDRESS__)
    void *fake[NL + 1];
    const void *main_bottom; size_t main_size;
#endif
};
extern thread_local WaveEmu *g_emu;

#if defined(DWQ_EMUL_IMPLEMENTATION)
thread_local WaveEmu *g_emu = nullptr;
static void emu_switch_to(WaveEmu *e, int from, int to) {
    // from/to: lane index or -1 for the scheduler
    void **fsp = from < 0 ? &e->main_sp : &e->sp[from];
    void *tsp = to < 0 ? e->main_sp : e->sp[to];
    e->cur = to;
#if defined(__SANITIZE_ADDRESS__)
    if (to < 0) __sanitizer_start_switch_fiber(&e->fake[from], e->main_bottom, e->main_size);
    else __sanitizer_start_switch_fiber(from < 0 ? &e->fake[WaveEmu::NL] : &e->fake[from], e->stacks + (size_t)to * WaveEmu::STACK, WaveEmu::STACK);
#endif
    dwq_ctx_switch(fsp, tsp);
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(from < 0 ? e->fake[WaveEmu::NL] : e->fake[from], nullptr, nullptr);
#endif
}
static void emu_entry() {
    WaveEmu *e = g_emu;
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_finish_switch_fiber(nullptr, &e->main_bottom, &e->main_size);
#endif
    const int l = e->cur;
    e->body(e->arg, l);
    e->done[l] = true;
    // hand over to the next live lane, or back to the scheduler when every lane has finished
    for (;;) {
        int nxt = -1;
        for (int k = 1; k <= WaveEmu::NL; ++k) { const int c = (l + k) % WaveEmu::NL; if (!e->done[c]) { nxt = c; break; } }
#if defined(__SANITIZE_ADDRESS__)
        // this fiber never resumes: tell ASan its fake stack can go (null save slot)
        if (nxt < 0) __sanitizer_start_switch_fiber(nullptr, e->main_bottom, e->main_size);
        else __sanitizer_start_switch_fiber(nullptr, e->stacks + (size_t)nxt * WaveEmu::STACK, WaveEmu::STACK);
        e->cur = nxt;
        dwq_ctx_switch(&e->sp[l], nxt < 0 ? e->main_sp : e->sp[nxt]);
#else
        emu_switch_to(e, l, nxt);
#endif
        fprintf(stderr, "dwq emulation: finished lane %d resumed\n", l);
        abort();
    }
}
// Runs body(arg, lane) for the 64 lanes of one wave as fibers; returns false if the lanes disagreed on the number of
// cross-lane operations (a cross-lane call in divergent control flow).
bool run_wave(void (*body)(void *, int), void *arg) {
    WaveEmu *e = (WaveEmu *)calloc(1, sizeof(WaveEmu));
    e->stacks = (char *)aligned_alloc(64, WaveEmu::STACK * WaveEmu::NL);
    e->body = body; e->arg = arg;
    for (int l = 0; l < WaveEmu::NL; ++l) {
        char *top = e->stacks + (size_t)(l + 1) * WaveEmu::STACK;
        void **s = (void **)(((uintptr_t)top - 64) & ~(uintptr_t)15);
        // frame popped by dwq_ctx_switch: r15 r14 r13 r12 rbx rbp, then `ret` into emu_entry with rsp = 8 mod 16
        s -= 1; *s = nullptr;                    // fake return address of emu_entry (alignment slot)
        s -= 1; *s = (void *)&emu_entry;
        for (int i = 0; i < 6; ++i) { s -= 1; *s = nullptr; }
        e->sp[l] = (void *)s;
    }
    WaveEmu *prev = g_emu;
    g_emu = e;
    emu_switch_to(e, -1, 0);
    g_emu = prev;
    bool ok = true;
    for (int l = 0; l < WaveEmu::NL; ++l) ok = ok && e->done[l] && e->nsync[l] == e->nsync[0];
    free(e->stacks);
    free(e);
    return ok;
}
#else
bool run_wave(void (*body)(void *, int), void *arg);
#endif

// Every lane calls this at a cross-lane operation: run the other lanes up to the same point, then continue.  With
// round-robin order "switch to the next live lane" IS the barrier: when control comes back, all lanes have arrived.
static inline void emu_barrier() {
    WaveEmu *e = g_emu;
    const int l = e->cur;
    e->nsync[l] += 1;
    int nxt = l;
    for (int k = 1; k <= WaveEmu::NL; ++k) { const int c = (l + k) % WaveEmu::NL; if (!e->done[c]) { nxt = c; break; } }
    if (nxt == l) return;
    extern void emu_switch_public(WaveEmu *, int, int);
    emu_switch_public(e, l, nxt);
}
#if defined(DWQ_EMUL_IMPLEMENTATION)
void emu_switch_public(WaveEmu *e, int from, int to) { emu_switch_to(e, from, to); }
#endif

DQ_HD int lane_id() { return g_emu->cur; }
static inline float emu_xchg(float x, int src_lane) {
    WaveEmu *e = g_emu;
    const int l = e->cur, par = (int)(e->nsync[l] & 1);
    e->xf[par][l] = x;
    emu_barrier();
    return e->xf[par][src_lane];
}
template <int J> DQ_HD float quad_bcast(float x) { return emu_xchg(x, (g_emu->cur & ~3) | J); }
DQ_HD float quad_xor1(float x) { return emu_xchg(x, g_emu->cur ^ 1); }
DQ_HD float quad_xor2(float x) { return emu_xchg(x, g_emu->cur ^ 2); }
DQ_HD float oct_xor4(float x) { return emu_xchg(x, g_emu->cur ^ 4); }
DQ_HD float oct_lo(float x) { return emu_xchg(x, g_emu->cur & ~4); }
DQ_HD float oct_hi(float x) { return emu_xchg(x, g_emu->cur | 4); }
DQ_HD float oct_take_lo(float own, float src) { const float t = emu_xchg(src, g_emu->cur & ~4); return (g_emu->cur & 4) ? t : own; }
DQ_HD float oct_take_hi(float own, float src) { const float t = emu_xchg(src, g_emu->cur | 4); return (g_emu->cur & 4) ? own : t; }
template <int LPE_, int K> DQ_HD float rs_take(float a, float b) {
    const int l = g_emu->cur, q = (l >> 2) & (LPE_ / 4 - 1), src = l - 4 * q + 4 * K;
    const float ta = emu_xchg(a, src), tb = emu_xchg(b, src);
    return q >= 2 ? a : ((q & 1) ? tb : ta);          // (quarters 2, 3 of a hex env: a value nobody reads)
}
template <int LPE_, int K> DQ_HD float rs_all(float x) {
    const int l = g_emu->cur, q = (l >> 2) & (LPE_ / 4 - 1);
    return emu_xchg(x, l - 4 * q + 4 * K);
}
DQ_HD float quad_xor1_hi(float x) { return emu_xchg(x, (g_emu->cur & 4) ? (g_emu->cur ^ 1) : g_emu->cur); }
DQ_HD float oct_fetch(float x, int src) { return emu_xchg(x, (g_emu->cur & ~7) | src); }
DQ_HD float half_bits_to_float(int h) {          // (positive normal numbers and zero: all the tables hold)
    const unsigned int e = ((unsigned int)h >> 10) & 31u, m = (unsigned int)h & 1023u;
    if (e == 0) return 0.0f;
    const unsigned int u = ((e - 15u + 127u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
DQ_HD float quad_pair_lo(float x) { return emu_xchg(x, g_emu->cur & ~2); }
DQ_HD float quad_pair_hi(float x) { return emu_xchg(x, g_emu->cur | 2); }
DQ_HD float hex_xor8(float x) { return emu_xchg(x, g_emu->cur ^ 8); }
template <int K> DQ_HD float quarter_take(float x) { return emu_xchg(x, (g_emu->cur & 12) == 0 ? g_emu->cur + 4 * K : g_emu->cur); }
DQ_HD float quarter0_all(float x) { return emu_xchg(x, g_emu->cur & ~12); }
DQ_HD bool wave_any(bool p) {
    WaveEmu *e = g_emu;
    const int l = e->cur, par = (int)(e->nsync[l] & 1);
    e->xi[par][l] = p ? 1 : 0;
    emu_barrier();
    int any = 0;
    for (int k = 0; k < WaveEmu::NL; ++k) any |= e->xi[par][k];
    return any != 0;
}
DQ_HD unsigned long long wave_ballot(bool p) {
    WaveEmu *e = g_emu;
    const int l = e->cur, par = (int)(e->nsync[l] & 1);
    e->xi[par][l] = p ? 1 : 0;
    emu_barrier();
    unsigned long long m = 0;
    for (int k = 0; k < WaveEmu::NL; ++k) if (!e->done[k] && e->xi[par][k]) m |= 1ull << k;
    return m;
}
DQ_HD void wave_sync() { emu_barrier(); }
DQ_HD void wave_sync_global() { emu_barrier(); }
DQ_HD float rsqrt_nr(float x) { return 1.0f / sqrtf(x); }
DQ_HD float rcp_nr(float x) { return 1.0f / x; }
DQ_HD float rcp_fast(float x) { return 1.0f / x; }
DQ_HD void atomic_add_u64(unsigned long long *p, unsigned long long v) { *p += v; }

}  // namespace dwq
