// dw_quad_wave.h -- execution-model shim of the QUAD kernels: 4 lanes (one DPP quad) per environment, 16 environments
// per wavefront, one wavefront per workgroup.
//
// Why quads: the first-generation kernels (dw_physics.h / dw_task.h) give a whole 64-lane wave to one env and run the
// tree recursions on 6..12 of its lanes; the PMC passes of round 1 show 15.8 k VALU instructions per env-step for
// ~0.05 M useful FMAs (5-10 % lane utilisation) on a kernel whose limit is VALU issue.  Here a lane owns a LIMB of its env
// (leg / leg / trunk+arm / arm) and walks it body by body with every instruction doing useful arithmetic; the four
// lanes of an env exchange data where limbs meet with DPP quad permutes (full rate, no LDS), and a wave carries 16 envs.
//
// Cross-lane vocabulary (all of it must be called in wave-uniform control flow):
//   quad_bcast<J>(x)       value of x in lane J of the caller's quad            (v_mov_dpp quad_perm:[J,J,J,J])
//   quad_xor1(x)/quad_xor2 value of x in lane (l ^ 1) / (l ^ 2)                 (quad_perm:[1,0,3,2] / [2,3,0,1])
//   oct_xor4(x)            value of x in lane (l ^ 4)  (octet kernels)           (row_shl:4 / row_shr:4, complementary bank masks)
//   oct_lo(x) / oct_hi(x)  value of x in lane (l & ~4) / (l | 4) (octet kernels) (row_shr:4 into the high quads / row_shl:4 into the low quads)
//   oct_take_lo / _hi(own, src): one half of a limb takes ANOTHER register of the other half's lane, the other half keeps `own`
//   oct_fetch(x, src)      x of lane (l & ~7) | src of the caller's octet, src per lane (ds_bpermute)
//   quad_xor1_hi / quad_pair_lo / quad_pair_hi: one-move forms of `cond ? exchanged : own` (below)
//   wave_any(p)            true in every lane iff p holds in some lane           (v_cmp + s_cmp on the ballot)
//   wave_ballot(p)         64-bit mask of p over the lanes, the same in every lane (v_cmp into an SGPR pair)
//   wave_sync_global()     as wave_sync, for global memory too (workgroup-scope release / acquire).
//   wave_sync()            LDS written before it by any lane is visible to every lane after it.  One wave = one
//                          workgroup and a wave's LDS operations complete in order, so on the device this is a compiler
//                          fence and no instruction.
//
// The second half of this file is NOT a product path: it runs the identical kernel source on the host, one FIBER per
// lane (tests/emul/), switching fibers at every cross-lane operation, so indexing and synchronisation mistakes are found
// by the CPU test-suite (and by ASan/UBSan) before a GPU run that could fault.  Unlike the device, host lanes do NOT run in
// lock step between cross-lane operations, so a missing wave_sync() is an error there even where the device would forgive it.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DQ_HD __device__ __forceinline__
#define DQ_OPAQUE(i) asm volatile("" : "+v"(i))
// a wave-uniform value pinned in a scalar register: the compiler may otherwise re-load a launch-invariant parameter from memory at
// every use instead of keeping it (a scalar-memory round trip on the critical path of a lone wave)
#define DQ_SGPR_KEEP(x) asm volatile("" : "+s"(x))
// the machine scheduler moves nothing across this point: bounds how many iterations of an unrolled loop it overlaps (and with
// them the registers their loads occupy)
#define DQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
namespace dwq {

DQ_HD int lane_id() { return (int)(threadIdx.x & 63u); }      // (the octet kernels run two waves per workgroup)

template <int CTRL> DQ_HD float dpp_quad(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int J> DQ_HD float quad_bcast(float x) { return dpp_quad<J | (J << 2) | (J << 4) | (J << 6)>(x); }
DQ_HD float quad_xor1(float x) { return dpp_quad<1 | (0 << 2) | (3 << 4) | (2 << 6)>(x); }
DQ_HD float quad_xor2(float x) { return dpp_quad<2 | (3 << 2) | (0 << 4) | (1 << 6)>(x); }
// value of x in lane (l ^ 4): the other quad of an 8-lane octet (dw_oct.h).  gfx9 DPP has no xor across quads; the low quads
// of a 16-lane row (banks 0, 2) take the value 4 lanes up (row_shl:4), the high quads (banks 1, 3) 4 lanes down (row_shr:4):
// two moves with complementary bank masks.
DQ_HD float oct_xor4(float x) {
    const int xi = __builtin_bit_cast(int, x);
    int r = __builtin_amdgcn_update_dpp(xi, xi, 0x104, 0xF, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, xi, 0x114, 0xF, 0xA, false);
    return __builtin_bit_cast(float, r);
}
// value of x in the half-0 lane of my limb (lane l & ~4): one move, the high quads take the value 4 lanes down
DQ_HD float oct_lo(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x114, 0xF, 0xA, false));
}
// value of x in the half-1 lane of my limb (lane l | 4): one move, the low quads take the value 4 lanes up
DQ_HD float oct_hi(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x104, 0xF, 0x5, false));
}
// selections across the halves of a limb with DIFFERENT registers on the two sides (the row-split recursion of dw_oct.h):
//   oct_take_lo(own, src)   half-1 lanes take src of lane l - 4, half-0 lanes keep `own`   (row_shr:4 into the high quads)
//   oct_take_hi(own, src)   half-0 lanes take src of lane l + 4, half-1 lanes keep `own`   (row_shl:4 into the low quads)
DQ_HD float oct_take_lo(float own, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own), __builtin_bit_cast(int, src), 0x114, 0xF, 0xA, false));
}
DQ_HD float oct_take_hi(float own, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own), __builtin_bit_cast(int, src), 0x104, 0xF, 0x5, false));
}
// selections that a lane-dependent `cond ? exchanged : own` would spend a move AND a select on, as ONE move:
//   quad_xor1_hi(x)   half-1 lanes (l & 4) take x of lane l ^ 1, half-0 lanes keep their own (bank mask: the high quads of a row)
//   quad_pair_lo(x)   x of the lane (l & ~2) of my pair {l, l ^ 2};  quad_pair_hi(x): of the lane (l | 2)
DQ_HD float quad_xor1_hi(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 1 | (0 << 2) | (3 << 4) | (2 << 6), 0xF, 0xA, false));
}
// x of lane (l & ~7) | src of my octet, src per lane (ds_bpermute: through the LDS crossbar, no memory touched)
DQ_HD float oct_fetch(float x, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 56u) | (unsigned)src) << 2), __builtin_bit_cast(int, x)));
}
DQ_HD float half_bits_to_float(int h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
DQ_HD float quad_pair_lo(float x) { return dpp_quad<0 | (1 << 2) | (0 << 4) | (1 << 6)>(x); }
DQ_HD float quad_pair_hi(float x) { return dpp_quad<2 | (3 << 2) | (2 << 4) | (3 << 6)>(x); }
// The 16-lanes-per-env ("hex") instantiation of the octet kernels (dw_oct.h, OCT_LPE = 16): an env is one DPP row, four QUARTERS
// q = (l >> 2) & 3 of four limb lanes each.
//   hex_xor8(x)          x of lane l ^ 8 (the other octet of the row)                      (row_ror:8)
//   quarter_take<K>(x)   quarter-0 lanes take x of lane l + 4 K (quarter K, same limb); other lanes keep theirs  (row_shl:4K, bank 0)
//   quarter0_all(x)      every lane takes x of lane l & ~12 (quarter 0, same limb)          (three bank-masked row_shr moves)
DQ_HD float hex_xor8(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x128, 0xF, 0xF, false));
}
template <int K> DQ_HD float quarter_take(float x) {
    static_assert(K >= 1 && K <= 3, "quarter_take: K = 1..3");
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x100 + 4 * K, 0xF, 0x1, false));
}
DQ_HD float quarter0_all(float x) {
    const int xi = __builtin_bit_cast(int, x);
    int r = __builtin_amdgcn_update_dpp(xi, xi, 0x114, 0xF, 0x2, false);          // row_shr:4  into bank 1
    r = __builtin_amdgcn_update_dpp(r, xi, 0x118, 0xF, 0x4, false);               // row_shr:8  into bank 2
    r = __builtin_amdgcn_update_dpp(r, xi, 0x11C, 0xF, 0x8, false);               // row_shr:12 into bank 3
    return __builtin_bit_cast(float, r);
}
DQ_HD bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
DQ_HD unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }   // bit l = p of lane l, the same in every lane
DQ_HD void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// as wave_sync, and global-memory writes of any lane before it are visible to every lane's loads after it (one wave =
// one workgroup on one CU: workgroup scope)
DQ_HD void wave_sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
DQ_HD float rsqrt_nr(float x) { const float y = __builtin_amdgcn_rsqf(x); return y * (1.5f - 0.5f * x * y * y); }
DQ_HD float rcp_nr(float x) { const float y = __builtin_amdgcn_rcpf(x); return y * (2.0f - x * y); }
DQ_HD float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }          // 1 ulp
DQ_HD void atomic_add_u64(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
#if defined(__HIP_DEVICE_COMPILE__)          // (the same through a global pointer: global_atomic instead of flat_atomic, dw_bufg.h)
DQ_HD void atomic_add_u64(unsigned long long __attribute__((address_space(1))) *p, unsigned long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

}  // namespace dwq
#else
// ------------------------------------------------------------------------------------------------ host emulation
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
#endif
