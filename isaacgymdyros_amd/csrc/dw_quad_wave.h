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
// This header holds DEVICE code only.  The test-suite compiles the identical kernel source for the host, one FIBER per lane, switching
// fibers at every cross-lane operation, so that indexing and synchronisation mistakes are found on the CPU (and by ASan/UBSan) before
// a GPU run that could fault; that emulation of this vocabulary lives with the tests (tests/emul/dw_quad_wave_host.h) and is named
// by the emulation build (-DDWQ_HOST_SHIM_HEADER=...): the product's build never sees it.
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
// The row-split recursion's hand-over for both layouts.  An env's working PAIR is its quarters 0 and 1 (LPE = 8: its two halves); the lane of
// half 0 wants register a, the lane of half 1 register b, of the lane of quarter K of its limb (K < LPE / 4).  One move per word where the
// source quarter is in the pair (it keeps its own), two where it is not (hex, K = 2, 3); quarters 2, 3 of a hex env get values nobody reads.
template <int LPE_, int K> DQ_HD float rs_take(float a, float b) {
    static_assert((LPE_ == 8 && K < 2) || (LPE_ == 16 && K < 4), "rs_take: quarter out of range");
    const int ai = __builtin_bit_cast(int, a), bi = __builtin_bit_cast(int, b);
    int r;
    if (LPE_ == 8) {
        r = K == 0 ? __builtin_amdgcn_update_dpp(ai, bi, 0x114, 0xF, 0xA, false) : __builtin_amdgcn_update_dpp(bi, ai, 0x104, 0xF, 0x5, false);
    } else if (K == 0) {
        r = __builtin_amdgcn_update_dpp(ai, bi, 0x114, 0xF, 0x2, false);          // quarter 1 takes b of lane l - 4
    } else if (K == 1) {
        r = __builtin_amdgcn_update_dpp(bi, ai, 0x104, 0xF, 0x1, false);          // quarter 0 takes a of lane l + 4
    } else {
        r = __builtin_amdgcn_update_dpp(ai, ai, 0x100 + 4 * K, 0xF, 0x1, false);          // quarter 0: a of lane l + 4 K
        r = __builtin_amdgcn_update_dpp(r, bi, 0x100 + 4 * (K - 1), 0xF, 0x2, false);       // quarter 1: b of lane l + 4 (K - 1)
    }
    return __builtin_bit_cast(float, r);
}
// x of the quarter-K lane of my limb in EVERY lane of the env (K = 0, 1: the working pair's two lanes)
template <int LPE_, int K> DQ_HD float rs_all(float x) {
    static_assert(K < 2, "rs_all: the pair's quarters");
    if (LPE_ == 8) return K == 0 ? oct_lo(x) : oct_hi(x);
    if (K == 0) return quarter0_all(x);
    const int xi = __builtin_bit_cast(int, x);
    int r = __builtin_amdgcn_update_dpp(xi, xi, 0x104, 0xF, 0x1, false);          // row_shl:4 into quarter 0
    r = __builtin_amdgcn_update_dpp(r, xi, 0x114, 0xF, 0x4, false);               // row_shr:4 into quarter 2
    r = __builtin_amdgcn_update_dpp(r, xi, 0x118, 0xF, 0x8, false);               // row_shr:8 into quarter 3
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
// a host compiler: only the test-suite's emulation build does that, and it brings its own implementation of the vocabulary above
#if !defined(DWQ_HOST_SHIM_HEADER)
#error "dw_quad_wave.h is device code (hipcc).  The host emulation of its cross-lane vocabulary is test infrastructure: tests/emul/dw_quad_wave_host.h, selected with -DDWQ_HOST_SHIM_HEADER (tests/emul/Makefile)"
#endif
#include DWQ_HOST_SHIM_HEADER
#endif
