// dw_wave_host.h -- TEST INFRASTRUCTURE: the host definition of dw::Wave and its cross-lane helpers (isaacgymdyros_amd/csrc/dw_wave.h
// documents them): a region is a loop over the 64 lanes, a `simt` region one fiber per lane (dw_quad_wave_host.h).  Nothing in the product
// includes this file: dw_wave.h pulls it in only when a host compiler names it (-DDW_HOST_SHIM_HEADER, tests/emul/Makefile).
#pragma once
#include <type_traits>
#define DW_HD static inline
#define DW_OPAQUE(i) ((void)0)
#include "dw_quad_wave_host.h"          // one fiber per lane: run_wave / emu_xchg
namespace dw {
struct Wave {
    template <class F> void par(F &&f) const {
        for (int lane = 0; lane < 64; ++lane) f(lane);
    }
    template <class F> void simt(F &&f) const {
        using Fn = typename std::remove_reference<F>::type;
        if (!dwq::run_wave([](void *a, int lane) { (*static_cast<Fn *>(a))(lane); }, (void *)&f)) abort();
    }
};
static inline int uniform(int x) { return x; }
template <int J> static inline float half_bcast(float x) { return dwq::emu_xchg(x, (dwq::lane_id() & 32) | J); }
static inline float lane_bcast(float x, int lane) { return dwq::emu_xchg(x, lane); }
}  // namespace dw
