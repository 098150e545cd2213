// dw_wave.h -- execution-model shim of the DyrosDynamicWalk kernels.
//
// The kernels are written for ONE WAVEFRONT (64 lanes) PER ENVIRONMENT: every phase of the step is a
// "region" executed by all 64 lanes, regions exchange data only through the env's LDS block, and a
// workgroup is exactly one wave, so the boundary between regions is a wavefront-scope fence (no instruction).
// `Wave::par(f)` runs f(lane) for the calling lane and ends the region.  `Wave::simt(f)` is a region whose lanes also
// talk to each other in registers (swz / readlane below), which a lane loop cannot express: the host runs it with one
// fiber per lane (dw_quad_wave.h).  `uniform(x)` turns a value
// every lane read from the same LDS word into a scalar-register value so that loops and branches that
// contain regions are provably wave-uniform.
//
// This header holds DEVICE code only.  The test-suite compiles the identical kernel source with g++ into a lane-loop emulation, so that
// indexing and synchronisation mistakes are found on the CPU (and by ASan/UBSan) instead of by a GPU fault that can take a whole node
// down; that second definition of `Wave` lives with the tests (tests/emul/dw_wave_host.h, named by -DDW_HOST_SHIM_HEADER).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DW_HD __device__ __forceinline__
namespace dw {

// Makes an index opaque to the optimiser at this point.  Used where a lane's addresses into the device-resident model
// (64-bit pointers) would otherwise be computed once at kernel entry for both substeps and then live -- i.e. be spilled
// to scratch -- across the whole kernel; recomputing them costs two instructions.
#define DW_OPAQUE(i) asm volatile("" : "+v"(i))

struct Wave {
    template <class F> DW_HD void par(F &&f) const {
        // (tried: laundering the lane id through an empty asm per region, to stop lane-derived LDS addresses
        //  from being kept live across regions -- fewer registers, but the recomputed address arithmetic cost 8 %)
        const int lane = (int)threadIdx.x;
        f(lane);
        // End of region.  The workgroup IS the wave, and a wave's LDS instructions execute in program order, so
        // no s_barrier (and no vmcnt drain) is needed: a wavefront-scope fence keeps the compiler from moving
        // LDS/global accesses across the region boundary and emits no instruction.
#if defined(DW_REGION_SYNCTHREADS)
        __syncthreads();
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    }
    template <class F> DW_HD void simt(F &&f) const { par(f); }
};
DW_HD int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// x of lane J of the caller's 32-lane half (ds_swizzle broadcast: cross-lane only, no LDS memory); x of lane `lane`
template <int J> DW_HD float half_bcast(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), J << 5)); }
DW_HD float lane_bcast(float x, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane)); }
}  // namespace dw
#else
#if !defined(DW_HOST_SHIM_HEADER)
#error "dw_wave.h is device code (hipcc).  The host emulation is test infrastructure: tests/emul/dw_wave_host.h, selected with -DDW_HOST_SHIM_HEADER (tests/emul/Makefile)"
#endif
#include DW_HOST_SHIM_HEADER
#endif
