// dw_lane_kernels.hip -- the gfx950 entry points of the lane kernels (bodies: dw_lane_kernels.h, dw_lane.h, dw_lane_post.h)
// and their launchers.  A translation unit of its own; linked into libdyroswalk_hip.so next to dw_hip.hip, which owns the C-ABI.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "dw_params.h"
#include "dw_lane_kernels.h"

// The whole VecTask.step of 64 envs per workgroup: four wavefronts, one per limb, one lane per env (grid = ceil(N / 64)
// workgroups of 256 threads).  157 KB of LDS per workgroup: one workgroup per CU, one wave per SIMD, so a wave may use the
// whole register file (512 VGPRs + AGPRs).  The buffer table travels split as for the octet kernels (dw_bufg.h).
// (GPUF: the torch flavour of the post phase's norms, compiled in -- dw_oct_post.h)
template <bool TERRAIN, int GPUF>
__global__ __launch_bounds__(dwl::NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dw_k_step_lane(const dwl::LaneModel *__restrict__ LM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB, const float *mocap,
                    const float *actions, const float *noise, long long step, const long long *step_dev) {
    __shared__ dwl::LLds L;
    if (step_dev) step = *step_dev;          // (dw_step_dev: the counter lives in device memory so that a captured launch can be replayed)
    dwl::lane_step<TERRAIN, GPUF>(L, *LM, *M, P->C, make_obuf(HB, &P->B), actions, mocap, noise, step, (int)blockIdx.x);
}
// One physics substep at the Gym boundary, same layout.
template <bool TERRAIN>
__global__ __launch_bounds__(dwl::NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dw_k_simulate_lane(const dwl::LaneModel *__restrict__ LM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB, const float *tau,
                        const float *push) {
    __shared__ dwl::LLds L;
    dwl::lane_simulate<TERRAIN>(L, *LM, *M, P->C.phys, P->C.friction, P->C.num_envs, make_obuf(HB, &P->B), tau, push, (int)blockIdx.x);
}

namespace dwl {

static int groups(int num_envs) { return (num_envs + EPW - 1) / EPW; }

void launch_step(bool terrain, int gpu_flavour, int num_envs, hipStream_t stream, const LaneModel *LM, const dw::DevModel *M, const dw::DevParams *P,
                 const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev) {
    const dim3 grid(groups(num_envs)), block(NT);
    // gpu_flavour < 0: the build that reads every switch at run time (injected noise record, frozen physics: tests)
#define DWL_LAUNCH(T, F) hipLaunchKernelGGL((dw_k_step_lane<T, F>), grid, block, 0, stream, LM, M, P, make_hot(B), mocap, actions, noise, step, step_dev)
    if (terrain) { if (gpu_flavour < 0) DWL_LAUNCH(true, -1); else if (gpu_flavour) DWL_LAUNCH(true, 1); else DWL_LAUNCH(true, 0); }
    else { if (gpu_flavour < 0) DWL_LAUNCH(false, -1); else if (gpu_flavour) DWL_LAUNCH(false, 1); else DWL_LAUNCH(false, 0); }
#undef DWL_LAUNCH
}
void launch_simulate(bool terrain, int num_envs, hipStream_t stream, const LaneModel *LM, const dw::DevModel *M, const dw::DevParams *P,
                     const DwBuffers &B, const float *tau, const float *push) {
    const dim3 grid(groups(num_envs)), block(NT);
    if (terrain) hipLaunchKernelGGL((dw_k_simulate_lane<true>), grid, block, 0, stream, LM, M, P, make_hot(B), tau, push);
    else hipLaunchKernelGGL((dw_k_simulate_lane<false>), grid, block, 0, stream, LM, M, P, make_hot(B), tau, push);
}
int build_lanemodel_host(const dw::DevModel *hm, LaneModel **out, const char **err) {
    LaneModel *q = (LaneModel *)malloc(sizeof(LaneModel));
    if (!q) { *err = "out of host memory"; return DW_ENOMEM; }
    const int rc = build_lanemodel(hm, q, err);
    if (rc) { free(q); return rc; }
    *out = q;
    return DW_OK;
}
size_t lanemodel_bytes() { return sizeof(LaneModel); }
int lane_lds_bytes() { return (int)sizeof(LLds); }
size_t sc_park_floats(int num_envs) { return (size_t)num_envs * DW_MAX_SC_PAIRS * SC_PARK_WORDS; }

}  // namespace dwl
