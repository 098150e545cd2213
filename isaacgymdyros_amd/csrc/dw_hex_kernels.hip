// dw_hex_kernels.hip -- the gfx950 entry points of the HEX instantiation of the octet kernels: the same source (dw_oct.h,
// dw_oct_kernels.h, dw_oct_post.h) compiled with 16 lanes per env (OCT_LPE = 16: one DPP row per env, four quarters of four limb
// lanes, 4 envs per wavefront).  For launches of at most 4096 envs on an MI355X (BASELINE config 2): there the octet layout has
// 512 wavefronts for 1024 SIMDs; this one has 1024, each with half the envs' map / item work per lane.  One wave per SIMD by
// construction, so there is one build (the register-resident form of the step, KEEP).  A translation unit of its own, built with
// the flags of dw_oct_kernels.hip (isaacgymdyros_amd/build.py).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#define OCT_LPE 16
// The rolled chain loops (kinematics and the two velocity / acceleration passes: 11 steps each) two steps per iteration: the lone wave
// of a SIMD is bound by the latency of its dependent chains, and with two steps in one block the scheduler starts a step's table and
// slot reads under the previous step's arithmetic (A/B at 4096 envs: -1.7 %; 3 steps per iteration: +2.6 %, 4: as 2; the octet build
// for two waves per SIMD loses 2 % to it -- a second wave hides that latency already and the longer code costs instruction fetch).
#define OCT_CHAIN_UNROLL 2
// spatial 6-vectors of the chain passes and the contact phase as three register pairs on the packed fp32 instructions (dw_limb.h V6)
#define OQ_PACKED 1
#include "dw_params.h"
#include "dw_oct_kernels.h"

template <bool TERRAIN, int GPUF>
__global__ __launch_bounds__(64 * dwx::WPG) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dw_k_step_hex(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB, const float *mocap,
                   const float *actions, const float *noise, long long step, const long long *step_dev) {
    __shared__ dwx::OLds L;
    if (step_dev) step = *step_dev;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    dwx::oct_step<TERRAIN, GPUF, true>(L.w[w], L.hot, *QM, *M, P->C, make_obuf(HB, &P->B), actions, mocap, noise, step, (int)blockIdx.x * dwx::WPG + w);
}
template <bool TERRAIN>
__global__ __launch_bounds__(64 * dwx::WPG) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dw_k_simulate_hex(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB, const float *tau,
                       const float *push) {
    __shared__ dwx::OLds L;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    dwx::oct_simulate<TERRAIN>(L.w[w], L.hot, *QM, *M, P->C.phys, P->C.friction, P->C.num_envs, make_obuf(HB, &P->B), tau, push, (int)blockIdx.x * dwx::WPG + w);
}

namespace dwx {

int groups(int num_envs) { return (num_envs + EPO * WPG - 1) / (EPO * WPG); }
int waves(int num_envs) { return groups(num_envs) * WPG; }

void launch_step(bool terrain, int gpu_flavour, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                 const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev) {
    const dim3 grid(groups(num_envs)), block(64 * WPG);
    if (terrain && gpu_flavour) hipLaunchKernelGGL((dw_k_step_hex<true, 1>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
    else if (terrain) hipLaunchKernelGGL((dw_k_step_hex<true, 0>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
    else if (gpu_flavour) hipLaunchKernelGGL((dw_k_step_hex<false, 1>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
    else hipLaunchKernelGGL((dw_k_step_hex<false, 0>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
}
void launch_simulate(bool terrain, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                     const DwBuffers &B, const float *tau, const float *push) {
    const dim3 grid(groups(num_envs)), block(64 * WPG);
    if (terrain) hipLaunchKernelGGL((dw_k_simulate_hex<true>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
    else hipLaunchKernelGGL((dw_k_simulate_hex<false>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
}
int hex_lds_bytes() { return (int)sizeof(OLds); }

}  // namespace dwx
