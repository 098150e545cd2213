// dw_quad_kernels.hip -- the gfx950 entry points of the quad kernels (bodies: dw_quad_kernels.h, dw_quad.h, dw_quad_post.h)
// and their launchers.  A translation unit of its own so that it is compiled with the default machine scheduler (see
// dw_hip.hip and isaacgymdyros_amd/build.py); linked into libdyroswalk_hip.so next to dw_hip.hip, which owns the C-ABI.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "dw_params.h"
#include "dw_quad_kernels.h"

// The whole VecTask.step of 16 envs per wavefront (grid = ceil(N / 16)): 4 lanes per env.  40 KB of LDS per wave: 4 waves
// per CU, one per SIMD, so the whole register file (256 VGPRs + 256 AGPRs) belongs to the wave.
// (DwBuffers travels BY VALUE: pointer members of a by-value kernel argument are known to be global addresses, whereas
//  pointers loaded from a parameter block in memory are generic and every access through them is a FLAT instruction --
//  counted on the LDS counter as well, so that each wait for an LDS read also waited for all global requests in flight.)
template <bool TERRAIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dw_k_step_quad(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwBuffers B, const float *mocap,
                    const float *actions, const float *noise, long long step, const long long *step_dev) {
    __shared__ dwq::QLds L;
    if (step_dev) step = *step_dev;
#if defined(DQ_STAGGER)
#ifndef DQ_STAGGER_MULT
#define DQ_STAGGER_MULT 1
#endif
          // (timing experiment: de-phase the waves' memory bursts; DQ_STAGGER = groups, DQ_STAGGER_SLEEP = s_sleep units)
    for (int i = 0; i < (int)(blockIdx.x % DQ_STAGGER) * DQ_STAGGER_MULT; ++i) __builtin_amdgcn_s_sleep(DQ_STAGGER_SLEEP);
#endif
    dwq::quad_step<TERRAIN>(L, *QM, *M, P->C, B, actions, mocap, noise, step, (int)blockIdx.x);
}
// One physics substep at the Gym boundary, same layout.
template <bool TERRAIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dw_k_simulate_quad(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwBuffers B, const float *tau,
                        const float *push) {
    __shared__ dwq::QLds L;
    dwq::quad_simulate<TERRAIN>(L, *QM, *M, P->C.phys, P->C.friction, P->C.num_envs, B, tau, push, (int)blockIdx.x);
}

namespace dwq {

void launch_step(bool terrain, int num_envs, hipStream_t stream, const QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                 const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev) {
    const dim3 grid((num_envs + EPW - 1) / EPW);
    if (terrain) hipLaunchKernelGGL(dw_k_step_quad<true>, grid, dim3(64), 0, stream, QM, M, P, B, mocap, actions, noise, step, step_dev);
    else hipLaunchKernelGGL(dw_k_step_quad<false>, grid, dim3(64), 0, stream, QM, M, P, B, mocap, actions, noise, step, step_dev);
}
void launch_simulate(bool terrain, int num_envs, hipStream_t stream, const QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                     const DwBuffers &B, const float *tau, const float *push) {
    const dim3 grid((num_envs + EPW - 1) / EPW);
    if (terrain) hipLaunchKernelGGL(dw_k_simulate_quad<true>, grid, dim3(64), 0, stream, QM, M, P, B, tau, push);
    else hipLaunchKernelGGL(dw_k_simulate_quad<false>, grid, dim3(64), 0, stream, QM, M, P, B, tau, push);
}
int build_quadmodel_host(const dw::DevModel *hm, const DwModel *model, QuadModel **out, const char **err, bool octet) {
    QuadModel *q = (QuadModel *)malloc(sizeof(QuadModel));
    if (!q) { *err = "out of host memory"; return DW_ENOMEM; }
    int rc = build_quadmodel(hm, model, q, err, octet);
    // (the octet kernels carry no parking registers: build_quadmodel(accumulate) must have scheduled every hand-over)
    if (rc == DW_OK && octet)
        for (int s = 0; s < QS_MAX; ++s) for (int l = 0; l < 4; ++l)
            if (q->in[s][l].body >= 0 && (q->in[s][l].flags & 2)) { rc = DW_EINVAL; *err = "octet kernels: the schedule parks a chain"; }
    if (rc == DW_OK && octet && q->nsteps != QS_MAX) { rc = DW_EINVAL; *err = "octet kernels: built for a schedule of exactly QS_MAX steps"; }
    if (rc) { free(q); return rc; }
    *out = q;
    return DW_OK;
}
size_t quadmodel_bytes() { return sizeof(QuadModel); }
int quad_lds_bytes() { return (int)sizeof(QLds); }

}  // namespace dwq
