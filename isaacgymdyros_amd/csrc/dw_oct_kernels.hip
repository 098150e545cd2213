// dw_oct_kernels.hip -- the gfx950 entry points of the octet kernels (bodies: dw_oct_kernels.h, dw_oct.h, dw_oct_post.h) and
// their launchers.  A translation unit of its own (built with -mllvm -amdgpu-sched-strategy=iterative-ilp, isaacgymdyros_amd/build.py);
// linked into libdyroswalk_hip.so next to dw_hip.hip, which owns the C-ABI.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <string.h>

#include "dw_params.h"
#include "dw_handle.h"
#include "dw_oct_kernels.h"
#include "dw_amp_step.h"

// The whole VecTask.step of 8 envs per wavefront, 8 lanes per env; a workgroup is two wavefronts that share one copy of the
// hot tables and nothing else (grid = ceil(N / 16) workgroups of 128 threads).  40 KB of LDS per workgroup: 4 workgroups =
// 8 waves per CU, TWO per SIMD, so a wave may use 256 registers (VGPRs + AGPRs; the register file is unified on gfx950).
// (The buffer table travels split: DwHot by value -- the pointers of the physics and the item loops, global not generic as kernel
// arguments -- and the rest read from the parameter block through DwBuffersG: dw_bufg.h.)
// Two builds of each kernel from the same source.  WPE = 2: two waves per SIMD (the register budget the code is written for),
// for launches with more waves than the device has SIMDs.  WPE = 1: declared occupancy one wave per SIMD -- the same 250
// registers, but the hardware then never puts two of the launch's waves on one SIMD while another SIMD is idle, which it
// otherwise does as soon as a CU holds two workgroups (measured at 8192 envs: 0.1329 ms against 0.1428 ms).
// GPUF: the torch flavour (DwConfig.torch_gpu_div) the post phase sums its norms in, compiled in (dw_oct_post.h).
template <bool TERRAIN, int WPE, int GPUF>
__global__ __launch_bounds__(64 * dwo::WPG) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void dw_k_step_oct(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB, const float *mocap,
                   const float *actions, const float *noise, long long step, const long long *step_dev) {
    __shared__ dwo::OLds L;
    if (step_dev) step = *step_dev;          // (dw_step_dev: the counter lives in device memory so that a captured launch can be replayed)
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));          // (wave-uniform: keep it in a scalar register)
#if defined(OCT_STAGGER_SHIFT)      // (timing experiment: hold back every other group of workgroups so that the two waves of a SIMD are in different phases)
    if ((blockIdx.x >> OCT_STAGGER_SHIFT) & 1) for (int i = 0; i < OCT_STAGGER_SLEEP; ++i) __builtin_amdgcn_s_sleep(127);
#endif
#if defined(OCT_FORCE_SCRATCH)      // (timing experiment: what a launch pays for HAVING private memory, with none of it on a hot path)
    volatile int pad[4];
    pad[threadIdx.x & 3] = (int)step;
#endif
    dwo::oct_step<TERRAIN, GPUF, WPE == 1>(L.w[w], L.hot, *QM, *M, P->C, make_obuf(HB, &P->B), actions, mocap, noise, step, (int)blockIdx.x * dwo::WPG + w);
#if defined(OCT_FORCE_SCRATCH)
    if (pad[(threadIdx.x + 1) & 3] == 0x7fffffff) __builtin_trap();
#endif
}
// One physics substep at the Gym boundary, same layout.
template <bool TERRAIN, int WPE>
__global__ __launch_bounds__(64 * dwo::WPG) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void dw_k_simulate_oct(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB, const float *tau,
                       const float *push) {
    __shared__ dwo::OLds L;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    dwo::oct_simulate<TERRAIN>(L.w[w], L.hot, *QM, *M, P->C.phys, P->C.friction, P->C.num_envs, make_obuf(HB, &P->B), tau, push, (int)blockIdx.x * dwo::WPG + w);
}

// The whole TocabiAMPLower step in ONE launch (row f-3; tasks/amp/tocabi_amp_lower_base.py:642-804): the task's regions (dw_amp_step.h:
// action clamp / history / command ramp / torques | encoder model + next torques | encoder model, counters, foot positions, observation,
// reward, termination, discriminator observation, histories) around K physics substeps of the octet kernels.  An octet workgroup is two
// wavefronts = 16 envs, and so is the task code's env group: the same 16 envs, so what one part leaves in the envs' rows in global memory
// the next part of the same workgroup finds there after a workgroup barrier -- no grid-wide hand-over, no launch boundary.  The task
// regions run with 128 threads per group (EnvGroupT<128>; the four serial per-env functions two to a wavefront) instead of the 256 of the
// three separate kernels; LDS is the octet slots and the task's staging rows in turn (a union: 40.9 KB, 8 waves per CU as before).
// Until round 6 a step was begin | simulate | mid | simulate | end = five launches of a replayed graph (0.206 ms at 16384 envs).
// (the k-th of eight pointers held in registers: selects, no indexed private array)
__device__ __forceinline__ const float *zsel(const float *const (&z)[8], int k) {
    const float *r = z[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) r = k == i ? z[i] : r;
    return r;
}
struct AmpZ { const float *z[8]; };          // the caller's encoder draws, one [N, 33] array per substep (NULL: device draws / no noise)
struct AmpArgs { DwAmpConfig C; DwAmpBuffers B; };          // the task's two tables in device memory (DwHandle::d_amp_args)
static_assert(sizeof(AmpArgs) <= sizeof(((DwHandle *)nullptr)->amp_args_host), "DwHandle::amp_args_host holds the task's tables");
__global__ __launch_bounds__(64 * dwo::WPG) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dw_k_amp_step_oct(const dwq::QuadModel *__restrict__ QM, const dw::DevModel *__restrict__ M, const dw::DevParams *__restrict__ P, const DwHot HB,
                       const AmpArgs *__restrict__ A, const dwa::GymRows G, const float *actions_in, const int64_t *ramp_dur, const float *ramp_u,
                       const float *z0, const float *z1, const float *z2, const float *z3, const float *z4, const float *z5, const float *z6, const float *z7,
                       const float *rootvel_noise, int K) {
    const DwAmpConfig &C = A->C;
    const DwAmpBuffers &B = A->B;
    const float *const zs[8] = {z0, z1, z2, z3, z4, z5, z6, z7};
    static_assert(dwo::WPG == 2 && dwo::EPO * dwo::WPG == dwa::GE, "an octet workgroup and an env group are the same 16 envs");
    union Lds { dwo::OLds L; dwa::GroupLds G; dwa::BeginLds Bg; __device__ Lds() {} };
    __shared__ Lds S;
    using WG = dwa::EnvGroupT<64 * dwo::WPG>;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), group = (int)blockIdx.x;
    dwa::step_begin(WG(), S.Bg, C, B, G.dof_state, actions_in, ramp_dur, ramp_u, group);
    for (int k = 0; k < K; ++k) {
        // (every region ends with a workgroup barrier: the torques of the group's envs are in DwAmpBuffers.tau, the state in the Gym tensors)
        dwo::oct_simulate<false>(S.L.w[w], S.L.hot, *QM, *M, P->C.phys, P->C.friction, P->C.num_envs, make_obuf(HB, &P->B), B.tau, nullptr, group * dwo::WPG + w);
        __syncthreads();
        if (k + 1 < K) dwa::step_mid(WG(), C, B, G.dof_state, zsel(zs, k), k, group);
    }
    dwa::step_end(WG(), S.G, *M, C, B, G, zsel(zs, K - 1), K - 1, rootvel_noise, group);
}

namespace dwq {
// the limb schedule and its tables (dw_quad_model.h) for the octet kernels, built on the host at dw_create
int build_quadmodel_host(const dw::DevModel *hm, const DwModel *model, QuadModel **out, const char **err) {
    QuadModel *q = (QuadModel *)malloc(sizeof(QuadModel));
    if (!q) { *err = "out of host memory"; return DW_ENOMEM; }
    int rc = build_quadmodel(hm, model, q, err, true);
    // (the octet kernels carry no parking registers: build_quadmodel(accumulate) must have scheduled every hand-over)
    if (rc == DW_OK)
        for (int s = 0; s < QS_MAX; ++s) for (int l = 0; l < 4; ++l)
            if (q->in[s][l].body >= 0 && (q->in[s][l].flags & 2)) { rc = DW_EINVAL; *err = "octet kernels: the schedule parks a chain"; }
    if (rc == DW_OK && q->nsteps != QS_MAX) { rc = DW_EINVAL; *err = "octet kernels: built for a schedule of exactly QS_MAX steps"; }
    if (rc) { free(q); return rc; }
    *out = q;
    return DW_OK;
}
size_t quadmodel_bytes() { return sizeof(QuadModel); }
}  // namespace dwq

namespace dwo {

static int groups(int num_envs) { return (num_envs + EPO * WPG - 1) / (EPO * WPG); }

// waves of the launch <= SIMDs of the device: the one-wave-per-SIMD build (wave_build: DwConfig.debug_wave_build, 1 / 2 force a build)
int device_simds() {
    static int simds = 0;
    if (!simds) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        simds = 4 * cus;
    }
    return simds;
}
int waves(int num_envs) { return groups(num_envs) * WPG; }
static bool spread(int num_envs, int wave_build) {
    if (wave_build == 1) return true;
    if (wave_build == 2) return false;
    return groups(num_envs) * WPG <= device_simds();
}
template <int GPUF>
static void launch_step_f(bool terrain, int wave_build, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                          const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev) {
    const dim3 grid(groups(num_envs)), block(64 * WPG);
    const bool sp = spread(num_envs, wave_build);
    if (terrain && sp) hipLaunchKernelGGL((dw_k_step_oct<true, 1, GPUF>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
    else if (terrain) hipLaunchKernelGGL((dw_k_step_oct<true, 2, GPUF>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
    else if (sp) hipLaunchKernelGGL((dw_k_step_oct<false, 1, GPUF>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
    else hipLaunchKernelGGL((dw_k_step_oct<false, 2, GPUF>), grid, block, 0, stream, QM, M, P, make_hot(B), mocap, actions, noise, step, step_dev);
}
void launch_step(bool terrain, int gpu_flavour, int wave_build, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                 const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev) {
    if (gpu_flavour) launch_step_f<1>(terrain, wave_build, num_envs, stream, QM, M, P, B, mocap, actions, noise, step, step_dev);
    else launch_step_f<0>(terrain, wave_build, num_envs, stream, QM, M, P, B, mocap, actions, noise, step, step_dev);
}
void launch_simulate(bool terrain, int wave_build, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                     const DwBuffers &B, const float *tau, const float *push) {
    const dim3 grid(groups(num_envs)), block(64 * WPG);
    const bool sp = spread(num_envs, wave_build);
    if (terrain && sp) hipLaunchKernelGGL((dw_k_simulate_oct<true, 1>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
#if defined(OCT_SKIP_TSIM2)          // (A/B builds only)
    else if (terrain) hipLaunchKernelGGL((dw_k_simulate_oct<true, 1>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
#else
    else if (terrain) hipLaunchKernelGGL((dw_k_simulate_oct<true, 2>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
#endif
    else if (sp) hipLaunchKernelGGL((dw_k_simulate_oct<false, 1>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
    else hipLaunchKernelGGL((dw_k_simulate_oct<false, 2>), grid, block, 0, stream, QM, M, P, make_hot(B), tau, push);
}
void launch_amp_step(int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P, const DwBuffers &Bf, const void *d_args,
                     const float *actions_in, const int64_t *ramp_dur, const float *ramp_u, const float *const *z, int K, const float *rootvel_noise) {
    AmpZ Z;
    for (int k = 0; k < 8; ++k) Z.z[k] = (z && k < K) ? z[k] : nullptr;
    const dwa::GymRows G{Bf.root_states, Bf.dof_state, Bf.contact_forces, Bf.dof_damping, Bf.dof_armature};
    hipLaunchKernelGGL(dw_k_amp_step_oct, dim3(groups(num_envs)), dim3(64 * WPG), 0, stream, QM, M, P, make_hot(Bf), (const AmpArgs *)d_args, G, actions_in, ramp_dur, ramp_u,
                       Z.z[0], Z.z[1], Z.z[2], Z.z[3], Z.z[4], Z.z[5], Z.z[6], Z.z[7], rootvel_noise, K);
}
// the task's tables in the handle's device copy: copied when they differ from what is there (in steady state never: the tables of an env do
// not change between steps), on the launch's stream so that the launch behind it reads the new ones
int amp_args_to_device(DwHandle *h, const DwAmpConfig &C, const DwAmpBuffers &B, hipStream_t stream) {
    AmpArgs a;
    memset(&a, 0, sizeof a);
    a.C = C; a.B = B;
    if (!h->d_amp_args && hipMalloc(&h->d_amp_args, sizeof(AmpArgs)) != hipSuccess) return DW_ENOMEM;
    if (!h->amp_args_valid || memcmp(h->amp_args_host, &a, sizeof a) != 0) {
        memcpy(h->amp_args_host, &a, sizeof a);
        if (hipMemcpyAsync(h->d_amp_args, h->amp_args_host, sizeof a, hipMemcpyHostToDevice, stream) != hipSuccess) { h->amp_args_valid = 0; return DW_EHIP; }
        h->amp_args_valid = 1;
    }
    return DW_OK;
}
int oct_lds_bytes() { return (int)sizeof(OLds); }
int sc_park_words() { return SC_PARK_WORDS; }

}  // namespace dwo
