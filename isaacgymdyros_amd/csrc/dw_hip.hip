// dw_hip.hip -- the C-ABI of include/dyros_walk.h (libdyroswalk_hip.so): argument validation, the read-only model / mocap tables
// in device memory, dispatch.  One launch per policy step.  The step and substep kernels live in their own translation units:
// the octet kernels (DwConfig.pipeline 0 / 3: 8 lanes per env, 8 envs per wavefront, two wavefronts per SIMD; dw_oct*.h,
// dw_oct_kernels.hip).  This file holds dw_k_reset, the kernel behind dw_reset_idx (one wavefront per listed env; the resets
// inside a step are the step kernels' own).  Nothing here allocates, synchronises or copies per call.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dw_params.h"

#include "dw_handle.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
// (for the other translation unit with C-ABI entry points, dw_amp.hip; not exported)
extern "C" __attribute__((visibility("hidden"))) void dw_set_error(int code, const char *msg) { (void)code; snprintf(g_err, sizeof(g_err), "%s", msg); }
static int fail_hip(const char *what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return DW_EHIP;
}

// The limb schedule and its tables (dw_quad_model.h), built by the octet unit.
namespace dwq {
struct QuadModel;
int  build_quadmodel_host(const dw::DevModel *hm, const DwModel *model, QuadModel **out, const char **err);      // malloc'ed
size_t quadmodel_bytes();
}  // namespace dwq
// The octet kernels (DwConfig.pipeline = 3): dw_oct_kernels.hip.
namespace dwo {
void launch_step(bool terrain, int gpu_flavour, int wave_build, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                 const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev);
void launch_simulate(bool terrain, int wave_build, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                     const DwBuffers &B, const float *tau, const float *push);
int  oct_lds_bytes();
int  sc_park_words();
int  device_simds();
int  waves(int num_envs);
}  // namespace dwo
// The hex instantiation of the same source (16 lanes per env; launches of at most one wavefront per SIMD, N <= 4096): dw_hex_kernels.hip.
namespace dwx {
void launch_step(bool terrain, int gpu_flavour, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                 const DwBuffers &B, const float *mocap, const float *actions, const float *noise, long long step, const long long *step_dev);
void launch_simulate(bool terrain, int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P,
                     const DwBuffers &B, const float *tau, const float *push);
int  hex_lds_bytes();
int  waves(int num_envs);
}  // namespace dwx

// the coarse bound table of the height field (dw_physics.h terrain_bound), one thread per cell; runs once, at dw_bind
__global__ __launch_bounds__(256) void dw_k_terrain_bound(const int16_t *hs, int rows, int cols, int cell, int reach, int hm_rows, int hm_cols, int16_t *out) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= hm_rows * hm_cols) return;
    out[i] = dw::terrain_bound_cell(hs, rows, cols, cell, reach, i / hm_cols, i % hm_cols);
}

__global__ __launch_bounds__(64) void dw_k_reset(const dw::DevModel *M, const dw::DevParams *P, const float *noise,
                                                 long long step, const int32_t *ids, int n) {
    __shared__ dw::TaskLds S;
    dw::Wave w;
    const int e = ids[blockIdx.x];
    if (e < 0 || e >= P->C.num_envs) return;     // wave-uniform: the whole workgroup leaves
    dw::TaskBuffers T;
    T.b = &P->B; T.actions = nullptr; T.noise = noise; T.mocap = P->mocap; T.step = step;
    dw::reset_only_env(w, S, *M, P->C, T, e);
}

// The handle's tables, parameter block and the caller's buffers live on the device that was current at dw_create.
// Entry points that touch the device make that device current for the duration of the call (one process may drive
// several GPUs, or torch may have switched devices in between) and restore the caller's choice afterwards.
struct DeviceGuard {
    int prev = -1, want;
    explicit DeviceGuard(int dev) : want(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

extern "C" {

int dw_abi_version(void) { return DW_ABI_VERSION; }
const char *dw_last_error(void) { return g_err; }
void dw_default_config(DwConfig *c) { if (c) dw::default_config(c); }

int dw_create(const DwConfig *cfg, const DwModel *model, const DwTaskConst *task, DwHandle **out) {
    if (!cfg || !model || !out) return fail(DW_EINVAL, "dw_create: null argument");
    if (const char *m = dw::check_config(cfg)) return fail(DW_EINVAL, m);
    if (task && (!task->kp || !task->kv || !task->action_high || !task->initial_dof_pos || !task->mocap ||
                 !task->obs_mean || !task->obs_var || !task->dof_armature_nominal || !task->dof_damping_nominal))
        return fail(DW_EINVAL, "dw_create: task constants incomplete");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(DW_EHIP, "dw_create: no HIP device visible (this library has no CPU path)");
    DwHandle *h = (DwHandle *)calloc(1, sizeof(DwHandle));
    if (h) h->next_step = -1;
    if (!h) return fail(DW_ENOMEM, "dw_create: out of host memory");
    h->cfg = *cfg;
    h->params = dw::make_task_params(cfg);
    dw::DevModel *hm = (dw::DevModel *)malloc(sizeof(dw::DevModel));
    if (!hm) { free(h); return fail(DW_ENOMEM, "dw_create: out of host memory"); }
    const char *err = "";
    int rc = dw::build_devmodel(model, task, hm, &err);
    if (rc) { free(hm); free(h); return fail(rc, err); }
    h->pipeline = cfg->pipeline == 0 ? DW_DEFAULT_PIPELINE : cfg->pipeline;
    h->reach = dw::model_reach(*hm);
    dwq::QuadModel *hq = nullptr;
    rc = dwq::build_quadmodel_host(hm, model, &hq, &err);
    if (rc) { free(hm); free(h); return fail(rc, err); }
    (void)hipGetDevice(&h->device);
    e = hipMalloc((void **)&h->d_model, sizeof(dw::DevModel));
    if (e == hipSuccess) e = hipMemcpy(h->d_model, hm, sizeof(dw::DevModel), hipMemcpyHostToDevice);
    free(hm);
    if (e == hipSuccess && hq) {
        e = hipMalloc((void **)&h->d_qmodel, dwq::quadmodel_bytes());
        if (e == hipSuccess) e = hipMemcpy(h->d_qmodel, hq, dwq::quadmodel_bytes(), hipMemcpyHostToDevice);
    }
    free(hq);
    if (e != hipSuccess) { dw_destroy(h); return fail_hip("dw_create: model upload", e); }
    // which lane layout runs this handle's launches: the hex instantiation (16 lanes per env) when all its wavefronts find a SIMD of
    // their own -- N <= 4096 on an MI355X -- or when a test asks for it (debug_wave_build = 3); the octet kernels otherwise
    h->hex = cfg->debug_wave_build == 3 || (cfg->debug_wave_build == 0 && dwx::waves(cfg->num_envs) <= dwo::device_simds());
    {
        const size_t waves = (size_t)(h->hex ? dwx::waves(cfg->num_envs) : dwo::waves(cfg->num_envs));
        e = hipMalloc((void **)&h->d_sc_park, waves * 64 * dwo::sc_park_words() * sizeof(float));
        if (e != hipSuccess) { dw_destroy(h); return fail_hip("dw_create: self-collision park buffer", e); }
        h->params.phys.sc_park = h->d_sc_park;
    }
    if (cfg->terrain && cfg->terrain_curriculum) {
        const size_t bytes = sizeof(unsigned long long) * dw::lvl_acc_words(cfg->terrain_num_types);
        e = hipMalloc((void **)&h->d_lvl_acc, bytes);
        if (e == hipSuccess) e = hipMemset(h->d_lvl_acc, 0, bytes);
        if (e != hipSuccess) { dw_destroy(h); return fail_hip("dw_create: curriculum level sums", e); }
        h->params.terrain_lvl_acc = h->d_lvl_acc;
    }
    e = hipMalloc((void **)&h->d_params, sizeof(dw::DevParams));
    if (e == hipSuccess) e = hipMemset(h->d_params, 0, sizeof(dw::DevParams));
    if (e == hipSuccess) e = hipMemcpy(&h->d_params->C, &h->params, sizeof(dw::TaskParams), hipMemcpyHostToDevice);
    if (e != hipSuccess) { dw_destroy(h); return fail_hip("dw_create: parameter upload", e); }
    if (task) {
        const size_t bytes = sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS;
        e = hipMalloc((void **)&h->d_mocap, bytes);
        if (e == hipSuccess) e = hipMemcpy(h->d_mocap, task->mocap, bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(&h->d_params->mocap, &h->d_mocap, sizeof(float *), hipMemcpyHostToDevice);
        if (e != hipSuccess) { dw_destroy(h); return fail_hip("dw_create: mocap upload", e); }
        h->has_task = 1;
    }
    *out = h;
    return DW_OK;
}

int dw_destroy(DwHandle *h) {
    if (!h) return fail(DW_EINVAL, "dw_destroy: null handle");
    DeviceGuard guard(h->device);
    if (h->d_model) (void)hipFree(h->d_model);
    if (h->d_qmodel) (void)hipFree(h->d_qmodel);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_mocap) (void)hipFree(h->d_mocap);
    if (h->d_sc_park) (void)hipFree(h->d_sc_park);
    if (h->d_hmax) (void)hipFree(h->d_hmax);
    if (h->d_lvl_acc) (void)hipFree(h->d_lvl_acc);
    if (h->d_amp_args) (void)hipFree(h->d_amp_args);
    free(h);
    return DW_OK;
}

int dw_bind(DwHandle *h, const DwBuffers *b) {
    if (!h || !b) return fail(DW_EINVAL, "dw_bind: null argument");
    if (const char *m = dw::check_buffers(b, false)) return fail(DW_EINVAL, m);
    if (const char *m = dw::check_terrain_buffers(&h->cfg, b)) return fail(DW_EINVAL, m);
    DeviceGuard guard(h->device);
    // height field: the coarse bound table is built from the NEW samples first (height_samples is read at bind: bind again after
    // changing the terrain); only when that has succeeded are the old table freed and the handle / parameter block switched over, so
    // a failed rebind leaves the handle as it was (bound to the old buffers and the old, still allocated table)
    int16_t *new_hmax = nullptr;
    int cell = 0, hr = 0, hc = 0;
    hipError_t e;
    if (h->cfg.terrain) {
        cell = dw::hm_cell_samples(h->cfg.terrain_hscale);
        const int reach = dw::hm_reach_samples(h->cfg.terrain_hscale, h->reach);
        hr = (h->cfg.terrain_rows + cell - 1) / cell; hc = (h->cfg.terrain_cols + cell - 1) / cell;
        e = hipMalloc((void **)&new_hmax, sizeof(int16_t) * (size_t)hr * hc);
        if (e != hipSuccess) return fail_hip("dw_bind: terrain bound table", e);
        hipLaunchKernelGGL(dw_k_terrain_bound, dim3((hr * hc + 255) / 256), dim3(256), 0, 0, b->height_samples, h->cfg.terrain_rows, h->cfg.terrain_cols,
                           cell, reach, hr, hc, new_hmax);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(new_hmax); return fail_hip("dw_bind: terrain bound kernel", e); }
    }
    dw::PhysParams phys = h->params.phys;          // (sc_park: set at dw_create)
    phys.hs = h->cfg.terrain ? b->height_samples : nullptr;
    phys.hmax = new_hmax;
    if (new_hmax) { phys.hm_cell = cell; phys.hm_rows = hr; phys.hm_cols = hc; }
    // bind time, not step time: two small synchronous copies into the parameter block (the pointer table, the physics parameters)
    e = hipMemcpy(&h->d_params->C.phys, &phys, sizeof(dw::PhysParams), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(&h->d_params->B, b, sizeof(DwBuffers), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        // the device block may now be half new: nothing may be launched against it until a bind succeeds
        h->bound = 0;
        if (new_hmax) (void)hipFree(new_hmax);
        return fail_hip("dw_bind: parameter block upload", e);
    }
    if (h->d_hmax) (void)hipFree(h->d_hmax);
    h->d_hmax = new_hmax;
    h->params.phys = phys;
    h->buf = *b;
    h->bound = 1;
    return DW_OK;
}

int dw_simulate(DwHandle *h, const float *tau, const float *push_xy, void *stream) {
    if (!h || !h->bound) return fail(DW_ESTATE, "dw_simulate: buffers not bound");
    if (!tau) return fail(DW_EINVAL, "dw_simulate: tau is null");
    if (h->cfg.debug_freeze_physics) return DW_OK;
    DeviceGuard guard(h->device);
    if (h->hex) dwx::launch_simulate(h->cfg.terrain != 0, h->cfg.num_envs, (hipStream_t)stream, h->d_qmodel, h->d_model, h->d_params, h->buf, tau, push_xy);
    else dwo::launch_simulate(h->cfg.terrain != 0, h->cfg.debug_wave_build, h->cfg.num_envs, (hipStream_t)stream, h->d_qmodel, h->d_model, h->d_params, h->buf, tau, push_xy);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip("dw_simulate: launch", e);
    return DW_OK;
}

// advances the device-resident step counter of dw_step_dev behind the step kernel (same stream)
__global__ void dw_k_bump(long long *counter) { *counter += 1; }

static int launch_step(DwHandle *h, const float *actions, const float *noise, long long step_index, const long long *step_dev, void *stream,
                       const char *who, float *obs_out = nullptr) {
    if (!h || !h->bound || !h->has_task) return fail(DW_ESTATE, "dw_step: handle has no task constants or no buffers bound");
    if (const char *m = dw::check_buffers(&h->buf, true)) return fail(DW_ESTATE, m);
    if (!actions) return fail(DW_EINVAL, "dw_step: actions is null");
    if (step_index < 0) return fail(DW_EINVAL, "dw_step: negative step index");
    DeviceGuard guard(h->device);
    // The curriculum's logging table rotates three slots by step index: step s adds into slot s % 3 and clears slot (s + 1) % 3, which is
    // right while the index advances by one per launch.  A caller that repeats or jumps the index would add into a slot nobody cleared (and
    // repeated adds could carry the 32-bit level sum into the count): the library owns the table, so it clears the slot this launch adds into.
    // (With a device counter the index advances by itself.  The perturbation gate's slots live in the caller's gate_acc and are the caller's
    // to restore together with the rest of the state: include/dyros_walk.h.)
    if (!step_dev) {
        if (h->d_lvl_acc && h->next_step >= 0 && step_index != h->next_step) {
            const size_t row = (size_t)h->cfg.terrain_num_types * dw::LVL_BUCKETS;
            hipError_t e0 = hipMemsetAsync(h->d_lvl_acc + (size_t)(step_index % 3) * row, 0, row * sizeof(unsigned long long), (hipStream_t)stream);
            if (e0 != hipSuccess) return fail_hip("dw_step: clearing the terrain log slot", e0);
        }
        h->next_step = step_index + 1;
    }
    // (the kernels take the buffer table by value: this launch's copy may name another observation buffer)
    DwBuffers bufs = h->buf;
    if (obs_out) bufs.obs_buf = obs_out;
    // the torch flavour of the post phase's norms is compiled into the step kernels
    const int flavour = h->cfg.torch_gpu_div != 0 ? 1 : 0;
    if (h->hex) dwx::launch_step(h->cfg.terrain != 0, flavour, h->cfg.num_envs, (hipStream_t)stream, h->d_qmodel, h->d_model, h->d_params, bufs, h->d_mocap,
                                 actions, noise, step_index, step_dev);
    else dwo::launch_step(h->cfg.terrain != 0, flavour, h->cfg.debug_wave_build, h->cfg.num_envs, (hipStream_t)stream, h->d_qmodel, h->d_model, h->d_params, bufs, h->d_mocap,
                          actions, noise, step_index, step_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(who, e);
    return DW_OK;
}

int dw_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *stream) {
    return launch_step(h, actions, noise, (long long)step_index, nullptr, stream, "dw_step: launch");
}

int dw_step_dev(DwHandle *h, const float *actions, const float *noise, int64_t *step_counter, void *stream) {
    if (!step_counter) return fail(DW_EINVAL, "dw_step_dev: step_counter is null");
    const int rc = launch_step(h, actions, noise, 0, (const long long *)step_counter, stream, "dw_step_dev: launch");
    if (rc != DW_OK) return rc;
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(dw_k_bump, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long *)step_counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip("dw_step_dev: counter launch", e);
    return DW_OK;
}

int dw_step_obs(DwHandle *h, const float *actions, const float *noise, int64_t step_index, int64_t *step_counter, float *obs_out, void *stream) {
    if (!obs_out) return fail(DW_EINVAL, "dw_step_obs: obs_out is null");
    const int rc = launch_step(h, actions, noise, step_counter ? 0 : (long long)step_index, (const long long *)step_counter, stream, "dw_step_obs: launch", obs_out);
    if (rc != DW_OK || !step_counter) return rc;
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(dw_k_bump, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long *)step_counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip("dw_step_obs: counter launch", e);
    return DW_OK;
}

// The curriculum's logging columns: out[e] = the env's DW_NUM_REW reward columns, then for every terrain type the mean level
// of its envs as the newest step left them (sum and count from the step kernel's accumulators; float(sum) / float(max(count, 1))).
__global__ __launch_bounds__(256) void dw_k_terrain_log(const float *__restrict__ stacked, const unsigned long long *__restrict__ acc, int types, int n, int gpu_div, float *__restrict__ out) {
    extern __shared__ float mean[];          // [types]
    const int row = types * dw::LVL_BUCKETS;
    for (int t = threadIdx.x; t < types; t += 256) {
        const unsigned long long *slot = acc + acc[3 * row] * row;
        unsigned long long w = 0;
        for (int b = 0; b < dw::LVL_BUCKETS; ++b) w += slot[b * types + t];
        const unsigned int cnt = (unsigned int)(w >> 32), sum = (unsigned int)w;
        // (torch-ROCm divides a tensor by a host scalar as a * (1 / b), BinaryDivTrueKernel; torch on the CPU divides: DwConfig.torch_gpu_div)
        const float fc = (float)(cnt ? cnt : 1u);
        mean[t] = gpu_div ? (float)sum * (1.0f / fc) : (float)sum / fc;
    }
    __syncthreads();
    const int width = DW_NUM_REW + types;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * width) return;
    const int e = (int)(i / width), c = (int)(i - (long long)e * width);
    out[i] = c < DW_NUM_REW ? stacked[(size_t)e * DW_NUM_REW + c] : mean[c - DW_NUM_REW];
}

int dw_terrain_log(DwHandle *h, float *out, void *stream) {
    if (!h || !h->bound || !h->has_task) return fail(DW_ESTATE, "dw_terrain_log: handle has no task constants or no buffers bound");
    if (!h->d_lvl_acc) return fail(DW_ESTATE, "dw_terrain_log: the handle was created without a terrain curriculum");
    if (!out) return fail(DW_EINVAL, "dw_terrain_log: out is null");
    DeviceGuard guard(h->device);
    const int types = h->cfg.terrain_num_types;
    const long long total = (long long)h->cfg.num_envs * (DW_NUM_REW + types);
    hipLaunchKernelGGL(dw_k_terrain_log, dim3((unsigned)((total + 255) / 256)), dim3(256), sizeof(float) * types, (hipStream_t)stream, h->buf.stacked_rewards, h->d_lvl_acc, types,
                       h->cfg.num_envs, h->cfg.torch_gpu_div != 0 ? 1 : 0, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip("dw_terrain_log: launch", e);
    return DW_OK;
}

int dw_reset_idx(DwHandle *h, const int32_t *env_ids, int32_t n, const float *noise, int64_t step_index, void *stream) {
    if (!h || !h->bound || !h->has_task) return fail(DW_ESTATE, "dw_reset_idx: handle has no task constants or no buffers bound");
    if (const char *m = dw::check_buffers(&h->buf, true)) return fail(DW_ESTATE, m);
    if (n < 0 || (n > 0 && !env_ids)) return fail(DW_EINVAL, "dw_reset_idx: bad env id list");
    if (n == 0) return DW_OK;
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(dw_k_reset, dim3(n), dim3(64), 0, (hipStream_t)stream, h->d_model, h->d_params, noise,
                       (long long)step_index, env_ids, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip("dw_reset_idx: launch", e);
    return DW_OK;
}

int dw_oct_lds_bytes(void) { return dwo::oct_lds_bytes(); }
int dw_hex_lds_bytes(void) { return dwx::hex_lds_bytes(); }

}  // extern "C"
