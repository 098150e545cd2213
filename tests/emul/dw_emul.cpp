// dw_emul.cpp -- lane-loop emulation of the HIP kernel body on the host.  TEST INFRASTRUCTURE ONLY.
//
// Compiles isaacgymdyros_amd/csrc/dw_task.h (the exact source hipcc builds into the shipped kernels)
// with the host definition of dw::Wave (64-iteration loop per region) and exports the C-ABI with the
// prefix dwe_ and HOST pointers.  It is not a fallback of the product -- nothing in isaacgymdyros_amd/
// can load it -- it exists so that the CPU test-suite (and ASan/UBSan) exercises the kernels' indexing
// and region structure before any GPU time is spent.
#define DWQ_EMUL_IMPLEMENTATION
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../isaacgymdyros_amd/csrc/dw_params.h"

struct DwHandle {
    DwConfig cfg;
    dw::DevModel model;
    dw::TaskParams params;
    DwBuffers buf;
    float *mocap;
    int bound;
};

static char g_err[256] = "";
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }

extern "C" {

int dwe_abi_version(void) { return DW_ABI_VERSION; }
const char *dwe_last_error(void) { return g_err; }
void dwe_default_config(DwConfig *c) { dw::default_config(c); }

int dwe_create(const DwConfig *cfg, const DwModel *model, const DwTaskConst *task, DwHandle **out) {
    if (!cfg || !model || !out) return fail(DW_EINVAL, "dwe_create: null argument");
    if (const char *m = dw::check_config(cfg)) return fail(DW_EINVAL, m);
    DwHandle *h = (DwHandle *)calloc(1, sizeof(DwHandle));
    if (!h) return fail(DW_ENOMEM, "out of memory");
    h->cfg = *cfg;
    const char *err = "";
    int rc = dw::build_devmodel(model, task, &h->model, &err);
    if (rc) { free(h); return fail(rc, err); }
    h->params = dw::make_task_params(cfg);
    if (task) {
        h->mocap = (float *)malloc(sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        memcpy(h->mocap, task->mocap, sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
    }
    *out = h;
    return DW_OK;
}
int dwe_destroy(DwHandle *h) { if (!h) return fail(DW_EINVAL, "null handle"); free(h->mocap); free(h); return DW_OK; }
int dwe_bind(DwHandle *h, const DwBuffers *b) {
    if (!h || !b) return fail(DW_EINVAL, "dwe_bind: null argument");
    if (const char *m = dw::check_buffers(b, false)) return fail(DW_EINVAL, m);
    if (const char *m = dw::check_terrain_buffers(&h->cfg, b)) return fail(DW_EINVAL, m);
    h->buf = *b; h->bound = 1;
    h->params.phys.hs = h->cfg.terrain ? b->height_samples : nullptr;
    return DW_OK;
}
int dwe_simulate(DwHandle *h, const float *tau, const float *push_xy, void *) {
    if (!h || !h->bound) return fail(DW_ESTATE, "buffers not bound");
    if (!tau) return fail(DW_EINVAL, "tau is null");
    if (h->cfg.debug_freeze_physics) return DW_OK;
    dw::Lds *S = new dw::Lds;
    dw::Wave w;
    for (int e = 0; e < h->cfg.num_envs; ++e) {
        if (h->cfg.terrain) dw::simulate_env<true>(w, *S, h->model, h->params, h->buf, tau, push_xy, e);
        else dw::simulate_env<false>(w, *S, h->model, h->params, h->buf, tau, push_xy, e);
    }
    delete S;
    return DW_OK;
}
int dwe_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *) {
    if (!h || !h->bound || !h->model.has_task) return fail(DW_ESTATE, "not ready");
    if (const char *m = dw::check_buffers(&h->buf, true)) return fail(DW_ESTATE, m);
    if (!actions) return fail(DW_EINVAL, "actions is null");
    dw::TaskBuffers T;
    T.b = &h->buf; T.actions = actions; T.noise = noise; T.mocap = h->mocap; T.step = step_index;
    dw::Lds *S = new dw::Lds;
    dw::Wave w;
    for (int e = 0; e < h->cfg.num_envs; ++e) {
        if (h->cfg.terrain) dw::step_env<true>(w, *S, h->model, h->params, T, e);
        else dw::step_env<false>(w, *S, h->model, h->params, T, e);
    }
    delete S;
    return DW_OK;
}
int dwe_step_dev(DwHandle *h, const float *actions, const float *noise, int64_t *step_counter, void *stream) {
    if (!step_counter) return fail(DW_EINVAL, "step_counter is null");
    const int rc = dwe_step(h, actions, noise, *step_counter, stream);
    if (rc == DW_OK) *step_counter += 1;
    return rc;
}
int dwe_reset_idx(DwHandle *h, const int32_t *ids, int32_t n, const float *noise, int64_t step_index, void *) {
    if (!h || !h->bound || !h->model.has_task) return fail(DW_ESTATE, "not ready");
    if (n < 0 || (n > 0 && !ids)) return fail(DW_EINVAL, "bad env id list");
    dw::TaskBuffers T;
    T.b = &h->buf; T.actions = nullptr; T.noise = noise; T.mocap = h->mocap; T.step = step_index;
    dw::Lds *S = new dw::Lds;
    dw::Wave w;
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= h->cfg.num_envs) { delete S; return fail(DW_EINVAL, "env id out of range"); }
        memset((void *)S, 0xff, sizeof(*S));      // NaN-fill per id: nothing may be read that this call did not load
        dw::reset_only_env(w, *S, h->model, h->params, T, ids[i]);
    }
    delete S;
    return DW_OK;
}
int dwe_lds_bytes(void) { return (int)sizeof(dw::Lds); }

}  // extern "C"
