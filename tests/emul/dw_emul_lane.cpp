// dw_emul_lane.cpp -- host emulation of the LANE kernels (isaacgymdyros_amd/csrc/dw_lane.h, dw_lane_kernels.h, dw_lane_post.h): the
// exact kernel source, one fiber per thread, 256 fibers per workgroup, switching at every workgroup barrier and wave vote
// (dw_lane_wave.h).  TEST INFRASTRUCTURE ONLY: nothing in isaacgymdyros_amd/ can load it.  Exports the C-ABI with the prefix
// dwe_ and HOST pointers, like the emulations of the other kernel generations.
#define DWQ_EMUL_IMPLEMENTATION
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../isaacgymdyros_amd/csrc/dw_params.h"
#include "../../isaacgymdyros_amd/csrc/dw_lane_kernels.h"

struct DwHandle {
    DwConfig cfg;
    dw::DevModel model;
    dwl::LaneModel lmodel;
    dw::DevParams dp;
    float *mocap;
    float *sc_park;
    int16_t *hmax;
    int bound;
};

static char g_err[256] = "";
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }

namespace {
struct WgArgs { DwHandle *h; const float *a0; const float *a1; long long step; int group; int kind; dwl::LLds *lds; };
void wg_body(void *p, int) {
    WgArgs *w = (WgArgs *)p;
    DwHandle *h = w->h;
    const OBuf B = make_obuf(make_hot(h->dp.B), &h->dp.B);
    switch (w->kind) {
    case 0:
        if (h->cfg.terrain) dwl::lane_simulate<true>(*w->lds, h->lmodel, h->model, h->dp.C.phys, h->dp.C.friction, h->cfg.num_envs, B, w->a0, w->a1, w->group);
        else dwl::lane_simulate<false>(*w->lds, h->lmodel, h->model, h->dp.C.phys, h->dp.C.friction, h->cfg.num_envs, B, w->a0, w->a1, w->group);
        break;
    case 1:
        if (h->cfg.terrain) dwl::lane_step<true>(*w->lds, h->lmodel, h->model, h->dp.C, B, w->a0, h->mocap, w->a1, w->step, w->group);
        else dwl::lane_step<false>(*w->lds, h->lmodel, h->model, h->dp.C, B, w->a0, h->mocap, w->a1, w->step, w->group);
        break;
    }
}
int run_groups(DwHandle *h, int kind, const float *a0, const float *a1, long long step) {
    const int ng = (h->cfg.num_envs + dwl::EPW - 1) / dwl::EPW;
    dwl::LLds *lds = (dwl::LLds *)aligned_alloc(64, (sizeof(dwl::LLds) + 63) / 64 * 64);
    int rc = DW_OK;
    for (int g = 0; g < ng && rc == DW_OK; ++g) {
        memset((void *)lds, 0xff, sizeof(*lds));           // NaN-fill per workgroup: a read of a never-written word poisons the result
        WgArgs a{h, a0, a1, step, g, kind, lds};
        if (!dwl::run_workgroup(wg_body, &a)) rc = fail(DW_ESTATE, "lane emulation: threads disagree on the number of barriers");
    }
    free(lds);
    return rc;
}
}  // namespace

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
    if (rc == DW_OK) rc = dwl::build_lanemodel(&h->model, &h->lmodel, &err);
    if (rc) { free(h); return fail(rc, err); }
    h->dp.C = dw::make_task_params(cfg);
    h->sc_park = (float *)calloc((size_t)cfg->num_envs * DW_MAX_SC_PAIRS * dwl::SC_PARK_WORDS, sizeof(float));
    h->dp.C.phys.sc_park = h->sc_park;
    if (task) {
        h->mocap = (float *)malloc(sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        memcpy(h->mocap, task->mocap, sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        h->dp.mocap = h->mocap;
    }
    *out = h;
    return DW_OK;
}
int dwe_destroy(DwHandle *h) { if (!h) return fail(DW_EINVAL, "null handle"); free(h->mocap); free(h->sc_park); free(h->hmax); free(h); return DW_OK; }
int dwe_bind(DwHandle *h, const DwBuffers *b) {
    if (!h || !b) return fail(DW_EINVAL, "dwe_bind: null argument");
    if (const char *m = dw::check_buffers(b, false)) return fail(DW_EINVAL, m);
    if (const char *m = dw::check_terrain_buffers(&h->cfg, b)) return fail(DW_EINVAL, m);
    h->dp.B = *b; h->bound = 1;
    h->dp.C.phys.hs = h->cfg.terrain ? b->height_samples : nullptr;
    free(h->hmax); h->hmax = nullptr; h->dp.C.phys.hmax = nullptr;
    if (h->cfg.terrain) {          // the coarse bound table, as dw_bind builds it
        const int cell = dw::hm_cell_samples(h->cfg.terrain_hscale), reach = dw::hm_reach_samples(h->cfg.terrain_hscale, dw::model_reach(h->model));
        const int hr = (h->cfg.terrain_rows + cell - 1) / cell, hc = (h->cfg.terrain_cols + cell - 1) / cell;
        h->hmax = (int16_t *)malloc(sizeof(int16_t) * (size_t)hr * hc);
        for (int i = 0; i < hr * hc; ++i) h->hmax[i] = dw::terrain_bound_cell(b->height_samples, h->cfg.terrain_rows, h->cfg.terrain_cols, cell, reach, i / hc, i % hc);
        h->dp.C.phys.hmax = h->hmax; h->dp.C.phys.hm_cell = cell; h->dp.C.phys.hm_rows = hr; h->dp.C.phys.hm_cols = hc;
    }
    return DW_OK;
}
int dwe_simulate(DwHandle *h, const float *tau, const float *push_xy, void *) {
    if (!h || !h->bound) return fail(DW_ESTATE, "buffers not bound");
    if (!tau) return fail(DW_EINVAL, "tau is null");
    if (h->cfg.debug_freeze_physics) return DW_OK;
    return run_groups(h, 0, tau, push_xy, 0);
}
int dwe_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *) {
    if (!h || !h->bound || !h->model.has_task) return fail(DW_ESTATE, "not ready");
    if (const char *m = dw::check_buffers(&h->dp.B, true)) return fail(DW_ESTATE, m);
    if (!actions) return fail(DW_EINVAL, "actions is null");
    return run_groups(h, 1, actions, noise, step_index);
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
    T.b = &h->dp.B; T.actions = nullptr; T.noise = noise; T.mocap = h->mocap; T.step = step_index;
    dw::TaskLds *S = new dw::TaskLds;
    dw::Wave w;
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= h->cfg.num_envs) { delete S; return fail(DW_EINVAL, "env id out of range"); }
        memset((void *)S, 0xff, sizeof(*S));      // NaN-fill per id: nothing may be read that this call did not load
        dw::reset_only_env(w, *S, h->model, h->dp.C, T, ids[i]);
    }
    delete S;
    return DW_OK;
}
int dwe_lds_bytes(void) { return (int)sizeof(dwl::LLds); }

}  // extern "C"
