// dw_emul_oct.cpp -- host emulation of the OCTET kernels (isaacgymdyros_amd/csrc/dw_oct.h, dw_oct_kernels.h, dw_oct_post.h): the exact
// kernel source, one fiber per lane, 64 fibers per wave, switching at every cross-lane operation (dw_quad_wave.h).
// TEST INFRASTRUCTURE ONLY: nothing in isaacgymdyros_amd/ can load it.  Exports the C-ABI with the prefix dwe_ and HOST
// pointers, so that one driver (oracle/oracle.py) runs the oracle, this library and the lane emulation.
#define DWQ_EMUL_IMPLEMENTATION
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../isaacgymdyros_amd/csrc/dw_params.h"
#include "../../isaacgymdyros_amd/csrc/dw_oct_kernels.h"

struct DwHandle {
    DwConfig cfg;
    dw::DevModel model;
    dwq::QuadModel qmodel;
    dw::DevParams dp;
    float *mocap;
    float *sc_park;
    int16_t *hmax;
    unsigned long long *lvl_acc;
    int bound;
};

static char g_err[256] = "";
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }

namespace {
struct WaveArgs { DwHandle *h; const float *a0; const float *a1; long long step; int wave; int kind; OCT_NS::OLds *lds; dw::TaskLds *tlds; };
void wave_body(void *p, int) {
    WaveArgs *w = (WaveArgs *)p;
    DwHandle *h = w->h;
    switch (w->kind) {
    case 0:
        if (h->cfg.terrain) OCT_NS::oct_simulate<true>(w->lds->w[w->wave % OCT_NS::WPG], w->lds->hot, h->qmodel, h->model, h->dp.C.phys, h->dp.C.friction, h->cfg.num_envs, make_obuf(make_hot(h->dp.B), &h->dp.B), w->a0, w->a1, w->wave);
        else OCT_NS::oct_simulate<false>(w->lds->w[w->wave % OCT_NS::WPG], w->lds->hot, h->qmodel, h->model, h->dp.C.phys, h->dp.C.friction, h->cfg.num_envs, make_obuf(make_hot(h->dp.B), &h->dp.B), w->a0, w->a1, w->wave);
        break;
    case 1: {
        // DwConfig.debug_wave_build = 1: the register-resident form of the one-wave-per-SIMD build (dw_oct_kernels.h KEEP), which the
        // HIP library uses for launches of at most one wave per SIMD; otherwise the two-waves form (what 16384 envs run)
        const bool keep = h->cfg.debug_wave_build == 1 || OCT_NS::LPE == 16;          // (the hex instantiation has the one-wave-per-SIMD form only)
        OCT_NS::OSlots &sl = w->lds->w[w->wave % OCT_NS::WPG];
        const OBuf ob = make_obuf(make_hot(h->dp.B), &h->dp.B);
        if (h->cfg.terrain && keep) OCT_NS::oct_step<true, -1, true>(sl, w->lds->hot, h->qmodel, h->model, h->dp.C, ob, w->a0, h->mocap, w->a1, w->step, w->wave);
        else if (h->cfg.terrain) OCT_NS::oct_step<true>(sl, w->lds->hot, h->qmodel, h->model, h->dp.C, ob, w->a0, h->mocap, w->a1, w->step, w->wave);
        else if (keep) OCT_NS::oct_step<false, -1, true>(sl, w->lds->hot, h->qmodel, h->model, h->dp.C, ob, w->a0, h->mocap, w->a1, w->step, w->wave);
        else OCT_NS::oct_step<false>(sl, w->lds->hot, h->qmodel, h->model, h->dp.C, ob, w->a0, h->mocap, w->a1, w->step, w->wave);
    }
        break;
    }
}
int run_waves(DwHandle *h, int kind, const float *a0, const float *a1, long long step) {
    // a workgroup = two waves with one copy of the hot tables; the waves are independent (each stages the tables itself), so
    // they run one after the other here.  Odd env counts leave the last workgroup's second wave without envs: it still runs.
    const int nw = (h->cfg.num_envs + OCT_NS::EPO * OCT_NS::WPG - 1) / (OCT_NS::EPO * OCT_NS::WPG) * OCT_NS::WPG;
    OCT_NS::OLds *lds = (OCT_NS::OLds *)aligned_alloc(64, (sizeof(OCT_NS::OLds) + 63) / 64 * 64);
    int rc = DW_OK;
    for (int w = 0; w < nw && rc == DW_OK; ++w) {
        if ((w % OCT_NS::WPG) == 0) memset(lds, 0xff, sizeof(*lds));           // NaN-fill per workgroup: a read of a never-written slot poisons the result
        WaveArgs a{h, a0, a1, step, w, kind, lds, nullptr};
        if (!dwq::run_wave(wave_body, &a)) rc = fail(DW_ESTATE, "octet emulation: lanes disagree on the number of cross-lane operations");
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
    if (rc == DW_OK) rc = dwq::build_quadmodel(&h->model, model, &h->qmodel, &err, true);
    if (rc == DW_OK) for (int s = 0; s < dwq::QS_MAX; ++s) for (int l = 0; l < 4; ++l) if (h->qmodel.in[s][l].body >= 0 && (h->qmodel.in[s][l].flags & 2)) { rc = DW_EINVAL; err = "octet kernels: the schedule parks a chain"; }
    if (rc == DW_OK && h->qmodel.nsteps != dwq::QS_MAX) { rc = DW_EINVAL; err = "octet kernels: built for a schedule of exactly QS_MAX steps"; }
    if (rc) { free(h); return fail(rc, err); }
    h->dp.C = dw::make_task_params(cfg);
    h->sc_park = (float *)calloc((size_t)((cfg->num_envs + OCT_NS::EPO * OCT_NS::WPG - 1) / (OCT_NS::EPO * OCT_NS::WPG)) * OCT_NS::WPG * 64 * OCT_NS::SC_PARK_WORDS, sizeof(float));
    h->dp.C.phys.sc_park = h->sc_park;
    if (cfg->terrain && cfg->terrain_curriculum) {
        h->lvl_acc = (unsigned long long *)calloc(dw::lvl_acc_words(cfg->terrain_num_types), sizeof(unsigned long long));
        h->dp.C.terrain_lvl_acc = h->lvl_acc;
    }
    if (task) {
        h->mocap = (float *)malloc(sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        memcpy(h->mocap, task->mocap, sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        h->dp.mocap = h->mocap;
    }
    *out = h;
    return DW_OK;
}
int dwe_destroy(DwHandle *h) { if (!h) return fail(DW_EINVAL, "null handle"); free(h->mocap); free(h->sc_park); free(h->hmax); free(h->lvl_acc); free(h); return DW_OK; }
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
    return run_waves(h, 0, tau, push_xy, 0);
}
int dwe_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *) {
    if (!h || !h->bound || !h->model.has_task) return fail(DW_ESTATE, "not ready");
    if (const char *m = dw::check_buffers(&h->dp.B, true)) return fail(DW_ESTATE, m);
    if (!actions) return fail(DW_EINVAL, "actions is null");
    dw::TaskBuffers T;
    T.b = &h->dp.B; T.actions = actions; T.noise = noise; T.mocap = h->mocap; T.step = step_index;
    dw::TaskLds *S = new dw::TaskLds;
    dw::Wave w;
    int rc = DW_OK;       // (pre_physics_step runs inside the quad kernel: quad_physics_step<.., PRE = true>)
    rc = run_waves(h, 1, actions, noise, step_index);     // with physics frozen it still runs the actuator and encoder models
    // (post_physics_step runs inside the quad kernel too: quad_physics_step<.., POST = true>)
    delete S;
    return rc;
}
int dwe_terrain_log(DwHandle *h, float *out, void *) {          // (the arithmetic of dw_k_terrain_log, dw_hip.hip)
    if (!h || !h->bound || !h->model.has_task) return fail(DW_ESTATE, "not ready");
    if (!h->lvl_acc) return fail(DW_ESTATE, "dwe_terrain_log: no terrain curriculum");
    if (!out) return fail(DW_EINVAL, "out is null");
    const int types = h->cfg.terrain_num_types, width = DW_NUM_REW + types, row = types * dw::LVL_BUCKETS;
    const unsigned long long *slot = h->lvl_acc + h->lvl_acc[3 * row] * row;
    for (int e = 0; e < h->cfg.num_envs; ++e)
        for (int c = 0; c < width; ++c) {
            if (c < DW_NUM_REW) { out[(size_t)e * width + c] = h->dp.B.stacked_rewards[(size_t)e * DW_NUM_REW + c]; continue; }
            unsigned long long w = 0;
            for (int b = 0; b < dw::LVL_BUCKETS; ++b) w += slot[b * types + (c - DW_NUM_REW)];
            const unsigned int cnt = (unsigned int)(w >> 32), sum = (unsigned int)w;
            const float fc = (float)(cnt ? cnt : 1u);
            out[(size_t)e * width + c] = h->cfg.torch_gpu_div ? (float)sum * (1.0f / fc) : (float)sum / fc;
        }
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
int dwe_lds_bytes(void) { return (int)sizeof(OCT_NS::OLds); }

// the derived schedule, for tests/test_quad_schedule.py: out[0] = nsteps, then [step][lane] bodies (outward order)
int dwe_quad_schedule(DwHandle *h, int32_t *out, int32_t cap) {
    if (!h || !out || cap < 1 + dwq::QS_MAX * 4 * 3 + 1) return fail(DW_EINVAL, "dwe_quad_schedule: buffer too small");
    int k = 0;
    out[k++] = h->qmodel.nsteps;
    for (int s = 0; s < dwq::QS_MAX; ++s) for (int l = 0; l < 4; ++l) out[k++] = h->qmodel.fk[s][l].body;
    for (int s = 0; s < dwq::QS_MAX; ++s) for (int l = 0; l < 4; ++l) out[k++] = h->qmodel.fk[s][l].psrc;
    for (int s = 0; s < dwq::QS_MAX; ++s) for (int l = 0; l < 4; ++l) out[k++] = (h->qmodel.in[s][l].flags << 16) | h->qmodel.in[s][l].gather;
    out[k++] = h->qmodel.base_gather;
    return k;
}

}  // extern "C"

#include "dw_emul_amp.inc"          // dwe_amp_step_begin / _mid / _end, dwe_amp_reset_rows / _done on this library's handle
