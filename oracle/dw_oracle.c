/*
 * dw_oracle.c -- CPU ORACLE of the DyrosDynamicWalk step.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement, one env at a time, of
 *   - the physics substep (dw_physics.c; parity UNPINNED against the closed PhysX engine, pinned by
 *     known-answer tests instead -- see dw_physics.h), and
 *   - the reference's task logic around it (dw_task.c; pinned by golden vectors recorded from the
 *     reference's own Python running in this container, tests/golden/).
 * It exports the C-ABI of include/dyros_walk.h with the prefix dwo_ and HOST pointers, so the tests can
 * drive the HIP library and this file with the same inputs.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never does.
 */
#include "dw_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[256] = "";
int dwo_fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
#define fail dwo_fail

int dwo_abi_version(void) { return DW_ABI_VERSION; }
const char *dwo_last_error(void) { return g_err; }
int dwo_real_bytes(void) { return (int)sizeof(real); }

void dwo_default_config(DwConfig *c) {
    memset(c, 0, sizeof(*c));
    c->num_envs = 64;
    c->dt = 0.002;
    c->control_freq_inv = 2;
    c->gravity[0] = 0; c->gravity[1] = 0; c->gravity[2] = -9.81f;
    c->solver_iterations = 5;
    c->contact_offset = 0.002f;
    c->max_depenetration_velocity = 10.0f;
    c->friction = 1.0f;
    c->erp = 0.2f;
    c->contact_cfm = 1e-3f;
    c->penalty_stiffness = 1.0e5f;
    c->penalty_damping = 1.0e3f;
    c->max_angular_velocity = 100.0f;
    c->max_episode_length = 8000.0f;
    c->initial_height = 0.93f;
    c->death_cost = 0.0f;
    c->perturb = 1;
    c->force_perturb_start = 0;
    c->randomize_dof_on_reset = 1;
    c->dr_damping_add[0] = 0.0f; c->dr_damping_add[1] = 2.9f;
    c->dr_armature_scale[0] = 0.8f; c->dr_armature_scale[1] = 1.2f;
    c->randomize_friction_on_reset = 0;
    c->dr_friction_scale[0] = 0.7f; c->dr_friction_scale[1] = 1.3f;
    c->timeout_fix = 0;
    c->root_vel_at_com = 1;
    c->torch_gpu_div = 0;
    c->self_collision = 1;
    c->seed = 42;
}

int dwo_create(const DwConfig *cfg, const DwModel *model, const DwTaskConst *task, DwHandle **out) {
    if (!cfg || !model || !out) return fail(DW_EINVAL, "dwo_create: null argument");
    if (cfg->num_envs <= 0) return fail(DW_EINVAL, "dwo_create: num_envs must be positive");
    if (cfg->control_freq_inv != 2) return fail(DW_EINVAL, "dwo_create: only controlFrequencyInv = 2 is supported");
    DwHandle *h = (DwHandle *)calloc(1, sizeof(DwHandle));
    if (!h) return fail(DW_ENOMEM, "dwo_create: out of memory");
    h->cfg = *cfg;
    h->model = *model;
    dwo_model_to_real(model, &h->rmodel);
    if (task) {
        memcpy(h->kp, task->kp, sizeof(h->kp));
        memcpy(h->kv, task->kv, sizeof(h->kv));
        memcpy(h->action_high, task->action_high, sizeof(h->action_high));
        memcpy(h->initial_dof_pos, task->initial_dof_pos, sizeof(h->initial_dof_pos));
        memcpy(h->obs_mean, task->obs_mean, sizeof(h->obs_mean));
        memcpy(h->obs_var, task->obs_var, sizeof(h->obs_var));
        memcpy(h->nominal_armature, task->dof_armature_nominal, sizeof(h->nominal_armature));
        memcpy(h->nominal_damping, task->dof_damping_nominal, sizeof(h->nominal_damping));
        h->mocap = (float *)malloc(sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        if (!h->mocap) { free(h); return fail(DW_ENOMEM, "dwo_create: out of memory"); }
        memcpy(h->mocap, task->mocap, sizeof(float) * DW_MOCAP_ROWS * DW_MOCAP_COLS);
        h->has_task = 1;
    }
    *out = h;
    return DW_OK;
}

int dwo_destroy(DwHandle *h) {
    if (!h) return fail(DW_EINVAL, "dwo_destroy: null handle");
    free(h->mocap);
    free(h);
    return DW_OK;
}

int dwo_bind(DwHandle *h, const DwBuffers *b) {
    if (!h || !b) return fail(DW_EINVAL, "dwo_bind: null argument");
    if (!b->root_states || !b->dof_state || !b->contact_forces || !b->mass_scale || !b->dof_damping ||
        !b->dof_armature || !b->friction_scale)
        return fail(DW_EINVAL, "dwo_bind: physics buffers missing");
    if (h->cfg.terrain && (!b->height_samples || h->cfg.terrain_rows < 2 || h->cfg.terrain_cols < 2 || !(h->cfg.terrain_hscale > 0)))
        return fail(DW_EINVAL, "dwo_bind: terrain configured without height samples / grid");
    if (h->cfg.terrain_curriculum && (!b->terrain_origins || !b->terrain_levels || !b->terrain_types))
        return fail(DW_EINVAL, "dwo_bind: terrain curriculum without level/type/origin buffers");
    h->buf = *b;
    h->bound = 1;
    return DW_OK;
}

void dwo_load_phys(const DwHandle *h, int e, DwoPhysIO *io) {
    const DwBuffers *b = &h->buf;
    for (int i = 0; i < 13; ++i) io->root[i] = b->root_states[13 * e + i];
    for (int j = 0; j < DW_NUM_DOF; ++j) {
        io->q[j] = b->dof_state[(DW_NUM_DOF * e + j) * 2];
        io->qd[j] = b->dof_state[(DW_NUM_DOF * e + j) * 2 + 1];
        io->damping[j] = b->dof_damping[DW_NUM_DOF * e + j];
        io->armature[j] = b->dof_armature[DW_NUM_DOF * e + j];
    }
    for (int k = 0; k < DW_NUM_BODIES; ++k) io->mass_scale[k] = b->mass_scale[DW_NUM_BODIES * e + k];
    io->mu = h->cfg.friction * b->friction_scale[e];
    io->height_samples = h->cfg.terrain ? b->height_samples : NULL;
}

void dwo_store_phys(const DwHandle *h, int e, const DwoPhysIO *io) {
    const DwBuffers *b = &h->buf;
    for (int i = 0; i < 13; ++i) b->root_states[13 * e + i] = (float)io->root[i];
    for (int j = 0; j < DW_NUM_DOF; ++j) {
        b->dof_state[(DW_NUM_DOF * e + j) * 2] = (float)io->q[j];
        b->dof_state[(DW_NUM_DOF * e + j) * 2 + 1] = (float)io->qd[j];
    }
    for (int k = 0; k < DW_NUM_BODIES * 3; ++k) b->contact_forces[DW_NUM_BODIES * 3 * e + k] = (float)io->contact[k];
}

int dwo_simulate(DwHandle *h, const float *tau, const float *push_xy, void *stream) {
    (void)stream;
    if (!h || !h->bound) return fail(DW_ESTATE, "dwo_simulate: buffers not bound");
    if (!tau) return fail(DW_EINVAL, "dwo_simulate: tau is null");
    const int N = h->cfg.num_envs;
    if (h->cfg.debug_freeze_physics) return DW_OK;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        DwoPhysIO io;
        dwo_load_phys(h, e, &io);
        for (int j = 0; j < DW_NUM_DOF; ++j) io.tau[j] = tau[DW_NUM_DOF * e + j];
        io.push[0] = push_xy ? push_xy[2 * e] : 0;
        io.push[1] = push_xy ? push_xy[2 * e + 1] : 0;
        float *warm = h->buf.env_state ? h->buf.env_state + (size_t)DW_ES_WORDS * e + DW_ES_WARM : NULL;
        for (int i = 0; i < 24; ++i) io.warm[i] = warm ? warm[i] : 0;
        dwo_phys_substep(&h->cfg, &h->rmodel, &io);
        if (warm) for (int i = 0; i < 24; ++i) warm[i] = (float)io.warm[i];
        dwo_store_phys(h, e, &io);
    }
    return DW_OK;
}

/* Debug: unconstrained forward dynamics of env e (no state update).  qdd [33], a0 [6] = base spatial
 * acceleration in base coordinates (linear part = classical acceleration minus w x v, gravity included). */
int dwo_forward_dynamics(DwHandle *h, int e, const float *tau33, double *qdd, double *a0) {
    if (!h || !h->bound) return fail(DW_ESTATE, "dwo_forward_dynamics: buffers not bound");
    DwoPhysIO io;
    dwo_load_phys(h, e, &io);
    for (int j = 0; j < DW_NUM_DOF; ++j) io.tau[j] = tau33[j];
    io.push[0] = io.push[1] = 0;
    for (int i = 0; i < 24; ++i) io.warm[i] = 0;
    dwo_phys_substep(&h->cfg, &h->rmodel, &io);
    for (int j = 0; j < DW_NUM_DOF; ++j) qdd[j] = io.qdd_free[j];
    for (int i = 0; i < 6; ++i) a0[i] = io.a0_free[i];
    return DW_OK;
}
