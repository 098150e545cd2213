/* dw_oracle.h -- CPU ORACLE internals.  TEST INFRASTRUCTURE ONLY (see dw_oracle.c). */
#ifndef DW_ORACLE_H
#define DW_ORACLE_H

#include "dw_physics.h"

struct DwHandle {
    DwConfig  cfg;
    DwModel   model;
    DwoModelR rmodel;
    DwBuffers buf;
    int       bound;
    int       has_task;
    float     kp[DW_NUM_DOF], kv[DW_NUM_DOF], action_high[DW_NUM_DOF], initial_dof_pos[DW_NUM_DOF];
    float     obs_mean[DW_NUM_OBS1], obs_var[DW_NUM_OBS1];
    float     nominal_armature[DW_NUM_DOF], nominal_damping[DW_NUM_DOF];
    float    *mocap;
};

int         dwo_fail(int code, const char *msg);
int         dwo_abi_version(void);
const char *dwo_last_error(void);
int         dwo_real_bytes(void);
void        dwo_default_config(DwConfig *cfg);
int dwo_create(const DwConfig *cfg, const DwModel *model, const DwTaskConst *task, DwHandle **out);
int dwo_destroy(DwHandle *h);
int dwo_bind(DwHandle *h, const DwBuffers *buffers);
int dwo_simulate(DwHandle *h, const float *tau, const float *push_xy, void *stream);
int dwo_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *stream);
int dwo_step_obs(DwHandle *h, const float *actions, const float *noise, int64_t step_index, int64_t *step_counter, float *obs_out, void *stream);
int dwo_reset_idx(DwHandle *h, const int32_t *env_ids, int32_t n, const float *noise, int64_t step_index,
                  void *stream);
int dwo_forward_dynamics(DwHandle *h, int e, const float *tau33, double *qdd, double *a0);

void dwo_load_phys(const DwHandle *h, int e, DwoPhysIO *io);
void dwo_store_phys(const DwHandle *h, int e, const DwoPhysIO *io);

#endif
