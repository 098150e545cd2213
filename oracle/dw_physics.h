/*
 * dw_physics.h -- CPU ORACLE, physics part.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar restatement of the simulation substep that stands in for the reference's closed
 * `gym.simulate` (reference call site: tasks/dyros_dynamic_walk.py:525; parameters:
 * cfg/task/DyrosDynamicWalk.yaml:37-56, tasks/dyros_dynamic_walk.py:286-291,363-373,
 * cfg/terrain/terrain_cfg.py:2-9; robot: assets/mjcf/dyros_tocabi/xml/dyros_tocabi.xml).
 *
 * PARITY UNPINNED against PhysX: the engine is a closed binary that is not in the reference checkout
 * (SURVEY.md section 8c), so this file fixes the physics by written decision (DESIGN.md "Physics
 * model") and is itself pinned by known-answer tests (tests/test_oracle_physics.py): pendulum period,
 * free fall, momentum conservation, ABA vs an independently written Jacobian/RNEA formulation in
 * numpy fp64, static stance load.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 *
 * Algorithm per env and substep (dt = 2 ms):
 *   1. forward kinematics + body-frame spatial velocities (floating base + 33 hinges)
 *   2. penalty ground forces for every non-sole collision primitive (deepest point of each), and penalty
 *      self-collision forces between capsule proxies of the two legs (closest points of the two segments)
 *   3. Featherstone articulated-body algorithm, armature and implicit joint damping on the
 *      diagonal (D = S'IaS + armature + dt*damping), gravity as a uniform field
 *   4. velocity-level contact solve for the 8 sole corners: Delassus matrix from 12 unit-wrench
 *      responses of the two foot bodies (delta-ABA), projected Gauss-Seidel with Coulomb cone,
 *      warm started, speculative margin for separated points, ERP*depth/dt bias for penetration
 *   5. impulse propagation through the whole tree, joint-velocity clamp, semi-implicit Euler,
 *      joint-limit clamp, quaternion exponential update
 */
#ifndef DW_PHYSICS_H
#define DW_PHYSICS_H

#include "../include/dyros_walk.h"

#ifdef DWO_DOUBLE
typedef double real;
#else
typedef float real;
#endif

typedef struct DwoPhysIO {
    /* state, Gym layout */
    real root[13];
    real q[DW_NUM_DOF], qd[DW_NUM_DOF];
    /* inputs */
    real tau[DW_NUM_DOF];
    real push[2];
    /* per-env parameters */
    real mass_scale[DW_NUM_BODIES];
    real damping[DW_NUM_DOF], armature[DW_NUM_DOF];
    real mu;
    /* warm start, in/out: impulses [8][3] (x,y,z; on terrain: tangent 1, tangent 2, normal) */
    real warm[DW_NUM_FOOT_PTS * 3];
    /* terrain: Terrain.heightsamples [terrain_rows, terrain_cols] or NULL for the plane z = 0 (row f-4) */
    const int16_t *height_samples;
    /* outputs */
    real contact[DW_NUM_BODIES * 3];
    /* debug outputs of the unconstrained dynamics */
    real qdd_free[DW_NUM_DOF];
    real a0_free[6];     /* base spatial acceleration, body coords, gravity included in the linear part */
} DwoPhysIO;

/* DwModel with `real`-typed numeric arrays, so that the fp64 build does its arithmetic in double. */
typedef struct DwoGeomR {
    int type, moving, gym, sole;
    real pos[3], rot[9], size[3];
} DwoGeomR;
typedef struct DwoModelR {
    int  mv_parent[DW_NUM_MOVING];
    real mv_pos[DW_NUM_MOVING][3], mv_rot0[DW_NUM_MOVING][9], mv_axis[DW_NUM_MOVING][3];
    real dof_lower[DW_NUM_DOF], dof_upper[DW_NUM_DOF], dof_vmax[DW_NUM_DOF];
    int  inert_mv[DW_NUM_INERT], inert_gym[DW_NUM_INERT];
    real inert_mass[DW_NUM_INERT], inert_com[DW_NUM_INERT][3], inert_I[DW_NUM_INERT][6];
    int  num_geoms;
    DwoGeomR geoms[DW_MAX_GEOMS];
    int  foot_mv[DW_NUM_FOOT_PTS], foot_gym[DW_NUM_FOOT_PTS];
    real foot_pos[DW_NUM_FOOT_PTS][3];
    int  num_sc_proxies, num_sc_pairs;
    int  sc_moving[DW_MAX_SC_PROXIES], sc_gym[DW_MAX_SC_PROXIES];
    real sc_p0[DW_MAX_SC_PROXIES][3], sc_p1[DW_MAX_SC_PROXIES][3], sc_radius[DW_MAX_SC_PROXIES];
    int  sc_pair[DW_MAX_SC_PAIRS][2];
} DwoModelR;

void dwo_model_to_real(const DwModel *m, DwoModelR *r);
void dwo_phys_substep(const DwConfig *cfg, const DwoModelR *m, DwoPhysIO *io);

#endif
