/*
 * dyros_walk.h -- C-ABI of the MI355X-native DyrosDynamicWalk simulation step.
 *
 * This is the drop-in boundary.  The reference task talks to a closed
 * native engine through the Isaac Gym tensor API; this library stands where that engine (and the
 * eager torch task logic around it) stood.  Every pointer in DwBuffers is DEVICE memory owned by
 * the caller (the Python host allocates it with torch), every call is enqueued on the hipStream_t
 * passed as `stream` (void* so the header needs no HIP include), nothing here synchronises the
 * device, allocates per call, or throws: all entry points return 0 on success and a negative
 * DW_E* code otherwise, with dw_last_error() holding the message.
 *
 * Reference interface each entry point replaces (paths relative to
 * python/IsaacGymEnvs/isaacgymenvs unless noted):
 *
 *   dw_create / dw_bind      gym.create_sim + load_asset + create_env/create_actor loop + prepare_sim +
 *                            acquire_{actor_root_state,dof_state,net_contact_force}_tensor
 *                            (tasks/base/vec_task.py:259-275,196; tasks/dyros_dynamic_walk.py:199-225,
 *                            272-385,73-107) -- except that state buffers are caller-owned here, so the
 *                            gymtorch.wrap_tensor aliasing shim (python/isaacgym/_bindings/src/gymtorch/
 *                            gymtorch.cpp:33-158) has no counterpart.
 *   dw_simulate              gym.set_dof_actuation_force_tensor + gym.apply_rigid_body_force_tensors +
 *                            gym.simulate + gym.refresh_{dof_state,actor_root_state,net_contact_force}_tensor
 *                            (tasks/dyros_dynamic_walk.py:502,520,525-526,547-549): ONE physics substep.
 *   dw_step                  VecTask.step: pre_physics_step + 2x simulate + post_physics_step
 *                            (tasks/base/vec_task.py:293-344; tasks/dyros_dynamic_walk.py:449-563,581-669,
 *                            750-947): ONE launch on the caller's stream, whichever kernels
 *                            DwConfig.pipeline selects.  During the launch the env's row of obs_buf is scratch
 *                            (it is rewritten with the new observation at the end of the same launch).
 *   dw_reset_idx             DyrosDynamicWalk.reset_idx + set_actor_root_state_tensor_indexed +
 *                            set_dof_state_tensor_indexed (tasks/dyros_dynamic_walk.py:598-669,720-748),
 *                            as reached from VecTask.reset_done (tasks/base/vec_task.py:376-391).
 *   dw_set_dof_properties    gym.set_actor_dof_properties / set_actor_rigid_body_properties as used by the
 *   (plain buffer writes)    domain randomisation (tasks/base/vec_task.py:655-721): the per-env parameter
 *                            arrays in DwBuffers ARE the properties; the host writes them directly.
 *
 * The CPU oracle (oracle/dw_oracle.c) exports the same functions with the prefix dwo_ and host
 * pointers, so one host class and one test-suite drive both.
 */
#ifndef DYROS_WALK_H
#define DYROS_WALK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DW_ABI_VERSION 10

/* ---- fixed sizes of the TOCABI model (reference: assets/mjcf/dyros_tocabi/xml/dyros_tocabi.xml) ---- */
#define DW_NUM_BODIES   38   /* Gym rigid bodies, XML depth-first                         */
#define DW_NUM_MOVING   34   /* floating base + 33 hinge links (welded feet merged)       */
#define DW_NUM_DOF      33
#define DW_NUM_INERT    36   /* bodies that carry an <inertial>                           */
#define DW_MAX_GEOMS    64
#define DW_NUM_FOOT_PTS  8   /* 4 sole corners per foot                                   */
#define DW_MAX_SC_PROXIES 16  /* capsule proxies for self-collision                        */
#define DW_MAX_SC_PAIRS   64  /* proxy pairs tested each substep                           */
#define DW_NUM_ACT      13   /* 12 leg torques + 1 gait-clock action                      */
#define DW_NUM_LOWER    12
#define DW_NUM_OBS1     37   /* single-step observation                                   */
#define DW_AMP_NUM_OBS1 36   /* TocabiAMPLower single-step observation (tasks/amp/tocabi_amp_lower_base.py:45) */
#define DW_AMP_DISC_BASE 28  /* discriminator observation per step without the key bodies: root height 1 + root euler 3 + leg
                               dof pos 12 + leg dof vel 12 (tasks/tocabi_amp_lower.py:47); + 3 per key body (two feet: 34) */
#define DW_AMP_NUM_ACT  12   /* TocabiAMPLower actions: the 12 leg torques (:46)          */
#define DW_NUM_HIS      10
#define DW_NUM_SKIP      2
#define DW_HIST_SLOTS   20   /* NumHis*NumSkip                                            */
#define DW_NUM_OBS     487   /* (37+13)*(10-1)+37, tasks/dyros_dynamic_walk.py:43         */
#define DW_NUM_REW      15   /* 14 reward terms + perturbation flag column                */
#define DW_ALOG_SLOTS    6   /* round(0.01/dt)+1 torque FIFO, tasks/dyros_dynamic_walk.py:166 */
#define DW_MOCAP_ROWS 3600
#define DW_MOCAP_COLS   36
#define DW_GATE_BUCKETS 32
#define DW_GATE_LATCH  192   /* 3 slots * 32 buckets * 2 words */
#define DW_GATE_WORDS  256

#define DW_OK            0
#define DW_EINVAL       -1
#define DW_ENOMEM       -2
#define DW_EHIP         -3
#define DW_ESTATE       -4

typedef struct DwGeom {
    int32_t type;        /* 0 box (size = half extents), 1 cylinder (size = radius, half height; axis = local z) */
    int32_t moving;      /* moving body that carries the primitive */
    int32_t gym;         /* Gym body the contact force is attributed to */
    int32_t sole;        /* 1 for the two *_Foot_Link sole boxes (handled by the contact solve) */
    float   pos[3];      /* primitive centre in the moving body's frame */
    float   rot[9];      /* primitive axes in the moving body's frame, row-major */
    float   size[3];
    float   _pad;
} DwGeom;

/* Capsule proxy of a link for self-collision (the reference collides every primitive with every other one:
 * create_actor(..., group=i, filter=0), tasks/dyros_dynamic_walk.py:354).  Segment end points in the moving body's
 * frame.  Shipped model: 4 proxies per leg (all 16 left x right pairs), upper arm / forearm / hand per arm, the torso and
 * the head (forearm and hand against torso, head and same-side thigh; hand against the other thigh and the same-side shank;
 * forearm against the other thigh; upper arm against torso, same-side thigh and the other arm; arm against arm):
 * 16 proxies (DW_MAX_SC_PROXIES, the table is full), 47 pairs of at most DW_MAX_SC_PAIRS = 64. */
typedef struct DwCapsule {
    int32_t moving, gym;
    float   p0[3], p1[3];
    float   radius;
} DwCapsule;

typedef struct DwModel {
    int32_t mv_parent[DW_NUM_MOVING];
    int32_t mv_gym[DW_NUM_MOVING];
    int32_t mv_depth[DW_NUM_MOVING];
    float   mv_pos[DW_NUM_MOVING][3];    /* body origin in the parent's frame                      */
    float   mv_rot0[DW_NUM_MOVING][9];   /* fixed body rotation (body coords -> parent coords)     */
    float   mv_axis[DW_NUM_MOVING][3];   /* hinge axis in body coords (entry 0 unused)             */
    float   dof_lower[DW_NUM_DOF];
    float   dof_upper[DW_NUM_DOF];
    float   dof_vmax[DW_NUM_DOF];        /* dof_prop['velocity'] = 4.03, tasks/dyros_dynamic_walk.py:372 */
    int32_t inert_mv[DW_NUM_INERT];
    int32_t inert_gym[DW_NUM_INERT];
    float   inert_mass[DW_NUM_INERT];
    float   inert_com[DW_NUM_INERT][3];  /* COM in the moving body's frame                         */
    float   inert_I[DW_NUM_INERT][6];    /* xx yy zz xy xz yz about the COM, moving body's frame   */
    int32_t num_geoms;
    DwGeom  geoms[DW_MAX_GEOMS];
    int32_t foot_mv[DW_NUM_FOOT_PTS];
    int32_t foot_gym[DW_NUM_FOOT_PTS];
    float   foot_pos[DW_NUM_FOOT_PTS][3];
    int32_t left_foot_gym, right_foot_gym, pelvis_gym;
    int32_t num_sc_proxies;
    DwCapsule sc_proxy[DW_MAX_SC_PROXIES];
    int32_t num_sc_pairs;
    int32_t sc_pair[DW_MAX_SC_PAIRS][2];   /* indices into sc_proxy */
} DwModel;

/* Task constants the reference hard-codes (tasks/dyros_dynamic_walk.py:58-70,95-100,296-301) or loads
 * from data files (:112-113,139-142).  HOST pointers, copied at dw_create. */
typedef struct DwTaskConst {
    const float *kp;            /* [33]  already divided by 9  */
    const float *kv;            /* [33]  already divided by 3  */
    const float *action_high;   /* [33]                         */
    const float *initial_dof_pos; /* [33]                       */
    const float *mocap;         /* [3600*36] float32 rows       */
    const float *obs_mean;      /* [37]                         */
    const float *obs_var;       /* [37]                         */
    const float *dof_armature_nominal; /* [33] tasks/dyros_dynamic_walk.py:366-371 (DR scales these)   */
    const float *dof_damping_nominal;  /* [33] = 0.1, tasks/dyros_dynamic_walk.py:365 (DR adds to these) */
} DwTaskConst;

typedef struct DwConfig {
    double  dt;                         /* sim.dt = 0.002 (cfg/task/DyrosDynamicWalk.yaml:38); double so that the
                                           python-side constant folding (dt*skipframe, 5*dt_policy, 8/dt_policy) is exact */
    int32_t num_envs;
    int32_t control_freq_inv;           /* env.controlFrequencyInv = 2 (:11); only 2 is supported     */
    float   gravity[3];                 /* sim.gravity (:42)                                          */
    int32_t solver_iterations;          /* physx.num_position_iterations + num_velocity_iterations    */
    float   contact_offset;             /* physx.contact_offset = 0.002 (:49)                         */
    float   max_depenetration_velocity; /* physx.max_depenetration_velocity = 10 (:52)                */
    float   friction;                   /* TerrainCfg static=dynamic friction = 1 (cfg/terrain/terrain_cfg.py:7-8) */
    float   erp;                        /* fraction of penetration removed per step (written decision) */
    float   contact_cfm;                /* relative diagonal regularisation of the Delassus matrix    */
    float   penalty_stiffness;          /* non-foot ground contact: N/m                               */
    float   penalty_damping;            /* non-foot ground contact: N s/m                             */
    float   max_angular_velocity;       /* asset_options.max_angular_velocity = 100 (tasks/dyros_dynamic_walk.py:289) */
    float   max_episode_length;         /* episodeLength/(dt*controlFrequencyInv) = 8000.0 (:35)      */
    float   initial_height;             /* env.initialHieght = 0.93                                   */
    float   death_cost;                 /* env.deathCost = 0                                          */
    int32_t perturb;                    /* env.perturbation                                           */
    int32_t force_perturb_start;        /* the commented override at tasks/dyros_dynamic_walk.py:491  */
    int32_t randomize_dof_on_reset;     /* task.randomize: damping/armature DR at every reset         */
    float   dr_damping_add[2];          /* [0, 2.9]   (cfg/task/DyrosDynamicWalk.yaml:103-108)        */
    float   dr_armature_scale[2];       /* [0.8, 1.2] (:109-115)                                      */
    int32_t randomize_friction_on_reset;/* BASELINE config 5; parameters of the commented block :89-95 */
    float   dr_friction_scale[2];       /* [0.7, 1.3]                                                 */
    int32_t timeout_fix;                /* 0 = reproduce SURVEY quirk Q16 (time_outs identically 0)   */
    int32_t root_vel_at_com;            /* 1 = root linear velocity is the COM's (PhysX convention)   */
    int32_t torch_gpu_div;              /* 1 = `tensor / python_scalar` is tensor * (1/scalar), as torch's GPU
                                           kernels compute it; 0 = true division, as torch's CPU kernels do */
    int32_t self_collision;             /* 1 = capsule self-collision of legs, arms and torso (DwCapsule); 0 = ground contacts only */
    int32_t debug_freeze_physics;       /* 1 = simulate() leaves the state untouched (task-logic parity tests) */
    uint64_t seed;                      /* key of the counter-based in-kernel RNG                     */
    /* Terrain (reference cfg/terrain/terrain_cfg.py:1-22, tasks/dyros_dynamic_walk.py:203-270 create,
     * :671-691 curriculum, :693-708 origins, :729-732 spawn jitter).  terrain = 0 is the ground plane z = 0. */
    int32_t terrain;                    /* 1 = height field in DwBuffers.height_samples ('heightfield' and 'trimesh') */
    int32_t terrain_rows, terrain_cols; /* samples along x / y (Terrain.tot_rows, tot_cols)           */
    float   terrain_hscale;             /* horizontal_scale: sample spacing [m]                       */
    float   terrain_vscale;             /* vertical_scale: height per sample unit [m]                 */
    float   terrain_border;             /* border_size [m]: world (x,y) = index * hscale - border     */
    int32_t terrain_curriculum;         /* TerrainCfg.curriculum: change level at reset (:671-691)    */
    int32_t terrain_num_levels;         /* num_rows of the tile grid = max_terrain_level              */
    int32_t terrain_num_types;          /* num_cols of the tile grid                                  */
    float   terrain_env_length;         /* terrain_length [m]: walked more than half of it => level up */
    float   max_episode_length_s;       /* env.episodeLength as the curriculum uses it (:34, :685)    */
    int32_t custom_origins;             /* 1 = reset adds U(-1,1) m of xy jitter to the origin (:729-732) */
    int32_t pipeline;                   /* which kernels run dw_step / dw_simulate (one launch per policy step):
                                           0 = default (3); 3 = the octet kernels (8 lanes per env, 8 envs per wavefront,
                                           two waves per SIMD).  1, 2 and 4 (the wave-per-env, quad and lane kernels of
                                           rounds 1, 2 and 4) are retired: DW_EINVAL */
    int32_t debug_wave_build;           /* tests only.  The step / substep kernels exist in two builds of one source: one declared for
                                           one wavefront per SIMD (keeps its per-joint state in registers; chosen when the launch
                                           has no more wavefronts than the device has SIMDs, N <= 8192 on MI355X) and one for two
                                           (parks that state in HBM; the 16384-env path).  0 = choose by launch size; 1 / 2 = force
                                           that build, so that parity tests at small N can run the production-size code */
} DwConfig;

/* Layout of the injected-noise record, one per env per step (floats).  When the `noise` argument of
 * dw_step / dw_reset_idx is NULL the kernel draws the same logical record from Philox4x32-10 keyed by
 * (seed; env, step).  Slots hold FINAL encoder noise values (already N(0, 0.00016/3)) and raw U[0,1)
 * draws for everything else, so a recorded reference run can be replayed bit for bit. */
#define DW_NZ_ENC      0    /* [2][33] encoder noise, substep-major (tasks/dyros_dynamic_walk.py:528)     */
#define DW_NZ_VEL     66    /* [6]  root velocity noise draw (:766)                                       */
#define DW_NZ_QPOS_BIAS 72  /* [12] (:615)   */
#define DW_NZ_QUAT_BIAS 84  /* [3]  (:616)   */
#define DW_NZ_TARGET_VEL 87 /* [1]  (:624)   */
#define DW_NZ_INIT_MOCAP 88 /* [1]  (:630)   */
#define DW_NZ_MOTOR   89    /* [12] (:645)   */
#define DW_NZ_DELAY  101    /* [1]  (:652)   */
#define DW_NZ_PTIMING 102   /* [1]  (:665)   */
#define DW_NZ_DR_DAMP 103   /* [33]          */
#define DW_NZ_DR_ARM  136   /* [33]          */
#define DW_NZ_DR_FRIC 169   /* [1]           */
#define DW_NZ_PERT   170    /* [3] impulse, duration, phase (:440-443) */
#define DW_NZ_TERRAIN_LVL 173 /* [1] randint_like for robots that solved the last level (:689) */
#define DW_NZ_ROOT_JITTER 174 /* [2] spawn jitter on terrain (:732)                            */
#define DW_NOISE_WORDS 176
#define DW_NZ_UBLOCK 0x10000 /* generated (not injected) uniform words: Philox counter word of the block holding words 4b..4b+3 is DW_NZ_UBLOCK + b */

/* Per-env task-state record: DW_ES_WORDS 32-bit words, 16-byte aligned, one contiguous row per env
 * so a wavefront moves it with dwordx4 loads.  Offsets in words; `i:` marks int32 fields. */
#define DW_ES_QPOS_NOISE      0   /* [33] */
#define DW_ES_QVEL_NOISE     33   /* [33] */
#define DW_ES_QPOS_PRE       66   /* [33] */
#define DW_ES_PRE_QVEL       99   /* [33] pre_joint_velocity_states */
#define DW_ES_TARGET_QPOS   132   /* [33] target_data_qpos of the last step */
#define DW_ES_TARGET_FORCE  165   /* [2]  */
#define DW_ES_TARGET_VEL    167   /* [2]  */
#define DW_ES_MOTOR_SCALE   169   /* [12] */
#define DW_ES_QPOS_BIAS     181   /* [12] */
#define DW_ES_QUAT_BIAS     193   /* [3]  */
#define DW_ES_ACTION_LOG    196   /* [6][12] */
#define DW_ES_ACTIONS       268   /* [13] */
#define DW_ES_ACTIONS_PRE   281   /* [13] */
#define DW_ES_ACTION_TORQUE 294   /* [12] */
#define DW_ES_ACTION_TORQUE_PRE 306 /* [12] */
#define DW_ES_FOOT_FORCE_PRE 318  /* [2][3] contact_forces_pre rows of L_Foot_Link, R_Foot_Link */
#define DW_ES_TIME          324
#define DW_ES_EPI_LEN       325
#define DW_ES_EPI_LEN_LOG   326
#define DW_ES_CRS           327   /* contact_reward_sum  */
#define DW_ES_CRM           328   /* contact_reward_mean */
#define DW_ES_MAGNITUDE     329
#define DW_ES_PHASE         330
#define DW_ES_INIT_MOCAP    331   /* i: */
#define DW_ES_MOCAP_IDX     332   /* i: */
#define DW_ES_DELAY_IDX     333   /* i: */
#define DW_ES_SIMUL_LEN     334   /* i: */
#define DW_ES_PERT_COUNT    335   /* i: */
#define DW_ES_PERT_DURATION 336   /* i: */
#define DW_ES_PERT_ON       337   /* i: */
#define DW_ES_IMPULSE       338   /* i: */
#define DW_ES_PERT_TIMING   339   /* i: */
#define DW_ES_PERT_START    340   /* i: */
#define DW_ES_HIST_HEAD     341   /* i: ring position of the OLDEST history slot */
#define DW_ES_NAN_RESETS    342   /* i: count of resets forced by a non-finite state */
#define DW_ES_EPI_RETURN    343   /* sum of rew_buf over the running episode (logging; a2c_common_dyros.py:661-687 keeps it in torch) */
#define DW_ES_WARM          344   /* [8][3] contact impulses of the previous substep (warm start) */
#define DW_ES_LAST_RETURN   368   /* return of the last finished episode */
#define DW_ES_EPISODES      369   /* i: finished episodes */
#define DW_ES_WORDS         372

/* Device buffers (caller-owned).  Shapes in [] with N = num_envs; float32 unless noted. */
typedef struct DwBuffers {
    /* Gym tensor API state (python/isaacgym docs: programming/tensors) */
    float   *root_states;     /* [N,13] pos3 quat(xyzw)4 linvel3 angvel3, world frame   */
    float   *dof_state;       /* [N,33,2] (pos, vel) interleaved                        */
    float   *contact_forces;  /* [N,38,3] net contact force per body, world frame       */
    /* per-env physical parameters (domain randomisation targets) */
    float   *mass_scale;      /* [N,38] per Gym body mass multiplier                    */
    float   *dof_damping;     /* [N,33]                                                 */
    float   *dof_armature;    /* [N,33]                                                 */
    float   *friction_scale;  /* [N]                                                    */
    float   *total_mass;      /* [N]                                                    */
    float   *env_origins;     /* [N,3]                                                  */
    /* VecTask buffers (tasks/base/vec_task.py:233-256) */
    float   *obs_buf;         /* [N,487]                                                */
    float   *rew_buf;         /* [N]                                                    */
    int64_t *reset_buf;       /* [N]                                                    */
    int64_t *progress_buf;    /* [N]                                                    */
    int64_t *timeout_buf;     /* [N]                                                    */
    int64_t *randomize_buf;   /* [N]                                                    */
    float   *stacked_rewards; /* [N,15] extras["stacked_rewards"]                       */
    /* task state */
    float   *env_state;       /* [N,DW_ES_WORDS]                                        */
    float   *obs_history;     /* [N,20,37] ring, see DW_ES_HIST_HEAD                    */
    float   *action_history;  /* [N,20,13] ring                                         */
    /* cross-env statistics for the perturbation gate (tasks/dyros_dynamic_walk.py:489): 3 rotating slots x
     * 32 buckets (env % 32) of {sum epi_len_log, sum contact_reward_mean * 2^32} as int64 (integer sums are
     * order-independent, so the gate is deterministic), latch word at [DW_GATE_LATCH] */
    int64_t *gate_acc;        /* [DW_GATE_WORDS]                                        */
    /* terrain (may be NULL when DwConfig.terrain == 0) */
    int16_t *height_samples;  /* [terrain_rows, terrain_cols] Terrain.heightsamples; READ AT dw_bind as well (a coarse bound table of
                                 the field is built from it there): bind again after changing the terrain */
    float   *terrain_origins; /* [terrain_num_levels, terrain_num_types, 3] tile spawn origins */
    int64_t *terrain_levels;  /* [N] current difficulty level of each env                */
    int64_t *terrain_types;   /* [N] terrain type (column) of each env, fixed            */
} DwBuffers;

typedef struct DwHandle DwHandle;

int         dw_abi_version(void);
const char *dw_last_error(void);
void        dw_default_config(DwConfig *cfg);

int dw_create(const DwConfig *cfg, const DwModel *model, const DwTaskConst *task, DwHandle **out);
int dw_destroy(DwHandle *h);
int dw_bind(DwHandle *h, const DwBuffers *buffers);

/* One physics substep at the Gym boundary: tau [N,33] joint efforts, push_xy [N,2] world force on
 * base_link's COM (may be NULL).  Reads and writes root_states/dof_state, writes contact_forces. */
int dw_simulate(DwHandle *h, const float *tau, const float *push_xy, void *stream);

/* One VecTask.step.  actions [N,13] (clamped to +-1 inside, vec_task.py:304-307); noise [N,DW_NOISE_WORDS]
 * or NULL; step_index = number of dw_step calls made before this one.  It is expected to advance by ONE per call on a handle: the counter-based
 * generator is keyed by it, and two pieces of cross-env bookkeeping rotate three slots by it (slot = step_index % 3: this step adds, the next slot is
 * cleared, the previous one is read) -- the perturbation gate's sums in the CALLER's gate_acc, and the terrain curriculum's logging sums in a
 * library-owned table.  A repeated or jumped index is legal (replays do it): the library then clears the terrain slot the launch adds into; gate_acc is
 * caller state like every other bound buffer -- whoever rewinds the step index restores it together with env_state (DyrosDynamicWalk.state_dict /
 * load_state_dict do), otherwise the gate sees the sums of whatever step last used the slot. */
int dw_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *stream);

/* The same step with the step counter in DEVICE memory (int64, caller-owned, initialised by the caller): the kernel reads it and
 * a one-thread launch behind it adds 1.  No argument of the call changes from step to step, so a rollout step -- this call, or
 * this call with the policy network around it -- can be captured in a hipGraph once and replayed; every replay draws from the
 * next block of the counter-based generator, exactly as successive dw_step calls do.  (dw_step itself takes step_index by
 * value: replaying a captured dw_step would replay its noise.) */
int dw_step_dev(DwHandle *h, const float *actions, const float *noise, int64_t *step_counter, void *stream);

/* dw_step with the 487-word observations written to `obs_out` [N,487] (device memory) instead of DwBuffers.obs_buf, for this call
 * only.  The reference's step returns a FRESH tensor every call (torch.clamp(self.obs_buf, ...), tasks/base/vec_task.py:338): a host
 * that hands a newly allocated tensor here keeps that contract without a copy of 1 948 B per env after the kernel.  step_counter:
 * NULL (step_index is used, as dw_step) or the device counter of dw_step_dev.
 * (dw_reset_idx never touches obs_buf: a reset env's observations are rebuilt by the next step, as in the reference.) */
int dw_step_obs(DwHandle *h, const float *actions, const float *noise, int64_t step_index, int64_t *step_counter, float *obs_out,
                void *stream);

/* The logging columns of the terrain curriculum (tasks/dyros_dynamic_walk.py:415-421: the reference loops over the terrain types with a
 * nonzero() + sum + cat each): out [N, DW_NUM_REW + terrain_num_types] (device memory) = every env's row of DwBuffers.stacked_rewards
 * followed, for each terrain type, by the mean terrain level of the envs of that type as the NEWEST step found DwBuffers.terrain_levels
 * (the reference forms the columns in compute_reward, before the step's reset_idx moves any level; float(sum) / float(count), or float(sum) * (1 / float(count)) with
 * DwConfig.torch_gpu_div as torch-ROCm divides by a host scalar; count = 0 reads as 1).  The step kernels sum levels and counts per type while they run, from the bound tables -- a caller's edits of
 * terrain_levels / terrain_types between steps are seen by the next step; call this after a dw_step /
 * dw_step_dev / dw_step_obs on the same stream.  One launch.  DW_ESTATE unless DwConfig.terrain and terrain_curriculum are set. */
int dw_terrain_log(DwHandle *h, float *out, void *stream);

/* reset_idx for the env ids listed (int32, device memory). */
int dw_reset_idx(DwHandle *h, const int32_t *env_ids, int32_t n, const float *noise,
                 int64_t step_index, void *stream);

/* ---- Row f-3: env-side functions of the sibling TOCABI tasks on the same physics. -------------------
 * Stateless, one thread per env, device pointers, asynchronous on `stream`; argument order and meaning are those of the
 * reference's TorchScript functions so that a maintainer binds them at the reference's own call sites:
 *   dw_amp_observations  tasks/amp/tocabi_amp_lower_base.py:918-962  compute_humanoid_observations -> obs [N,DW_AMP_NUM_OBS1]
 *                        (key_pos, which the reference passes and never reads, is omitted)
 *   dw_amp_disc_observations  tasks/tocabi_amp_lower.py:310-350 build_amp_observations -> obs [N, DW_AMP_DISC_BASE + 3 n_key]: the
 *                        discriminator's per-step observation of the AMP subclass (root height, root euler, 12 leg dof positions and
 *                        velocities, key-body positions relative to the root in the heading frame; local_root_obs != 0: key_pos copied
 *                        as given).  dof_pos / dof_vel: 12 leading dofs of each row, rows dof_row_stride elements apart, dofs
 *                        dof_elem_stride apart (33, 1 for a [N,33] tensor; 66, 2 for the two halves of dof_state [N,33,2]; 12, 1 for
 *                        the motion library's [M,12]); key_pos [N,n_key,3]
 *   dw_amp_reward        :964-1023 compute_humanoid_reward -> reward [N], reward_values [N,9]; actions [N,12], contact_force [N,38,3]
 *   dw_amp_reset         :1025-1069 compute_humanoid_reset -> reset [N], terminated [N]; contact_body_ids: Gym bodies whose contact
 *                        does not terminate (the feet), rigid_body_pos [N,38,3] (rows 0, 8, 16 are read), rigid_body_rot [N,38,4] (row 0)
 *   dw_newwalk_reward    tasks/tocabi_new_walk.py:384-496 compute_humanoid_walk_reward -> total_reward [N], reset [N], reward8 [N,8]
 *                        (hard-codes contact rows 7 and 14 as the reference does; num_dof <= 64)
 *   dw_body_positions    world position of the origin of up to DW_MAX_BODY_QUERY MOVING bodies (HOST array of indices) of every env,
 *                        out [N,nb,3]: the rows of acquire_rigid_body_state_tensor (:108) those functions need, from the bound
 *                        root_states / dof_state */
#define DW_MAX_BODY_QUERY 8
int dw_amp_observations(int n, const float *root_states, const float *rootvel_noise, const float *dof_pos, const float *dof_pos_bias,
                        const float *quat_bias, const float *dof_vel, const float *commands, float *obs, void *stream);
int dw_amp_disc_observations(int n, const float *root_states, const float *dof_pos, const float *dof_vel, int dof_row_stride,
                             int dof_elem_stride, int local_root_obs, const float *key_pos, int n_key, float *obs, void *stream);
int dw_amp_reward(int n, const float *root_states, const float *dof_vel, const float *dof_vel_pre, const float *commands,
                  const float *actions, const float *actions_pre, const float *motor_efforts, const float *contact_force,
                  const float *total_mass, float *reward, float *reward_values, void *stream);
int dw_amp_reset(int n, const int64_t *progress_buf, const float *contact_buf, const int32_t *contact_body_ids, int n_contact_ids,
                 const float *rigid_body_pos, const float *rigid_body_rot, float max_episode_length, int enable_early_termination,
                 float termination_height, int64_t *reset, int64_t *terminated, void *stream);
int dw_newwalk_reward(int n, const int64_t *reset_buf, const int64_t *progress_buf, const float *target_vel, const float *root_pose_states,
                      const float *joint_position_states, const float *joint_velocity_states, const int32_t *non_feet_idxs, int n_non_feet,
                      const float *contact_forces, int num_bodies, float termination_height, float death_cost, float max_episode_length,
                      const float *q_nominal, int num_dof, const float *head_states, const float *lfoot_states, const float *rfoot_states,
                      const float *phase, float *total_reward, int64_t *reset, float *reward8, void *stream);
int dw_body_positions(DwHandle *h, const int32_t *moving_bodies, int nb, float *out, void *stream);

/* ---- Row f-3, fused: the bookkeeping of one TocabiAMPLower step between the physics launches, as three kernels instead of ~100
 * elementwise torch launches (tasks/amp/tocabi_amp_lower_base.py:642-804, tasks/tocabi_amp_lower.py:88-96).  The host class
 * (isaacgymdyros_amd/tocabi_amp_lower.py, cfg sim.mi355.amp_fused) calls, per step of controlFrequencyInv = K substeps:
 *   dw_amp_step_begin                 action clamp + action history, command ramp, torques of substep 0 (PD / delayed-torque FIFO)
 *   dw_simulate
 *   (K - 1) x [ dw_amp_step_mid(k), dw_simulate ]   encoder model of substep k - 1, torques of substep k
 *   dw_amp_step_end(K - 1)            encoder model of the last substep, counters, foot positions, observation + history stacking,
 *                                     reward, termination, discriminator observation history, time-outs, clamped observation
 * and dw_amp_reset_done between two steps.  One wavefront per env (csrc/dw_amp_step.h; the same source runs under g++ in
 * tests/emul/ and under ASan / UBSan).
 * Random numbers: tensors the caller drew (torch's generator, in the torch implementation's order: every expression is the one
 * of the torch implementation of the same class, so the fused step is bit-identical to it, tests/test_amp_gpu.py) -- or, with
 * DwAmpConfig.device_draws = 1 and NULL draw arguments, drawn inside the kernels (Philox4x32-10 keyed by DwAmpConfig.seed, the env
 * and the env's counter DwAmpBuffers.draw_ctr, which the kernels advance: a step recorded in a hipGraph draws fresh numbers at
 * every replay).  Histories: DwAmpConfig.hist_ring = 0 is the reference's layout (newest slot last, the whole row shifts every
 * step); 1 keeps action_history / obs_history as rings, DwAmpBuffers.hist_head [N,2] = physical slot of the oldest entry of the
 * two (logical slot i = physical slot (head + i) mod num_his * num_skip).  All pointers are device memory, [N, ...] row-major
 * float32 unless noted; int64 where the reference holds torch.long. */
typedef struct DwAmpBuffers {
    float   *actions, *actions_pre;                 /* [N,12]                                                        */
    float   *action_history, *obs_history;          /* [N, his*skip*12], [N, his*skip*36]                            */
    float   *commands, *start_target_vel, *final_target_vel;       /* [N,3]                                          */
    int64_t *vel_change_duration, *cur_vel_change_duration;        /* [N]                                            */
    float   *epi_len;                               /* [N]                                                           */
    float   *power_scale;                           /* [N,12]                                                        */
    float   *action_log;                            /* [N, log_slots, 12] delayed-torque FIFO, newest last           */
    int64_t *delay_idx, *simul_len;                 /* [N]                                                           */
    float   *qpos_noise, *qvel_noise, *qpos_pre;    /* [N,33]                                                        */
    float   *qpos_bias, *quat_bias;                 /* [N,12], [N,3]                                                 */
    float   *dof_vel_pre;                           /* [N,33]                                                        */
    float   *tau;                                   /* [N,33] what dw_simulate is handed                             */
    int64_t *progress_buf, *randomize_buf, *reset_buf, *terminate_buf;   /* [N]                                      */
    uint8_t *timeout_buf;                           /* [N] torch.bool                                                */
    float   *rigid_body_pos, *rigid_body_rot;       /* [N,38,3], [N,38,4]: rows 0, 8, 16 / row 0 are maintained      */
    float   *foot_pos;                              /* [N,2,3]                                                       */
    float   *obs1, *obs_buf, *obs_out;              /* [N,36], [N,num_obs], [N,num_obs] (= clamp(obs_buf, +-clip))   */
    float   *rew_buf, *reward_values;               /* [N], [N,9]                                                    */
    float   *total_mass;                            /* [N]                                                           */
    float   *amp_obs_buf, *amp_obs1;                /* [N, amp_steps, 34], [N,34]                                    */
    const float *motor_efforts, *p_gains, *d_gains, *init_angle;   /* [12], [33], [33], [33]                         */
    const float *pd_action_offset, *pd_action_scale;               /* [12] each (pd_control only, else NULL)         */
    /* touched by dw_amp_reset_rows only */
    float   *epi_len_log;                           /* [N]                                                           */
    int64_t *perturbation_count, *perturb_timing;   /* [N]                                                           */
    uint8_t *pert_on;                               /* [N] torch.bool                                                */
    const float *initial_root_states;               /* [N,13]                                                        */
    /* rings and device draws (NULL unless DwAmpConfig.hist_ring / device_draws) */
    int32_t *hist_head;                             /* [N,2] action / observation history: slot of the oldest entry  */
    int64_t *draw_ctr;                              /* [N] the env's draw counter                                    */
    /* dw_amp_reset_done with DwAmpConfig.dr_damping / dr_armature */
    const float *nominal_damping, *nominal_armature;               /* [33] each                                      */
} DwAmpBuffers;
typedef struct DwAmpConfig {
    int32_t num_envs, num_his, num_skip, log_slots, amp_steps;
    int32_t pd_control, noise, vel_change, local_root_obs, enable_early_termination;
    float   clip_actions, clip_obs;                 /* +inf = no clipping                                            */
    float   max_episode_length, termination_height;
    float   inv_dt;                                 /* float(1 / dt): torch's GPU kernels multiply by it for `x / dt` */
    float   dt;
    int32_t gpu_div;                                /* 1: qvel = d * inv_dt (torch on a GPU), 0: d / dt (torch on a CPU) */
    float   cmd_lo[3], cmd_scale[3];                /* command ranges x, y, yaw: float(lo), float(hi - lo)            */
    int32_t hist_ring;                              /* 1: action_history / obs_history are rings (hist_head)         */
    int32_t device_draws;                           /* 1: NULL draw arguments are drawn in the kernels (draw_ctr)    */
    int32_t randomize;                              /* task.randomize: power_scale (and the dof properties) at reset */
    int32_t dr_damping, dr_armature, dr_frequency;  /* dw_amp_reset_done: dof-property randomisation of a resetting env whose
                                                       randomize_buf has reached dr_frequency (tasks/base/vec_task.py:519-733) */
    float   dr_damping_range[2], dr_armature_range[2];             /* additive / scaling, uniform                    */
    int32_t delay_idx_range[2];                     /* device draws: delay_idx in [lo, hi) = [1 + int(0.002 / dt), 1 + round(0.01 / dt)) (:302) */
    uint64_t seed;                                  /* key of the device draws                                       */
} DwAmpConfig;
/* actions_in [N,12] (clamped to +-clip_actions inside).  ramp_dur [N] int64 in [1,250), ramp_u [N,3] uniform in [0,1): the draws of
 * the command ramp for EVERY env (kept where an env changes its command); NULL with vel_change = 0 or device_draws = 1.  Reads the
 * bound dof_state of h. */
int dw_amp_step_begin(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *actions_in, const int64_t *ramp_dur,
                      const float *ramp_u, void *stream);
/* between substep `substep` - 1 and substep `substep` (1 <= substep < 8).  z [N,33]: normal draws with sigma 0.00016 / 3 for the
 * encoder model of the substep that ended (NULL with noise = 0 or device_draws = 1) */
int dw_amp_step_mid(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *z, int substep, void *stream);
/* after the last substep (index `substep`).  z as above; rootvel_noise [N,6]: uniform in +-0.025 (NULL: zeros with noise = 0, drawn
 * with device_draws = 1); reads the bound root_states / dof_state / contact_forces of h */
int dw_amp_step_end(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *z, int substep, const float *rootvel_noise,
                    void *stream);
/* The same step -- dw_amp_step_begin, `substeps` x dw_simulate with dw_amp_step_mid between them, dw_amp_step_end -- as ONE launch (ABI 9; plane
 * only): the task's code around the octet kernels' physics substep inside one kernel (csrc/dw_oct_kernels.hip dw_k_amp_step_oct), a workgroup of two
 * wavefronts per 16 envs; what tasks/amp/tocabi_amp_lower_base.py:642-804 does between two policy forwards.  Same arithmetic on the same state in the
 * same order as the separate entry points: the results are the same bits (tests/test_amp_gpu.py).  z: `substeps` device pointers to the [N,33] encoder
 * draws of the substeps, in order (the draws do not depend on the physics, so the caller makes them up front, in the order it would have between the
 * launches); NULL (or NULL entries) with noise = 0 or device_draws = 1.  The handle's step torques are DwAmpBuffers.tau; applied forces (push) are
 * not part of this entry point, as they are not of the three kernels. */
int dw_amp_step(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *actions_in, const int64_t *ramp_dur, const float *ramp_u,
                const float *const *z, int substeps, const float *rootvel_noise, void *stream);
/* reset_idx of the listed envs (tasks/amp/tocabi_amp_lower_base.py:238-305 with the default state initialisation, then
 * tasks/tocabi_amp_lower.py:144-147,258-272) as ONE launch instead of ~60 indexed assignments: a wavefront per listed env writes its rows
 * of the Gym tensors (initial pose, zero contact), the reset observation (computed from the episode's last encoder reading, as the
 * reference does), every piece of task state, and the env's discriminator history (copies of its current observation).  ids [n] int64 in
 * device memory, distinct.  The draws are the caller's RAW uniforms in [0,1), one row per listed env in the order of ids, turned into
 * values here with torch's own arithmetic: power_scale_u [n,12] -> 0.8 + 0.4 u (NULL: keep the scales), cmd_x_u / cmd_y_u / cmd_yaw_u [n]
 * -> DwAmpConfig.cmd_lo + cmd_scale u, qpos_bias_u [n,12] -> u 6.28 / 100 - 3.14 / 100, quat_bias_u [n,3] -> u 6.28 / 150 - 3.14 / 150 (the
 * division as DwAmpConfig.gpu_div says); perturb_timing [n], delay_idx [n] int64 as drawn; rootvel_noise [N,6] is indexed by env. */
int dw_amp_reset_rows(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const int64_t *ids, int n, const float *power_scale_u,
                      const float *rootvel_noise, const float *cmd_x_u, const float *cmd_y_u, const float *cmd_yaw_u, const float *qpos_bias_u,
                      const float *quat_bias_u, const int64_t *perturb_timing, const int64_t *delay_idx, void *stream);
/* VecTask.reset_done's reset_idx (tasks/base/vec_task.py:376-391 -> tasks/amp/tocabi_amp_lower_base.py:238-305) for EVERY env whose
 * reset_buf is non-zero, in one launch and without an id list (no host round trip before the launch): what dw_amp_reset_rows does for a
 * listed env, plus the dof-property randomisation (DwAmpConfig.dr_*: damping = nominal + U, armature = nominal * U into the bound
 * dof_damping / dof_armature of h) and the clamped observation row (obs_out).  Draws: rows of the caller's arrays indexed BY ENV
 * ([N, ...]: raw uniforms as for dw_amp_reset_rows; damping_u / armature_u [N,33]) or, with device_draws = 1, NULL = drawn here. */
typedef struct DwAmpResetDraws {
    const float *power_scale_u, *rootvel_noise, *cmd_x_u, *cmd_y_u, *cmd_yaw_u, *qpos_bias_u, *quat_bias_u, *damping_u, *armature_u;
    const int64_t *perturb_timing, *delay_idx;
} DwAmpResetDraws;
int dw_amp_reset_done(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const DwAmpResetDraws *draws, void *stream);
/* The other half of VecTask.reset_done (tasks/base/vec_task.py:381, `done_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)`): the ids of
 * the envs whose reset_buf [n] is non-zero, ascending, into ids [n], and their number into *count (device memory) and -- if given -- *count_host
 * (pinned host memory mapped to the device: the caller waits for an event recorded behind this launch and reads the number there, without a copy
 * of its own).  One launch; queue it BEFORE dw_amp_reset_done, which clears the flags.  ABI 10. */
int dw_amp_reset_ids(const int64_t *reset_buf, int n, int64_t *ids, int64_t *count, int64_t *count_host, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* DYROS_WALK_H */
