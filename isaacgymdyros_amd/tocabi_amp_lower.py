"""TocabiAMPLower's env side on the MI355X physics.

Mirrors `TocabiAMPLowerBase` of the reference (paths relative to python/IsaacGymEnvs/isaacgymenvs):
  tasks/amp/tocabi_amp_lower_base.py:50-236   constructor state (gains, biases, histories, delay FIFO, command schedule)
                                    :238-305  reset_idx
                                    :540-580  _compute_observations (history stacking: NumHis x NumSkip)
                                    :642-748  pre_physics_step (push perturbation, command ramp, torque / PD actuation with the
                                              delayed-torque FIFO, controlFrequencyInv substeps, encoder model)
                                    :750-804  post_physics_step
                                    :918-1069 the three TorchScript functions -- here the HIP entry points dw_amp_observations,
                                              dw_amp_reward, dw_amp_reset (include/dyros_walk.h), pinned bit for bit against the
                                              reference's own functions (tests/golden/amp_lower_ref.npz)
  cfg/task/TocabiAMPLower.yaml                default_amp_cfg below

What runs where.  `gym.simulate` is dw_simulate of the bound physics handle (one launch per substep, the same octet kernel the
DyrosDynamicWalk step fuses); observation, reward and termination are one HIP launch each; the rows of the rigid-body state
tensor they need (base, the two foot links) come from dw_body_positions.  The bookkeeping between them -- history shifts,
the command ramp, the torque FIFO -- is either ~100 elementwise torch launches per step on [N,12]..[N,480] tensors (the form the
replay tests pin to the reference class) or, with cfg sim.mi355.amp_fused, HIP kernels that give the torch form's bits
(dw_amp_step_begin / _mid / _end around the physics launches, dw_amp_reset_rows / _done; csrc/dw_amp_step.h, DESIGN.md section 9;
histories as rings, sim.mi355.amp_hist_ring); enable_graph_step() records a step of either form in a hipGraph.  Random draws are
torch's device generator, as in the reference -- or, with cfg sim.mi355.amp_device_draws, made inside the fused kernels (no draw
launches, reset_done() one launch; same distributions, another stream).  The generator and
`simulate` are injectable, and tests/test_amp_gpu.py replays the reference CLASS' recorded draws and physics states through this
class for 60 steps (tests/golden/amp_class_ref.npz): every state field, the torques handed to the engine and the reset flow are
bit-identical, the observation and reward to the rounding of atan2f / expf.  Four things that replay found and that are now as
the reference has them: the gains are quotients rounded once from double; the TorchScript observation function adds the
encoder bias to `qpos_noise` IN PLACE; `x / dt` divides on the CPU and multiplies by a reciprocal on the GPU (cfg
sim.mi355.torch_gpu_div picks the flavour); the command ramp multiplies before it divides.

The AMP subclass' layer (tasks/tocabi_amp_lower.py) is part of this class:
  :47                NUM_AMP_OBS_PER_STEP = 34; `numAMPObsSteps` of them form the discriminator's observation
  :88-96             post_physics_step: shift the AMP history, compute the newest step, extras["amp_obs"]
  :105-131           fetch_amp_obs_demo: demonstration observations from the motion library (isaacgymdyros_amd/motion_lib.py)
  :144-178,180-256   reset_idx / _reset_actors: stateInit Default | Start | Random | Hybrid (reference state initialisation
                     from the motion library), then _init_amp_obs (history of a reset env: copies of its current observation for
                     a default start, the motion's earlier frames for a reference start)
  :310-350           build_amp_observations -- here the HIP entry point dw_amp_disc_observations, pinned against the reference's
                     function (tests/golden/amp_disc_ref.npz)
The reference's motion tables (assets/amp/tocabi_motions/*.txt) are not in its checkout: `cfg.env.motion_file` must name the
user's own (a .yaml list or one .txt) for every stateInit but Default and for fetch_amp_obs_demo; the motion library is pinned
against the reference's on synthetic tables (tests/test_amp_motion.py).  The key-body rows of the rigid-body state a reset
env's observation reads are recomputed from the state just set (dw_body_positions); Isaac Gym's deferred setters would still
show the pre-reset rows there -- irrelevant for the history, which _init_amp_obs overwrites.

Not built: `stateInit: Custom` (the task's triangle-mesh terrain origins; the terrain of this task is not wired).

Where this class departs from the task yaml, all of it on purpose: the spawn height is 0.93 m, not the reference's 0.89 (which
starts the soles 2.2 cm inside the plane; cfg sim.mi355.amp_initial_height restores the number); env.contactBodies must be the two
foot links (the kernels test Gym rows 8 and 16; anything else is refused); terrainType other than 'plane' and stateInit 'Custom'
are refused.  physx.num_position_iterations + num_velocity_iterations is taken as given (4 + 0 for this task).
"""
from __future__ import annotations

import copy
import ctypes as C
from typing import Any, Dict

import numpy as np
import torch

from . import _lib
from .config import default_cfg
from .dyros_dynamic_walk import DyrosDynamicWalk
from .task_constants import ACTION_HIGH
from .vec_task import Box, VecTask

NUM_OBS = 3 + 6 + 3 + 12 + 12          # reference :45 (root euler, root velocities, command, leg dof pos, leg dof vel)
NUM_ACTIONS = 12                        # :46
NUM_AMP_OBS_PER_STEP = 1 + 3 + 12 + 12 + 6          # tasks/tocabi_amp_lower.py:47 (root height, root euler, leg dof pos / vel, two foot positions)
STATE_INITS = ("Default", "Start", "Random", "Hybrid")
REWARD_NAMES = ["x_vel_tracking", "y_vel_tracking", "yaw_vel_tracking", "contact_force_threshold", "contact_force_penalty",
                "joint_velocity_regulation", "joint_acceleration_regulation", "torque_regulation", "torque_diff_regulation"]   # :1010-1012
INIT_ANGLE = [0.0, 0.0, -0.28, 0.6, -0.32, 0.0, 0.0, 0.0, -0.28, 0.6, -0.32, 0.0, 0.0, 0.0, 0.0,
              0.3, 0.174533, 1.22173, -1.27, -1.57, 0.0, -1.0, 0.0, 0.0, 0.0,
              -0.3, -0.174533, -1.22173, 1.27, 1.57, 0.0, 1.0, 0.0]                                    # :90-95
ARMATURE = [0.614, 0.862, 1.09, 1.09, 1.09, 0.360, 0.614, 0.862, 1.09, 1.09, 1.09, 0.360, 0.078, 0.078, 0.078,
            0.18, 0.18, 0.18, 0.18, 0.0032, 0.0032, 0.0032, 0.0032, 0.0032, 0.0032,
            0.18, 0.18, 0.18, 0.18, 0.0032, 0.0032, 0.0032, 0.0032]                                   # :443-447
# reference :1073-1103 (class control), by DoF order; applied as p / 9 and d / 3 (:167-168)
P_GAINS = [2000, 5000, 4000, 3700, 3200, 3200] * 2 + [6000, 10000, 10000] + [400, 1000, 400, 400, 400, 400, 100, 100] + [100, 100] + \
          [400, 1000, 400, 400, 400, 400, 100, 100]
D_GAINS = [15, 50, 20, 25, 24, 24] * 2 + [200, 100, 100] + [10, 28, 10, 10, 10, 10, 3, 3] + [100, 100] + [10, 28, 10, 10, 10, 10, 3, 3]


def default_amp_cfg(num_envs: int = 4096, sim_device: str = "cuda:0") -> dict:
    """cfg/task/TocabiAMPLower.yaml with the Hydra interpolations resolved (plane terrain)."""
    return {
        "name": "TocabiAMPLower", "physics_engine": "physx", "rl_device": sim_device, "seed": 42,
        "env": {"numEnvs": num_envs, "envSpacing": 5, "episodeLength": 8000, "enableDebugVis": False, "pdControl": False,
                "powerScale": 1.0, "controlFrequencyInv": 2, "stateInit": "Default", "hybridInitProb": 0.5, "numAMPObsSteps": 2,
                "motion_file": None, "localRootObs": False,
                "contactBodies": ["L_Foot_Link", "R_Foot_Link"], "terminationHeight": 0.6, "enableEarlyTermination": True,
                "perturbation": False, "velChange": True, "NumHis": 10, "NumSkip": 2,
                "command": {"x": [-0.5, 1.0], "y": [-0.0, 0.0], "yaw": [-0.0, 0.0]},
                "terrain": {"terrainType": "plane", "staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0, "curriculum": True}},
        "sim": {"dt": 0.002, "substeps": 1, "up_axis": "z", "use_gpu_pipeline": True, "gravity": [0.0, 0.0, -9.81],
                "physx": {"num_position_iterations": 4, "num_velocity_iterations": 0, "contact_offset": 0.002, "rest_offset": 0.0,
                          "bounce_threshold_velocity": 0.04, "max_depenetration_velocity": 10.0}},
        "task": {"noise": True, "randomize": True,
                 "randomization_params": {"frequency": 1, "actor_params": {"humanoid": {
                     "rigid_body_properties": {"mass": {"range": [0.8, 1.2], "operation": "scaling", "distribution": "uniform", "setup_only": True}},
                     "dof_properties": {"damping": {"range": [0.0, 2.9], "operation": "additive", "distribution": "uniform"},
                                        "armature": {"range": [0.8, 1.2], "operation": "scaling", "distribution": "uniform"}}}}}},
    }


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class TorchDraws:
    """The class' source of random numbers: torch's device generator, one method per kind of draw the reference makes
    (torch.rand, torch.randint, torch.normal).  A test replaces it with a replay of the reference's recorded draws
    (tests/test_amp_gpu.py), which is why every draw of the class goes through here and happens in the reference's order --
    including the draws of size zero the reference makes every step for the command ramp."""

    def __init__(self, device, seed):
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))

    def rand(self, *shape):
        return torch.rand(*shape, device=self.device, generator=self.gen)

    def randint(self, lo, hi, shape):
        return torch.randint(lo, hi, shape, device=self.device, generator=self.gen)

    def normal(self, shape, std):
        return torch.randn(*shape, device=self.device, generator=self.gen) * std

    def bernoulli(self, n, p):
        return torch.bernoulli(torch.full((n,), float(p), device=self.device), generator=self.gen)


class TocabiAMPLower(VecTask):

    def __init__(self, cfg: Dict[str, Any], sim_device: str, graphics_device_id: int = 0, headless: bool = True):
        self.cfg = cfg
        e = cfg["env"]
        if e["terrain"]["terrainType"] not in ("plane", "none"):
            raise ValueError("TocabiAMPLower on the MI355X physics: only terrainType 'plane' is wired (module docstring)")
        self._state_init = e.get("stateInit", "Default")
        if self._state_init not in STATE_INITS:
            raise ValueError("TocabiAMPLower on the MI355X physics: stateInit must be one of %s (module docstring)" % (STATE_INITS,))
        self._hybrid_init_prob = float(e.get("hybridInitProb", 0.5))
        self._num_amp_obs_steps = int(e.get("numAMPObsSteps", 2))
        if self._num_amp_obs_steps < 2:
            raise ValueError("numAMPObsSteps must be at least 2 (tasks/tocabi_amp_lower.py:65)")
        self._local_root_obs = bool(e.get("localRootObs", False))
        if self._state_init != "Default" and not e.get("motion_file"):
            raise ValueError("stateInit %r draws its start states from the motion library: set cfg.env.motion_file (the reference's "
                             "tables are not in its checkout)" % self._state_init)
        # the bodies allowed to touch the ground are the two foot links in the kernels (Gym rows 8 and 16: dw_amp_step_end's
        # termination, dw_amp_reset's caller); a cfg that names others is refused rather than silently given the feet
        if list(e.get("contactBodies", ["L_Foot_Link", "R_Foot_Link"])) != ["L_Foot_Link", "R_Foot_Link"]:
            raise ValueError("TocabiAMPLower on the MI355X physics: env.contactBodies must be ['L_Foot_Link', 'R_Foot_Link'] (module docstring)")
        self._pd_control = e["pdControl"]
        self.randomize = cfg["task"]["randomize"]
        self.noise = cfg["task"]["noise"]
        self.randomization_params = cfg["task"]["randomization_params"]
        self.max_episode_length = e["episodeLength"]
        self._termination_height = e["terminationHeight"]
        self._enable_early_termination = e["enableEarlyTermination"]
        self.perturb = e["perturbation"]
        self.c_x, self.c_y, self.c_yaw = e["command"]["x"], e["command"]["y"], e["command"]["yaw"]
        self.num_obs_his, self.num_obs_skip = e["NumHis"], e["NumSkip"]
        e["numObservations"] = (NUM_OBS + NUM_ACTIONS) * self.num_obs_his - NUM_ACTIONS            # :83
        e["numActions"] = NUM_ACTIONS
        super().__init__(config=cfg, sim_device=sim_device, graphics_device_id=graphics_device_id, headless=headless)

        # ---- the physics: a bound DwHandle with the Gym tensors, carried by the DyrosDynamicWalk host class (its task state is unused)
        pc = default_cfg(self.num_envs, self.device)
        pc["seed"] = cfg.get("seed", 42)
        pc["sim"].update({k: copy.deepcopy(v) for k, v in cfg["sim"].items() if k != "mi355"})
        # (physx.num_position_iterations + num_velocity_iterations = the contact solver's iteration count, as the task yaml gives them:
        #  4 + 0 for this task, cfg/task/TocabiAMPLower.yaml)
        pc["sim"]["mi355"].update(cfg["sim"].get("mi355", {}))
        pc["env"]["perturbation"] = False
        # spawn height: the reference's 0.89 (:394) puts the soles 2.2 cm INTO the plane in the initial pose; PhysX resolves that
        # positionally, this simulator's contact model through a velocity bias (erp), which launches the robot at 2 m/s.  Default
        # here: the DyrosDynamicWalk height (soles 1.8 cm above the plane); cfg sim.mi355.amp_initial_height = 0.89 restores the number
        self._spawn_z = float(cfg["sim"].get("mi355", {}).get("amp_initial_height", 0.93))
        pc["env"]["initialHieght"] = self._spawn_z
        pc["task"]["randomize"] = bool(self.randomize)
        pc["task"]["randomization_params"] = {"frequency": 1, "actor_params": {"humanoid": {
            "rigid_body_properties": copy.deepcopy(self.randomization_params["actor_params"]["humanoid"].get("rigid_body_properties", {}))}}}
        self._phys = DyrosDynamicWalk(pc, self.device, graphics_device_id, headless)
        self._lib, self._api = self._phys._lib, self._phys._api
        dev = self._tdev = self._phys._tdev
        N = self.num_envs
        f = dict(dtype=torch.float, device=dev)
        self.dt = cfg["sim"]["dt"]                                                             # :98
        self.num_bodies, self.num_dof = 38, 33
        self.pelvis_idx, self.left_foot_idx, self.right_foot_idx = 0, 8, 16
        b = self._phys._buf
        self._root_states = b["root_states"]
        self._dof_state = b["dof_state"]
        self._dof_pos, self._dof_vel = b["dof_state"][..., 0], b["dof_state"][..., 1]
        self._contact_forces = b["contact_forces"]
        self.total_mass = b["total_mass"].view(N, 1)
        self.init_angle = torch.tensor(INIT_ANGLE, **f)
        self.motor_efforts = torch.tensor(ACTION_HIGH[:12], **f)                              # :357-358 (ctrlrange upper limits)
        self.max_motor_effort = float(self.motor_efforts.max())
        # (the quotients are formed in double and rounded once, as the reference's element-wise assignment does, :167-168)
        self.p_gains = torch.tensor([p / 9.0 for p in P_GAINS], **f)
        self.d_gains = torch.tensor([d / 3.0 for d in D_GAINS], **f)
        self._nominal_damping = torch.full((33,), 0.1, **f)                                    # :441
        self._nominal_armature = torch.tensor(ARMATURE, **f)
        b["dof_damping"][:] = self._nominal_damping
        b["dof_armature"][:] = self._nominal_armature
        self._initial_root_states = torch.zeros(N, 13, **f)
        self._initial_root_states[:, 2] = self._spawn_z
        self._initial_root_states[:, 6] = 1.0
        self._root_states[:] = self._initial_root_states
        self._initial_dof_pos = self.init_angle.repeat(N, 1)
        self._dof_pos[:] = self._initial_dof_pos
        self._dof_vel[:] = 0.0
        self._contact_forces.zero_()
        self.power_scale = torch.ones(N, 12, **f)
        self.actions = torch.zeros(N, NUM_ACTIONS, **f)
        self.actions_pre = self.actions.clone()
        self._dof_vel_pre = self._dof_vel.clone().contiguous()
        # rows 0, 8, 16 of the rigid-body state (what compute_humanoid_reset reads); the other rows stay zero
        self._rigid_body_pos = torch.zeros(N, 38, 3, **f)
        self._rigid_body_rot = torch.zeros(N, 38, 4, **f)
        self._rigid_body_rot[..., 3] = 1.0
        self._foot_pos = torch.zeros(N, 2, 3, **f)
        self._foot_mv = (C.c_int32 * 2)(6, 12)                                                  # moving bodies of L_/R_Foot_Link's parent joint
        self._contact_body_ids = torch.tensor([8, 16], dtype=torch.int32, device=dev)
        self.commands = torch.zeros(N, 3, **f)
        self._terminate_buf = torch.ones(N, device=dev, dtype=torch.long)
        self.reset_buf = torch.ones(N, device=dev, dtype=torch.long)
        self.progress_buf = torch.zeros(N, device=dev, dtype=torch.long)
        self.randomize_buf = torch.zeros(N, device=dev, dtype=torch.long)
        self.timeout_buf = torch.zeros(N, device=dev, dtype=torch.bool)
        self.obs_buf = torch.zeros(N, self.num_obs, **f)
        self.rew_buf = torch.zeros(N, **f)
        self._obs1 = torch.zeros(N, NUM_OBS, **f)
        self._reward_values = torch.zeros(N, 9, **f)
        self.qpos_bias, self.quat_bias = torch.zeros(N, 12, **f), torch.zeros(N, 3, **f)
        self.qpos_noise, self.qvel_noise, self.qpos_pre = (torch.zeros(N, 33, **f) for _ in range(3))
        self.epi_len, self.epi_len_log = torch.zeros(N, **f), torch.zeros(N, **f)
        self._gpu_div = bool(cfg["sim"].get("mi355", {}).get("torch_gpu_div", True))
        self._rng = TorchDraws(dev, cfg.get("seed", 42))
        self._simulate = self._phys.simulate          # (`gym.simulate`; a test injects recorded physics states here)
        self.perturbation_count = torch.zeros(N, device=dev, dtype=torch.long)
        self.pert_duration = self._rng.randint(1, 100, (N,))
        self.pert_on = torch.zeros(N, device=dev, dtype=torch.bool)
        self.impulse = torch.zeros(N, device=dev, dtype=torch.long)
        self.magnitude, self.phase = torch.zeros(N, **f), torch.zeros(N, **f)
        self.perturb_timing = torch.ones(N, device=dev, dtype=torch.long)
        self.perturb_start = False
        self.vel_change = e["velChange"]
        self.vel_change_duration = torch.zeros(N, device=dev, dtype=torch.long)
        self.cur_vel_change_duration = torch.zeros(N, device=dev, dtype=torch.long)
        self.start_target_vel, self.final_target_vel = torch.zeros(N, 3, **f), torch.zeros(N, 3, **f)
        self.obs_history = torch.zeros(N, self.num_obs_his * self.num_obs_skip * NUM_OBS, **f)
        self.action_history = torch.zeros(N, self.num_obs_his * self.num_obs_skip * NUM_ACTIONS, **f)
        self._log_slots = round(0.01 / self.dt) + 1                                            # :225
        self.action_log = torch.zeros(N, self._log_slots, 12, **f)
        self.delay_idx = torch.ones(N, device=dev, dtype=torch.long)                           # delay_idx_tensor[:,1]
        self.simul_len = torch.zeros(N, device=dev, dtype=torch.long)                          # simul_len_tensor[:,1]
        self._env_ar = torch.arange(N, device=dev)
        self._push = torch.zeros(N, 2, **f)
        self.time_step = 0
        self.extras["reward_names"] = list(REWARD_NAMES)
        # ---- the AMP subclass' state (tasks/tocabi_amp_lower.py:58-86)
        self._motion_lib = None
        if e.get("motion_file"):
            from .motion_lib import TocabiLowerMotionLib
            self._motion_lib = TocabiLowerMotionLib(e["motion_file"], self.num_dof, dev)
        self.num_amp_obs = self._num_amp_obs_steps * NUM_AMP_OBS_PER_STEP
        self._amp_obs_space = Box(np.ones(self.num_amp_obs) * -np.inf, np.ones(self.num_amp_obs) * np.inf)
        self._amp_obs_buf = torch.zeros(N, self._num_amp_obs_steps, NUM_AMP_OBS_PER_STEP, **f)
        self._curr_amp_obs_buf = self._amp_obs_buf[:, 0]
        self._hist_amp_obs_buf = self._amp_obs_buf[:, 1:]
        self._amp_obs_demo_buf = None
        self._graph, self._capturing, self._g_actions, self._g_out = None, False, None, None
        # cfg sim.mi355.amp_fused: the step's bookkeeping as four HIP kernels (dw_amp_step_*) instead of ~100 torch launches
        self._fused = bool(cfg["sim"].get("mi355", {}).get("amp_fused", False))
        # (the fused reset makes the reference's draws in the reference's order, so it can be switched on by itself and replayed
        #  against the reference class; the fused step cannot: its command ramp draws for every env)
        self._fused_reset = bool(cfg["sim"].get("mi355", {}).get("amp_fused_reset", self._fused))
        # cfg sim.mi355.amp_hist_ring (fused step only, default on with it): action_history / obs_history are rings in memory
        # (include/dyros_walk.h DwAmpConfig.hist_ring); history_linear() gives the reference's layout.
        # cfg sim.mi355.amp_device_draws (fused step only): every draw of step() and reset_done() is made inside the kernels
        # (Philox keyed by the seed, the env and the env's draw counter) instead of by torch's generator: no draw kernels, no host
        # round trip; same distributions, another stream -- the form bench.py measures.
        mi = cfg["sim"].get("mi355", {})
        # The ring layout is understood by the fused kernels only (dw_amp_step_*, dw_amp_reset_rows / _done).  A reset that takes the
        # torch path -- stateInit Start / Random / Hybrid, or amp_fused_reset off -- reads the histories in the reference's shifting
        # layout (_compute_observations(env_ids)), so the rings default to ON only where every reset is the fused one, and asking
        # for them elsewhere is an error rather than a silently rotated observation.
        ring_ok = self._fused and self._fused_reset and self._state_init == "Default"
        if "amp_hist_ring" in mi and bool(mi["amp_hist_ring"]) and not ring_ok:
            raise ValueError("sim.mi355.amp_hist_ring needs amp_fused with the fused reset (amp_fused_reset, stateInit 'Default'): "
                             "the torch reset path reads the histories in the reference's shifting layout")
        self._hist_ring = bool(mi.get("amp_hist_ring", ring_ok)) and ring_ok
        self._device_draws = bool(mi.get("amp_device_draws", False))
        if self._device_draws and not (self._fused and self._fused_reset):
            raise ValueError("sim.mi355.amp_device_draws needs amp_fused (step and reset)")
        if self._device_draws and self._state_init != "Default":
            raise ValueError("sim.mi355.amp_device_draws: the fused reset covers stateInit 'Default' only")
        # cfg sim.mi355.amp_one_launch (fused step on the plane; default OFF): the whole step -- the three task kernels and the K physics substeps
        # between them -- as ONE launch (dw_amp_step, include/dyros_walk.h): same arithmetic in the same order, same bits
        # (tests/test_amp_gpu.py).  Measured at 16384 envs in a replayed graph: 0.254 ms against 0.194 ms for the five launches
        # (profiles/r06_amp_one_launch.txt) -- the task regions run with 16 wavefronts per CU and ~70 registers as kernels of their own, with 8
        # wavefronts per CU under the substep's 256-register budget inside the octet workgroup: what the launch boundaries cost (~20 us) is
        # less than what the fusion loses (~60 us).  Kept as an entry point and a negative result.
        # (A handle of at most 4096 envs takes the hex instantiation for dw_simulate, whose sums round differently from the octet substep the
        #  one-launch step carries: comparisons pin sim.mi355.debug_wave_build = 2.)
        self._one_launch = bool(mi.get("amp_one_launch", False)) and self._fused
        if self._one_launch and (self._phys.custom_origins or self.control_freq_inv > 8):
            if "amp_one_launch" in mi and bool(mi["amp_one_launch"]):
                raise ValueError("sim.mi355.amp_one_launch: the one-launch step is built for the plane and at most 8 substeps")
            self._one_launch = False
        self._z_ptrs = None
        self._hist_head = torch.zeros(N, 2, dtype=torch.int32, device=dev)
        self._draw_ctr = torch.zeros(N, dtype=torch.int64, device=dev)
        self._tau = torch.zeros(N, 33, **f)
        self._obs_out = torch.zeros(N, self.num_obs, **f)
        self._amp_obs1 = torch.zeros(N, NUM_AMP_OBS_PER_STEP, **f)
        self._reset_default_env_ids, self._reset_ref_env_ids = [], []
        self._reset_ref_motion_ids = self._reset_ref_motion_times = None
        if self._pd_control:
            lo, hi = self._phys.model.dof_lower, self._phys.model.dof_upper
            lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
            mid, sc = 0.5 * (hi + lo), 1.0 * (hi - lo)                                         # :485-503
            self._pd_action_offset = torch.tensor(0.5 * ((mid + sc) + (mid - sc)), **f)
            self._pd_action_scale = torch.tensor(0.5 * ((mid + sc) - (mid - sc)), **f)

    # ------------------------------------------------------------------ helpers
    def _rand(self, *shape):
        return self._rng.rand(*shape)

    def _div(self, x, s):
        """`tensor / python_scalar` as torch does it: its GPU kernels multiply by float(1 / s), its CPU kernels divide (the last
        bit differs).  cfg sim.mi355.torch_gpu_div (default True) = what the reference would compute on a GPU; False = the CPU
        flavour the fixtures were recorded with (a tensor divisor makes the GPU divide too)."""
        return x / s if self._gpu_div else x / torch.tensor(float(s), device=x.device, dtype=x.dtype)

    def _rand_float(self, lo, hi, shape):                 # torch_rand_float (python/isaacgym/torch_utils.py:50-52)
        return (hi - lo) * self._rand(*shape) + lo

    def _chk(self, rc):
        _lib.check(self._api, rc)

    def _stream(self):
        """torch's current stream on the task's device: the entry points launch where torch's own kernels of this step run (the
        null stream in eager use, the capturing stream while enable_graph_step() records a step)."""
        return C.c_void_p(torch.cuda.current_stream(self._tdev).cuda_stream)

    def _foot_positions(self):
        self._chk(self._api["body_positions"](self._phys._h, self._foot_mv, 2, _p(self._foot_pos), self._stream()))

    def _refresh_sim_tensors(self):
        """refresh_*_tensor of the reference (:527-538): the Gym tensors are the physics' own buffers; the three rigid-body rows
        the reset needs are recomputed from them."""
        self._foot_positions()
        self._rigid_body_pos[:, 0] = self._root_states[:, 0:3]
        self._rigid_body_pos[:, 8] = self._foot_pos[:, 0]
        self._rigid_body_pos[:, 16] = self._foot_pos[:, 1]
        self._rigid_body_rot[:, 0] = self._root_states[:, 3:7]

    # ------------------------------------------------------------------ reset (:238-305)
    def reset_idx(self, env_ids):
        env_ids = env_ids.to(self._tdev).long()
        n = len(env_ids)
        if n == 0:
            return
        if self._fused_reset and self._state_init == "Default":
            return self._reset_fused(env_ids)
        if self.randomize:
            self.power_scale[env_ids] = self._rand_float(0.8, 1.2, (n, 12))
            self._randomize_dof_properties(env_ids)
        self._reset_actors(env_ids)
        self._contact_forces[env_ids] = 0.0
        self._refresh_sim_tensors()
        # the reference computes the observation of the reset envs HERE (:253), before the new command, encoder state and biases
        # are drawn (:266-279) and before it zeroes the two histories (:296-297): the reset env's obs_buf rows are made of the
        # episode's last encoder reading, and its history starts from zeros
        self._compute_observations(env_ids)
        self._dof_vel_pre[env_ids] = 0.0
        self.actions_pre[env_ids] = 0.0
        # the reference now writes the INITIAL root state over the reset envs' rows (:262-263, no custom origins) -- also over the root
        # pose and velocity a reference start has just taken from the motion library: a motion start keeps the motion's joint state and
        # its reset observation, but stands at the default place (its key-body rows, refreshed above, are still the motion's when
        # _init_amp_obs reads them)
        self._root_states[env_ids] = self._initial_root_states[env_ids]
        self.commands[env_ids, 0] = self._rand_float(self.c_x[0], self.c_x[1], (n,))
        self.commands[env_ids, 1] = self._rand_float(self.c_y[0], self.c_y[1], (n,))
        self.commands[env_ids, 2] = self._rand_float(self.c_yaw[0], self.c_yaw[1], (n,))
        self.qpos_noise[env_ids] = self._initial_dof_pos[env_ids]
        self.qpos_pre[env_ids] = self._initial_dof_pos[env_ids]
        self.qvel_noise[env_ids] = 0.0
        if self.noise:
            self.qpos_bias[env_ids] = self._div(self._rand(n, 12) * 6.28, 100) - 3.14 / 100
            self.quat_bias[env_ids] = self._div(self._rand(n, 3) * 6.28, 150) - 3.14 / 150
        else:
            self.qpos_bias[env_ids] = 0.0
            self.quat_bias[env_ids] = 0.0
        self.time_step = 0
        self.epi_len_log[env_ids] = self.epi_len[env_ids]
        self.epi_len[env_ids] = 0
        self.perturbation_count[env_ids] = 0
        self.pert_on[env_ids] = False
        self.perturb_timing[env_ids] = self._rng.randint(0, int(8 / 0.002), (n,))
        self.obs_history[env_ids] = 0
        self.action_history[env_ids] = 0
        self.action_log[env_ids] = 0
        self.delay_idx[env_ids] = self._rng.randint(1 + int(0.002 / self.dt), 1 + round(0.01 / self.dt), (n,))
        self.simul_len[env_ids] = 0
        self._init_amp_obs(env_ids)                       # (tasks/tocabi_amp_lower.py:144-147)

    def _randomize_dof_properties(self, env_ids):
        """apply_randomizations (tasks/base/vec_task.py:519-733) for the dof properties: envs that are resetting and whose
        randomize_buf has reached the frequency draw damping (additive) and armature (scaling) from the ORIGINAL values."""
        freq = self.randomization_params.get("frequency", 1)
        dofp = self.randomization_params["actor_params"]["humanoid"].get("dof_properties", {})
        sel = env_ids[(self.randomize_buf[env_ids] >= freq) & (self.reset_buf[env_ids] != 0)]
        if len(sel) > 0:
            b = self._phys._buf
            if "damping" in dofp:
                lo, hi = dofp["damping"]["range"]
                b["dof_damping"][sel] = self._nominal_damping + self._rand_float(lo, hi, (len(sel), 33))
            if "armature" in dofp:
                lo, hi = dofp["armature"]["range"]
                b["dof_armature"][sel] = self._nominal_armature * self._rand_float(lo, hi, (len(sel), 33))
            self.randomize_buf[sel] = 0

    def _reset_fused(self, env_ids):
        """reset_idx with one launch for all the row writes (dw_amp_reset_rows): the draws are made here, in reset_idx's order and
        sizes, so the result is reset_idx's bit for bit (tests/test_amp_gpu.py); default state initialisation only."""
        n, N = len(env_ids), self.num_envs
        # (raw uniforms in reset_idx's order and sizes; the kernel forms the values with the same float32 arithmetic)
        ps = None
        if self.randomize:
            ps = self._rand(n, 12)
            self._randomize_dof_properties(env_ids)
        self.time_step += 1
        nz = self._rand(N, 6) * 0.05 - 0.025 if self.noise else torch.zeros(N, 6, device=self._tdev)
        cx, cy, cyaw = self._rand(n), self._rand(n), self._rand(n)
        qb = quatb = None
        if self.noise:
            qb, quatb = self._rand(n, 12), self._rand(n, 3)
        ptime = self._rng.randint(0, int(8 / 0.002), (n,))
        didx = self._rng.randint(1 + int(0.002 / self.dt), 1 + round(0.01 / self.dt), (n,))
        c, b = self._fused_tables()
        self._chk(self._api["amp_reset_rows"](self._phys._h, C.byref(c), C.byref(b), _p(env_ids.contiguous()), n, _p(ps), _p(nz), _p(cx), _p(cy), _p(cyaw),
                                              _p(qb), _p(quatb), _p(ptime), _p(didx), self._stream()))
        self.time_step = 0
        self._reset_default_env_ids = env_ids

    # ------------------------------------------------------------------ state initialisation (tasks/tocabi_amp_lower.py:149-256)
    def _reset_actors(self, env_ids):
        # (as in the reference, _reset_default_env_ids / _reset_ref_env_ids are NOT cleared here: the ids of an earlier reset of the
        #  other kind stay listed, and _init_amp_obs re-initialises those envs' AMP history too -- tasks/tocabi_amp_lower.py:258-267;
        #  the class-level fixture tests/golden/amp_subclass_ref.npz holds this class to it)
        if self._state_init == "Default":
            self._reset_default(env_ids)
        elif self._state_init in ("Start", "Random"):
            self._reset_ref_state_init(env_ids)
        else:                                             # Hybrid: a reference start with probability hybridInitProb, else the default pose
            ref = self._rng.bernoulli(len(env_ids), self._hybrid_init_prob) == 1.0
            if bool(ref.any()):
                self._reset_ref_state_init(env_ids[ref])
            if bool((~ref).any()):
                self._reset_default(env_ids[~ref])
        self.progress_buf[env_ids] = 0
        self.reset_buf[env_ids] = 0
        self._terminate_buf[env_ids] = 0

    def _reset_default(self, env_ids):
        self._dof_pos[env_ids] = self._initial_dof_pos[env_ids]
        self._dof_vel[env_ids] = 0.0
        self._root_states[env_ids] = self._initial_root_states[env_ids]
        self._reset_default_env_ids = env_ids

    def _reset_ref_state_init(self, env_ids):
        ml, n = self._motion_lib, len(env_ids)
        motion_ids = ml.sample_motions(n)
        motion_times = ml.sample_time(motion_ids) if self._state_init in ("Random", "Hybrid") else np.zeros(n)
        root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, _ = ml.get_motion_state(motion_ids, motion_times)
        # the legs from the motion, the upper body at its initial pose and at rest (:214-215)
        self._root_states[env_ids, 0:3] = root_pos
        self._root_states[env_ids, 3:7] = root_rot
        self._root_states[env_ids, 7:10] = root_vel
        self._root_states[env_ids, 10:13] = root_ang_vel
        self._dof_pos[env_ids] = torch.cat([dof_pos, self._initial_dof_pos[env_ids][:, 12:]], dim=-1)
        self._dof_vel[env_ids] = torch.cat([dof_vel, torch.zeros(n, 21, device=self._tdev)], dim=-1)
        self._reset_ref_env_ids = env_ids
        self._reset_ref_motion_ids, self._reset_ref_motion_times = motion_ids, motion_times

    # ------------------------------------------------------------------ discriminator observations (tasks/tocabi_amp_lower.py:88-131,258-305)
    def get_num_amp_obs(self):
        return self.num_amp_obs

    @property
    def amp_observation_space(self):
        return self._amp_obs_space

    def _build_amp_obs(self, root_states, dof_pos, dof_vel, key_pos, out):
        """build_amp_observations (:310-350) on device tensors; dof_pos / dof_vel either the two halves of dof_state or [M,12]."""
        n = root_states.shape[0]
        assert root_states.is_contiguous() and key_pos.is_contiguous() and out.is_contiguous()
        assert dof_pos.stride() == dof_vel.stride()
        self._chk(self._api["amp_disc_observations"](n, _p(root_states), _p(dof_pos), _p(dof_vel), dof_pos.stride(0), dof_pos.stride(1),
                                                     int(self._local_root_obs), _p(key_pos), key_pos.shape[1], _p(out), self._stream()))
        return out

    def _compute_amp_observations(self, env_ids=None):
        # key bodies = the two foot links (:47-48 of the base): rows 8 and 16 of the rigid-body state, kept in _foot_pos
        obs = self._build_amp_obs(self._root_states, self._dof_pos, self._dof_vel, self._foot_pos, self._amp_obs1)
        if env_ids is None:
            self._curr_amp_obs_buf[:] = obs
        else:
            self._curr_amp_obs_buf[env_ids] = obs[env_ids]

    def _update_hist_amp_obs(self):
        # slot i -> slot i + 1, oldest dropped (:283-290); one copy through a temporary instead of the reference's reversed loop
        self._amp_obs_buf[:, 1:] = self._amp_obs_buf[:, :-1].clone()

    def _init_amp_obs(self, env_ids):
        self._compute_amp_observations(env_ids)
        if len(self._reset_default_env_ids) > 0:
            ids = self._reset_default_env_ids
            self._hist_amp_obs_buf[ids] = self._curr_amp_obs_buf[ids].unsqueeze(-2)
        if len(self._reset_ref_env_ids) > 0:
            ids, steps = self._reset_ref_env_ids, self._num_amp_obs_steps - 1
            mids = np.tile(np.expand_dims(self._reset_ref_motion_ids, axis=-1), [1, steps]).flatten()
            times = (np.expand_dims(self._reset_ref_motion_times, axis=-1) + (-self.dt * (np.arange(0, steps) + 1))).flatten()
            self._hist_amp_obs_buf[ids] = self._motion_amp_obs(mids, times).view(len(ids), steps, NUM_AMP_OBS_PER_STEP)

    def _motion_amp_obs(self, motion_ids, motion_times):
        root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_pos = self._motion_lib.get_motion_state(motion_ids, motion_times)
        root_states = torch.cat([root_pos, root_rot, root_vel, root_ang_vel], dim=-1).contiguous()
        out = torch.empty(root_states.shape[0], NUM_AMP_OBS_PER_STEP, dtype=torch.float, device=self._tdev)
        return self._build_amp_obs(root_states, dof_pos.contiguous(), dof_vel.contiguous(), key_pos.contiguous(), out)

    def fetch_amp_obs_demo(self, num_samples):
        """[num_samples, numAMPObsSteps * 34] demonstration observations: a random motion and time per sample and the steps before it."""
        if self._motion_lib is None:
            raise RuntimeError("fetch_amp_obs_demo needs the motion library: set cfg.env.motion_file")
        ml, steps = self._motion_lib, self._num_amp_obs_steps
        motion_ids = ml.sample_motions(num_samples)
        if self._amp_obs_demo_buf is None:
            self._amp_obs_demo_buf = torch.zeros(num_samples, steps, NUM_AMP_OBS_PER_STEP, dtype=torch.float, device=self._tdev)
        else:
            assert self._amp_obs_demo_buf.shape[0] == num_samples
        times0 = ml.sample_time(motion_ids)
        mids = np.tile(np.expand_dims(motion_ids, axis=-1), [1, steps]).flatten()
        times = (np.expand_dims(times0, axis=-1) + (-self.dt * np.arange(0, steps))).flatten()
        self._amp_obs_demo_buf[:] = self._motion_amp_obs(mids, times).view(self._amp_obs_demo_buf.shape)
        return self._amp_obs_demo_buf.view(-1, self.num_amp_obs)

    # ------------------------------------------------------------------ observations (:540-610)
    def _compute_humanoid_obs(self):
        self.time_step += 1
        N = self.num_envs
        nz = self._rand(N, 6) * 0.05 - 0.025 if self.noise else torch.zeros(N, 6, device=self._tdev)
        self._chk(self._api["amp_observations"](N, _p(self._root_states), _p(nz), _p(self.qpos_noise), _p(self.qpos_bias), _p(self.quat_bias),
                                                _p(self.qvel_noise), _p(self.commands), _p(self._obs1), self._stream()))
        return self._obs1

    def _compute_observations(self, env_ids=None):
        obs = self._compute_humanoid_obs()
        if env_ids is None:
            # the reference's TorchScript function adds the bias to the tensor it is handed, IN PLACE (`dof_pos[:, :12] +=
            # dof_pos_bias`, :945), and on this path that tensor is self.qpos_noise itself: the encoder reading carries the
            # bias from here to the next substep -- and an env that resets now shows it twice in its reset observation
            self.qpos_noise[:, :12] += self.qpos_bias
            self.obs_history.copy_(torch.cat((self.obs_history[:, NUM_OBS:], obs), dim=-1))
        else:
            self.obs_history[env_ids] = obs[env_ids].repeat(1, self.num_obs_his * self.num_obs_skip)
        H, S = self.num_obs_his, self.num_obs_skip
        oh = self.obs_history.view(self.num_envs, H * S, NUM_OBS)
        ah = self.action_history.view(self.num_envs, H * S, NUM_ACTIONS)
        self.obs_buf[:, :NUM_OBS * H] = oh[:, S - 1::S].reshape(self.num_envs, -1)                       # slots S(i+1)-1
        self.obs_buf[:, NUM_OBS * H:] = ah[:, S::S][:, :H - 1].reshape(self.num_envs, -1)               # slots S(i+1), i < H-1

    # ------------------------------------------------------------------ pre-physics (:642-748)
    def pre_physics_step(self, actions):
        N, dev = self.num_envs, self._tdev
        # (every piece of state is updated IN PLACE, in tensors that live as long as the env: a step recorded in a hipGraph,
        #  enable_graph_step(), then reads and writes the same memory at every replay)
        self.actions.copy_(actions.to(dev))
        self.action_history.copy_(torch.cat((self.action_history[:, NUM_ACTIONS:], self.actions), dim=-1))
        push = None
        if self.perturb and float(self.epi_len_log.mean()) > self.max_episode_length - 8 / 0.002:
            self.perturb_start = True
        if self.perturb_start:
            start = (self.epi_len % (8 / 0.002)) == self.perturb_timing
            ns = int(start.sum())
            if ns:
                self.pert_on[start] = True
                self.impulse[start] = self._rng.randint(50, 250, (ns,))
                self.pert_duration[start] = self._rng.randint(int(0.1 / 0.002), int(1 / 0.002), (ns,))
                self.magnitude[start] = self.impulse[start] / (self.pert_duration[start] * 0.002)
                self.phase[start] = self._rand(ns) * 2 * 3.14159265358979
            self.perturbation_count = torch.where(self.pert_on, self.perturbation_count + 1, self.perturbation_count)
            self._push[:, 0] = torch.where(self.pert_on, self.magnitude * torch.cos(self.phase), torch.zeros_like(self.magnitude))
            self._push[:, 1] = torch.where(self.pert_on, self.magnitude * torch.sin(self.phase), torch.zeros_like(self.magnitude))
            done = self.perturbation_count == self.pert_duration
            self.pert_on[done] = False
            self.perturbation_count[done] = 0
            push = self._push
        if self.vel_change and (self._graph is not None or self._capturing):
            # the same command ramp without a host round trip (a recorded step cannot size its draws by a count): draws for every
            # env, kept where the env changes its command.  Same distributions, another use of the generator's stream than the
            # reference's, which draws exactly as many numbers as envs change (the eager branch below, pinned by the replay tests)
            change = (self.epi_len % int(self.max_episode_length / 2)) == int(self.max_episode_length / 4 - 1)
            self.vel_change_duration.copy_(torch.where(change, self._rng.randint(1, 250, (N,)), self.vel_change_duration))
            self.cur_vel_change_duration.copy_(torch.where(change, torch.zeros_like(self.cur_vel_change_duration), self.cur_vel_change_duration))
            ch = change.unsqueeze(-1)
            self.start_target_vel.copy_(torch.where(ch, self.commands, self.start_target_vel))
            fin = torch.stack((self._rand_float(self.c_x[0], self.c_x[1], (N,)), self._rand_float(self.c_y[0], self.c_y[1], (N,)),
                               self._rand_float(self.c_yaw[0], self.c_yaw[1], (N,))), dim=-1)
            self.final_target_vel.copy_(torch.where(ch, fin, self.final_target_vel))
            mask = self.cur_vel_change_duration < self.vel_change_duration
            ramp = self.start_target_vel + (self.final_target_vel - self.start_target_vel) * self.cur_vel_change_duration.unsqueeze(-1) / self.vel_change_duration.unsqueeze(-1)
            self.commands.copy_(torch.where(mask.unsqueeze(-1), ramp, self.commands))
            self.cur_vel_change_duration += mask.long()
        elif self.vel_change:
            change = (self.epi_len % int(self.max_episode_length / 2)) == int(self.max_episode_length / 4 - 1)
            nc = int(change.sum())
            # (the four draws happen every step, of size zero when no env changes its command: the reference's generator stream)
            self.vel_change_duration[change] = self._rng.randint(1, 250, (nc,))
            self.cur_vel_change_duration[change] = 0
            self.start_target_vel[change] = self.commands[change]
            self.final_target_vel[change, 0] = self._rand_float(self.c_x[0], self.c_x[1], (nc,))
            self.final_target_vel[change, 1] = self._rand_float(self.c_y[0], self.c_y[1], (nc,))
            self.final_target_vel[change, 2] = self._rand_float(self.c_yaw[0], self.c_yaw[1], (nc,))
            mask = self.cur_vel_change_duration < self.vel_change_duration
            # (the reference's order of operations, :690-691: difference times elapsed, then divided by the duration)
            ramp = self.start_target_vel + (self.final_target_vel - self.start_target_vel) * self.cur_vel_change_duration.unsqueeze(-1) / self.vel_change_duration.unsqueeze(-1)
            self.commands.copy_(torch.where(mask.unsqueeze(-1), ramp, self.commands))
            self.cur_vel_change_duration += mask.long()
        for _ in range(self.control_freq_inv):
            upper = self.p_gains[12:] * (self.init_angle[12:] - self._dof_pos[:, 12:]) + self.d_gains[12:] * (-self._dof_vel[:, 12:])
            if self._pd_control:
                tar = self._pd_action_offset[:12] + self._pd_action_scale[:12] * self.actions
                lower = self.p_gains[:12] * (tar - self._dof_pos[:, :12]) + self.d_gains[:12] * (-self._dof_vel[:, :12])
            else:
                lower = self.actions * self.motor_efforts.unsqueeze(0) * self.power_scale
                lower = torch.max(torch.min(lower, self.motor_efforts.unsqueeze(0)), -self.motor_efforts.unsqueeze(0))
                # delayed-torque FIFO (:712-724): newest at the end, read `delay_idx` back once the FIFO has filled that far
                self.action_log.copy_(torch.cat((self.action_log[:, 1:], lower.unsqueeze(1)), dim=1))
                self.simul_len.copy_((self.simul_len + 1).clamp(max=self._log_slots, min=0))
                filled = self.simul_len > self.delay_idx
                delayed = torch.where(filled.unsqueeze(-1), self.action_log[self._env_ar, self.delay_idx], self.action_log[self._env_ar, -self.simul_len])
                if self.noise:
                    lower = delayed
            self._simulate(torch.cat((lower, upper), dim=1).contiguous(), push)
            push = None                                  # (applied forces last one simulate())
            if self.noise:
                z = self._rng.normal((N, 33), 0.00016 / 3.0)
                self.qpos_noise.copy_(self._dof_pos + torch.clamp(z, min=-0.00016, max=0.00016))
            else:
                self.qpos_noise.copy_(self._dof_pos)
            self.qvel_noise.copy_(self._div(self.qpos_noise - self.qpos_pre, self.dt))
            self.qpos_pre.copy_(self.qpos_noise)
        self.epi_len += 1

    # ------------------------------------------------------------------ post-physics (:750-804)
    def post_physics_step(self):
        N = self.num_envs
        self.progress_buf += 1
        self.randomize_buf += 1
        self._refresh_sim_tensors()
        self._compute_observations()
        dv = self._dof_vel.contiguous()
        self._chk(self._api["amp_reward"](N, _p(self._root_states), _p(dv), _p(self._dof_vel_pre), _p(self.commands), _p(self.actions),
                                          _p(self.actions_pre), _p(self.motor_efforts), _p(self._contact_forces), _p(self.total_mass),
                                          _p(self.rew_buf), _p(self._reward_values), self._stream()))
        self.extras["reward_names"] = list(REWARD_NAMES)
        self.extras["reward_values"] = self._reward_values
        self._chk(self._api["amp_reset"](N, _p(self.progress_buf), _p(self._contact_forces), _p(self._contact_body_ids), 2,
                                         _p(self._rigid_body_pos), _p(self._rigid_body_rot), float(self.max_episode_length),
                                         int(bool(self._enable_early_termination)), float(self._termination_height),
                                         _p(self.reset_buf), _p(self._terminate_buf), self._stream()))
        self.extras["terminate"] = self._terminate_buf
        self._dof_vel_pre.copy_(dv)
        self.actions_pre.copy_(self.actions)
        # the AMP subclass' part (tasks/tocabi_amp_lower.py:88-96)
        self._update_hist_amp_obs()
        self._compute_amp_observations()
        self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.num_amp_obs)

    # ------------------------------------------------------------------ the fused step (include/dyros_walk.h: dw_amp_step_*)
    def _fused_tables(self):
        """DwAmpConfig / DwAmpBuffers over the class' own tensors (they are updated in place and never re-bound, so the table is
        built once)."""
        from . import abi
        if getattr(self, "_amp_cfg", None) is not None:
            return self._amp_cfg, self._amp_buf
        c = abi.DwAmpConfig()
        c.num_envs, c.num_his, c.num_skip, c.log_slots, c.amp_steps = self.num_envs, self.num_obs_his, self.num_obs_skip, self._log_slots, self._num_amp_obs_steps
        c.pd_control, c.noise, c.vel_change = int(bool(self._pd_control)), int(bool(self.noise)), int(bool(self.vel_change))
        c.local_root_obs, c.enable_early_termination = int(self._local_root_obs), int(bool(self._enable_early_termination))
        c.clip_actions, c.clip_obs = float(self.clip_actions), float(self.clip_obs)
        c.max_episode_length, c.termination_height = float(self.max_episode_length), float(self._termination_height)
        c.inv_dt, c.dt, c.gpu_div = float(np.float32(1.0 / self.dt)), float(self.dt), int(self._gpu_div)
        for i, (lo, hi) in enumerate((self.c_x, self.c_y, self.c_yaw)):
            c.cmd_lo[i], c.cmd_scale[i] = float(lo), float(hi - lo)
        c.hist_ring, c.device_draws, c.randomize = int(self._hist_ring), int(self._device_draws), int(bool(self.randomize))
        c.seed = int(self.cfg.get("seed", 42)) & (2 ** 64 - 1)
        c.delay_idx_range[0], c.delay_idx_range[1] = 1 + int(0.002 / self.dt), 1 + round(0.01 / self.dt)
        dofp = self.randomization_params["actor_params"]["humanoid"].get("dof_properties", {}) if self.randomize else {}
        c.dr_frequency = int(self.randomization_params.get("frequency", 1))
        c.dr_damping, c.dr_armature = int("damping" in dofp), int("armature" in dofp)
        if "damping" in dofp:
            c.dr_damping_range[0], c.dr_damping_range[1] = (float(v) for v in dofp["damping"]["range"])
        if "armature" in dofp:
            c.dr_armature_range[0], c.dr_armature_range[1] = (float(v) for v in dofp["armature"]["range"])
        t = {"actions": self.actions, "actions_pre": self.actions_pre, "action_history": self.action_history, "obs_history": self.obs_history,
             "commands": self.commands, "start_target_vel": self.start_target_vel, "final_target_vel": self.final_target_vel,
             "vel_change_duration": self.vel_change_duration, "cur_vel_change_duration": self.cur_vel_change_duration, "epi_len": self.epi_len,
             "power_scale": self.power_scale, "action_log": self.action_log, "delay_idx": self.delay_idx, "simul_len": self.simul_len,
             "qpos_noise": self.qpos_noise, "qvel_noise": self.qvel_noise, "qpos_pre": self.qpos_pre, "qpos_bias": self.qpos_bias,
             "quat_bias": self.quat_bias, "dof_vel_pre": self._dof_vel_pre, "tau": self._tau, "progress_buf": self.progress_buf,
             "randomize_buf": self.randomize_buf, "reset_buf": self.reset_buf, "terminate_buf": self._terminate_buf, "timeout_buf": self.timeout_buf,
             "rigid_body_pos": self._rigid_body_pos, "rigid_body_rot": self._rigid_body_rot, "foot_pos": self._foot_pos, "obs1": self._obs1,
             "obs_buf": self.obs_buf, "obs_out": self._obs_out, "rew_buf": self.rew_buf, "reward_values": self._reward_values,
             "total_mass": self.total_mass, "amp_obs_buf": self._amp_obs_buf, "amp_obs1": self._amp_obs1, "motor_efforts": self.motor_efforts,
             "p_gains": self.p_gains, "d_gains": self.d_gains, "init_angle": self.init_angle,
             "pd_action_offset": self._pd_action_offset if self._pd_control else None, "pd_action_scale": self._pd_action_scale if self._pd_control else None,
             "epi_len_log": self.epi_len_log, "perturbation_count": self.perturbation_count, "perturb_timing": self.perturb_timing,
             "pert_on": self.pert_on, "initial_root_states": self._initial_root_states, "hist_head": self._hist_head, "draw_ctr": self._draw_ctr,
             "nominal_damping": self._nominal_damping, "nominal_armature": self._nominal_armature}
        b = abi.DwAmpBuffers()
        for name in abi.AMP_BUFFER_NAMES:
            v = t[name]
            if v is not None:
                assert v.is_contiguous() and v.device == self.actions.device, name
            setattr(b, name, v.data_ptr() if v is not None else None)
        self._amp_cfg, self._amp_buf, self._amp_keep = c, b, t
        return c, b

    def _step_fused(self, actions):
        """One step with the bookkeeping in three kernels (dw_amp_step_begin / _mid / _end) around the physics launches.  The draws
        are torch's, in the order of the torch implementation's recorded branch (pre_physics_step), so both give the same numbers
        (tests/test_amp_gpu.py) -- or, with sim.mi355.amp_device_draws, the kernels' own."""
        if self.perturb:
            raise ValueError("amp_fused: env.perturbation is not part of the fused step")
        N, api, st, h = self.num_envs, self._api, self._stream(), self._phys._h
        c, b = self._fused_tables()
        a = actions.to(self._tdev).float().contiguous()
        dd = self._device_draws
        rd = ru = None
        if self.vel_change and not dd:
            rd = self._rng.randint(1, 250, (N,))
            ru = torch.stack((self._rand(N), self._rand(N), self._rand(N)), dim=-1).contiguous()
        K = self.control_freq_inv
        if self._one_launch:
            # ONE launch.  The draws do not depend on the physics, so they are made up front, in the order the separate launches make them
            # between the substeps (z of substep 0 .. K - 1, then the root-velocity noise): the same numbers.
            zs = [self._rng.normal((N, 33), 0.00016 / 3.0).contiguous() if self.noise and not dd else None for _ in range(K)]
            nz = None
            if not dd:
                nz = self._rand(N, 6) * 0.05 - 0.025 if self.noise else torch.zeros(N, 6, device=self._tdev)
            zp = (C.c_void_p * K)(*[_p(z) for z in zs])
            self._z_keep = (zs, zp, nz, a, rd, ru)          # (alive until the launch has read them; a captured step keeps them for its replays)
            self._chk(api["amp_step"](h, C.byref(c), C.byref(b), _p(a), _p(rd), _p(ru), zp, K, _p(nz), st))
            self.time_step += 1
        else:
            self._chk(api["amp_step_begin"](h, C.byref(c), C.byref(b), _p(a), _p(rd), _p(ru), st))
            z = None
            for k in range(K):
                self._simulate(self._tau, None)
                z = self._rng.normal((N, 33), 0.00016 / 3.0).contiguous() if self.noise and not dd else None
                if k + 1 < K:
                    self._chk(api["amp_step_mid"](h, C.byref(c), C.byref(b), _p(z), k + 1, st))
            self.time_step += 1
            nz = None
            if not dd:
                nz = self._rand(N, 6) * 0.05 - 0.025 if self.noise else torch.zeros(N, 6, device=self._tdev)
            self._chk(api["amp_step_end"](h, C.byref(c), C.byref(b), _p(z), K - 1, _p(nz), st))
        self.extras["reward_names"] = list(REWARD_NAMES)
        self.extras["reward_values"] = self._reward_values
        self.extras["terminate"] = self._terminate_buf
        self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.num_amp_obs)
        self.extras["time_outs"] = self.timeout_buf
        self.obs_dict["obs"] = self._obs_out
        return self.obs_dict, self.rew_buf, self.reset_buf, self.extras

    def history_linear(self):
        """(action_history, obs_history) in the reference's layout (newest slot last), whatever the layout in memory: with
        sim.mi355.amp_hist_ring the two tensors are rings and this gathers through the heads."""
        if not self._hist_ring:
            return self.action_history, self.obs_history
        NH = self.num_obs_his * self.num_obs_skip
        ar = torch.arange(NH, device=self._tdev).unsqueeze(0)
        out = []
        for k, (t, w) in enumerate(((self.action_history, NUM_ACTIONS), (self.obs_history, NUM_OBS))):
            idx = (self._hist_head[:, k].long().unsqueeze(1) + ar) % NH
            out.append(torch.gather(t.view(self.num_envs, NH, w), 1, idx.unsqueeze(-1).expand(-1, -1, w)).reshape(self.num_envs, NH * w))
        return tuple(out)

    def reset_done(self):
        """VecTask.reset_done (tasks/base/vec_task.py:376-391).  With the fused reset and device draws: ONE launch over all envs
        (dw_amp_reset_done acts on the envs whose reset_buf is set), queued before the host asks which envs those were."""
        if not (self._device_draws and self._state_init == "Default"):
            return super().reset_done()
        # The ids go back to the host as a tensor of their own length, so the host has to learn the count.  dw_amp_reset_ids (ONE launch, queued
        # BEFORE the reset, which clears the flags) compacts the ids on the device and stores their number straight into pinned host memory; the
        # host waits for the event behind that launch only -- the reset kernel runs meanwhile -- and slices.  (Round 5: done.nonzero() behind the
        # reset launch: 0.094 ms per call at 16384 envs; round 6 first: ne / sum / pinned copy / nonzero_static, seven torch launches: 0.07 ms.)
        if getattr(self, "_cnt_pin", None) is None:
            self._cnt_pin = torch.zeros(1, dtype=torch.int64).pin_memory()
            self._cnt_dev = torch.zeros(1, dtype=torch.int64, device=self._tdev)
            self._cnt_evt = torch.cuda.Event()
        if self.reset_buf.dtype != torch.int64 or not self.reset_buf.is_contiguous():
            raise ValueError("reset_done: reset_buf must be a contiguous int64 tensor")
        # (the ids land in a tensor of this call's own -- room for every env, 8 B each, recycled by torch's allocator when the caller lets go of
        #  the slice it gets: no copy into a right-sized tensor afterwards, which was one more launch on the stream)
        ids_all = torch.empty(self.num_envs, dtype=torch.int64, device=self._tdev)
        self._chk(self._api["amp_reset_ids"](self.reset_buf.data_ptr(), self.num_envs, ids_all.data_ptr(), self._cnt_dev.data_ptr(),
                                             self._cnt_pin.data_ptr(), self._stream()))
        self._cnt_evt.record(torch.cuda.current_stream(self._tdev))
        c, b = self._fused_tables()
        self._chk(self._api["amp_reset_done"](self._phys._h, C.byref(c), C.byref(b), None, self._stream()))
        self.obs_dict["obs"] = self._obs_out
        self._cnt_evt.synchronize()
        ids = ids_all[:int(self._cnt_pin[0])]
        if len(ids) > 0:
            self.time_step = 0          # (as reset_idx: only when some env was reset)
            self._reset_default_env_ids = ids
        return self.obs_dict, ids

    def _step_body(self, actions):
        if self._fused:
            return self._step_fused(actions)
        action_tensor = torch.clamp(actions, -self.clip_actions, self.clip_actions)
        self.pre_physics_step(action_tensor)
        self.post_physics_step()
        self.timeout_buf.copy_((self.progress_buf >= self.max_episode_length - 1) & (self.reset_buf != 0))
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.obs_dict["obs"] = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def step(self, actions):                              # :806-845
        if self._graph is None:
            return self._step_body(actions)
        self._g_actions.copy_(actions)
        self._graph.replay()
        self.obs_dict["obs"] = self._g_obs               # (reset_done() puts its own tensor there between two steps)
        return self._g_out

    def enable_graph_step(self, warmup: int = 3):
        """Records one step() -- the ~100 elementwise torch launches, the two dw_simulate launches and the HIP entry points between
        them -- in a hipGraph and replays it from then on (reset_done() stays eager: it returns ids to the host).  What changes for
        the caller: the tensors step() returns are the same objects every time, overwritten by the next step (copy what you keep);
        the command ramp draws for every env instead of for the envs that change (pre_physics_step).  `warmup` real steps with
        zero actions run first, as a capture requires.  Needs perturbation off (its schedule reads a mean back to the host)."""
        if self.perturb:
            raise ValueError("enable_graph_step: env.perturbation reads a population mean on the host every step; not capturable")
        if self.rl_device != self.device:
            raise ValueError("enable_graph_step: rl_device must be the simulation device")
        if not self._gpu_div and not self._fused:
            raise ValueError("enable_graph_step: sim.mi355.torch_gpu_div = False makes the torch step build a tensor from a host scalar "
                             "every substep (the CPU flavour of `x / dt`, for replaying CPU fixtures); not capturable")
        dev = self._tdev
        self._g_actions = torch.zeros(self.num_envs, NUM_ACTIONS, device=dev)
        self._capturing = True
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self._step_body(self._g_actions)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        graph.register_generator_state(self._rng.gen)
        with torch.no_grad(), torch.cuda.graph(graph, stream=side):
            self._g_out = self._step_body(self._g_actions)
        self._g_obs = self.obs_dict["obs"]
        self._graph = graph              # (_capturing stays set: pre_physics_step keeps to the recorded branch if it is ever run eagerly again)

    def close(self):
        self._phys.close()
