"""DyrosDynamicWalk on MI355X: the host side of the fused step.

Public surface = the reference's task class (python/IsaacGymEnvs/isaacgymenvs/tasks/dyros_dynamic_walk.py) as
rl_games drives it through RLGPUEnv (utils/rlgames_utils.py:157-186): constructor
`(cfg, sim_device, graphics_device_id, headless)`, `step / reset / reset_done / reset_idx`, spaces, `num_envs`,
`max_episode_length`, `extras` with `time_outs`, `stacked_rewards`, `reward_names`, and the reference's attribute
names for every per-env tensor.  Everything numeric happens in libdyroswalk_hip.so (one launch per step); torch
owns the device buffers and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict

import numpy as np
import torch

from . import _lib, abi
from .model import NUM_BODIES, NUM_DOF, load_model
from .task_constants import REWARD_NAMES, load_task_constants
from .vec_task import VecTask

_TORCH_DT = {"f4": torch.float32, "i8": torch.int64, "i2": torch.int16}


def _u(gen, shape, device):
    return torch.rand(shape, generator=gen, device=device, dtype=torch.float32)


class DyrosDynamicWalk(VecTask):

    def __init__(self, cfg: Dict[str, Any], sim_device: str, graphics_device_id: int = 0, headless: bool = True):
        from .config import validate_cfg
        validate_cfg(cfg)                       # host-side range checks before the library is loaded or memory allocated
        self.cfg = cfg
        env_cfg = cfg["env"]
        self.randomization_params = cfg["task"]["randomization_params"]
        self.randomize = cfg["task"]["randomize"]
        self.death_cost = env_cfg["deathCost"]
        self.termination_height = env_cfg["terminationHeight"]
        self.max_episode_length_s = env_cfg["episodeLength"]
        # reference :35 -- a float, 8000.0
        self.max_episode_length = self.max_episode_length_s / (cfg["sim"].get("dt") * env_cfg.get("controlFrequencyInv", 8))
        self.num_obs_his = env_cfg["NumHis"]
        self.num_obs_skip = env_cfg["NumSkip"]
        self.initial_height = env_cfg["initialHieght"]
        self.num_single_step_obs = env_cfg["NumSingleStepObs"]
        self.num_action = env_cfg["NumAction"]
        env_cfg["numObservations"] = (self.num_single_step_obs + self.num_action) * (self.num_obs_his - 1) + self.num_single_step_obs
        env_cfg["numActions"] = self.num_action
        self.perturb = env_cfg["perturbation"]
        if (self.num_obs_his, self.num_obs_skip, self.num_single_step_obs, self.num_action) != (10, 2, 37, 13):
            raise ValueError("the MI355X step kernel is specialised for NumHis=10, NumSkip=2, NumSingleStepObs=37, NumAction=13")

        super().__init__(config=cfg, sim_device=sim_device, graphics_device_id=graphics_device_id, headless=headless)

        if not torch.cuda.is_available():
            raise _lib.DyrosWalkLibraryError("no GPU visible to torch; the MI355X step cannot run")
        self._lib, self._api = _lib.load()
        self._tdev = torch.device(self.device)
        torch.cuda.set_device(self._tdev)

        self.model = load_model()
        self.num_bodies, self.num_dof = NUM_BODIES, NUM_DOF
        self.left_foot_idx, self.right_foot_idx = self.model.left_foot_idx, self.model.right_foot_idx
        self.pelvis_idx = self.model.pelvis_idx
        self.non_feet_idxs = self.model.non_feet_idxs()
        self.dt = cfg["sim"]["dt"]
        self.skipframe = env_cfg.get("controlFrequencyInv", 8)
        self.dt_policy = self.dt * self.skipframe
        self.mocap_data_num = 3599
        self.mocap_cycle_dt = 0.0005
        self.mocap_cycle_period = self.mocap_data_num * self.mocap_cycle_dt

        # terrain (reference :47, :203-213): a TerrainCfg with the reference's defaults ('plane'), overridden by an
        # optional cfg["terrain"] mapping since this package has no class to edit
        from .terrain import Terrain, TerrainCfg
        self.terrain_cfg = TerrainCfg(**dict(cfg.get("terrain", {}) or {}))
        if self.terrain_cfg.mesh_type not in (None, "none", "plane", "heightfield", "trimesh"):
            raise ValueError("Terrain mesh type not recognised. Allowed types are [None, plane, heightfield, trimesh]")
        self.custom_origins = self.terrain_cfg.mesh_type in ("heightfield", "trimesh")
        if self.custom_origins:
            self.terrain = Terrain(self.terrain_cfg, self.num_envs, seed=int(cfg.get("seed", 42)))
        self._tc = load_task_constants()
        self._make_config()
        self._create_native()
        self._allocate()
        self._initial_state()
        self._bind()
        self._step_count = 0
        # cfg sim.mi355.device_step_counter: the step counter lives in device memory (dw_step_dev), so that step() can be captured
        # in a hipGraph and replayed
        self._step_dev = None
        if self.cfg["sim"].get("mi355", {}).get("device_step_counter", False):
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=self._tdev)
        self._bound_obs = self._buf["obs_buf"]
        self._fresh_obs = (not self.alias_obs) and np.isinf(self.clip_obs)
        self.extras["reward_names"] = list(REWARD_NAMES)

    # ------------------------------------------------------------------ native handle
    def _make_config(self):
        c = abi.DwConfig()
        self._api["default_config"](C.byref(c))
        sim, px, mi = self.cfg["sim"], self.cfg["sim"]["physx"], self.cfg["sim"].get("mi355", {})
        c.dt = float(sim["dt"])
        c.num_envs = self.num_envs
        c.control_freq_inv = int(self.skipframe)
        for i in range(3):
            c.gravity[i] = float(sim["gravity"][i])
        c.solver_iterations = int(px.get("num_position_iterations", 4)) + int(px.get("num_velocity_iterations", 1))
        c.contact_offset = float(px.get("contact_offset", 0.002))
        c.max_depenetration_velocity = float(px.get("max_depenetration_velocity", 10.0))
        c.friction = float(mi.get("plane_friction", 1.0))
        c.erp = float(mi.get("erp", 0.2))
        c.contact_cfm = float(mi.get("contact_cfm", 1e-3))
        c.penalty_stiffness = float(mi.get("penalty_stiffness", 1e5))
        c.penalty_damping = float(mi.get("penalty_damping", 1e3))
        c.max_angular_velocity = 100.0                         # reference :289
        c.max_episode_length = float(self.max_episode_length)
        c.initial_height = float(self.initial_height)
        c.death_cost = float(self.death_cost)
        c.perturb = int(bool(self.perturb))
        c.force_perturb_start = int(bool(mi.get("force_perturb_start", False)))
        ap = self.randomization_params.get("actor_params", {}).get("humanoid", {}) if self.randomize else {}
        dofp = ap.get("dof_properties", {})
        c.randomize_dof_on_reset = int("damping" in dofp or "armature" in dofp)
        if "damping" in dofp:
            c.dr_damping_add[0], c.dr_damping_add[1] = dofp["damping"]["range"]
        else:
            c.dr_damping_add[0] = c.dr_damping_add[1] = 0.0
        if "armature" in dofp:
            c.dr_armature_scale[0], c.dr_armature_scale[1] = dofp["armature"]["range"]
        else:
            c.dr_armature_scale[0] = c.dr_armature_scale[1] = 1.0
        fr = ap.get("rigid_shape_properties", {}).get("friction")
        c.randomize_friction_on_reset = int(fr is not None)
        if fr is not None:
            c.dr_friction_scale[0], c.dr_friction_scale[1] = fr["range"]
        c.timeout_fix = int(bool(mi.get("timeout_fix", False)))
        c.root_vel_at_com = int(bool(mi.get("root_vel_at_com", True)))
        c.torch_gpu_div = int(bool(mi.get("torch_gpu_div", True)))
        c.self_collision = int(bool(mi.get("self_collision", True)))
        c.debug_freeze_physics = int(bool(mi.get("debug_freeze_physics", False)))
        c.seed = int(self.cfg.get("seed", 42)) & 0xFFFFFFFFFFFFFFFF
        # which kernels: 0/3 = octet kernels, 8 lanes per env, two waves per SIMD (the only generation shipped)
        # kernels of round 1; one launch per policy step in all three
        c.pipeline = {"auto": 0, "oct": 3}.get(mi.get("pipeline", 0), mi.get("pipeline", 0))
        c.debug_wave_build = int(mi.get("debug_wave_build", 0))          # tests: force the one- / two-waves-per-SIMD build of the kernels
        tc = self.terrain_cfg
        c.terrain = int(self.custom_origins)
        c.custom_origins = int(self.custom_origins)
        if self.custom_origins:
            c.terrain_rows, c.terrain_cols = int(self.terrain.tot_rows), int(self.terrain.tot_cols)
            c.terrain_hscale, c.terrain_vscale = float(tc.horizontal_scale), float(tc.vertical_scale)
            c.terrain_border = float(tc.border_size)
            c.terrain_curriculum = int(bool(tc.curriculum))
            c.terrain_num_levels, c.terrain_num_types = int(tc.num_rows), int(tc.num_cols)
            c.terrain_env_length = float(self.terrain.env_length)
            c.max_episode_length_s = float(self.max_episode_length_s)
            c.friction = float(mi.get("plane_friction", tc.static_friction))
        self._ccfg = c

    def _create_native(self):
        tc = abi.DwTaskConst()
        self._tc_keep = {}
        for k, _ in abi.DwTaskConst._fields_:
            arr = np.ascontiguousarray(self._tc[k], dtype=np.float32).ravel()
            self._tc_keep[k] = arr
            setattr(tc, k, arr.ctypes.data_as(C.POINTER(C.c_float)))
        self._cmodel = self.model.to_c()
        h = C.c_void_p()
        _lib.check(self._api, self._api["create"](C.byref(self._ccfg), C.byref(self._cmodel), C.byref(tc), C.byref(h)))
        self._h = h

    def _allocate(self):
        N, dev = self.num_envs, self._tdev
        self._buf = {}
        for name, (shape, dt) in abi.BUFFER_SPECS.items():
            full = (abi.GLOBAL_WORDS[name],) if shape is None else (N,) + tuple(shape)
            self._buf[name] = torch.zeros(full, dtype=_TORCH_DT[dt], device=dev)
        b = self._buf
        # VecTask.allocate_buffers (reference vec_task.py:233-256)
        self.obs_buf, self.rew_buf = b["obs_buf"], b["rew_buf"]
        self.reset_buf, self.progress_buf = b["reset_buf"], b["progress_buf"]
        self.timeout_buf, self.randomize_buf = b["timeout_buf"], b["randomize_buf"]
        self.states_buf = torch.zeros((N, self.num_states), device=dev, dtype=torch.float)
        self.reset_buf.fill_(1)
        # Gym tensors (reference :73-107)
        self.root_states = b["root_states"]
        self.dof_state = b["dof_state"].view(N * NUM_DOF, 2)
        self.dof_pos = b["dof_state"][..., 0]
        self.dof_vel = b["dof_state"][..., 1]
        self.contact_forces = b["contact_forces"]
        self.total_mass = b["total_mass"].view(N, 1)
        self.env_origins = b["env_origins"]
        if self.custom_origins:         # the two terrain tables replace their one-element placeholders (reference :253-254, :703)
            b["height_samples"] = torch.from_numpy(np.ascontiguousarray(self.terrain.heightsamples)).to(dev)
            b["terrain_origins"] = torch.from_numpy(self.terrain.env_origins).to(dev).to(torch.float).contiguous()
            self.height_samples = b["height_samples"].view(self.terrain.tot_rows, self.terrain.tot_cols)
            self.terrain_origins = b["terrain_origins"]
            self.terrain_levels, self.terrain_types = b["terrain_levels"], b["terrain_types"]
            self.max_terrain_level = self.terrain_cfg.num_rows

    def _bind(self):
        db = abi.DwBuffers()
        for name in abi.BUFFER_NAMES:
            setattr(db, name, self._buf[name].data_ptr())
        _lib.check(self._api, self._api["bind"](self._h, C.byref(db)))

    # ------------------------------------------------------------------ initial state (reference :94-195, :199-225)
    def _initial_state(self):
        N, dev, b = self.num_envs, self._tdev, self._buf
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(self.cfg.get("seed", 42)))
        self._gen = gen
        es = b["env_state"]
        tc = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in self._tc.items() if k != "mocap"}
        self.Kp, self.Kv = tc["kp"], tc["kv"]
        self.action_high = tc["action_high"]
        self.obs_mean, self.obs_var = tc["obs_mean"], tc["obs_var"]
        self.initial_dof_pos = tc["initial_dof_pos"].unsqueeze(0).expand(N, NUM_DOF)
        self.initial_dof_vel = torch.zeros((N, NUM_DOF), device=dev)
        self.dof_limits_lower = torch.tensor(self.model.dof_lower, device=dev, dtype=torch.float)
        self.dof_limits_upper = torch.tensor(self.model.dof_upper, device=dev, dtype=torch.float)
        if self.custom_origins:
            # origins on the terrain tiles (reference :697-707): a random starting level, the type from the env index
            tcf = self.terrain_cfg
            max_init_level = tcf.max_init_terrain_level if tcf.curriculum else tcf.num_rows - 1   # range: validate_cfg
            lv = torch.randint(0, max_init_level + 1, (N,), generator=gen, device=dev)
            ty = torch.div(torch.arange(N, device=dev), (N / tcf.num_cols), rounding_mode="floor").to(torch.long)
            ty = ty.clamp_(0, tcf.num_cols - 1)               # float rounding of N / num_cols must not index past the last column
            b["terrain_levels"].copy_(lv)
            b["terrain_types"].copy_(ty)
            b["env_origins"].copy_(b["terrain_origins"].view(tcf.num_rows, tcf.num_cols, 3)[lv, ty])
        else:
            # env origins: grid (reference :709-718), robots start at origin + U(-1,1) xy jitter (:350-353)
            num_cols = np.floor(np.sqrt(N))
            num_rows = np.ceil(N / num_cols)
            xx, yy = torch.meshgrid(torch.arange(num_rows), torch.arange(num_cols), indexing="ij")
            spacing = self.cfg["env"]["envSpacing"]
            org = torch.zeros((N, 3))
            org[:, 0] = spacing * xx.flatten()[:N]
            org[:, 1] = spacing * yy.flatten()[:N]
            b["env_origins"].copy_(org.to(dev))
        root = b["root_states"]
        root.zero_()
        root[:, 0:3] = b["env_origins"]
        root[:, 0:2] += _u(gen, (N, 2), dev) * 2.0 - 1.0
        root[:, 2] += self.initial_height
        root[:, 6] = 1.0
        b["dof_state"][..., 0] = tc["initial_dof_pos"]
        b["dof_state"][..., 1] = 0.0
        # physical parameters + the "first randomization" of apply_randomizations (reference :218-219, vec_task.py:519-733)
        b["friction_scale"].fill_(1.0)
        b["mass_scale"].fill_(1.0)
        b["dof_damping"][:] = tc["dof_damping_nominal"]
        b["dof_armature"][:] = tc["dof_armature_nominal"]
        if self.randomize:
            self._apply_setup_randomization(gen)
        masses = torch.zeros(NUM_BODIES, device=dev)
        for k, g in enumerate(self.model.inert_gym):
            masses[g] = float(self.model.inert_mass[k])
        b["total_mass"][:] = (b["mass_scale"] * masses).sum(dim=1)
        # task state (reference :110-195)
        v = lambda n: abi.es_view(es, n)          # noqa: E731
        v("qpos_noise").zero_(); v("qvel_noise").zero_(); v("qpos_pre").zero_()
        v("target_vel")[:, 0] = (_u(gen, (N, 1), dev) * 0.8)[:, 0]
        v("motor_constant_scale")[:] = _u(gen, (N, 12), dev) * 0.4 + 0.8
        v("qpos_bias")[:] = _u(gen, (N, 12), dev) * 6.28 / 100 - 3.14 / 100
        v("quat_bias")[:] = _u(gen, (N, 3), dev) * 6.28 / 150 - 3.14 / 150
        v("delay_idx")[:] = 1                      # delay_idx_tensor[:,1] = 1 (:161)
        v("pert_duration")[:] = torch.randint(1, 100, (N,), generator=gen, device=dev, dtype=torch.int32)
        v("perturb_timing")[:] = 1
        if self._ccfg.force_perturb_start:
            v("perturb_start")[:] = 1

    def _apply_setup_randomization(self, gen):
        N, dev, b = self.num_envs, self._tdev, self._buf
        ap = self.randomization_params.get("actor_params", {}).get("humanoid", {})
        m = ap.get("rigid_body_properties", {}).get("mass")
        if m is not None:
            lo, hi = m["range"]
            b["mass_scale"][:] = lo + _u(gen, (N, NUM_BODIES), dev) * (hi - lo)     # independent per body
        d = ap.get("dof_properties", {})
        if "damping" in d:
            lo, hi = d["damping"]["range"]
            b["dof_damping"][:] = b["dof_damping"] + (lo + _u(gen, (N, NUM_DOF), dev) * (hi - lo))
        if "armature" in d:
            lo, hi = d["armature"]["range"]
            b["dof_armature"][:] = b["dof_armature"] * (lo + _u(gen, (N, NUM_DOF), dev) * (hi - lo))
        f = ap.get("rigid_shape_properties", {}).get("friction")
        if f is not None:
            lo, hi = f["range"]
            b["friction_scale"][:] = lo + _u(gen, (N,), dev) * (hi - lo)

    # ------------------------------------------------------------------ reference attribute names (views of the record)
    def __getattr__(self, name):
        if name in abi.ES_FIELDS and "_buf" in self.__dict__:
            return abi.es_view(self._buf["env_state"], name)
        raise AttributeError(name)

    @property
    def delay_idx_tensor(self):
        N = self.num_envs
        return torch.stack([torch.arange(N, device=self._tdev), abi.es_view(self._buf["env_state"], "delay_idx").long()], 1)

    @property
    def simul_len_tensor(self):
        N = self.num_envs
        return torch.stack([torch.arange(N, device=self._tdev), abi.es_view(self._buf["env_state"], "simul_len").long()], 1)

    def _logical(self, ring):
        head = abi.es_view(self._buf["env_state"], "hist_head").long()
        idx = (head[:, None] + torch.arange(20, device=self._tdev)[None, :]) % 20
        return torch.gather(ring, 1, idx[:, :, None].expand(-1, -1, ring.shape[2])).reshape(self.num_envs, -1)

    @property
    def obs_history(self):
        """[N, 740] in the reference's order (slot 0 oldest)."""
        return self._logical(self._buf["obs_history"])

    @property
    def action_history(self):
        return self._logical(self._buf["action_history"])

    # ------------------------------------------------------------------ VecTask API
    def step(self, actions: torch.Tensor, noise: torch.Tensor = None):
        """One policy step (reference vec_task.py:293-344).  `noise`: optional [N, DW_NOISE_WORDS] injected-noise
        record (tests); None = in-kernel counter-based RNG."""
        a = actions.to(self._tdev)
        if a.dtype != torch.float32 or not a.is_contiguous():
            a = a.float().contiguous()
        if a.shape != (self.num_envs, self.num_actions):
            raise ValueError("actions must be [%d, %d]" % (self.num_envs, self.num_actions))
        nz = 0
        if noise is not None:
            if noise.shape != (self.num_envs, abi.K["DW_NOISE_WORDS"]) or noise.dtype != torch.float32 or not noise.is_contiguous():
                raise ValueError("noise must be a contiguous float32 [N, DW_NOISE_WORDS] tensor")
            nz = noise.data_ptr()
        stream = torch.cuda.current_stream(self._tdev).cuda_stream
        # The reference returns a FRESH observation tensor every step (torch.clamp(self.obs_buf, ...), vec_task.py:338).  Here the
        # kernel itself writes this step's observations into a newly allocated tensor (dw_step_obs: the kernels take the
        # buffer table by value, so the destination is a per-launch argument), which becomes self.obs_buf: the contract without the
        # 1 948 B-per-env copy after the kernel.  alias_obs, a clipping range and a captured graph (whose
        # pointers must not change) keep the bound buffer.
        fresh = self._fresh_obs and self._step_dev is None
        if fresh:
            out = torch.empty_like(self._bound_obs)
            _lib.check(self._api, self._api["step_obs"](self._h, a.data_ptr(), nz, self._step_count, None, out.data_ptr(), stream))
            self.obs_buf = self._buf["obs_buf"] = out          # (the newest observations; self._bound_obs stays the tensor dw_bind named)
        elif self._step_dev is not None:
            # (graph-capturable form: no argument changes from step to step; the kernel reads the counter, a one-thread launch adds 1)
            _lib.check(self._api, self._api["step_dev"](self._h, a.data_ptr(), nz, self._step_dev.data_ptr(), stream))
        else:
            _lib.check(self._api, self._api["step"](self._h, a.data_ptr(), nz, self._step_count, stream))
        self._step_count += 1
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.extras["stacked_rewards"] = self._buf["stacked_rewards"]
        if self.custom_origins and self.terrain_cfg.curriculum:
            # logging columns of the curriculum (reference :417-426): mean level of the envs of each terrain type.  The step kernel has
            # summed levels and counts per type while it ran; one small launch writes the [N, 15 + types] tensor (the reference loops
            # over the types with a nonzero() + sum + cat each)
            nc = self.terrain_cfg.num_cols
            if len(self.extras["reward_names"]) == len(REWARD_NAMES):
                self.extras["reward_names"] = list(REWARD_NAMES) + ["terrain %d level" % i for i in range(nc)]
            log = torch.empty(self.num_envs, self._buf["stacked_rewards"].shape[1] + nc, device=self._tdev, dtype=torch.float)
            _lib.check(self._api, self._api["terrain_log"](self._h, log.data_ptr(), stream))
            self.extras["stacked_rewards"] = log
        self.obs_dict["obs"] = (self.obs_buf if fresh else self._clip_obs(self.obs_buf)).to(self.rl_device)
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def reset_idx(self, env_ids: torch.Tensor, noise: torch.Tensor = None):
        ids = env_ids.to(device=self._tdev, dtype=torch.int32).contiguous()
        if ids.numel() == 0:
            return
        nz = noise.data_ptr() if noise is not None else 0
        stream = torch.cuda.current_stream(self._tdev).cuda_stream
        _lib.check(self._api, self._api["reset_idx"](self._h, ids.data_ptr(), int(ids.numel()), nz, self._current_step(), stream))
        # (reset_idx does not touch the observations: the reference recomputes them in the next step's post-physics, :655-669)

    def _current_step(self) -> int:
        """The step index that keys the in-kernel RNG.  With cfg sim.mi355.device_step_counter the truth is the device word:
        a replayed hipGraph advances it without running any Python, so the host's count is stale (one host sync here; do not
        call while a capture is open)."""
        if self._step_dev is not None:
            self._step_count = int(self._step_dev.item())
        return self._step_count

    def sync_perturbation_gate(self, group=None):
        """Optional, sharded runs only: lets this rank's push-perturbation gate latch on the means over ALL ranks' envs
        (tasks/dyros_dynamic_walk.py:489 as ONE simulation of every env would evaluate it) instead of its own -- one all-reduce of 3 doubles,
        called once per rollout horizon (isaacgymdyros_amd/dist.py sync_perturbation_gate).  Without it the gates are per rank, which is the
        reference's Horovod layout.  Reads the step counter (a host sync with sim.mi355.device_step_counter)."""
        from . import dist as dwdist
        steps = self._current_step()
        if steps < 1:
            return None
        dtp = float(self.cfg["sim"]["dt"]) * float(self.cfg["env"].get("controlFrequencyInv", 2))
        return dwdist.sync_perturbation_gate(self._buf["gate_acc"], steps, self.num_envs, float(self.max_episode_length), 8.0 / dtp, group)

    def kernel_info(self) -> dict:
        """Which device kernel one step() launches (bench.py names it in its roofline object; the rocprofv3 summaries under
        profiles/ carry the same name)."""
        return {"kernels": "dw_k_step_oct", "pipeline": "octet (8 lanes per env, 8 envs per wave, 2 waves per SIMD)", "launches_per_step": 1}

    def simulate(self, tau: torch.Tensor, push_xy: torch.Tensor = None):
        """One physics substep at the Gym boundary: set_dof_actuation_force_tensor + apply_rigid_body_force_tensors
        (base_link x/y) + simulate + the three refreshes (reference :502,520,525-526,547-549)."""
        t = tau.to(self._tdev).float().contiguous()
        if t.shape != (self.num_envs, NUM_DOF):
            raise ValueError("tau must be [%d, %d]" % (self.num_envs, NUM_DOF))
        p = 0
        if push_xy is not None:
            push_xy = push_xy.to(self._tdev).float().contiguous()
            p = push_xy.data_ptr()
        stream = torch.cuda.current_stream(self._tdev).cuda_stream
        _lib.check(self._api, self._api["simulate"](self._h, t.data_ptr(), p, stream))

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything the next step() depends on: the device buffers and the step counter that keys the in-kernel
        RNG.  The reference never checkpoints simulation state (rl_games only saves the policy); here it is free."""
        torch.cuda.synchronize(self._tdev)
        d = {k: v.detach().clone() for k, v in self._buf.items()}
        d["_step_count"] = self._current_step()
        return d

    def load_state_dict(self, d):
        """Restores a state_dict().  Shapes, dtypes and every field the kernels use as an index are checked on the host
        first: the buffers are restored verbatim and the step kernel gathers with them."""
        for k, v in self._buf.items():
            if k not in d:
                raise ValueError("state dict lacks buffer %r" % k)
            if tuple(d[k].shape) != tuple(v.shape) or d[k].dtype != v.dtype:
                raise ValueError("state dict buffer %r is %s %s, expected %s %s" % (k, tuple(d[k].shape), d[k].dtype, tuple(v.shape), v.dtype))
        if self.custom_origins:
            lv, ty = d["terrain_levels"], d["terrain_types"]
            if lv.numel() and (int(lv.min()) < 0 or int(lv.max()) >= self.terrain_cfg.num_rows):
                raise ValueError("state dict terrain_levels out of range [0, %d)" % self.terrain_cfg.num_rows)
            if ty.numel() and (int(ty.min()) < 0 or int(ty.max()) >= self.terrain_cfg.num_cols):
                raise ValueError("state dict terrain_types out of range [0, %d)" % self.terrain_cfg.num_cols)
        es = d["env_state"]
        for name, lo, hi in (("hist_head", 0, 19), ("delay_idx", 0, 5), ("init_mocap_data_idx", 0, 3599)):
            if name in abi.ES_FIELDS:
                f = abi.es_view(es, name)
                if f.numel() and (int(f.min()) < lo or int(f.max()) > hi):
                    raise ValueError("state dict env_state field %r out of range [%d, %d]" % (name, lo, hi))
        for k, v in self._buf.items():
            v.copy_(d[k].to(v.device))
        self._step_count = int(d["_step_count"])
        if self._step_dev is not None:
            self._step_dev.fill_(self._step_count)

    def close(self):
        if getattr(self, "_h", None) is not None:
            torch.cuda.synchronize(self._tdev)
            self._api["destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
