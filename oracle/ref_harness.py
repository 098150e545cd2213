"""In-container harness that runs the REFERENCE's own task class against a fake `gym`.

TEST INFRASTRUCTURE ONLY, and only usable where the reference checkout is mounted
(/root/reference; never on the GPU box).  It exists to (i) check the CPU oracle's task logic against the
reference's Python bit for bit and (ii) mint the committed fixtures under tests/golden/
(oracle/make_goldens.py).  Nothing of the reference is copied: its files are imported from where they lie.

Recipe: SURVEY.md appendix C.  The closed engine behind `gym.simulate` is replaced by the oracle's
physics substep (oracle/dw_physics.c) operating IN PLACE on the tensors the fake gym hands to the task,
so a whole `DyrosDynamicWalk.step()` of the reference runs end to end.  Every torch RNG draw the task
makes is recorded so the same numbers can be replayed into dw_step/dwo_step as the injected-noise record.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("DW_REFERENCE", "/root/reference")
PY = os.path.join(REF, "python")
IGE = os.path.join(PY, "IsaacGymEnvs", "isaacgymenvs")


def available() -> bool:
    return os.path.isfile(os.path.join(IGE, "tasks", "dyros_dynamic_walk.py"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)


class _Quat:
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class _Transform:
    def __init__(self, p=None, r=None):
        self.p = p or _Vec3()
        self.r = r or _Quat()


class _SimParams:
    def __init__(self):
        self.dt = 0.0
        self.substeps = 1
        self.up_axis = 1
        self.gravity = _Vec3()
        self.use_gpu_pipeline = False
        self.num_client_threads = 0
        self.physx = _Bag()
        self.flex = _Bag()


class _BodyProp:
    def __init__(self, mass):
        self.mass = mass


DOF_PROP_DTYPE = np.dtype([("hasLimits", "?"), ("lower", "f4"), ("upper", "f4"), ("driveMode", "i4"),
                           ("stiffness", "f4"), ("damping", "f4"), ("velocity", "f4"), ("effort", "f4"),
                           ("friction", "f4"), ("armature", "f4")])


class FakeGym:
    """The ~25 gymapi.Gym methods the task touches, backed by an OracleSim."""

    def __init__(self, sim):
        self.osim = sim                      # oracle.OracleSim: owns the state buffers
        self.model = sim.model
        self.N = sim.N
        self.calls = []
        self.frame = 0
        self.tau = None
        self.push = None
        self.root = torch.from_numpy(sim.buf["root_states"])
        self.dof = torch.from_numpy(sim.buf["dof_state"]).view(self.N * 33, 2)
        self.contact = torch.from_numpy(sim.buf["contact_forces"]).view(self.N * 38, 3)
        masses = np.zeros(38)
        for k, g in enumerate(self.model.inert_gym):
            masses[g] = self.model.inert_mass[k]
        self.nominal_mass = masses
        self.body_props = [[_BodyProp(float(m)) for m in masses] for _ in range(self.N)]
        dp = np.zeros(33, dtype=DOF_PROP_DTYPE)
        dp["hasLimits"] = True
        dp["lower"] = np.asarray(self.model.dof_lower, dtype=np.float32)
        dp["upper"] = np.asarray(self.model.dof_upper, dtype=np.float32)
        self.dof_props = [dp.copy() for _ in range(self.N)]
        self.envs = 0

    # ---- setup ----
    def create_sim(self, *a): return "sim"
    def add_ground(self, *a): pass
    def add_heightfield(self, *a): pass
    def add_triangle_mesh(self, *a): pass
    def load_asset(self, *a): return "asset"
    def find_asset_rigid_body_index(self, asset, name): return self.model.body_names.index(name)
    def get_asset_rigid_body_count(self, a): return 38
    def get_asset_dof_count(self, a): return 33
    def get_asset_joint_count(self, a): return 37
    def create_env(self, *a):
        self.envs += 1
        return self.envs - 1
    def create_actor(self, env, asset, pose, name, group, filt, seg=0):
        self.root[env, 0] = pose.p.x
        self.root[env, 1] = pose.p.y
        self.root[env, 2] = pose.p.z
        return 0
    def set_rigid_body_color(self, *a): pass
    def get_actor_dof_properties(self, env, handle): return self.dof_props[env].copy()
    def set_actor_dof_properties(self, env, handle, props):
        self.dof_props[env] = props.copy()
        self.osim.buf["dof_damping"][env] = props["damping"]
        self.osim.buf["dof_armature"][env] = props["armature"]
        return True
    def get_actor_rigid_body_properties(self, env, handle): return self.body_props[env]
    def set_actor_rigid_body_properties(self, env, handle, props, recompute=True):
        self.body_props[env] = props
        m = np.array([float(np.asarray(p.mass).reshape(-1)[0]) for p in props])
        sc = np.ones(38)
        nz = self.nominal_mass > 0
        sc[nz] = m[nz] / self.nominal_mass[nz]
        self.osim.buf["mass_scale"][env] = sc
        return True
    def find_actor_handle(self, env, name): return 0
    def get_actor_count(self, env): return 1
    def get_actor_handle(self, env, i): return 0
    def get_actor_name(self, env, h): return "humanoid"
    def get_actor_rigid_shape_count(self, env, h): return 61
    def get_actor_rigid_body_count(self, env, h): return 38
    def get_frame_count(self, sim): return self.frame
    def prepare_sim(self, sim): return True
    def acquire_actor_root_state_tensor(self, sim): return self.root
    def acquire_dof_state_tensor(self, sim): return self.dof
    def acquire_net_contact_force_tensor(self, sim): return self.contact
    # ---- per step ----
    def refresh_dof_state_tensor(self, sim): self.calls.append("refresh_dof"); return True
    def refresh_actor_root_state_tensor(self, sim): self.calls.append("refresh_root"); return True
    def refresh_net_contact_force_tensor(self, sim): self.calls.append("refresh_contact"); return True
    def set_dof_state_tensor(self, sim, t): return True
    def set_dof_actuation_force_tensor(self, sim, t):
        self.calls.append("set_tau")
        self.tau = t.detach().clone().view(self.N, 33).numpy()
        return True
    def apply_rigid_body_force_tensors(self, sim, forces, torques, space):
        self.calls.append("apply_forces")
        self.push = forces.view(self.N, 38, 3)[:, 0, 0:2].clone().numpy()
        return True
    def simulate(self, sim):
        self.calls.append("simulate")
        self.osim.simulate(self.tau, self.push)
        self.push = None           # applied forces last one simulate()
        self.frame += 1
    def fetch_results(self, sim, wait): pass
    def set_actor_root_state_tensor_indexed(self, sim, t, ids, n): self.calls.append("set_root_idx"); return True
    def set_dof_state_tensor_indexed(self, sim, t, ids, n): self.calls.append("set_dof_idx"); return True
    def get_sim_params(self, sim): return _SimParams()
    def set_sim_params(self, sim, p): pass
    def get_actor_tendon_properties(self, env, h): return []
    def set_actor_tendon_properties(self, env, h, p): return True
    def get_actor_rigid_shape_properties(self, env, h): return []
    def set_actor_rigid_shape_properties(self, env, h, p): return True


_loaded = {}


def load_reference(fake_gym_factory):
    """Import the reference task module with stubbed `isaacgym` / `gym` packages.  Returns the module dict."""
    if _loaded:
        _loaded["gymapi"].acquire_gym = fake_gym_factory
        return _loaded
    if not available():
        raise RuntimeError("reference checkout not present")
    np.float = float          # python/isaacgym/torch_utils.py:135 default argument
    np.Inf = np.inf           # tasks/base/vec_task.py:92-98

    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        sys.modules[name] = m
        return m

    isaacgym = pkg("isaacgym")
    gymapi = types.ModuleType("isaacgym.gymapi")
    gymtorch = types.ModuleType("isaacgym.gymtorch")
    sys.modules["isaacgym.gymapi"] = gymapi
    sys.modules["isaacgym.gymtorch"] = gymtorch
    isaacgym.gymapi, isaacgym.gymtorch = gymapi, gymtorch
    gymapi.Vec3, gymapi.Quat, gymapi.Transform, gymapi.SimParams = _Vec3, _Quat, _Transform, _SimParams
    for n in ("PlaneParams", "AssetOptions", "CameraProperties"):
        setattr(gymapi, n, type(n, (_Bag,), {}))
    for n in ("HeightFieldParams", "TriangleMeshParams"):      # these carry a transform (tasks/dyros_dynamic_walk.py:246-248)
        setattr(gymapi, n, type(n, (_Bag,), {"__init__": lambda self, **kw: (_Bag.__init__(self, **kw), setattr(self, "transform", _Transform()))[0]}))
    for i, n in enumerate(("SIM_PHYSX", "SIM_FLEX", "UP_AXIS_Y", "UP_AXIS_Z", "DOF_MODE_NONE", "MESH_VISUAL",
                           "ENV_SPACE")):
        setattr(gymapi, n, i)
    gymapi.ContactCollection = lambda v: v
    gymapi.acquire_gym = fake_gym_factory
    gymtorch.wrap_tensor = lambda t: t
    gymtorch.unwrap_tensor = lambda t: t
    gym = pkg("gym")
    spaces = types.ModuleType("gym.spaces")
    sys.modules["gym.spaces"] = spaces
    gym.spaces = spaces
    gym.Space = object

    class Box:
        def __init__(self, low, high):
            self.low, self.high, self.shape = low, high, low.shape
    spaces.Box = Box

    pkg("isaacgymenvs", IGE)
    pkg("isaacgymenvs.utils", os.path.join(IGE, "utils"))
    pkg("isaacgymenvs.cfg", os.path.join(IGE, "cfg"))
    pkg("isaacgymenvs.cfg.terrain", os.path.join(IGE, "cfg", "terrain"))
    pkg("isaacgymenvs.tasks", os.path.join(IGE, "tasks"))
    pkg("isaacgymenvs.tasks.base", os.path.join(IGE, "tasks", "base"))

    tu = _load("isaacgym.torch_utils", os.path.join(PY, "isaacgym", "torch_utils.py"))
    isaacgym.torch_utils = tu
    gu = _load("isaacgym.gymutil", os.path.join(PY, "isaacgym", "gymutil.py"))
    isaacgym.gymutil = gu
    _load("isaacgym.terrain_utils", os.path.join(PY, "isaacgym", "terrain_utils.py"))
    ju = _load("isaacgymenvs.utils.torch_jit_utils", os.path.join(IGE, "utils", "torch_jit_utils.py"))
    _load("isaacgymenvs.cfg.terrain.terrain_cfg", os.path.join(IGE, "cfg", "terrain", "terrain_cfg.py"))
    _load("isaacgymenvs.utils.terrain", os.path.join(IGE, "utils", "terrain.py"))
    vt = _load("isaacgymenvs.tasks.base.vec_task", os.path.join(IGE, "tasks", "base", "vec_task.py"))
    task = _load("isaacgymenvs.tasks.dyros_dynamic_walk", os.path.join(IGE, "tasks", "dyros_dynamic_walk.py"))
    _loaded.update(torch_utils=tu, gymutil=gu, jit_utils=ju, vec_task=vt, task=task, gymapi=gymapi)
    return _loaded


def reference_cfg(num_envs: int, randomize: bool = True, perturbation: bool = True):
    import yaml
    with open(os.path.join(IGE, "cfg", "task", "DyrosDynamicWalk.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["physics_engine"] = "physx"
    cfg["env"]["numEnvs"] = num_envs
    cfg["env"]["perturbation"] = perturbation
    cfg["sim"]["use_gpu_pipeline"] = False
    cfg["sim"]["physx"].update(num_threads=4, solver_type=1, use_gpu=False, num_subscenes=4)
    cfg["rl_device"] = "cpu"
    cfg["task"]["randomize"] = randomize
    return cfg


class RngRecorder:
    """Records torch.rand / torch.randint / torch.normal draws in call order while active."""

    def __init__(self):
        self.log = []
        self._orig = {}

    def __enter__(self):
        for name in ("rand", "randint", "normal", "randint_like"):
            self._orig[name] = getattr(torch, name)

            def make(n):
                def wrapped(*a, **k):
                    out = self._orig[n](*a, **k)
                    self.log.append((n, out.detach().clone()))
                    return out
                return wrapped
            setattr(torch, name, make(name))
        # `torch_rand_float` (python/isaacgym/torch_utils.py:50-52) is TorchScript: its draw never passes through the
        # python-level torch.rand above.  While recording, the task module calls an eager function with the same
        # arithmetic, (upper - lower) * rand(shape) + lower, so the spawn jitter of :732 shows up in the log.
        self._task = _loaded.get("task")
        if self._task is not None:
            self._jit_rand_float = self._task.torch_rand_float
            self._task.torch_rand_float = lambda lower, upper, shape, device: (upper - lower) * torch.rand(*shape, device=device) + lower
        return self

    def __exit__(self, *exc):
        for name, f in self._orig.items():
            setattr(torch, name, f)
        if self._task is not None:
            self._task.torch_rand_float = self._jit_rand_float


def make_reference_env(osim, num_envs: int, seed: int = 42, randomize: bool = True, perturbation: bool = True,
                       terrain: dict = None):
    """Construct the reference DyrosDynamicWalk on top of `osim` (an oracle.OracleSim with N envs).  `terrain`:
    values for the reference's TerrainCfg class attributes (the reference selects its terrain by editing that class,
    cfg/terrain/terrain_cfg.py:1-22); they are restored afterwards."""
    fake = FakeGym(osim)
    mods = load_reference(lambda: fake)
    cfg = reference_cfg(num_envs, randomize, perturbation)
    torch.manual_seed(seed)
    np.random.seed(seed)
    tcls = sys.modules["isaacgymenvs.cfg.terrain.terrain_cfg"].TerrainCfg
    saved = {k: getattr(tcls, k) for k in (terrain or {})}
    for k, v in (terrain or {}).items():
        setattr(tcls, k, v)
    cwd = os.getcwd()
    os.chdir(IGE)
    try:
        env = mods["task"].DyrosDynamicWalk(cfg, "cpu", 0, True)
        for k, v in (terrain or {}).items():       # the env reads its TerrainCfg instance later (:603): pin the values on it
            setattr(env.terrain_cfg, k, v)
    finally:
        os.chdir(cwd)
        for k, v in saved.items():
            setattr(tcls, k, v)
    return env, fake, mods
