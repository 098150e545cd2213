"""Task configuration as the nested dict the reference's task constructor takes (cfg["env"], cfg["sim"],
cfg["task"]).  Values restate cfg/task/DyrosDynamicWalk.yaml with the Hydra interpolations resolved to the
defaults of cfg/config.yaml (Hydra/OmegaConf are not needed: a plain dict is the interface)."""
from __future__ import annotations

import copy


def default_cfg(num_envs: int = 4096, sim_device: str = "cuda:0") -> dict:
    return {
        "name": "DyrosDynamicWalk",
        "physics_engine": "physx",                       # yaml:3
        "rl_device": sim_device,                         # default to the sim device
        "seed": 42,                                      # cfg/config.yaml:11
        "env": {
            "numEnvs": num_envs,                         # yaml:7
            "envSpacing": 5,                             # yaml:8
            "episodeLength": 32,                         # yaml:9 (seconds)
            "enableDebugVis": False,
            "controlFrequencyInv": 2,                    # yaml:11
            "clipActions": 1.0,                          # yaml:13
            "NumSingleStepObs": 37, "NumAction": 13,     # yaml:15-16
            "perturbation": True,                        # yaml:18
            "NumHis": 10, "NumSkip": 2,                  # yaml:19-20
            "initialHieght": 0.93,                       # yaml:22 (sic)
            "deathCost": 0.0, "terminationHeight": 0.6,  # yaml:26-27
            "enableCameraSensors": False,
        },
        "sim": {
            "dt": 0.002, "substeps": 1, "up_axis": "z",  # yaml:38-40
            "use_gpu_pipeline": True,
            "gravity": [0.0, 0.0, -9.81],                # yaml:42
            "physx": {
                "num_position_iterations": 4, "num_velocity_iterations": 1,     # yaml:45-48
                "contact_offset": 0.002, "rest_offset": 0.0,                    # yaml:49-50
                "bounce_threshold_velocity": 0.04, "max_depenetration_velocity": 10.0,   # yaml:51-52
            },
            # knobs of THIS simulator's contact model (no PhysX counterpart; DESIGN.md "Physics model")
            "mi355": {"erp": 0.2, "contact_cfm": 1e-3, "penalty_stiffness": 1.0e5, "penalty_damping": 1.0e3,
                      "plane_friction": 1.0,             # cfg/terrain/terrain_cfg.py:7-8
                      "self_collision": True, "root_vel_at_com": True, "torch_gpu_div": True, "timeout_fix": False,
                      "force_perturb_start": False},
        },
        "task": {
            "randomize": True,                           # yaml:59
            "randomization_params": {
                "frequency": 1,                          # yaml:62
                "actor_params": {"humanoid": {
                    "rigid_body_properties": {"mass": {"range": [0.8, 1.2], "operation": "scaling",
                                                       "distribution": "uniform", "setup_only": True}},   # yaml:82-88
                    "dof_properties": {
                        "damping": {"range": [0.0, 2.9], "operation": "additive", "distribution": "uniform"},   # yaml:103-108
                        "armature": {"range": [0.8, 1.2], "operation": "scaling", "distribution": "uniform"},   # yaml:109-115
                    },
                    # present but commented out in the reference (yaml:89-95); BASELINE config 5 enables it
                    # "rigid_shape_properties": {"friction": {"range": [0.7, 1.3], "operation": "scaling"}},
                }},
            },
        },
    }


def with_friction_randomization(cfg: dict) -> dict:
    cfg = copy.deepcopy(cfg)
    cfg["task"]["randomization_params"]["actor_params"]["humanoid"]["rigid_shape_properties"] = {
        "friction": {"range": [0.7, 1.3], "operation": "scaling", "distribution": "uniform"}}
    return cfg


def with_terrain(cfg: dict, **terrain) -> dict:
    """Height-field terrain.  The reference selects it by editing its `TerrainCfg` class
    (cfg/terrain/terrain_cfg.py:1-22); here the same fields travel in cfg["terrain"], e.g.
    `with_terrain(cfg, mesh_type="heightfield", curriculum=True)`."""
    cfg = copy.deepcopy(cfg)
    cfg["terrain"] = dict(cfg.get("terrain") or {}, **terrain)
    return cfg


def validate_cfg(cfg: dict) -> None:
    """Pure-Python range checks of everything the constructor later turns into device indices or kernel sizes.  Called
    BEFORE the native library is loaded or a byte of device memory is allocated, so a bad configuration is a ValueError
    on the host, never an out-of-range gather on the GPU (a device-side assert aborts the process; round-1 post-mortem).
    The reference has no such checks: `terrain_origins[levels, types]` (tasks/dyros_dynamic_walk.py:703-707) indexes out
    of range for max_init_terrain_level >= num_rows."""
    env, sim = cfg["env"], cfg["sim"]
    n = env["numEnvs"]
    if not isinstance(n, int) or n < 1:
        raise ValueError("env.numEnvs must be a positive integer, got %r" % (n,))
    if n > (1 << 20):
        raise ValueError("env.numEnvs = %d: at most 2^20 envs per GPU (the kernels address the per-env streams with 32-bit byte offsets; shard over "
                         "more GPUs, isaacgymdyros_amd.dist)" % n)
    if (env.get("NumHis"), env.get("NumSkip"), env.get("NumSingleStepObs"), env.get("NumAction")) != (10, 2, 37, 13):
        raise ValueError("the MI355X step kernel is specialised for NumHis=10, NumSkip=2, NumSingleStepObs=37, NumAction=13")
    if env.get("controlFrequencyInv", 2) != 2:
        raise ValueError("only env.controlFrequencyInv = 2 is supported (DyrosDynamicWalk.yaml:11)")
    if not float(sim["dt"]) > 0:
        raise ValueError("sim.dt must be positive")
    if len(sim.get("gravity", [0, 0, -9.81])) != 3:
        raise ValueError("sim.gravity must have three components")
    px = sim.get("physx", {})
    if int(sim.get("mi355", {}).get("debug_wave_build", 0)) not in (0, 1, 2, 3):
        raise ValueError("sim.mi355.debug_wave_build must be 0 (by launch size), 1 / 2 (octet kernels: one / two waves per SIMD) or 3 (hex instantiation)")
    if sim.get("mi355", {}).get("pipeline", 0) not in (0, 3, "auto", "oct"):
        raise ValueError("sim.mi355.pipeline must be 0/'auto' or 3/'oct' (1/'fused', 2/'quad' and 4/'lane', the kernels of rounds 1, 2 and 4, are retired)")
    iters = int(px.get("num_position_iterations", 4)) + int(px.get("num_velocity_iterations", 1))
    if not 1 <= iters <= 64:
        raise ValueError("physx.num_position_iterations + num_velocity_iterations must be in 1..64")
    t = dict(cfg.get("terrain") or {})
    mesh = t.get("mesh_type", "plane")
    if mesh not in (None, "none", "plane", "heightfield", "trimesh"):
        raise ValueError("Terrain mesh type not recognised. Allowed types are [None, plane, heightfield, trimesh]")
    if mesh in ("heightfield", "trimesh"):
        from .terrain import TerrainCfg
        tc = TerrainCfg(**t)                           # AttributeError for unknown fields
        if tc.selected:
            raise ValueError("TerrainCfg.selected is not supported (it cannot run in the reference either, DESIGN.md section 9)")
        if not (isinstance(tc.num_rows, int) and isinstance(tc.num_cols, int) and tc.num_rows >= 1 and tc.num_cols >= 1):
            raise ValueError("TerrainCfg.num_rows and num_cols must be positive integers")
        if tc.curriculum and not 0 <= int(tc.max_init_terrain_level) < tc.num_rows:
            raise ValueError("TerrainCfg.max_init_terrain_level (%d) must be in [0, num_rows = %d)"
                             % (tc.max_init_terrain_level, tc.num_rows))
        if not (tc.horizontal_scale > 0 and tc.vertical_scale > 0):
            raise ValueError("TerrainCfg.horizontal_scale and vertical_scale must be positive")
        if tc.border_size < 0:
            raise ValueError("TerrainCfg.border_size must be non-negative")
        if tc.terrain_length < 2 * tc.horizontal_scale or tc.terrain_width < 2 * tc.horizontal_scale:
            raise ValueError("TerrainCfg.terrain_length / terrain_width must span at least two samples")
        if abs(tc.terrain_length - tc.terrain_width) > 1e-9:
            raise ValueError("TerrainCfg tiles must be square (the reference builds every tile width x width, utils/terrain.py:108)")
        props = list(tc.terrain_proportions)
        if not props or any(p < 0 for p in props) or sum(props) > 1.0 + 1e-6:
            raise ValueError("TerrainCfg.terrain_proportions must be non-negative and sum to at most 1")
        rows = int(tc.num_rows * int(tc.terrain_length / tc.horizontal_scale)) + 2 * int(tc.border_size / tc.horizontal_scale)
        cols = int(tc.num_cols * int(tc.terrain_width / tc.horizontal_scale)) + 2 * int(tc.border_size / tc.horizontal_scale)
        if rows < 2 or cols < 2 or rows * cols > 2 ** 31 - 1:
            raise ValueError("terrain sample grid %d x %d is out of range" % (rows, cols))


# ---------------------------------------------------------------------------------------------- reference YAML
# The reference hands its task constructor `omegaconf_to_dict(cfg.task)` (train.py:104-110): the task YAML with Hydra /
# OmegaConf interpolations resolved against the root config (cfg/config.yaml) by four custom resolvers
# (train.py:58-63).  Hydra and OmegaConf are not needed for that: the loader below resolves the same expressions.
ROOT_DEFAULTS = {            # cfg/config.yaml:6-32
    "num_envs": "", "seed": 42, "physics_engine": "physx", "pipeline": "gpu", "sim_device": "cuda:0",
    "rl_device": "cuda:0", "graphics_device_id": 0, "num_threads": 4, "solver_type": 1, "num_subscenes": 4,
    "headless": True,
}


def _split_args(s: str):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
            continue
        depth += ch == "{"
        depth -= ch == "}"
        cur += ch
    out.append(cur)
    return [a.strip() for a in out]


def _resolve(value, root: dict):
    """One scalar of the task YAML: `${..key}` (any number of dots: a key of the root config), `${eq:a,b}`,
    `${contains:a,b}`, `${if:c,a,b}`, `${resolve_default:d,v}`, nested."""
    if not isinstance(value, str) or "${" not in value:
        return value
    s = value.strip()
    if not (s.startswith("${") and s.endswith("}")):
        raise ValueError("unsupported interpolation inside a string: %r" % value)
    body = s[2:-1]
    name, sep, rest = body.partition(":")
    if not sep or name.startswith("."):
        key = body.lstrip(".")
        if key not in root:
            raise KeyError("interpolation %r: the root config has no key %r" % (value, key))
        return root[key]
    args = [_resolve(a.strip('"').strip("'") if "${" not in a else a, root) for a in _split_args(rest)]
    if name == "eq":
        return str(args[0]).lower() == str(args[1]).lower()
    if name == "contains":
        return str(args[0]).lower() in str(args[1]).lower()
    if name == "if":
        return args[1] if args[0] else args[2]
    if name == "resolve_default":
        d = args[0]
        if args[1] != "":
            return args[1]
        try:
            return int(d)
        except (TypeError, ValueError):
            return d
    raise ValueError("unknown resolver %r in %r" % (name, value))


def load_task_yaml(path_or_text: str, **root_overrides) -> dict:
    """The constructor argument of `DyrosDynamicWalk` from the reference's own task YAML
    (cfg/task/DyrosDynamicWalk.yaml), with its `${...}` expressions resolved against cfg/config.yaml's defaults and
    `root_overrides` (e.g. `num_envs=16384, sim_device="cuda:1"`), as Hydra would on the reference's command line."""
    import os
    import yaml
    text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
    root = dict(ROOT_DEFAULTS, **root_overrides)

    def walk(x):
        if isinstance(x, dict):
            return {k: walk(v) for k, v in x.items()}
        if isinstance(x, list):
            return [walk(v) for v in x]
        return _resolve(x, root)
    cfg = walk(yaml.safe_load(text))
    cfg.setdefault("rl_device", root["rl_device"])
    cfg.setdefault("seed", root["seed"])
    return cfg
