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
