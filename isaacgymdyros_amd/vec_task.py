"""Env / VecTask: the base-class contract of the reference, without the Gym engine behind it.

Mirrors tasks/base/vec_task.py of the reference (paths relative to python/IsaacGymEnvs/isaacgymenvs):
  Env.__init__                 :51-98    device parsing, sizes, spaces, clip limits
  properties                   :129-152  observation_space, action_space, num_envs, num_acts, num_obs
  VecTask.allocate_buffers     :233-256
  VecTask.step                 :293-344
  VecTask.zero_actions         :346-354
  VecTask.reset / reset_done   :362-391
What is gone: create_sim / set_viewer / render (the engine and the viewer), __parse_sim_params (a gymapi struct),
and the python-loop domain randomisation (:477-733), whose actor-parameter part lives in the step kernel and in
DyrosDynamicWalk._apply_setup_randomization.
"""
from __future__ import annotations

import abc
from typing import Any, Dict, Tuple

import numpy as np
import torch


class Box:
    """Stand-in for gym.spaces.Box (the `gym` package is not a dependency): low, high, shape, dtype."""

    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low.min(), self.high.max(), self.shape)


class Env(abc.ABC):
    def __init__(self, config: Dict[str, Any], sim_device: str, graphics_device_id: int, headless: bool):
        split_device = sim_device.split(":")
        self.device_type = split_device[0]
        self.device_id = int(split_device[1]) if len(split_device) > 1 else 0
        if self.device_type.lower() not in ("cuda", "gpu"):
            # reference: vec_task.py:61-71 falls back to the CPU pipeline of PhysX; this framework has no CPU path
            raise ValueError("sim_device must be a GPU ('cuda:N'): the MI355X step has no CPU pipeline "
                             "(the CPU restatement under oracle/ is test infrastructure, not a backend)")
        self.device = "cuda:" + str(self.device_id)
        self.rl_device = config.get("rl_device", self.device)
        self.headless = headless
        self.graphics_device_id = -1

        self.num_environments = config["env"]["numEnvs"]
        self.num_agents = config["env"].get("numAgents", 1)
        self.num_observations = config["env"]["numObservations"]
        self.num_states = config["env"].get("numStates", 0)
        self.num_actions = config["env"]["numActions"]
        self.control_freq_inv = config["env"].get("controlFrequencyInv", 1)

        self.obs_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.act_space = Box(np.ones(self.num_actions) * -1., np.ones(self.num_actions) * 1.)
        self.clip_obs = config["env"].get("clipObservations", np.inf)
        self.clip_actions = config["env"].get("clipActions", np.inf)

    @abc.abstractmethod
    def step(self, actions: torch.Tensor) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor, Dict[str, Any]]:
        ...

    @abc.abstractmethod
    def reset(self) -> Dict[str, torch.Tensor]:
        ...

    @abc.abstractmethod
    def reset_idx(self, env_ids: torch.Tensor):
        ...

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self) -> int:
        return self.num_environments

    @property
    def num_acts(self) -> int:
        return self.num_actions

    @property
    def num_obs(self) -> int:
        return self.num_observations


class VecTask(Env):
    def __init__(self, config, sim_device, graphics_device_id, headless):
        super().__init__(config, sim_device, graphics_device_id, headless)
        if config.get("physics_engine", "physx") not in ("physx",):
            raise ValueError(f"Invalid physics engine backend: {config.get('physics_engine')}")   # vec_task.py:173-175
        if config["sim"]["up_axis"] not in ["z"]:
            raise ValueError(f"Invalid physics up-axis: {config['sim']['up_axis']}")              # vec_task.py:435-438
        self.viewer = None
        self.alias_obs = bool(config["sim"].get("mi355", {}).get("alias_obs", False))
        self.obs_dict: Dict[str, torch.Tensor] = {}
        self.extras: Dict[str, Any] = {}

    def _clip_obs(self, t: torch.Tensor) -> torch.Tensor:
        # reference: torch.clamp(obs_buf, -clip_obs, clip_obs) (vec_task.py:338) -- always a FRESH tensor, also with the
        # task's clip of +-inf.  That contract is the default here.  cfg["sim"]["mi355"]["alias_obs"] = True opts into
        # returning the persistent buffer itself (saves a 32 MB copy per step at 16384 envs; the caller must then copy
        # what it keeps across steps -- rl_games does, a2c_common_dyros.py:642-661; bench.py sets it).
        if np.isinf(self.clip_obs):
            return t if self.alias_obs else t.clone()
        return torch.clamp(t, -self.clip_obs, self.clip_obs)

    def zero_actions(self) -> torch.Tensor:
        return torch.zeros([self.num_envs, self.num_actions], dtype=torch.float32, device=self.rl_device)

    def reset(self):
        """Called once when the environment starts; does not compute observations (vec_task.py:362-374)."""
        self.obs_dict["obs"] = self._clip_obs(self.obs_buf).to(self.rl_device)
        return self.obs_dict

    def get_observations(self):
        """Not in the reference; same dictionary as reset()."""
        return self.reset()

    def reset_done(self):
        done_env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(done_env_ids) > 0:
            self.reset_idx(done_env_ids)
        self.obs_dict["obs"] = self._clip_obs(self.obs_buf).to(self.rl_device)
        return self.obs_dict, done_env_ids

    def render(self):
        pass
