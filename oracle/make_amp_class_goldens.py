#!/usr/bin/env python3
"""Mint tests/golden/amp_class_ref.npz: the reference's `TocabiAMPLowerBase` CLASS stepping over a fake gym.  TEST
INFRASTRUCTURE ONLY; runs only where /root/reference is mounted.

The class (tasks/amp/tocabi_amp_lower_base.py) is imported from where it lies and driven the way the AMP learner drives it --
`reset_done()` then `step(actions)` -- over a fake `gym` whose `simulate` is the CPU oracle's physics substep (the closed engine
cannot run).  What is committed is DATA:
  * inputs: the actions, every torch RNG draw the class made (in call order, with kind and shape), and the physics state the
    fake gym held after every `simulate` (root state, dof state, net contact forces, the three rigid-body rows the class reads);
  * the reference's outputs after every step: obs_buf, rew_buf, reset_buf, _terminate_buf, progress_buf, commands, the two
    history tensors, qpos_noise / qvel_noise, the delayed-torque FIFO and its counters, the torque tensor handed to the engine.
tests/test_amp_gpu.py replays the draws and the physics states into `isaacgymdyros_amd.tocabi_amp_lower.TocabiAMPLower`
(its generator and its `simulate` are injectable for exactly this) and compares every output: the class logic -- history
stacking, command ramp, torque FIFO with delay, encoder model, reset order -- is then pinned, not only the three pure functions.
Configuration: the reference's TocabiAMPLower.yaml with `randomize: False` (the dof-property randomisation is VecTask code
pinned by dr_reset.npz; its numpy draws are not torch draws), 16 envs x 60 steps, episode length 31.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from isaacgymdyros_amd.task_constants import ACTION_HIGH          # noqa: E402
from oracle import ref_harness as RH                                 # noqa: E402
from oracle.oracle import OracleSim                                  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "amp_class_ref.npz")
N, STEPS = 16, 60


class AmpFakeGym(RH.FakeGym):
    """FakeGym plus the handful of calls only the AMP task makes (rigid-body state tensor, actuator limits, dof names)."""

    def __init__(self, sim):
        super().__init__(sim)
        self.rb = torch.zeros(self.N * 38, 13)
        self.rb[:, 6] = 1.0
        self.after_sim = []          # physics state after every simulate(): what the replay injects

    def get_asset_actuator_properties(self, asset):
        return [RH._Bag(upper_control_limit=float(v)) for v in ACTION_HIGH]
    def get_asset_dof_name(self, asset, i): return self.model.dof_names[i]
    def find_actor_rigid_body_handle(self, env, handle, name): return self.model.body_names.index(name)
    def acquire_rigid_body_state_tensor(self, sim): return self.rb
    def acquire_force_sensor_tensor(self, sim): return torch.zeros(self.N, 12)
    def acquire_dof_force_tensor(self, sim): return torch.zeros(self.N, 33)
    def create_asset_force_sensor(self, *a): return 0
    def enable_actor_dof_force_sensors(self, *a): return True
    def refresh_force_sensor_tensor(self, sim): return True
    def refresh_dof_force_tensor(self, sim): return True

    def _update_rb(self):
        import ctypes as C
        out = np.zeros((self.N, 2, 3), np.float32)
        arr = (C.c_int32 * 2)(6, 12)
        assert self.osim.api["body_positions"](self.osim.h, arr, 2, out.ctypes.data_as(C.c_void_p), None) == 0
        rb = self.rb.view(self.N, 38, 13)
        rb[:, 0, :] = self.root
        rb[:, 8, 0:3] = torch.from_numpy(out[:, 0])
        rb[:, 16, 0:3] = torch.from_numpy(out[:, 1])

    def refresh_rigid_body_state_tensor(self, sim):
        self._update_rb()
        return True

    def simulate(self, sim):
        super().simulate(sim)
        self._update_rb()
        rb = self.rb.view(self.N, 38, 13)
        self.after_sim.append(dict(root=self.root.clone().numpy(), dof=self.dof.clone().numpy().reshape(self.N, 33, 2),
                                   contact=self.contact.clone().numpy().reshape(self.N, 38, 3),
                                   feet=rb[:, [8, 16], 0:3].clone().numpy(), tau=self.tau.copy()))

    def set_actor_root_state_tensor_indexed(self, sim, t, ids, n):
        # the reference passes its `_initial_root_states`: the engine copies those rows into the live tensor
        idx = ids.long()
        self.root[idx] = t[idx]
        return True

    def set_dof_state_tensor_indexed(self, sim, t, ids, n):
        return True                     # (`t` IS the live dof tensor, already written by the task)


def load_amp_module():
    RH.load_reference(lambda: None)
    RH.sys.modules.setdefault("isaacgymenvs.tasks.amp", RH.types.ModuleType("isaacgymenvs.tasks.amp")).__path__ = [os.path.join(RH.IGE, "tasks", "amp")]
    return RH._load("isaacgymenvs.tasks.amp.tocabi_amp_lower_base", os.path.join(RH.IGE, "tasks", "amp", "tocabi_amp_lower_base.py"))


def amp_cfg(num_envs):
    with open(os.path.join(RH.IGE, "cfg", "task", "TocabiAMPLower.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["physics_engine"] = "physx"
    cfg["env"]["numEnvs"] = num_envs
    cfg["env"]["stateInit"] = "Default"
    cfg["env"]["episodeLength"] = 31                       # so that falls, time-outs (the robots that are still up at step 30) and the command ramp all occur inside 60 steps
    cfg["sim"]["use_gpu_pipeline"] = False
    cfg["sim"]["physx"].update(num_threads=4, solver_type=1, use_gpu=False, num_subscenes=4)
    cfg["rl_device"] = "cpu"
    cfg["task"]["randomize"] = False
    return cfg


def main():
    osim = OracleSim(N)
    fake = AmpFakeGym(osim)
    RH.load_reference(lambda: fake)
    amp = load_amp_module()
    RH._loaded["gymapi"].acquire_gym = lambda: fake
    cfg = amp_cfg(N)
    torch.manual_seed(7)
    np.random.seed(7)
    cwd = os.getcwd()
    os.chdir(RH.IGE)
    try:
        env = amp.TocabiAMPLowerBase(cfg, "cpu", 0, True)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(11)
    rec = RH.RngRecorder()
    # `torch_rand_float` (python/isaacgym/torch_utils.py:50-52) is TorchScript: its draw never passes through the python-level
    # torch.rand the recorder wraps.  While recording, the AMP module calls an eager function with the same arithmetic.
    jit_rand_float = amp.torch_rand_float
    amp.torch_rand_float = lambda lower, upper, shape, device: (upper - lower) * torch.rand(*shape, device=device) + lower
    d = {"actions": [], "draw_kind": [], "draw_shape": [], "draw_data": []}
    keys = ["obs_buf", "rew_buf", "reset_buf", "_terminate_buf", "progress_buf", "commands", "obs_history", "action_history", "qpos_noise",
            "qvel_noise", "qpos_pre", "action_log", "epi_len", "qpos_bias", "quat_bias", "_dof_vel_pre", "actions_pre", "timeout_buf"]
    per = {k: [] for k in keys}
    per.update(delay_idx=[], simul_len=[], reset_ids=[])
    with rec:
        for t in range(STEPS):
            _, ids = env.reset_done()
            per["reset_ids"].append(np.pad(ids.numpy(), (0, N - len(ids)), constant_values=-1))
            a = (rng.uniform(-1, 1, size=(N, 12)) * (0.2 if t < 10 else 1.0)).astype(np.float32)
            d["actions"].append(a)
            env.step(torch.from_numpy(a))
            for k in keys:
                v = getattr(env, k)
                per[k].append(v.clone().numpy() if torch.is_tensor(v) else np.asarray(v))
            per["delay_idx"].append(env.delay_idx_tensor[:, 1].clone().numpy())
            per["simul_len"].append(env.simul_len_tensor[:, 1].clone().numpy())
    amp.torch_rand_float = jit_rand_float
    flat = []
    for kind, tns in rec.log:
        d["draw_kind"].append(kind)
        d["draw_shape"].append(np.array(list(tns.shape) + [0] * (3 - tns.dim()), np.int64))
        flat.append(tns.float().numpy().ravel())          # (float32 holds every draw exactly: uniform / normal floats, integers below 2^24)
    out = {"actions": np.stack(d["actions"]), "draw_kind": np.array(d["draw_kind"]), "draw_shape": np.stack(d["draw_shape"]),
           "draw_ndim": np.array([tns.dim() for _, tns in rec.log], np.int64),
           "draw_offset": np.cumsum([0] + [len(x) for x in flat]).astype(np.int64), "draw_data": np.concatenate(flat),
           "num_envs": np.array(N), "steps": np.array(STEPS), "episode_length": np.array(cfg["env"]["episodeLength"])}
    for k, v in per.items():
        out["ref_" + k] = np.stack(v)
    for name in ("root", "dof", "contact", "feet", "tau"):
        out["sim_" + name] = np.stack([s[name] for s in fake.after_sim])
    out["motor_efforts"] = env.motor_efforts.numpy()
    out["p_gains"], out["d_gains"] = env.p_gains.numpy(), env.d_gains.numpy()
    out["total_mass"] = env.total_mass.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "draws", len(rec.log), "simulates", len(fake.after_sim), "resets per step", [int((r >= 0).sum()) for r in per["reset_ids"]][:20],
          "terminated total", int(np.stack(per["_terminate_buf"]).sum()))


if __name__ == "__main__":
    main()
