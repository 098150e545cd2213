"""Replay of the committed golden fixtures through a backend (CPU oracle or HIP library).  Test helper."""
import os

import numpy as np

from isaacgymdyros_amd import abi
from oracle import parity as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fields that involve no transcendental and no physics: must match the reference bit for bit
EXACT_LOGIC = ["reset_buf", "progress_buf", "timeout_buf", "delay_idx", "simul_len", "init_mocap_data_idx",
               "mocap_data_idx", "perturbation_count", "pert_on", "perturb_timing", "action_torque",
               "target_data_qpos", "target_data_force", "time", "epi_len", "epi_len_log", "motor_constant_scale",
               "qpos_bias", "quat_bias", "target_vel", "contact_reward_sum", "contact_reward_mean"]
# exp / sin / cos / asin / atan2 sit between these and their inputs: glibc (oracle), SLEEF (torch CPU, the
# goldens) and OCML (GPU) each round the last bit their own way
TRANSCENDENTAL = {"rew_buf": (1e-6, 2e-6), "stacked_rewards": (1e-6, 2e-6), "obs_buf": (2e-6, 4e-6),
                  "magnitude": (0.0, 0.0), "phase": (0.0, 0.0)}


class GoldenTerrain:
    """The height field of a terrain fixture, with the attributes OracleSim(terrain=...) reads."""

    def __init__(self, g):
        import types
        rows, cols = int(g["cfg_terrain_rows"]), int(g["cfg_terrain_cols"])
        lv, ty = int(g["cfg_terrain_num_levels"]), int(g["cfg_terrain_num_types"])
        self.heightsamples = g["init_height_samples"].reshape(rows, cols)
        self.env_origins = g["init_terrain_origins"].reshape(lv, ty, 3)
        self.tot_rows, self.tot_cols = rows, cols
        self.env_length = float(g["cfg_terrain_env_length"])
        self.cfg = types.SimpleNamespace(horizontal_scale=float(g["cfg_terrain_hscale"]), vertical_scale=float(g["cfg_terrain_vscale"]),
                                         border_size=float(g["cfg_terrain_border"]), curriculum=bool(g["cfg_terrain_curriculum"]),
                                         num_rows=lv, num_cols=ty)


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


class OracleBackend:
    def __init__(self, N, task_const, **cfg):
        from oracle.oracle import OracleSim
        self.sim = OracleSim(N, task_const=task_const, **cfg)

    def load_buffers(self, bufs):
        for k, v in bufs.items():
            self.sim.buf[k][...] = v

    def write_state(self, root, dof, cf):
        self.sim.buf["root_states"][...] = root
        self.sim.buf["dof_state"][...] = dof
        self.sim.buf["contact_forces"][...] = cf

    def step(self, a, nz, t):
        self.sim.step(a, nz, t)

    def read_buffers(self):
        return self.sim.buf


def replay(golden, backend, on_step=None):
    """Feeds the golden's inputs to `backend`, yields (t, reference_snapshot, backend_snapshot)."""
    N, steps = int(golden["N"]), int(golden["steps"])
    backend.load_buffers({k[5:]: v for k, v in golden.items() if k.startswith("init_")})
    frozen = "inj_root" in golden
    for t in range(steps):
        if frozen:
            backend.write_state(golden["inj_root"][t], golden["inj_dof"][t], golden["inj_cf"][t])
        if t == int(golden["force_perturb_step"]):
            bufs = backend.read_buffers()
            es = np.array(bufs["env_state"], copy=True)
            abi.es_view(es, "perturb_start")[...] = 1
            ga = np.array(bufs["gate_acc"], copy=True)
            ga[abi.K["DW_GATE_LATCH"]] = 1
            backend.load_buffers({"env_state": es, "gate_acc": ga})
        backend.step(golden["actions"][t], golden["noise"][t], t)
        got = P.snapshot_buffers(backend.read_buffers())
        ref = {k[5:]: v[t] for k, v in golden.items() if k.startswith("step_")}
        if t == steps - 1:
            ref.update({k[6:]: v for k, v in golden.items() if k.startswith("final_")})
        yield t, ref, got
