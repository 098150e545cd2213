"""Replay machinery shared by oracle/make_goldens.py and the tests.  TEST INFRASTRUCTURE ONLY.

* sync_from_reference: copy the state of a reference DyrosDynamicWalk instance (running over the fake gym of
  ref_harness.py) into a DwBuffers-shaped set of numpy arrays.
* noise_from_log: turn the torch RNG draws one reference step() made into the injected-noise record
  (layout DW_NZ_* of include/dyros_walk.h).
* snapshot / FIELDS: the named outputs compared between reference, oracle and HIP library.
"""
from __future__ import annotations

import numpy as np
import torch

from isaacgymdyros_amd import abi

K = abi.K


def sync_from_reference(env, buf):
    """env: reference task instance; buf: dict of numpy arrays (oracle.alloc_buffers layout)."""
    N = env.num_envs
    es = buf["env_state"]
    es[:] = 0

    def put(name, t):
        v = abi.es_view(es, name)
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        v[...] = a.reshape(v.shape).astype(v.dtype)

    buf["root_states"][:] = env.root_states.numpy()
    buf["dof_state"][:] = env.dof_state.view(N, 33, 2).numpy()
    buf["contact_forces"][:] = env.contact_forces.numpy()
    buf["total_mass"][:] = env.total_mass[:, 0].numpy()
    buf["env_origins"][:] = env.env_origins.numpy()
    if getattr(env, "custom_origins", False):
        buf["terrain_levels"][:] = env.terrain_levels.numpy()
        buf["terrain_types"][:] = env.terrain_types.numpy()
        buf["terrain_origins"][...] = env.terrain_origins.numpy().astype(np.float32).reshape(buf["terrain_origins"].shape)
        buf["height_samples"][...] = np.asarray(env.terrain.heightsamples, dtype=np.int16).reshape(buf["height_samples"].shape)
    buf["friction_scale"][:] = 1.0
    buf["obs_buf"][:] = env.obs_buf.numpy()
    buf["rew_buf"][:] = env.rew_buf.numpy()
    buf["reset_buf"][:] = env.reset_buf.numpy()
    buf["progress_buf"][:] = env.progress_buf.numpy()
    buf["timeout_buf"][:] = env.timeout_buf.numpy()
    buf["randomize_buf"][:] = env.randomize_buf.numpy()
    for name in ("qpos_noise", "qvel_noise", "qpos_pre", "pre_joint_velocity_states", "target_data_qpos",
                 "target_data_force", "target_vel", "motor_constant_scale", "qpos_bias", "quat_bias", "action_log",
                 "actions", "actions_pre", "action_torque", "action_torque_pre", "time", "epi_len", "epi_len_log",
                 "contact_reward_sum", "contact_reward_mean", "magnitude", "phase", "init_mocap_data_idx",
                 "mocap_data_idx", "perturbation_count", "pert_duration", "pert_on", "impulse", "perturb_timing",
                 "perturb_start"):
        put(name, getattr(env, name))
    put("delay_idx", env.delay_idx_tensor[:, 1])
    put("simul_len", env.simul_len_tensor[:, 1])
    cfp = env.contact_forces_pre
    put("foot_force_pre", torch.stack([cfp[:, env.left_foot_idx], cfp[:, env.right_foot_idx]], 1))
    # histories: logical order, head = 0
    buf["obs_history"][:] = env.obs_history.view(N, 20, 37).numpy()
    buf["action_history"][:] = env.action_history.view(N, 20, 13).numpy()
    abi.es_view(es, "hist_head")[...] = 0
    buf["gate_acc"][:] = 0
    # slot (step-1)%3 for step 0 is slot 2; bucket = env % 32
    el = env.epi_len_log.numpy().astype(np.int64)
    cm = np.rint(env.contact_reward_mean.numpy().astype(np.float32) * np.float32(4294967296.0)).astype(np.int64)
    for e in range(N):
        buf["gate_acc"][(2 * K["DW_GATE_BUCKETS"] + e % K["DW_GATE_BUCKETS"]) * 2] += el[e]
        buf["gate_acc"][(2 * K["DW_GATE_BUCKETS"] + e % K["DW_GATE_BUCKETS"]) * 2 + 1] += cm[e]
    buf["gate_acc"][K["DW_GATE_LATCH"]] = int(bool(env.perturb_start[0, 0]))


def logical_history(hist, head):
    """[N,20,W] ring + [N] head -> logical order (slot 0 oldest)."""
    N = hist.shape[0]
    idx = (head.reshape(N, 1) + np.arange(20)[None, :]) % 20
    return np.take_along_axis(hist, idx[:, :, None], axis=1)


def noise_from_log(log, N, pert_ids, reset_ids, terrain_levels: int = 0, terrain_curriculum: bool = False):
    """log: list of (kind, tensor) in call order for ONE reference step().  terrain_levels > 0: the env runs on a
    height field (spawn jitter is drawn at reset); terrain_curriculum: the level draw of :689 precedes the other resets."""
    nz = np.zeros((N, K["DW_NOISE_WORDS"]), dtype=np.float32)
    it = iter(log)

    def nxt(kind):
        k, t = next(it)
        assert k == kind, (k, kind)
        return t.numpy()

    if pert_ids is not None:
        imp = nxt("randint").reshape(-1)
        dur = nxt("randint").reshape(-1)
        ph = nxt("rand").reshape(-1)
        nz[pert_ids, K["DW_NZ_PERT"] + 0] = (imp - 50 + 0.5) / 200.0
        nz[pert_ids, K["DW_NZ_PERT"] + 1] = (dur - 25 + 0.5) / 225.0
        nz[pert_ids, K["DW_NZ_PERT"] + 2] = ph
    for sub in range(2):
        nz[:, K["DW_NZ_ENC"] + 33 * sub: K["DW_NZ_ENC"] + 33 * (sub + 1)] = nxt("normal")
    if len(reset_ids) > 0:
        if terrain_curriculum:
            nz[reset_ids, K["DW_NZ_TERRAIN_LVL"]] = (nxt("randint_like").reshape(-1) + 0.5) / float(terrain_levels)
        nz[reset_ids, K["DW_NZ_QPOS_BIAS"]:K["DW_NZ_QPOS_BIAS"] + 12] = nxt("rand")
        nz[reset_ids, K["DW_NZ_QUAT_BIAS"]:K["DW_NZ_QUAT_BIAS"] + 3] = nxt("rand")
        nxt("rand")      # ft_bias (never read)
        nxt("rand")      # m_bias
        if terrain_levels > 0:
            nz[reset_ids, K["DW_NZ_ROOT_JITTER"]:K["DW_NZ_ROOT_JITTER"] + 2] = nxt("rand")    # torch_rand_float(-1, 1, (n, 2))
        nz[reset_ids, K["DW_NZ_TARGET_VEL"]] = nxt("rand").reshape(-1)
        nxt("rand")      # vel_theta * 0.0
        nz[reset_ids, K["DW_NZ_INIT_MOCAP"]] = nxt("rand").reshape(-1)
        nz[reset_ids, K["DW_NZ_MOTOR"]:K["DW_NZ_MOTOR"] + 12] = nxt("rand")
        nz[reset_ids, K["DW_NZ_DELAY"]] = (nxt("randint").reshape(-1) - 2 + 0.5) / 4.0
        nz[reset_ids, K["DW_NZ_PTIMING"]] = (nxt("randint").reshape(-1) + 0.5) / 2000.0
    nz[:, K["DW_NZ_VEL"]:K["DW_NZ_VEL"] + 6] = nxt("rand")
    rest = list(it)
    assert not rest, [k for k, _ in rest]
    return nz


# named outputs of one step: (source on the reference env, extractor on a DwBuffers-shaped dict)
def snapshot_reference(env, extras):
    N = env.num_envs
    cfp = env.contact_forces_pre
    d = dict(
        obs_buf=env.obs_buf.numpy().copy(), rew_buf=env.rew_buf.numpy().copy(),
        reset_buf=env.reset_buf.numpy().copy(), progress_buf=env.progress_buf.numpy().copy(),
        timeout_buf=env.timeout_buf.numpy().copy(),
        stacked_rewards=extras["stacked_rewards"].numpy().astype(np.float32).copy(),
        root_states=env.root_states.numpy().copy(), dof_state=env.dof_state.view(N, 33, 2).numpy().copy(),
        contact_forces=env.contact_forces.numpy().copy(),
        obs_history=env.obs_history.view(N, 20, 37).numpy().copy(),
        action_history=env.action_history.view(N, 20, 13).numpy().copy(),
        foot_force_pre=torch.stack([cfp[:, env.left_foot_idx], cfp[:, env.right_foot_idx]], 1).numpy().copy(),
        delay_idx=env.delay_idx_tensor[:, 1].numpy().astype(np.int32).copy(),
        simul_len=env.simul_len_tensor[:, 1].numpy().astype(np.int32).copy(),
    )
    for name in STATE_FIELDS:
        t = getattr(env, name)
        a = t.numpy().copy()
        off, shape, kind = abi.ES_FIELDS[name]
        d[name] = a.reshape((N,) + tuple(shape)).astype(np.int32 if kind == "i" else np.float32)
    if getattr(env, "custom_origins", False):
        d["terrain_levels"] = env.terrain_levels.numpy().astype(np.int64).copy()
        d["env_origins"] = env.env_origins.numpy().astype(np.float32).copy()
    return d


STATE_FIELDS = ["qpos_noise", "qvel_noise", "qpos_pre", "pre_joint_velocity_states", "target_data_qpos",
                "target_data_force", "target_vel", "motor_constant_scale", "qpos_bias", "quat_bias", "action_log",
                "actions", "actions_pre", "action_torque", "action_torque_pre", "time", "epi_len", "epi_len_log",
                "contact_reward_sum", "contact_reward_mean", "magnitude", "phase", "init_mocap_data_idx",
                "mocap_data_idx", "perturbation_count", "pert_duration", "pert_on", "impulse", "perturb_timing",
                "perturb_start"]


def snapshot_buffers(buf):
    """Same keys as snapshot_reference, from a DwBuffers-shaped dict of numpy arrays."""
    es = buf["env_state"]
    head = abi.es_view(es, "hist_head").copy()
    d = dict(
        obs_buf=buf["obs_buf"].copy(), rew_buf=buf["rew_buf"].copy(), reset_buf=buf["reset_buf"].copy(),
        progress_buf=buf["progress_buf"].copy(), timeout_buf=buf["timeout_buf"].copy(),
        stacked_rewards=buf["stacked_rewards"].copy(), root_states=buf["root_states"].copy(),
        dof_state=buf["dof_state"].copy(), contact_forces=buf["contact_forces"].copy(),
        obs_history=logical_history(buf["obs_history"], head),
        action_history=logical_history(buf["action_history"], head),
        foot_force_pre=abi.es_view(es, "foot_force_pre").copy(),
        delay_idx=abi.es_view(es, "delay_idx").copy(), simul_len=abi.es_view(es, "simul_len").copy(),
    )
    for name in STATE_FIELDS:
        d[name] = abi.es_view(es, name).copy()
    d["terrain_levels"] = np.array(buf["terrain_levels"], dtype=np.int64, copy=True)
    d["env_origins"] = np.array(buf["env_origins"], dtype=np.float32, copy=True)
    return d


def ulp_diff(a, b):
    """Distance in units of last place between two float32 arrays (NaN == NaN counts as 0)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def compare(ref, got, exact=(), ulp=None, atol=None):
    """Returns list of (field, worst) mismatches.  exact: fields compared bitwise (ints by value; -0.0 != +0.0
    for floats).  ulp: dict field -> max ulp.  atol: dict field -> (max abs, max rel)."""
    bad = []
    for k in exact:
        a, b = ref[k], got[k]
        if a.dtype.kind == "f":
            same = np.array_equal(np.ascontiguousarray(a, np.float32).view(np.int32),
                                  np.ascontiguousarray(b, np.float32).view(np.int32)) or \
                (ulp_diff(a, b).max() == 0 and np.array_equal(np.signbit(a), np.signbit(b)))
        else:
            same = np.array_equal(a.astype(np.int64), b.astype(np.int64))
        if not same:
            bad.append((k, "bitwise"))
    for k, u in (ulp or {}).items():
        d = ulp_diff(ref[k], got[k]).max()
        if d > u:
            bad.append((k, "ulp %d > %d" % (d, u)))
    for k, (ab, rl) in (atol or {}).items():
        a, b = ref[k].astype(np.float64), got[k].astype(np.float64)
        err = np.abs(a - b)
        lim = ab + rl * np.abs(a)
        if not np.all((err <= lim) | (np.isnan(a) & np.isnan(b))):
            bad.append((k, "abs %.3g" % err.max()))
    return bad
