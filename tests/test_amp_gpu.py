"""Row f-3 on the GPU: the HIP entry points dw_amp_* / dw_newwalk_reward / dw_body_positions against the CPU oracle and the
reference fixture, and the TocabiAMPLower env class stepping on the MI355X physics."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import amp_calls

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "amp_lower_ref.npz")
_NP2T = {"f4": torch.float32, "i8": torch.int64, "i4": torch.int32}


def ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def hip_outputs(g, early=True):
    from isaacgymdyros_amd import _lib
    lib, api = _lib.load()

    def chk(rc):
        assert rc == 0, lib.dw_last_error()
    out = amp_calls.run_all(api, g, lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda(),
                            lambda s, d: torch.zeros(s, dtype=_NP2T[d], device="cuda"), chk, early)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def oracle_outputs(g, early=True):
    from oracle import oracle
    lib, api = oracle.load()

    def chk(rc):
        assert rc == 0
    return amp_calls.run_all(api, g, lambda a: np.ascontiguousarray(a), lambda s, d: np.zeros(s, d), chk, early)


def test_amp_functions_hip_vs_oracle_and_reference():
    g = np.load(G)
    h, o = hip_outputs(g), oracle_outputs(g)
    # observations: 33 of the 36 entries are the reference's bit patterns; the three Euler angles go through atan2f, where the
    # device library and torch's SLEEF kernel round differently (measured: <= 2.4e-7 rad, 1 ulp of the largest angles; tools/amp_diag.py)
    d = ulps(h["obs"], g["ref_obs"])
    assert d[:, 3:].max() == 0, np.argwhere(d[:, 3:] > 0)[:5]
    assert np.abs(h["obs"][:, :3] - g["ref_obs"][:, :3]).max() <= 2.4e-7
    assert np.abs(h["obs"] - o["obs"]).max() <= 2.4e-7
    # reward: exp of the device library against torch's / glibc's: <= 2 ulp per term
    assert ulps(h["reward_values"], g["ref_reward_values"]).max() <= 2
    assert ulps(h["reward_values"], o["reward_values"]).max() <= 2
    assert np.abs(h["reward"] - g["ref_reward"]).max() <= 1.2e-7
    # termination: exact
    for early in (True, False):
        hh = hip_outputs(g, early)
        assert np.array_equal(hh["reset"], g["ref_reset_early%d" % early])
        assert np.array_equal(hh["terminated"], g["ref_terminated_early%d" % early])
    # TocabiNewWalk's reward
    assert np.array_equal(h["nw_reset"], g["ref_nw_reset"])
    assert np.array_equal(h["nw_total"] == -1.0, g["ref_nw_total"] == -1.0)
    assert np.abs(h["nw_reward8"] - g["ref_nw_reward8"]).max() <= 5e-7
    assert np.abs(h["nw_total"] - g["ref_nw_total"]).max() <= 2e-7


def test_amp_argument_checks():
    from isaacgymdyros_amd import _lib
    lib, api = _lib.load()
    x = torch.zeros(16, 64, device="cuda")
    p = C.c_void_p(x.data_ptr())
    assert api["amp_observations"](0, p, p, p, p, p, p, p, p, None) != 0
    assert api["amp_observations"](4, None, p, p, p, p, p, p, p, None) != 0
    assert b"dw_amp_observations" in lib.dw_last_error()
    assert api["newwalk_reward"](4, p, p, p, p, p, p, p, 2, p, 15, 0.6, -1.0, 1000.0, p, 65, p, p, p, p, p, p, p, None) != 0


def test_body_positions_hip_vs_oracle():
    """dw_body_positions (the rows of the rigid-body state tensor the reset reads) against the oracle's double-precision
    quaternion chain, on random joint angles and base poses."""
    from isaacgymdyros_amd.config import default_cfg
    from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
    from oracle.oracle import OracleSim
    N = 64
    env = DyrosDynamicWalk(default_cfg(N, "cuda:0"), "cuda:0", 0, True)
    rng = np.random.default_rng(5)
    root = np.zeros((N, 13), np.float32)
    root[:, :3] = rng.normal(size=(N, 3))
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 3:7] = q
    dof = np.zeros((N, 33, 2), np.float32)
    dof[..., 0] = rng.uniform(-1.2, 1.2, size=(N, 33))
    env._buf["root_states"].copy_(torch.from_numpy(root))
    env._buf["dof_state"].copy_(torch.from_numpy(dof))
    bodies = [6, 12, 0, 3, 22, 33, 15, 30]
    arr = (C.c_int32 * len(bodies))(*bodies)
    out = torch.zeros(N, len(bodies), 3, device="cuda")
    assert env._api["body_positions"](env._h, arr, len(bodies), C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    osim = OracleSim(N)
    osim.buf["root_states"][:] = root
    osim.buf["dof_state"][:] = dof
    ref = np.zeros((N, len(bodies), 3), np.float32)
    assert osim.api["body_positions"](osim.h, arr, len(bodies), ref.ctypes.data_as(C.c_void_p), None) == 0
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err < 5e-6, err                     # fp32 matrix chain of up to 8 links against double: ~1e-6 m
    assert np.abs(ref[:, 2] - root[:, :3]).max() == 0          # moving body 0 is the base itself
    # out-of-range body index is refused on the host
    bad = (C.c_int32 * 1)(34)
    assert env._api["body_positions"](env._h, bad, 1, C.c_void_p(out.data_ptr()), None) != 0
    env.close()


def test_tocabi_amp_lower_env_steps():
    """The env class on the MI355X physics: shapes, the reset flow driven by reset_done() as the AMP learner drives it, finite
    observations, rewards inside the reference's range, terminations from falls, and the robots standing on their feet under
    zero actions at the start (the physics really is under the class)."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg, NUM_OBS, NUM_ACTIONS
    N = 256
    env = TocabiAMPLower(default_amp_cfg(N, "cuda:0"), "cuda:0", 0, True)
    assert env.num_obs == (NUM_OBS + NUM_ACTIONS) * 10 - NUM_ACTIONS == 468 and env.num_actions == 12
    obs, ids = env.reset_done()
    assert len(ids) == N and obs["obs"].shape == (N, 468)
    g = torch.Generator(device="cuda").manual_seed(3)
    total_resets, rew_sum = 0, 0.0
    for t in range(300):
        a = (torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * (0.0 if t < 40 else 0.6)
        obs, rew, reset, extras = env.step(a)
        assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all()
        assert float(rew.min()) >= -0.2 - 1e-6 and float(rew.max()) <= 0.8 + 0.6 + 0.1 + 0.05 + 0.05 + 0.08 + 0.6 + 1e-5
        if t == 39:
            # 40 steps of zero torque on the legs (upper body under its PD): nobody has terminated yet and both feet carry load
            z = env._root_states[:, 2]
            assert int(reset.sum()) == 0 or float((reset != 0).float().mean()) < 0.2, (int(reset.sum()), float(z.min()))
            fz = env._contact_forces[:, [8, 16], 2].sum(dim=1)
            assert float((fz > 100.0).float().mean()) > 0.5, float(fz.mean())
        rew_sum += float(rew.mean())
        total_resets += int(reset.sum())
        # the newest observation slot of obs_buf is this step's 36-word observation
        assert torch.equal(env.obs_buf[:, 36 * 9:36 * 10], env._obs1)
        _, ids = env.reset_done()
        assert int(env.reset_buf[ids].sum()) == 0 and int(env.progress_buf[ids].sum()) == 0
    assert total_resets > 0                       # random torques make robots fall: the termination path ran
    assert extras["reward_values"].shape == (N, 9) and len(extras["reward_names"]) == 9
    env.close()


class _ReplayDraws:
    """The reference class' recorded torch draws, handed out in order (oracle/make_amp_class_goldens.py).  Kind and element
    count of every request must match the recording: the class then makes the reference's draws in the reference's order."""

    def __init__(self, g):
        self.kind, self.off, self.data = [str(k) for k in g["draw_kind"]], g["draw_offset"], g["draw_data"]
        self.i = 0

    def _next(self, kind, shape):
        assert self.i < len(self.kind), "the class draws more than the reference did"
        assert self.kind[self.i] == kind, (self.i, self.kind[self.i], kind)
        v = self.data[self.off[self.i]:self.off[self.i + 1]]
        n = int(np.prod(shape)) if len(shape) else 1
        assert len(v) == n, (self.i, kind, len(v), shape)
        self.i += 1
        return torch.from_numpy(np.ascontiguousarray(v)).cuda().reshape(shape)

    def rand(self, *shape):
        return self._next("rand", shape)

    def randint(self, lo, hi, shape):
        return self._next("randint", tuple(shape)).long()

    def normal(self, shape, std):
        return self._next("normal", tuple(shape))

    def bernoulli(self, n, p):
        return self._next("bernoulli", (n,))


@pytest.mark.parametrize("fused_reset", [False, True])
def test_tocabi_amp_lower_class_replays_the_reference_class(fused_reset):
    """The reference's TocabiAMPLowerBase stepped 60 times over the oracle's physics (fixture tests/golden/amp_class_ref.npz:
    its draws, the physics state after every simulate, its outputs).  The host class here is given the same actions, the same
    draws and the same physics states (its generator and its simulate are injectable) and must produce the reference's numbers:
    torques handed to the engine, encoder model, command ramp, delayed-torque FIFO, histories, the 468-word observation,
    reward, termination, time-outs, and the reset flow in the reference's order."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    g = np.load(os.path.join(os.path.dirname(G), "amp_class_ref.npz"))
    N, STEPS = int(g["num_envs"]), int(g["steps"])
    cfg = default_amp_cfg(N, "cuda:0")
    cfg["env"]["episodeLength"] = int(g["episode_length"])
    cfg["task"]["randomize"] = False
    # (the fixture is torch-CPU arithmetic; fused_reset: reset_idx as the one launch of dw_amp_reset_rows -- it makes the reference's draws in
    #  the reference's order, so the reference class pins it directly)
    cfg["sim"]["mi355"] = {"amp_initial_height": 0.89, "torch_gpu_div": False, "amp_fused_reset": fused_reset}
    env = TocabiAMPLower(cfg, "cuda:0", 0, True)
    assert np.array_equal(env.motor_efforts.cpu().numpy(), g["motor_efforts"])
    assert np.array_equal(env.p_gains.cpu().numpy(), g["p_gains"]) and np.array_equal(env.d_gains.cpu().numpy(), g["d_gains"])
    env.total_mass[:] = torch.from_numpy(g["total_mass"]).cuda()
    env._rng = _ReplayDraws(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()          # noqa: E731
    state = {"k": 0, "fresh": False, "tau_err": 0.0}
    default_feet = env._foot_positions

    def inject(tau, push=None):
        k = state["k"]
        state["tau_err"] = max(state["tau_err"], float((tau - T(g["sim_tau"][k])).abs().max()))
        env._root_states.copy_(T(g["sim_root"][k]))
        env._dof_state.copy_(T(g["sim_dof"][k]))
        env._contact_forces.copy_(T(g["sim_contact"][k]))
        state["k"], state["fresh"] = k + 1, True

    def feet():
        if state["fresh"]:
            env._foot_pos.copy_(T(g["sim_feet"][state["k"] - 1]))
            state["fresh"] = False
        else:
            default_feet()
    env._simulate, env._foot_positions = inject, feet

    def ulps(a, b):
        return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    exact = {"reset_buf": "reset_buf", "_terminate_buf": "_terminate_buf", "progress_buf": "progress_buf", "commands": "commands",
             "qpos_noise": "qpos_noise", "qvel_noise": "qvel_noise", "qpos_pre": "qpos_pre", "action_log": "action_log", "epi_len": "epi_len",
             "qpos_bias": "qpos_bias", "quat_bias": "quat_bias", "_dof_vel_pre": "_dof_vel_pre", "actions_pre": "actions_pre",
             "action_history": "action_history", "delay_idx": "delay_idx", "simul_len": "simul_len"}
    for t in range(STEPS):
        _, ids = env.reset_done()
        want = g["ref_reset_ids"][t]
        assert np.array_equal(ids.cpu().numpy(), want[want >= 0]), t
        env.step(T(g["actions"][t]))
        for mine, ref in exact.items():
            a = getattr(env, mine).cpu().numpy()
            b = g["ref_" + ref][t]
            assert np.array_equal(a.reshape(b.shape).astype(b.dtype), b), (t, mine, np.abs(a.reshape(b.shape).astype(np.float64) - b).max())
        assert np.array_equal(env.timeout_buf.cpu().numpy(), g["ref_timeout_buf"][t]), t
        # the observation: exact but for the three Euler angles of every history slot (atan2f, see the function test)
        ob, rob = env.obs_buf.cpu().numpy(), g["ref_obs_buf"][t]
        eul = np.zeros(468, bool)
        for i in range(10):
            eul[36 * i:36 * i + 3] = True
        assert np.array_equal(ob[:, ~eul], rob[:, ~eul]), (t, np.argwhere(ob[:, ~eul] != rob[:, ~eul])[:4])
        assert np.abs(ob[:, eul] - rob[:, eul]).max() <= 2.4e-7, t
        oh, roh = env.obs_history.cpu().numpy(), g["ref_obs_history"][t]
        assert np.abs(oh - roh).max() <= 2.4e-7, t
        assert np.abs(env.rew_buf.cpu().numpy() - g["ref_rew_buf"][t]).max() <= 2.4e-7, t
    assert state["k"] == 2 * STEPS and env._rng.i == len(env._rng.kind)          # every simulate and every draw consumed
    assert state["tau_err"] == 0.0, state["tau_err"]                               # the torques handed to the engine, bit for bit
    assert g["ref__terminate_buf"].sum() > 0 and (g["ref_reset_ids"][1:] >= 0).sum() > 0          # the fixture terminates and resets envs
    env.close()


# ---------------------------------------------------------------------------------------------------------------------
# The AMP subclass' layer (reference tasks/tocabi_amp_lower.py): discriminator observations, reference state initialisation
GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "amp_disc_ref.npz")


def _disc(api, chk, root, dp, dv, row, elem, local, key, cuda):
    N, nk = root.shape[0], key.shape[1]
    if cuda:
        out = torch.zeros(N, 28 + 3 * nk, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
    else:
        out = np.zeros((N, 28 + 3 * nk), np.float32)
        p = lambda a: C.c_void_p(a.ctypes.data)
    chk(api["amp_disc_observations"](N, p(root), p(dp), p(dv), row, elem, int(local), p(key), nk, p(out), None))
    return out


def test_amp_disc_observations_hip_vs_oracle_and_reference():
    from isaacgymdyros_amd import _lib
    from oracle import oracle
    g = np.load(GD)
    lib, api = _lib.load()
    olib, oapi = oracle.load()

    def chk(rc):
        assert rc == 0, lib.dw_last_error()
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    root, key, dp, dv = (np.ascontiguousarray(g[k]) for k in ("root_states", "key_pos", "dof_pos", "dof_vel"))
    for local in (0, 1):
        h = _disc(api, chk, cu(root), cu(dp), cu(dv), 33, 1, local, cu(key), True)
        torch.cuda.synchronize()
        h = h.cpu().numpy()
        o = _disc(oapi, chk, root, dp, dv, 33, 1, local, key, False)
        ref = g["ref_obs_local%d" % local]
        d = ulps(h, ref)
        copies = [0] + list(range(4, 28))
        assert d[:, copies].max() == 0                                  # root height and the 24 dof entries: the reference's bits
        assert np.abs(h[:, 1:4] - ref[:, 1:4]).max() <= 2.4e-7          # Euler angles: atan2f (as dw_amp_observations above)
        if local:
            assert d[:, 28:].max() == 0
        else:
            # key bodies in the heading frame: downstream of atan2f / sinf / cosf of the heading (positions of order 1 m)
            assert np.abs(h[:, 28:] - ref[:, 28:]).max() <= 1e-6, np.abs(h[:, 28:] - ref[:, 28:]).max()
        assert np.abs(h - o).max() <= 1e-6
    # the two halves of an interleaved dof_state and the motion library's 12-wide tensors
    ds = cu(np.stack([dp, dv], axis=-1))
    h_il = _disc(api, chk, cu(root), ds[..., 0], ds[..., 1], 66, 2, 0, cu(key), True)
    h_33 = _disc(api, chk, cu(root), cu(dp), cu(dv), 33, 1, 0, cu(key), True)
    h_12 = _disc(api, chk, cu(root), cu(dp[:, :12]), cu(dv[:, :12]), 12, 1, 0, cu(key), True)
    torch.cuda.synchronize()
    assert torch.equal(h_il, h_33) and torch.equal(h_12, h_33)
    # argument checks: key-body count and strides
    x = torch.zeros(64, 66, device="cuda")
    assert api["amp_disc_observations"](4, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 33, 1, 0,
                                        C.c_void_p(x.data_ptr()), 9, C.c_void_p(x.data_ptr()), None) != 0
    assert api["amp_disc_observations"](4, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 12, 2, 0,
                                        C.c_void_p(x.data_ptr()), 2, C.c_void_p(x.data_ptr()), None) != 0
    assert b"amp_disc_observations" in lib.dw_last_error()


def _oracle_disc_of_env(env):
    from oracle import oracle
    olib, oapi = oracle.load()
    root = env._root_states.cpu().numpy()
    ds = np.ascontiguousarray(env._dof_state.cpu().numpy()).reshape(-1)
    key = np.ascontiguousarray(env._foot_pos.cpu().numpy())

    def chk(rc):
        assert rc == 0
    return _disc(oapi, chk, np.ascontiguousarray(root), ds, ds[1:], 66, 2, 0, key, False)


def test_tocabi_amp_lower_amp_obs_history():
    """stateInit Default: a reset env's AMP history is copies of its current observation; stepping shifts the history by one slot
    and the newest slot is build_amp_observations of the current state (checked against the CPU oracle)."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg, NUM_AMP_OBS_PER_STEP
    N = 128
    cfg = default_amp_cfg(N, "cuda:0")
    cfg["env"]["numAMPObsSteps"] = 3
    env = TocabiAMPLower(cfg, "cuda:0", 0, True)
    assert env.get_num_amp_obs() == 3 * NUM_AMP_OBS_PER_STEP == 102 and env.amp_observation_space.shape == (102,)
    env.reset_done()
    b = env._amp_obs_buf
    assert torch.equal(b[:, 1], b[:, 0]) and torch.equal(b[:, 2], b[:, 0])
    assert float(b[:, 0, 0].min()) > 0.9 and torch.equal(b[:, 0, 4:16], env._dof_pos[:, :12])          # root height, leg angles
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(30):
        prev = env._amp_obs_buf.clone()
        _, _, reset, extras = env.step((torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.5)
        cur = env._amp_obs_buf
        assert extras["amp_obs"].shape == (N, 102) and extras["amp_obs"].data_ptr() == cur.data_ptr()
        assert torch.equal(cur[:, 1:], prev[:, :-1])
        assert np.abs(cur[:, 0].cpu().numpy() - _oracle_disc_of_env(env)).max() <= 1e-6
        _, ids = env.reset_done()
        if len(ids):
            assert torch.equal(env._amp_obs_buf[ids, 1], env._amp_obs_buf[ids, 0])
            keep = torch.ones(N, dtype=torch.bool, device="cuda")
            keep[ids] = False
            assert torch.equal(env._amp_obs_buf[keep], cur[keep])
    env.close()


def test_tocabi_amp_lower_reference_state_init(tmp_path):
    """stateInit Random / Start / Hybrid on synthetic motion tables: reset envs start at the motion library's state for the ids and
    times it drew (numpy's global generator, re-seeded here to know them) -- the joint state, that is: the root is put back to its
    initial state as the reference does --, their AMP history holds the motion's earlier frames,
    and fetch_amp_obs_demo returns the demonstration observations of the same function."""
    from isaacgymdyros_amd.motion_lib import TocabiLowerMotionLib
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    from oracle import oracle
    from tests import amp_motion_synth as SY
    olib, oapi = oracle.load()
    yml = SY.write(str(tmp_path))
    ml = TocabiLowerMotionLib(yml, 33, "cpu")
    N = 64

    def chk(rc):
        assert rc == 0

    def motion_obs(ids, times):
        rp, rr, rv, ra, dp, dv, kp = ml.get_motion_state(ids, times)
        root = torch.cat([rp, rr, rv, ra], dim=-1).numpy()
        return _disc(oapi, chk, np.ascontiguousarray(root), np.ascontiguousarray(dp.numpy()), np.ascontiguousarray(dv.numpy()), 12, 1, 0,
                     np.ascontiguousarray(kp.numpy()), False), (rp, rr, rv, ra, dp, dv)
    with pytest.raises(ValueError, match="motion_file"):
        c = default_amp_cfg(N, "cuda:0"); c["env"]["stateInit"] = "Random"
        TocabiAMPLower(c, "cuda:0", 0, True)
    for mode in ("Random", "Start", "Hybrid"):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["env"].update({"stateInit": mode, "motion_file": yml, "numAMPObsSteps": 3, "hybridInitProb": 0.5})
        env = TocabiAMPLower(cfg, "cuda:0", 0, True)
        np.random.seed(77)
        _, ids = env.reset_done()
        assert len(ids) == N
        ref_ids = env._reset_ref_env_ids
        n_ref = len(ref_ids)
        np.random.seed(77)
        mids = ml.sample_motions(n_ref)
        mt = ml.sample_time(mids) if mode != "Start" else np.zeros(n_ref)
        assert np.array_equal(mids, env._reset_ref_motion_ids) and np.array_equal(mt, env._reset_ref_motion_times)
        if mode == "Hybrid":
            assert 0 < n_ref < N and len(env._reset_default_env_ids) == N - n_ref
            dflt = env._reset_default_env_ids
            assert torch.equal(env._dof_pos[dflt], env._initial_dof_pos[dflt]) and torch.equal(env._amp_obs_buf[dflt, 1], env._amp_obs_buf[dflt, 0])
        else:
            assert n_ref == N
        _, (rp, rr, rv, ra, dp, dv) = motion_obs(mids, mt)
        tol = dict(rtol=0, atol=2e-6)           # (slerp's acos / sin on the device against torch's CPU kernels)
        # the joints start from the motion; the root does NOT: the reference's reset writes the initial root state over it again
        # (tasks/amp/tocabi_amp_lower_base.py:262-263; pinned at class level by the subclass replay below)
        assert torch.equal(env._root_states[ref_ids], env._initial_root_states[ref_ids])
        assert torch.allclose(env._dof_pos[ref_ids, :12].cpu(), dp, **tol) and torch.allclose(env._dof_vel[ref_ids, :12].cpu(), dv, **tol)
        assert torch.equal(env._dof_pos[ref_ids, 12:], env._initial_dof_pos[ref_ids, 12:]) and float(env._dof_vel[ref_ids, 12:].abs().max()) == 0.0
        # history slots 1, 2 = the motion 1 and 2 simulation steps earlier
        for k in (1, 2):
            exp, _ = motion_obs(mids, mt - env.dt * k)
            got = env._amp_obs_buf[ref_ids, k].cpu().numpy()
            assert np.allclose(got, exp, rtol=0, atol=2e-5, equal_nan=True), (mode, k, np.nanmax(np.abs(got - exp)))
        # demonstrations
        np.random.seed(5)
        demo = env.fetch_amp_obs_demo(32)
        assert demo.shape == (32, 102)
        np.random.seed(5)
        di = ml.sample_motions(32)
        dt0 = ml.sample_time(di)
        for k in range(3):
            exp, _ = motion_obs(di, dt0 - env.dt * k)
            got = demo[:, 34 * k:34 * (k + 1)].cpu().numpy()
            assert np.allclose(got, exp, rtol=0, atol=2e-5, equal_nan=True), (mode, k, np.nanmax(np.abs(got - exp)))
        # and the env steps from there
        for t in range(5):
            obs, rew, reset, extras = env.step(torch.zeros(N, 12, device="cuda"))
            assert torch.isfinite(obs["obs"]).all() and extras["amp_obs"].shape == (N, 102)
        env.close()


def test_tocabi_amp_lower_subclass_replays_the_reference_subclass(tmp_path):
    """The reference's TocabiAMPLower SUBCLASS (stateInit Hybrid, numAMPObsSteps 3, motion library on the synthetic tables) stepped
    40 times over the oracle's physics (tests/golden/amp_subclass_ref.npz, oracle/make_amp_subclass_goldens.py).  Given the same
    actions, torch draws (incl. the Bernoulli draws that pick the kind of start), physics states and numpy seed, the host class must
    reset the same envs, start the same ones from the motion library at the same states, and carry the same discriminator
    observation history -- including the reference's habit of never clearing its lists of default / reference starts, so that a
    later reset re-initialises the history of envs from an earlier one -- and return the same demonstrations."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    from tests import amp_motion_synth as SY
    g = np.load(os.path.join(os.path.dirname(G), "amp_subclass_ref.npz"))
    N, STEPS = int(g["num_envs"]), int(g["steps"])
    cfg = default_amp_cfg(N, "cuda:0")
    cfg["env"].update({"episodeLength": int(g["episode_length"]), "stateInit": "Hybrid", "hybridInitProb": 0.5, "numAMPObsSteps": 3,
                       "motion_file": SY.write(str(tmp_path))})
    cfg["task"]["randomize"] = False
    cfg["sim"]["mi355"] = {"amp_initial_height": 0.89, "torch_gpu_div": False}
    env = TocabiAMPLower(cfg, "cuda:0", 0, True)
    env.total_mass[:] = torch.from_numpy(g["total_mass"]).cuda()
    env._rng = _ReplayDraws(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()          # noqa: E731
    state = {"k": 0, "fresh": False, "tau_err": 0.0}
    default_feet = env._foot_positions

    def inject(tau, push=None):
        k = state["k"]
        state["tau_err"] = max(state["tau_err"], float((tau - T(g["sim_tau"][k])).abs().max()))
        env._root_states.copy_(T(g["sim_root"][k]))
        env._dof_state.copy_(T(g["sim_dof"][k]))
        env._contact_forces.copy_(T(g["sim_contact"][k]))
        state["k"], state["fresh"] = k + 1, True

    def feet():
        if state["fresh"]:
            env._foot_pos.copy_(T(g["sim_feet"][state["k"] - 1]))
            state["fresh"] = False
        else:
            default_feet()
    env._simulate, env._foot_positions = inject, feet
    ref_now = []
    inner = env._reset_ref_state_init

    def spy(ids):
        ref_now.append(ids.cpu().numpy())
        return inner(ids)
    env._reset_ref_state_init = spy
    # tolerances: the motion library's blends run on the GPU here and on the CPU in the fixture (slerp: acos / sin), the observation's
    # Euler angles and heading frame go through atan2f / sinf / cosf; positions and angles are of order 1
    tol_state, tol_amp = 2e-6, 1e-5
    np.random.seed(int(g["np_seed"]))
    demos, amp_err, n_ref, n_default = [], 0.0, 0, 0
    for t in range(STEPS):
        ref_now.clear()
        _, ids = env.reset_done()
        want = g["ref_reset_ids"][t]
        assert np.array_equal(ids.cpu().numpy(), want[want >= 0]), t
        wref = g["ref_ref_ids"][t]
        got_ref = np.concatenate(ref_now) if ref_now else np.zeros(0, np.int64)
        assert np.array_equal(got_ref, wref[wref >= 0]), (t, got_ref, wref)
        n_ref += len(got_ref); n_default += len(ids) - len(got_ref)
        assert np.abs(env._root_states.cpu().numpy() - g["ref_root_after_reset"][t]).max() <= tol_state, t
        assert np.abs(env._dof_pos.cpu().numpy() - g["ref_dof_pos_after_reset"][t]).max() <= tol_state, t
        assert np.abs(env._dof_vel.cpu().numpy() - g["ref_dof_vel_after_reset"][t]).max() <= tol_state, t
        e = np.abs(env._amp_obs_buf.cpu().numpy() - g["ref_amp_after_reset"][t]).max()
        assert e <= tol_amp, (t, e)
        if t % int(g["demo_every"]) == int(g["demo_every"]) - 1:
            demos.append(env.fetch_amp_obs_demo(int(g["demo_n"])).cpu().numpy().copy())
        _, _, _, extras = env.step(T(g["actions"][t]))
        e = np.abs(extras["amp_obs"].view(N, 3, 34).cpu().numpy() - g["ref_amp_after_step"][t]).max()
        amp_err = max(amp_err, float(e))
        assert e <= tol_amp, (t, e)
        assert np.array_equal(env.reset_buf.cpu().numpy(), g["ref_reset_buf"][t]) and np.array_equal(env.progress_buf.cpu().numpy(), g["ref_progress_buf"][t]), t
    assert state["k"] == 2 * STEPS and env._rng.i == len(env._rng.kind)          # every simulate and every draw consumed
    assert state["tau_err"] <= 1e-4, state["tau_err"]          # (reference starts put ~1e-6 into the PD torques of the upper body: gains of order 1e3)
    assert n_ref > 8 and n_default > 8                          # both kinds of start occurred
    assert np.abs(np.stack(demos) - g["ref_demos"]).max() <= tol_amp
    print("AMP observation history vs the reference subclass: max |diff| %.2e over %d steps (%d reference starts, %d default starts)" % (amp_err, STEPS, n_ref, n_default))
    env.close()


def test_tocabi_amp_lower_graph_step_equals_eager():
    """enable_graph_step(): the recorded step replayed 60 times (reset_done() eager between the replays) against a second env that
    runs the same branch of the step eagerly -- same seeds, same actions: observations, rewards, resets, the AMP history and the
    physics state agree bit for bit (the generator is registered with the graph, so both draw the same numbers)."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    N = 256
    envs = []
    for k in range(2):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["env"]["episodeLength"] = 40
        envs.append(TocabiAMPLower(cfg, "cuda:0", 0, True))
    a, b = envs
    for e in envs:
        e.reset_done()
    a.enable_graph_step(warmup=3)
    b._capturing = True                                   # the recorded branch of the command ramp, run eagerly
    for _ in range(3):
        b._step_body(torch.zeros(N, 12, device="cuda"))
    g = torch.Generator(device="cuda").manual_seed(9)
    resets = 0
    for t in range(60):
        for e in envs:
            e.reset_done()
        act = (torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.8
        oa, ra, da, xa = a.step(act)
        ob, rb, db, xb = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db), t
        assert torch.equal(xa["amp_obs"], xb["amp_obs"]) and torch.equal(xa["time_outs"], xb["time_outs"]), t
        assert torch.equal(a._root_states, b._root_states) and torch.equal(a._dof_state, b._dof_state) and torch.equal(a.commands, b.commands), t
        resets += int(da.sum())
    assert resets > N // 2                               # falls and the 40-step episode limit: the reset path ran between replays
    with pytest.raises(ValueError, match="perturbation"):
        c = default_amp_cfg(64, "cuda:0"); c["env"]["perturbation"] = True
        TocabiAMPLower(c, "cuda:0", 0, True).enable_graph_step()
    for e in envs:
        e.close()


@pytest.mark.parametrize("one_launch", [False, True], ids=["three_kernels", "one_launch"])
@pytest.mark.parametrize("pd_control,plain", [(False, False), (True, False), (False, True)])
def test_tocabi_amp_lower_fused_step_equals_torch_step(pd_control, plain, one_launch):
    """one_launch (round 6): the whole step -- those three kernels AND the physics substeps between them -- as ONE launch (dw_amp_step; the octet
    kernels' substep inside, so both envs are pinned to that build of the physics): still the torch form's bits.
    cfg sim.mi355.amp_fused: the step's bookkeeping in three HIP kernels (dw_amp_step_begin / _mid / _end; histories as rings, in the
    `plain` case in the reference's shifting layout) against the torch implementation of the same class (the branch of the command ramp that draws for every env; itself pinned to the
    reference class by the replay tests above): same seeds and actions for 80 steps with resets in between -- every output and
    every piece of state must be bit-identical.  Episode length 40 so that time-outs and the command ramp (episode step 9) occur."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    N = 201                                              # (not a multiple of the kernels' four envs per workgroup)
    envs = []
    for fused in (True, False):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["env"].update({"episodeLength": 40, "pdControl": pd_control, "numAMPObsSteps": 3})
        if plain:          # no encoder / observation noise, no command ramp, no randomisation: the entry points then get no draws at all
            cfg["task"]["noise"], cfg["task"]["randomize"], cfg["env"]["velChange"] = False, False, False
        cfg["sim"]["mi355"] = dict({"amp_fused": fused}, **({"amp_hist_ring": not plain, "amp_one_launch": one_launch} if fused else {}))          # (the rings exist with the fused step only)
        if one_launch:
            cfg["sim"]["mi355"]["debug_wave_build"] = 2          # (201 envs would take the hex instantiation: the one-launch step carries the octet substep)
        envs.append(TocabiAMPLower(cfg, "cuda:0", 0, True))
    a, b = envs
    assert a._hist_ring == (not plain) and not b._hist_ring and a._one_launch == one_launch
    b._capturing = True
    g = torch.Generator(device="cuda").manual_seed(4)

    def state(env, n):          # (the two histories in the reference's layout, whatever the layout in memory)
        if n in ("action_history", "obs_history"):
            return env.history_linear()[0 if n == "action_history" else 1]
        return getattr(env, n)
    names = ["actions", "actions_pre", "action_history", "obs_history", "commands", "start_target_vel", "final_target_vel", "vel_change_duration",
             "cur_vel_change_duration", "epi_len", "action_log", "simul_len", "qpos_noise", "qvel_noise", "qpos_pre", "_dof_vel_pre", "progress_buf",
             "randomize_buf", "reset_buf", "_terminate_buf", "timeout_buf", "_rigid_body_pos", "_rigid_body_rot", "_foot_pos", "obs_buf", "rew_buf",
             "_reward_values", "_amp_obs_buf", "_root_states", "_dof_state", "_contact_forces"]
    ramps = resets = 0
    after_reset = ["obs_buf", "_amp_obs_buf", "commands", "qpos_bias", "quat_bias", "delay_idx", "perturb_timing", "epi_len_log", "power_scale",
                   "obs_history", "action_history", "action_log", "qpos_noise", "_root_states", "_dof_state", "_contact_forces", "_foot_pos", "reset_buf"]
    for t in range(80):
        ra_, rb_ = a.reset_done(), b.reset_done()          # (a: dw_amp_reset_rows, b: reset_idx's indexed assignments)
        assert torch.equal(ra_[1], rb_[1]) and torch.equal(ra_[0]["obs"], rb_[0]["obs"]), t
        for n in after_reset:
            assert torch.equal(state(a, n), state(b, n)), (t, "after reset", n)
        assert torch.equal(a._phys._buf["dof_damping"], b._phys._buf["dof_damping"]) and torch.equal(a._phys._buf["dof_armature"], b._phys._buf["dof_armature"]), t
        act = (torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * (1.3 if t % 7 == 0 else 0.7)          # (beyond +-1 now and then: the clamp)
        oa, ra, da, xa = a.step(act)
        ob, rb, db, xb = b.step(act)
        for n in names:
            va, vb = state(a, n), state(b, n)
            assert torch.equal(va, vb), (t, n, float((va.double() - vb.double()).abs().max()))
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db), t
        assert torch.equal(xa["amp_obs"], xb["amp_obs"]) and torch.equal(xa["time_outs"], xb["time_outs"]) and torch.equal(xa["terminate"], xb["terminate"]), t
        ramps += int((a.cur_vel_change_duration > 0).sum())
        resets += int(da.sum())
    assert resets > N and (ramps > 0 or plain)
    # and recorded in a hipGraph: the same numbers again
    a.enable_graph_step(warmup=2)
    for _ in range(2):
        b._step_body(torch.zeros(N, 12, device="cuda"))
    for t in range(20):
        for e in envs:
            e.reset_done()
        act = (torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.7
        oa, ra, da, xa = a.step(act)
        ob, rb, db, xb = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(xa["amp_obs"], xb["amp_obs"]), t
    for e in envs:
        e.close()


def test_fused_step_with_a_motion_start_keeps_the_shifting_histories(tmp_path):
    """amp_fused with a stateInit other than 'Default' takes the torch reset path, which reads the histories in the
    reference's shifting layout -- so the rings must be OFF there (they default to on only where every reset is the fused one), and
    asking for them is an error.  Checked as behaviour: the fused step with stateInit 'Hybrid' against the non-fused class, same
    seeds and actions, resets in between -- the observation returned by reset_done(), obs_buf and both histories bit-identical."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    from tests import amp_motion_synth as SY
    N = 67
    yml = SY.write(str(tmp_path))
    envs = []
    for fused in (True, False):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["env"].update({"episodeLength": 30, "stateInit": "Hybrid", "hybridInitProb": 0.5, "numAMPObsSteps": 3, "motion_file": yml})
        cfg["sim"]["mi355"] = {"amp_fused": fused}
        envs.append(TocabiAMPLower(cfg, "cuda:0", 0, True))
    a, b = envs
    assert a._fused and not a._hist_ring and not b._hist_ring
    b._capturing = True          # (the torch step's branch of the command ramp that draws for every env, as the fused step does)
    with pytest.raises(ValueError, match="amp_hist_ring"):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["env"].update({"stateInit": "Hybrid", "motion_file": yml})
        cfg["sim"]["mi355"] = {"amp_fused": True, "amp_hist_ring": True}
        TocabiAMPLower(cfg, "cuda:0", 0, True)
    with pytest.raises(ValueError, match="amp_hist_ring"):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["sim"]["mi355"] = {"amp_fused": True, "amp_fused_reset": False, "amp_hist_ring": True}
        TocabiAMPLower(cfg, "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(9)
    resets = 0
    for t in range(70):
        np.random.seed(100 + t)
        ra_ = a.reset_done()
        np.random.seed(100 + t)
        rb_ = b.reset_done()
        assert torch.equal(ra_[1], rb_[1]) and torch.equal(ra_[0]["obs"], rb_[0]["obs"]), t
        for n in ("obs_buf", "obs_history", "action_history", "_amp_obs_buf", "_dof_state", "_root_states"):
            assert torch.equal(getattr(a, n), getattr(b, n)), (t, "after reset", n)
        act = (torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.7
        oa, ra, da, xa = a.step(act)
        ob, rb, db, xb = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(xa["amp_obs"], xb["amp_obs"]), t
        for n in ("obs_buf", "obs_history", "action_history"):
            assert torch.equal(getattr(a, n), getattr(b, n)), (t, n)
        resets += int(da.sum())
    assert resets > N
    for e in envs:
        e.close()


def test_amp_reset_ids_equals_nonzero():
    """dw_amp_reset_ids against `reset_buf.nonzero()` (tasks/base/vec_task.py:381): ids ascending, the count on the device and in pinned host
    memory, for no flag, every flag, one flag at either end, random flags, at sizes that are and are not multiples of the workgroup; a count
    pointer that is plain pageable host memory is refused before the launch."""
    from isaacgymdyros_amd import _lib
    lib, api = _lib.load()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    pin = torch.zeros(1, dtype=torch.int64).pin_memory()
    s = torch.cuda.current_stream().cuda_stream
    for n in (1, 7, 1024, 1025, 4096, 16384, 50001):
        cases = [torch.zeros(n, dtype=torch.int64, device=dev), torch.ones(n, dtype=torch.int64, device=dev)]
        one = torch.zeros(n, dtype=torch.int64, device=dev); one[0] = 1; cases.append(one)
        last = torch.zeros(n, dtype=torch.int64, device=dev); last[n - 1] = 5; cases.append(last)
        for pr in (0.01, 0.5):
            cases.append((torch.rand(n, generator=g, device=dev) < pr).to(torch.int64))
        for flags in cases:
            ids = torch.full((n,), -1, dtype=torch.int64, device=dev)
            cnt = torch.full((1,), -1, dtype=torch.int64, device=dev)
            pin[0] = -1
            assert api["amp_reset_ids"](flags.data_ptr(), n, ids.data_ptr(), cnt.data_ptr(), pin.data_ptr(), s) == 0, lib.dw_last_error()
            torch.cuda.synchronize()
            ref = flags.nonzero(as_tuple=False).squeeze(-1)
            assert int(cnt[0]) == len(ref) == int(pin[0]), (n, int(cnt[0]), len(ref), int(pin[0]))
            assert torch.equal(ids[:len(ref)], ref) and bool((ids[len(ref):] == -1).all())
    import numpy as np
    pageable = np.zeros(1, dtype=np.int64)
    assert api["amp_reset_ids"](flags.data_ptr(), n, ids.data_ptr(), cnt.data_ptr(), pageable.ctypes.data, s) != 0 and b"pinned" in lib.dw_last_error()
    assert api["amp_reset_ids"](None, n, ids.data_ptr(), cnt.data_ptr(), None, s) != 0


def test_amp_step_argument_checks():
    """The fused entry points refuse a table with a missing buffer, sizes beyond what a wave stages, missing draws or a handle of
    another size, with an error code and a message, before anything is launched."""
    from isaacgymdyros_amd import _lib, abi
    from hip_backend import make_env
    lib, api = _lib.load()
    env = make_env(4)
    h = env._h
    c, b = abi.DwAmpConfig(), abi.DwAmpBuffers()
    x = torch.zeros(4096, device="cuda")
    c.num_envs, c.num_his, c.num_skip, c.log_slots, c.amp_steps = 4, 10, 2, 6, 2
    for name in abi.AMP_BUFFER_NAMES:
        setattr(b, name, x.data_ptr())
    P = lambda t: C.c_void_p(t.data_ptr())
    b.obs_history = None
    assert api["amp_step_begin"](h, C.byref(c), C.byref(b), P(x), None, None, None) != 0
    assert b"dw_amp_step_begin" in lib.dw_last_error()
    b.obs_history = x.data_ptr()
    c.num_his = 40                                       # 40 x 2 x 36 words: more than a wave stages
    assert api["amp_step_mid"](h, C.byref(c), C.byref(b), None, 1, None) != 0
    c.num_his, c.vel_change = 10, 1
    assert api["amp_step_begin"](h, C.byref(c), C.byref(b), P(x), None, None, None) != 0          # the ramp draws are missing
    c.vel_change, c.noise = 0, 1
    assert api["amp_step_mid"](h, C.byref(c), C.byref(b), None, 1, None) != 0                     # the encoder draws are missing
    assert api["amp_step_mid"](h, C.byref(c), C.byref(b), P(x), 0, None) != 0                     # substep 0 has no "between"
    assert api["amp_step_end"](h, C.byref(c), C.byref(b), P(x), 1, None, None) != 0              # the root-velocity draws are missing
    assert api["amp_reset_done"](h, C.byref(c), C.byref(b), None, None) != 0                      # no draws and no device_draws
    c.device_draws = 1
    b.draw_ctr = None
    assert api["amp_reset_done"](h, C.byref(c), C.byref(b), None, None) != 0                      # the draw counters are missing
    b.draw_ctr = x.data_ptr()
    assert api["amp_reset_done"](h, C.byref(c), C.byref(b), None, None) != 0 and b"delay_idx_range" in lib.dw_last_error()
    c.hist_ring, b.hist_head = 1, None
    assert api["amp_step_begin"](h, C.byref(c), C.byref(b), P(x), None, None, None) != 0          # the ring heads are missing
    c.num_envs = 8
    b.hist_head = x.data_ptr()
    assert api["amp_step_end"](h, C.byref(c), C.byref(b), None, 1, None, None) != 0 and b"num_envs" in lib.dw_last_error()
    env.close()


class _HipAmp:
    """tests/amp_emul.py::AmpEmul with the tables in device memory and the HIP library behind them."""

    def __init__(self, N, **kw):
        from amp_emul import AmpEmul
        from hip_backend import make_env
        from isaacgymdyros_amd import abi

        class _Shim:          # what AmpEmul's constructor touches of a backend
            pass
        self.env = make_env(N, self_collision=0)
        shim = _Shim()
        shim.buf = {k: v.cpu().numpy() for k, v in self.env._buf.items()}
        self.host = AmpEmul(shim, N, **kw)          # builds the config and the initial tables (numpy)
        for k in ("root_states", "dof_state", "dof_damping", "dof_armature"):
            self.env._buf[k].copy_(torch.from_numpy(shim.buf[k]).cuda())
        self.c, self.N, self.K = self.host.c, N, self.host.K
        self.t = {n: torch.from_numpy(v).cuda() for n, v in self.host.a.items()}
        self.b = abi.DwAmpBuffers()
        for n in abi.AMP_BUFFER_NAMES:
            setattr(self.b, n, self.t[n].data_ptr())
        self.api, self.h = self.env._api, self.env._h

    def _chk(self, rc):
        from isaacgymdyros_amd import _lib
        _lib.check(self.api, rc)

    def reset_done(self):
        self._chk(self.api["amp_reset_done"](self.h, C.byref(self.c), C.byref(self.b), None, None))

    def begin(self, actions):
        self._act = torch.from_numpy(actions).cuda()
        self._chk(self.api["amp_step_begin"](self.h, C.byref(self.c), C.byref(self.b), C.c_void_p(self._act.data_ptr()), None, None, None))

    def mid(self, k):
        self._chk(self.api["amp_step_mid"](self.h, C.byref(self.c), C.byref(self.b), None, k, None))

    def end(self):
        self._chk(self.api["amp_step_end"](self.h, C.byref(self.c), C.byref(self.b), None, self.K - 1, None, None))

    def arrays(self):
        torch.cuda.synchronize()
        return {n: v.cpu().numpy() for n, v in self.t.items()}

    # ---- the AmpEmul-like driver interface of tests/test_amp_emulation.py::device_vs_caller_draws
    @property
    def a(self):
        return self.arrays()

    def _dev(self, x):
        if x is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        self._keep.append(t)
        return C.c_void_p(t.data_ptr())

    def reset_done_with(self, draws):
        from isaacgymdyros_amd import abi
        self._keep = []
        d = None
        if draws is not None:
            d = abi.DwAmpResetDraws()
            for n in abi.AMP_RESET_DRAW_NAMES:
                setattr(d, n, self._dev(draws[n]).value)
            d = C.byref(d)
        self._chk(self.api["amp_reset_done"](self.h, C.byref(self.c), C.byref(self.b), d, None))
        torch.cuda.synchronize()

    def step(self, actions, z=(None, None), rootvel_noise=None, ramp=(None, None), sync_physics=None):
        self._keep = []
        c, b = C.byref(self.c), C.byref(self.b)
        self._chk(self.api["amp_step_begin"](self.h, c, b, self._dev(actions), self._dev(ramp[0]), self._dev(ramp[1]), None))
        for k in range(self.K):
            self.env.simulate(self.t["tau"])
            if k + 1 < self.K:
                self._chk(self.api["amp_step_mid"](self.h, c, b, self._dev(z[k]), k + 1, None))
        self._chk(self.api["amp_step_end"](self.h, c, b, self._dev(z[self.K - 1]), self.K - 1, self._dev(rootvel_noise), None))
        torch.cuda.synchronize()


def test_fused_amp_kernels_with_device_draws_equal_their_host_emulation():
    """sim.mi355.amp_device_draws: the draws are made inside the kernels, so torch has nothing to compare with -- the checker is the
    host emulation of the same kernel source (tests/emul, g++, fp contraction off like the kernels).  Both run reset_done / begin /
    mid / end on the same inputs for 24 steps (the physics state after every dw_simulate is copied from the GPU to the emulation):
    integer state, uniform-derived state (commands, biases, power scale, dof properties), histories, torques and the encoder state
    without noise terms must be bit-identical; what passes through the device's fast log / cos (encoder noise) or through
    atan2f / expf / sinf / cosf (observation, reward, discriminator observation) agrees to the rounding of those functions."""
    from amp_emul import AmpEmul
    from emul_backend import EmulSim
    N = 7
    kw = dict(seed=19, episode_length=10.0, hist_ring=True, device_draws=True)
    g = _HipAmp(N, **kw)
    e = AmpEmul(EmulSim(N, self_collision=0), N, **kw)
    e.a["total_mass"][:] = g.t["total_mass"].cpu().numpy()          # (the GPU env randomised its link masses at setup)
    exact = ["actions", "actions_pre", "commands", "start_target_vel", "final_target_vel", "vel_change_duration", "cur_vel_change_duration",
             "epi_len", "power_scale", "delay_idx", "simul_len", "qpos_bias", "quat_bias", "progress_buf", "randomize_buf", "reset_buf", "terminate_buf",
             "timeout_buf", "epi_len_log", "perturbation_count", "perturb_timing", "pert_on", "hist_head", "draw_ctr", "action_history", "action_log", "tau",
             "dof_vel_pre", "rigid_body_rot"]
    # (encoder state: q + noise, where the two sides' noise differs by the rounding of the fast log / cos, ~1e-11 -- enough to round the SUM to the
    #  neighbouring float now and then: one ulp of a joint angle below 2 rad, 1.2e-7, and that over dt = 0.002 in the rate)
    close = {"qpos_noise": 1.2e-7, "qvel_noise": 6e-5, "qpos_pre": 1.2e-7, "obs1": 2e-5, "obs_buf": 2e-5, "obs_out": 2e-5, "obs_history": 2e-5, "reward_values": 2e-6, "rew_buf": 2e-6, "amp_obs_buf": 2e-6, "amp_obs1": 2e-6, "foot_pos": 2e-6, "rigid_body_pos": 2e-6}
    rng = np.random.default_rng(5)
    resets = 0

    def sync_physics():
        torch.cuda.synchronize()
        for k in ("root_states", "dof_state", "contact_forces", "dof_damping", "dof_armature"):
            e.sim.buf[k][...] = g.env._buf[k].cpu().numpy()

    def compare(tag):
        ga = g.arrays()
        for n in exact:
            assert np.array_equal(ga[n], e.a[n]), (tag, n)
        for n, tol in close.items():
            d = np.abs(ga[n].astype(np.float64) - e.a[n])
            assert d.max() <= tol, (tag, n, float(d.max()), np.argwhere(d > tol)[:4].tolist(), ga["reward_values"][:2].tolist(), e.a["reward_values"][:2].tolist())
        for k in ("dof_damping", "dof_armature"):
            assert np.array_equal(g.env._buf[k].cpu().numpy(), e.sim.buf[k]), (tag, k)

    for t in range(24):
        resets += int(e.a["reset_buf"].sum())
        g.reset_done(); e.reset_done()
        # (the reset wrote the Gym rows on both sides from the same tables: they must already agree)
        torch.cuda.synchronize()
        for k in ("root_states", "dof_state", "contact_forces"):
            assert np.array_equal(g.env._buf[k].cpu().numpy(), e.sim.buf[k]), (t, k)
        compare((t, "reset"))
        act = ((rng.random((N, 12), dtype=np.float32) * 2 - 1) * 0.9).astype(np.float32)
        api, h, c, b = e.sim.api, e.sim.h, C.byref(e.c), C.byref(e.b)
        g.begin(act)
        e._chk(api["amp_step_begin"](h, c, b, act.ctypes.data, None, None, None))
        compare((t, "begin"))
        for k in range(g.K):
            g.env.simulate(g.t["tau"])
            sync_physics()
            if k + 1 < g.K:
                g.mid(k + 1)
                e._chk(api["amp_step_mid"](h, c, b, None, k + 1, None))
                compare((t, "mid"))
        g.end()
        e._chk(api["amp_step_end"](h, c, b, None, g.K - 1, None, None))
        compare((t, "end"))
    assert resets >= 2 * N
    g.env.close()


def test_device_draws_equal_caller_draws_from_the_numpy_restatement():
    """the device-draws form of the fused kernels -- what bench.py's amp_lower leg times -- had no checker but the
    host emulation of its own source.  Here it runs next to the caller-draws form of the same entry points, every draw handed over from
    oracle/amp_draws.py, an independent numpy restatement of the generator (Philox4x32-10 pinned by Random123's known-answer vectors,
    the key / stream / word scheme restated from the reference's draw sites); the caller-draws form in turn equals the torch
    implementation of the class bit for bit (test_tocabi_amp_lower_fused_step_equals_torch_step), which replays the reference class.
    Bars: integer and uniform-derived state bit-identical; what passes through the device's fast log / cos (the encoder's Box-Muller)
    to their rounding; the draw counters equal the test's own count (one per step, one per reset)."""
    from test_amp_emulation import device_vs_caller_draws
    N = 37
    kw = dict(seed=29, episode_length=9.0, hist_ring=True)
    dev, cal = _HipAmp(N, device_draws=True, **kw), _HipAmp(N, device_draws=False, **kw)
    cal.t["total_mass"].copy_(dev.t["total_mass"])
    for k in ("mass_scale", "total_mass"):          # (each env randomised its link masses at setup from its own generator state)
        cal.env._buf[k].copy_(dev.env._buf[k])
    dev.reset_done = lambda: dev.reset_done_with(None)
    cal.reset_done = lambda u=None: cal.reset_done_with(u)
    exact = ["actions", "commands", "start_target_vel", "final_target_vel", "vel_change_duration", "cur_vel_change_duration", "epi_len", "power_scale",
             "delay_idx", "simul_len", "qpos_bias", "quat_bias", "progress_buf", "randomize_buf", "reset_buf", "terminate_buf", "perturb_timing", "hist_head",
             "action_history", "action_log"]
    # (the device's fast log / cos against numpy's: 1e-6 of a 5e-5 rad draw, which can move the sum q + draw by one ulp of q -- 1.2e-7
    #  below 2 rad -- and the encoder's rate (difference of two such sums / 2 ms) by 1.2e-4; the observation divides by its scales)
    close = {"qpos_noise": 1.3e-7, "qvel_noise": 2e-4, "tau": 1e-3, "obs_buf": 5e-3, "rew_buf": 1e-3}

    def after(tag):
        da, ca = dev.arrays(), cal.arrays()
        for n in exact:
            assert np.array_equal(da[n], ca[n]), (tag, n)
        for n, tol in close.items():
            assert np.abs(da[n].astype(np.float64) - ca[n]).max() <= tol, (tag, n, float(np.abs(da[n].astype(np.float64) - ca[n]).max()))
        for k in ("dof_damping", "dof_armature"):
            assert torch.equal(dev.env._buf[k], cal.env._buf[k]), (tag, k)
    resets, ctr = device_vs_caller_draws(dev, cal, 30, after=after)
    assert resets >= 2 * N
    assert np.array_equal(dev.arrays()["draw_ctr"].astype(np.uint64), ctr)
    dev.env.close(); cal.env.close()


def test_tocabi_amp_lower_device_draws_class_level():
    """The host class with sim.mi355.amp_device_draws: step() recorded in a hipGraph + reset_done() as one launch, no torch draw.  The
    run is reproducible from the seed, differs with another seed, keeps drawing fresh noise at every replay (the draw counters live
    in device memory), resets what has to be reset, and its draws have the reference's distributions."""
    from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
    N = 512

    def make(seed, graph):
        cfg = default_amp_cfg(N, "cuda:0")
        cfg["seed"] = seed
        cfg["env"]["episodeLength"] = 30
        cfg["sim"]["mi355"] = {"amp_fused": True, "amp_device_draws": True}
        env = TocabiAMPLower(cfg, "cuda:0", 0, True)
        env.reset_done()
        if graph:
            env.enable_graph_step(warmup=2)
        else:
            for _ in range(2):
                env._step_body(torch.zeros(N, 12, device="cuda"))
        return env
    a, b, c = make(3, True), make(3, False), make(4, True)
    g = torch.Generator(device="cuda").manual_seed(2)
    enc, nres = [], 0
    for t in range(45):
        ids = [e.reset_done()[1] for e in (a, b, c)]
        assert torch.equal(ids[0], ids[1]), t
        nres += len(ids[0])
        for e in (a, b, c):
            assert int(e.reset_buf.sum()) == 0 and int(e.progress_buf[ids[0]].sum()) == 0 if e is not c else True
        act = (torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.6
        oa, ra, da, xa = a.step(act)
        ob, rb, db, xb = b.step(act)
        c.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(xa["amp_obs"], xb["amp_obs"]), t
        assert torch.equal(a.commands, b.commands) and torch.equal(a.qpos_noise, b.qpos_noise), t
        enc.append((a.qpos_noise[:, 12:] - a._dof_pos[:, 12:]).clone())
    assert nres >= N                                            # the 30-step limit (and falls)
    assert not torch.equal(a.commands, c.commands) and not torch.equal(a.qpos_bias, c.qpos_bias)
    assert not torch.equal(enc[-1], enc[-2])                    # fresh draws at every replay
    z = torch.stack(enc).flatten()
    assert float(z.abs().max()) <= 0.00016 + 1e-7
    assert abs(float(z.mean())) < 2e-7 and abs(float(z.std()) / (0.00016 / 3.0) - 1.0) < 0.02           # sigma 0.00016 / 3, clamped at 3 sigma
    for e in (a, b):
        assert float(e.power_scale.min()) >= 0.8 and float(e.power_scale.max()) <= 1.2 and abs(float(e.power_scale.mean()) - 1.0) < 0.01
        assert float(e.qpos_bias.abs().max()) <= 0.0314 + 1e-6 and abs(float(e.qpos_bias.std()) / (0.0628 / 12 ** 0.5) - 1.0) < 0.05
        d = e._phys._buf["dof_damping"]
        assert float(d.min()) >= 0.1 and float(d.max()) <= 3.0 + 1e-5 and abs(float(d.mean()) - 1.55) < 0.02
        assert int(e.delay_idx.min()) >= 2 and int(e.delay_idx.max()) <= 5
        assert torch.equal(e.obs_dict["obs"], torch.clamp(e.obs_buf, -e.clip_obs, e.clip_obs))
    lin_a, lin_b = a.history_linear(), b.history_linear()
    assert torch.equal(lin_a[0], lin_b[0]) and torch.equal(lin_a[1], lin_b[1])
    for e in (a, b, c):
        e.close()
