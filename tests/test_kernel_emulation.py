"""The HIP kernel body, compiled by g++ as a 64-lane loop (tests/emul), against the goldens and the oracle.
Catches indexing / region-structure mistakes on the CPU; the GPU tests (-m gpu) repeat these through the
real library."""
import numpy as np
import pytest

import replay as R
from emul_backend import EmulBackend, EmulSim
from oracle import parity as P
from oracle.oracle import OracleSim


@pytest.fixture(params=[2, 1, 3], ids=["two_waves", "keep", "hex"])
def wave_build(request):
    """Both forms of the octet step (dw_oct_kernels.h) run the same checks: 2 = the two-waves-per-SIMD form, which parks its
    per-joint state in global memory across the physics (what 16384 envs run on the GPU), 1 = the register-resident form of the
    one-wave-per-SIMD build (KEEP; launches of N <= 8192), 3 = the hex instantiation (the same source with 16 lanes per env, what
    launches of N <= 4096 run).  DwConfig.debug_wave_build selects; dw_simulate has one form per lane layout."""
    return request.param


def test_task_logic_bitwise_vs_reference_goldens(task_const, wave_build):
    g = R.load("task_logic_frozen.npz")
    be = EmulBackend(int(g["N"]), task_const, debug_wave_build=wave_build, randomize_dof_on_reset=0, debug_freeze_physics=1, torch_gpu_div=0)
    for t, ref, got in R.replay(g, be):
        exact = R.EXACT_LOGIC + ["qpos_noise", "qvel_noise", "root_states", "dof_state"]
        if "obs_history" in ref:
            exact = exact + ["action_history", "action_log", "actions_pre", "pre_joint_velocity_states",
                             "foot_force_pre", "action_torque_pre", "qpos_pre"]
        bad = P.compare(ref, got, exact=exact, atol=R.TRANSCENDENTAL)
        assert not bad, (t, bad)
    assert P.compare(ref, got, atol={"obs_history": (2e-6, 4e-6)}) == []


def test_terrain_curriculum_bitwise_vs_reference_golden(task_const, wave_build):
    """Row f-4 through the kernel source: level changes, tile origins and spawn jitter of the reference's curriculum."""
    g = R.load("terrain_logic_frozen.npz")
    be = EmulBackend(int(g["N"]), task_const, debug_wave_build=wave_build, randomize_dof_on_reset=0, debug_freeze_physics=1, torch_gpu_div=0,
                     terrain=R.GoldenTerrain(g), max_episode_length_s=float(g["cfg_max_episode_length_s"]))
    for t, ref, got in R.replay(g, be):
        # the reference's stacked_rewards carries the curriculum's logging columns (mean level per terrain type, :417-421) behind the
        # 15 reward columns: here they come out of dw_terrain_log, from the sums the step kernel formed
        wide = ref["stacked_rewards"]
        log = be.sim.terrain_log()
        assert log.shape == wide.shape and np.array_equal(log[:, 15:], wide[:, 15:]), t
        assert np.array_equal(log[:, :15], got["stacked_rewards"]), t
        ref["stacked_rewards"] = wide[:, :15]
        bad = P.compare(ref, got, exact=R.EXACT_LOGIC + ["qpos_noise", "qvel_noise", "root_states", "dof_state"],
                        atol=R.TRANSCENDENTAL)
        assert not bad, (t, bad)
        assert np.array_equal(g["step_terrain_levels"][t], got["terrain_levels"]), t
        assert np.array_equal(g["step_env_origins"][t], got["env_origins"]), t


def test_kernel_body_equals_oracle_bitwise_when_physics_frozen(task_const, wave_build):
    """Same libm on both sides here, so with physics frozen the kernel body and the oracle agree on every bit,
    in-kernel Philox noise included (noise = None)."""
    g = R.load("task_logic_frozen.npz")
    N = int(g["N"])
    from replay import OracleBackend
    a = OracleBackend(N, task_const, debug_freeze_physics=1, torch_gpu_div=1, randomize_friction_on_reset=1)
    b = EmulBackend(N, task_const, debug_wave_build=wave_build, debug_freeze_physics=1, torch_gpu_div=1, randomize_friction_on_reset=1)
    init = {k[5:]: v for k, v in g.items() if k.startswith("init_")}
    a.load_buffers(init)
    b.load_buffers(init)
    for t in range(int(g["steps"])):
        for be in (a, b):
            be.write_state(g["inj_root"][t], g["inj_dof"][t], g["inj_cf"][t])
            be.sim.step(g["actions"][t], None, t)
        sa, sb = P.snapshot_buffers(a.read_buffers()), P.snapshot_buffers(b.read_buffers())
        bad = P.compare(sa, sb, exact=list(sa.keys()))
        assert not bad, (t, bad)
        for k in ("dof_damping", "dof_armature", "friction_scale", "randomize_buf", "gate_acc"):
            assert np.array_equal(a.read_buffers()[k], b.read_buffers()[k]), k


def test_whole_step_tracks_oracle_goldens(task_const, wave_build):
    """Physics differs from the oracle only in summation order (Cholesky solve vs explicit inverse, fused
    Gauss-Seidel update).  Stated tolerance, contacts active, random torques: after 10 policy steps (20 substeps)
    |dq| <= 1e-4 rad, |dqd| <= 2e-2 rad/s (0.5 % of the 4.03 rad/s joint-speed limit), root pose <= 1e-4; the trajectories then separate chaotically, so
    beyond that only a sanity bound and the reset pattern are held."""
    g = R.load("whole_step_oracle.npz")
    be = EmulBackend(int(g["N"]), task_const, debug_wave_build=wave_build, randomize_dof_on_reset=0, torch_gpu_div=0)
    for t, ref, got in R.replay(g, be):
        dq = np.abs(ref["dof_state"][:, :, 0] - got["dof_state"][:, :, 0]).max()
        dqd = np.abs(ref["dof_state"][:, :, 1] - got["dof_state"][:, :, 1]).max()
        if t < 10:
            assert dq < 1e-4 and dqd < 2e-2, (t, dq, dqd)
            assert np.abs(ref["root_states"][:, :7] - got["root_states"][:, :7]).max() < 1e-4, t
            assert np.abs(ref["rew_buf"] - got["rew_buf"]).max() < 5e-3, t     # ~1e-4 of reward per newton of sole load
        assert dq < 5e-2, (t, dq)
        assert np.array_equal(ref["reset_buf"], got["reset_buf"]), t


def test_physics_substep_vs_oracle_random_flight(wave_build):
    rng = np.random.default_rng(1)
    N = 16
    A, B = OracleSim(N), EmulSim(N, debug_wave_build=wave_build)
    A.buf["root_states"][:, 0:3] = rng.normal(size=(N, 3)) + np.array([0, 0, 3])
    q = rng.normal(size=(N, 4))
    A.buf["root_states"][:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    A.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6)) * 0.5
    A.buf["dof_state"][:, :, 0] = rng.uniform(-1, 1, size=(N, 33))
    A.buf["dof_state"][:, :, 1] = rng.uniform(-1, 1, size=(N, 33))
    A.buf["mass_scale"][:] = rng.uniform(0.8, 1.2, size=(N, 38))
    A.buf["dof_damping"][:] = 0.1 + rng.uniform(0, 2.9, size=(N, 33))
    for k in ("root_states", "dof_state", "mass_scale", "dof_damping", "dof_armature"):
        B.buf[k][:] = A.buf[k]
    tau = rng.uniform(-50, 50, size=(N, 33)).astype(np.float32)
    push = rng.uniform(-100, 100, size=(N, 2)).astype(np.float32)
    for i in range(100):
        A.simulate(tau, push)
        B.simulate(tau, push)
    assert np.abs(A.buf["dof_state"][:, :, 0] - B.buf["dof_state"][:, :, 0]).max() < 1e-4
    assert np.abs(A.buf["root_states"] - B.buf["root_states"]).max() < 1e-4


def _gate_roundtrip(make, N=40):
    """Population gate (tasks/dyros_dynamic_walk.py:489): means of epi_len_log and contact_reward_mean over ALL envs
    decide whether pushes start; here via deterministic int64 bucket sums written by step t and read by step t+1."""
    from isaacgymdyros_amd import abi
    sim = make(N)
    es = sim.buf["env_state"]
    acts = np.zeros((N, 13), np.float32)
    sim.step(acts, None, 0)
    assert not abi.es_view(es, "perturb_start").any() and sim.buf["gate_acc"][abi.K["DW_GATE_LATCH"]] == 0
    # pretend the population has learned to walk: long episodes, synchronised contacts
    abi.es_view(es, "epi_len_log")[:] = 7000.0
    abi.es_view(es, "contact_reward_mean")[:] = 0.18
    sim.step(acts, None, 1)              # accumulates the statistics of step 1
    assert not abi.es_view(es, "perturb_start").any()
    sim.step(acts, None, 2)              # reads them: gate opens, latch is set
    assert abi.es_view(es, "perturb_start").all() and sim.buf["gate_acc"][abi.K["DW_GATE_LATCH"]] == 1
    abi.es_view(es, "epi_len_log")[:] = 0.0
    sim.step(acts, None, 3)
    sim.step(acts, None, 4)              # statistics dropped, the latch keeps the pushes on
    assert abi.es_view(es, "perturb_start").all()
    return sim.buf["gate_acc"].copy()


def test_perturbation_gate_latches_identically(task_const, wave_build):
    a = _gate_roundtrip(lambda N: OracleSim(N, task_const=task_const, debug_freeze_physics=1))
    b = _gate_roundtrip(lambda N: EmulSim(N, task_const=task_const, debug_wave_build=wave_build, debug_freeze_physics=1))
    assert np.array_equal(a, b)


def _crossed(N, symmetric):
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    q = np.tile(np.asarray(INITIAL_DOF_POS, np.float32), (N, 1))
    roll = np.linspace(0.05, 0.25, N).astype(np.float32)
    q[:, 1] = -roll
    q[:, 7] = roll
    if not symmetric:
        q[:, 6] = 0.15
        q[:, 8] += 0.25
    return q


def test_self_collision_vs_oracle(wave_build):
    """Row f-1 through the kernel source: skew capsule axes (well-conditioned), one substep: forces 1e-3 relative, state 1e-5."""
    N = 48
    A, B = OracleSim(N), EmulSim(N, debug_wave_build=wave_build)
    for s in (A, B):
        s.buf["root_states"][:, 0:2] = 0
        s.buf["root_states"][:, 2] = 3.0
        s.buf["dof_state"][:, :, 0] = _crossed(N, False)
    tau = np.zeros((N, 33), np.float32)
    A.simulate(tau); B.simulate(tau)
    ca, cb = A.buf["contact_forces"], B.buf["contact_forces"]
    assert (np.linalg.norm(ca, axis=2) > 1.0).any(axis=1).sum() > N // 2
    assert np.abs(ca - cb).max() <= 1e-3 * np.abs(ca).max()
    assert np.abs(A.buf["dof_state"] - B.buf["dof_state"])[:, :, 0].max() < 1e-5


@pytest.mark.parametrize("which", ["oracle", "oct"])
def test_mirrored_legs_get_a_mirrored_response(which):
    """Exactly parallel capsules (mirror-symmetric legs): contact in the middle of the overlap, so the response is mirrored
    -- joint rates of the two legs are mirror images, the base neither yaws nor drifts sideways (the textbook closest-point
    rule put the contact at whichever end rounding chose).  Envs whose shank axes intersect are left out."""
    N = 40
    sim = OracleSim(N) if which == "oracle" else EmulSim(N)
    sim.buf["root_states"][:, 0:2] = 0
    sim.buf["root_states"][:, 2] = 3.0
    sim.buf["dof_state"][:, :, 0] = _crossed(N, True)
    sim.simulate(np.zeros((N, 33), np.float32))
    keep = np.linspace(0.05, 0.25, N) < 0.16
    qd = sim.buf["dof_state"][keep][:, :, 1]
    assert np.linalg.norm(sim.buf["contact_forces"][keep], axis=2).max() > 1000.0
    sign = np.array([-1.0, -1.0, 1.0, 1.0, 1.0, -1.0], np.float32)
    assert np.abs(qd[:, 0:6] - sign * qd[:, 6:12]).max() < 5e-2        # (the model itself is mirror-symmetric to ~1 % only; the end-point rule gave 0.9 rad/s)
    assert np.abs(sim.buf["root_states"][keep][:, 12]).max() < 2e-2
    assert np.abs(sim.buf["root_states"][keep][:, 8]).max() < 2e-2


def test_arms_into_torso_vs_oracle(wave_build):
    """Row f-1, second tranche (forearm / hand against torso and thigh, arm against arm) through the kernel source: the
    arm poses of tests/test_oracle_physics.py (inside the joint limits), one substep: forces 1e-3 relative, state 1e-5."""
    from test_oracle_physics import _arms_in
    N = 48
    A, B = OracleSim(N), EmulSim(N, debug_wave_build=wave_build)
    for s in (A, B):
        s.buf["root_states"][:, 0:2] = 0
        s.buf["root_states"][:, 2] = 3.0
        s.buf["dof_state"][:, :, 0] = _arms_in(N)
    tau = np.zeros((N, 33), np.float32)
    A.simulate(tau); B.simulate(tau)
    ca, cb = A.buf["contact_forces"], B.buf["contact_forces"]
    loaded = np.linalg.norm(ca, axis=2) > 1.0
    assert loaded[:, 19].sum() > N // 2 and loaded[:, [23, 25, 27, 33, 35, 37]].any(axis=1).sum() > N // 2
    assert loaded[:, [25, 27, 35, 37]].any(axis=1).sum() >= 4              # forearms and hands too
    assert np.abs(ca - cb).max() <= 1e-3 * np.abs(ca).max()
    assert np.abs(A.buf["dof_state"] - B.buf["dof_state"])[:, :, 0].max() < 1e-5
    assert np.abs(A.buf["dof_state"] - B.buf["dof_state"])[:, :, 1].max() < 2e-2


def _random_poses(N, seed=3):
    """Joint angles drawn wide inside the joint limits: legs, arms and torso interpenetrate somewhere in most envs."""
    from isaacgymdyros_amd.model import load_model
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    m = load_model()
    lo, hi = np.minimum(m.dof_lower, m.dof_upper), np.maximum(m.dof_lower, m.dof_upper)
    rng = np.random.default_rng(seed)
    q0 = np.asarray(INITIAL_DOF_POS, np.float32)
    q = q0 + rng.uniform(-0.9, 0.9, size=(N, 33)).astype(np.float32)
    return np.clip(q, lo + 1e-3, hi - 1e-3).astype(np.float32)


def _random_arm_poses(N, seed=7):
    """The arms over 90 % of their whole joint range, everything else at the initial pose: a few envs in a hundred put a hand or a
    forearm on the head (the 16th proxy, round 5) or an upper arm on a thigh."""
    import json
    import os
    from isaacgymdyros_amd.model import load_model, MODEL_JSON
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    m = load_model()
    dn = json.load(open(MODEL_JSON))["dof_names"]
    lo, hi = np.minimum(m.dof_lower, m.dof_upper), np.maximum(m.dof_lower, m.dof_upper)
    arm = [i for i, n in enumerate(dn) if any(k in n for k in ("Shoulder", "Armlink", "Elbow", "Forearm", "Wrist"))]
    rng = np.random.default_rng(seed)
    q = np.tile(np.asarray(INITIAL_DOF_POS, np.float32), (N, 1))
    q[:, arm] = rng.uniform(lo[arm] * 0.9, hi[arm] * 0.9, size=(N, len(arm))).astype(np.float32)
    return q


def test_self_collision_head_and_upper_arm_pairs(model):
    """The fourth tranche of pairs (head against forearms / hands, upper arm against thigh and against the other arm): arm poses over the
    whole joint range; the kernel body loads exactly the Gym bodies the oracle loads, the head among them, forces 1e-3."""
    N = 512
    A, B = OracleSim(N), EmulSim(N)
    q = _random_arm_poses(N)
    for s in (A, B):
        s.buf["root_states"][:, 0:2] = 0
        s.buf["root_states"][:, 2] = 3.0
        s.buf["dof_state"][:, :, 0] = q
    tau = np.zeros((N, 33), np.float32)
    A.simulate(tau); B.simulate(tau)
    ca, cb = A.buf["contact_forces"], B.buf["contact_forces"]
    la, lb = np.linalg.norm(ca, axis=2) > 1.0, np.linalg.norm(cb, axis=2) > 1.0
    names = list(model.body_names)
    assert la[:, names.index("Head_Link")].sum() >= 2
    assert la[:, [names.index("L_Armlink_Link"), names.index("R_Armlink_Link")]].any(axis=1).sum() >= 8
    assert np.array_equal(la, lb)
    assert np.abs(ca - cb).max() <= 1e-3 * np.abs(ca).max()


def test_self_collision_random_poses_touch_every_pair_class(wave_build, model):
    """Row f-1 detection (dw_oct.h: axes built once per proxy, pairs tested in rounds by class, a half-precision threshold rounded
    up) must flag every pair the oracle's exhaustive test finds touching: random poses wide enough that leg x leg, arm x torso /
    arm x arm and arm x thigh pairs all occur; every Gym body the oracle loads is loaded by the kernel body too, forces 1e-3."""
    N = 96
    A, B = OracleSim(N), EmulSim(N, debug_wave_build=wave_build)
    q = _random_poses(N)
    for s in (A, B):
        s.buf["root_states"][:, 0:2] = 0
        s.buf["root_states"][:, 2] = 3.0          # (far above the ground: every contact force is a self-collision)
        s.buf["dof_state"][:, :, 0] = q
    tau = np.zeros((N, 33), np.float32)
    A.simulate(tau); B.simulate(tau)
    ca, cb = A.buf["contact_forces"], B.buf["contact_forces"]
    la, lb = np.linalg.norm(ca, axis=2) > 1.0, np.linalg.norm(cb, axis=2) > 1.0
    names = list(model.body_names)
    leg = [i for i, n in enumerate(names) if any(k in n for k in ("Thigh", "Knee", "Ankle", "Foot"))]
    arm = [i for i, n in enumerate(names) if any(k in n for k in ("Armlink", "Forearm", "Wrist2"))]
    torso = names.index("Upperbody_Link")
    assert la[:, leg].any(axis=1).sum() > N // 4 and la[:, arm].any(axis=1).sum() > N // 4 and la[:, torso].sum() > N // 8
    # an arm and a thigh loaded in the same env with no torso / other-arm load: the mixed class (arm x thigh) at work
    thigh = [names.index("L_Thigh_Link"), names.index("R_Thigh_Link")]
    assert (la[:, arm].any(axis=1) & la[:, thigh].any(axis=1)).sum() >= 3
    assert np.array_equal(la, lb)
    assert np.abs(ca - cb).max() <= 1e-3 * np.abs(ca).max()


def _terrain_reset_case(sim, g):
    """All envs at level 1 of the golden's curriculum map; env 3 has walked 6 m from its tile origin (level up), env 5 has
    not moved (level down), env 6 is not in the id list.  Returns the buffers after reset_idx([3, 5])."""
    N = sim.N
    for k, v in g.items():
        if k.startswith("init_") and k[5:] in sim.buf and sim.buf[k[5:]].shape == v.shape:
            sim.buf[k[5:]][...] = v
    types = sim.buf["terrain_types"]
    sim.buf["terrain_levels"][:] = 1
    org = sim.buf["terrain_origins"].reshape(int(g["cfg_terrain_num_levels"]), int(g["cfg_terrain_num_types"]), 3)
    for e in range(N):
        sim.buf["env_origins"][e] = org[1, int(types[e])]
        sim.buf["root_states"][e, :3] = sim.buf["env_origins"][e] + np.array([0, 0, 0.93], dtype=np.float32)
    sim.buf["root_states"][3, 0] += 6.0
    from isaacgymdyros_amd import abi
    abi.es_view(sim.buf["env_state"], "target_vel")[...] = np.array([0.4, 0.0], dtype=np.float32)
    abi.es_view(sim.buf["env_state"], "epi_len")[...] = 10.0
    sim.buf["randomize_buf"][:] = 5
    sim.reset_idx([3, 5], None, 17)
    return {k: np.array(v, copy=True) for k, v in sim.buf.items()}


def test_reset_idx_with_terrain_curriculum_vs_oracle(task_const, wave_build):
    """the reset_done path read the base position from uninitialised LDS when the curriculum decided the
    level change.  A moved env, a stationary env and an untouched env, several ids in one call, against the oracle; the
    emulation NaN-fills its LDS block per id, so stale contents cannot pass."""
    g = R.load("terrain_logic_frozen.npz")
    N = int(g["N"])
    kw = dict(terrain=R.GoldenTerrain(g), max_episode_length_s=float(g["cfg_max_episode_length_s"]), torch_gpu_div=1)
    ora = _terrain_reset_case(OracleSim(N, task_const=task_const, **kw), g)
    emu = _terrain_reset_case(EmulSim(N, task_const=task_const, debug_wave_build=wave_build, **kw), g)
    assert int(ora["terrain_levels"][3]) == 2 and int(ora["terrain_levels"][5]) == 0 and int(ora["terrain_levels"][6]) == 1
    for k in ("terrain_levels", "env_origins", "root_states", "dof_state", "env_state", "reset_buf", "progress_buf",
              "randomize_buf", "dof_damping", "dof_armature"):
        assert np.array_equal(ora[k], emu[k]), k
    assert np.isfinite(emu["root_states"]).all() and np.isfinite(emu["env_state"]).all()
