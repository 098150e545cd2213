"""CPU oracle vs the golden vectors recorded from the reference's own Python."""
import numpy as np
import pytest

import replay as R
from oracle import parity as P


def test_task_logic_matches_reference_bitwise(task_const):
    """Physics frozen: everything compared is the reference's torch task logic.  Non-transcendental fields are
    bit-identical (including -0.0 of the clamped clock action); exp/sin/cos/asin/atan2-derived fields differ
    by at most abs 2e-6 + rel 4e-6 (glibc vs torch-CPU SLEEF last-bit rounding)."""
    g = R.load("task_logic_frozen.npz")
    be = R.OracleBackend(int(g["N"]), task_const, randomize_dof_on_reset=0, debug_freeze_physics=1)
    resets = 0
    for t, ref, got in R.replay(g, be):
        exact = R.EXACT_LOGIC + ["qpos_noise", "qvel_noise", "root_states", "dof_state"]
        if "obs_history" in ref:
            exact = exact + ["action_history", "action_log", "actions_pre", "pre_joint_velocity_states",
                             "foot_force_pre", "action_torque_pre", "qpos_pre"]
        bad = P.compare(ref, got, exact=exact, atol=R.TRANSCENDENTAL)
        assert not bad, (t, bad)
        resets += int(ref["reset_buf"].sum())
    assert resets > 50          # the fixture exercises reset_idx heavily
    assert P.compare(ref, got, atol={"obs_history": (2e-6, 4e-6)}) == []


def test_minus_zero_clock_action_survives(task_const):
    """SURVEY a-7: bool*float clamp of the 13th action yields -0.0, which surfaces in obs_buf."""
    g = R.load("task_logic_frozen.npz")
    ob = g["step_obs_buf"][-1]
    act_part = ob[:, 370:].reshape(ob.shape[0], 9, 13)[:, :, 12]
    assert np.any(np.signbit(act_part) & (act_part == 0)), "fixture should contain -0.0 clock actions"


def test_whole_step_over_oracle_physics(task_const):
    """Reference class stepping over the oracle's physics vs dwo_step: same physics code on both sides, so state
    is bit-identical and only the transcendental fields carry the libm tolerance."""
    g = R.load("whole_step_oracle.npz")
    be = R.OracleBackend(int(g["N"]), task_const, randomize_dof_on_reset=0)
    for t, ref, got in R.replay(g, be):
        bad = P.compare(ref, got, exact=R.EXACT_LOGIC + ["root_states", "dof_state", "qpos_noise", "qvel_noise"],
                        atol=R.TRANSCENDENTAL)
        assert not bad, (t, bad)


def test_torch_gpu_division_mode_changes_only_last_bits(task_const):
    """torch's GPU kernels compute tensor/python_scalar as tensor*(1/scalar); the oracle can follow either."""
    g = R.load("task_logic_frozen.npz")
    be = R.OracleBackend(int(g["N"]), task_const, randomize_dof_on_reset=0, debug_freeze_physics=1, torch_gpu_div=1)
    worst = 0.0
    for t, ref, got in R.replay(g, be):
        if t > 2:
            break
        worst = max(worst, float(np.abs(ref["qvel_noise"] - got["qvel_noise"]).max()))
        assert np.allclose(ref["qvel_noise"], got["qvel_noise"], rtol=1e-6, atol=1e-6)
    assert worst >= 0.0


def test_terrain_curriculum_and_spawn_match_reference_bitwise(task_const):
    """Row f-4: the reference class on a height-field curriculum map (TerrainCfg edited as tests/golden says), physics
    frozen, robots teleported around their tile each step.  Level changes (:671-691, including the random level after the
    last one), the new tile origin, the spawn jitter (:729-732) and everything downstream are bit-identical."""
    g = R.load("terrain_logic_frozen.npz")
    be = R.OracleBackend(int(g["N"]), task_const, randomize_dof_on_reset=0, debug_freeze_physics=1,
                         terrain=R.GoldenTerrain(g), max_episode_length_s=float(g["cfg_max_episode_length_s"]))
    ups = downs = resets = 0
    prev = g["init_terrain_levels"].copy()
    types = g["init_terrain_types"]
    for t, ref, got in R.replay(g, be):
        exact = R.EXACT_LOGIC + ["qpos_noise", "qvel_noise", "root_states", "dof_state"]
        # the reference appends one logging column per terrain type to stacked_rewards (:417-426): the mean level of that
        # type's envs before this step's resets; the kernel's buffer holds the 15 reward terms, the host class adds these
        extra = ref["stacked_rewards"][:, 15:]
        ref["stacked_rewards"] = ref["stacked_rewards"][:, :15]
        for i in range(extra.shape[1]):
            assert np.allclose(extra[:, i], prev[types == i].sum() / max((types == i).sum(), 1))
        bad = P.compare(ref, got, exact=exact, atol=R.TRANSCENDENTAL)
        assert not bad, (t, bad)
        assert np.array_equal(g["step_terrain_levels"][t], got["terrain_levels"]), t
        assert np.array_equal(g["step_env_origins"][t], got["env_origins"]), t
        ups += int((got["terrain_levels"] > prev).sum()); downs += int((got["terrain_levels"] < prev).sum())
        prev = got["terrain_levels"].copy()
        resets += int(ref["reset_buf"].sum())
    assert resets > 50 and ups > 5 and downs > 5      # the fixture moves robots both ways
