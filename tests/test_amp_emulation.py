"""The fused TocabiAMPLower step and reset (isaacgymdyros_amd/csrc/dw_amp_step.h) under the host emulation: the kernel source
compiled by g++, a region = a loop over the wave's 64 lanes, LDS NaN-filled per env.  CPU only: indexing, region discipline
(nothing read that this env's wave did not write), the ring histories against the reference's shifting layout, the device
draws' ranges and their dependence on seed / env / draw counter.  The bit-level pin of the arithmetic is the GPU test against the
torch class (tests/test_amp_gpu.py); here the physics between the kernels is the octet emulation's dwe_simulate."""
import numpy as np
import pytest

from amp_emul import AmpEmul
from emul_backend import EmulSim


def make(N, **kw):
    sim = EmulSim(N, self_collision=0)
    return AmpEmul(sim, N, **kw)


def rollout(env, steps, seed=3, scale=0.5):
    rng = np.random.default_rng(seed)
    log = []
    for t in range(steps):
        ids = env.reset_done()
        env.step((rng.random((env.N, 12), dtype=np.float32) * 2 - 1) * scale)
        log.append((ids.copy(), env.a["obs_buf"].copy(), env.a["rew_buf"].copy(), env.a["reset_buf"].copy()))
    return log


def test_fused_step_with_device_draws_runs_clean_and_resets():
    env = make(6, episode_length=12.0)
    log = rollout(env, 30)
    for ids, obs, rew, rs in log:
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
    for n, v in env.a.items():
        if v.dtype == np.float32:
            assert np.isfinite(v).all(), n                       # a NaN here = a read of LDS nobody wrote
    assert sum(len(l[0]) for l in log) >= 6 * 3                  # the 12-step episode limit: every env reset at least three times
    a = env.a
    assert (a["progress_buf"] >= 0).all() and (a["progress_buf"] <= 12).all()
    assert ((a["delay_idx"] >= env.c.delay_idx_range[0]) & (a["delay_idx"] < env.c.delay_idx_range[1])).all()
    assert ((a["perturb_timing"] >= 0) & (a["perturb_timing"] < 4000)).all()
    assert ((a["power_scale"] >= 0.8) & (a["power_scale"] <= 1.2)).all() and a["power_scale"].std() > 0.05
    assert (np.abs(a["qpos_bias"]) <= 0.0314 + 1e-6).all() and (np.abs(a["quat_bias"]) <= 3.14 / 150 + 1e-6).all()
    assert a["qpos_bias"].std() > 0.005
    d, arm = env.sim.buf["dof_damping"], env.sim.buf["dof_armature"]
    assert ((d >= 0.1) & (d <= 3.0 + 1e-5)).all() and d.std() > 0.3
    assert ((arm >= 0.8 * a["nominal_armature"] - 1e-6) & (arm <= 1.2 * a["nominal_armature"] + 1e-6)).all()
    lo, sc = np.array(env.c.cmd_lo[:]), np.array(env.c.cmd_scale[:])
    # (a command is either a reset draw inside its range or on a ramp between two such values)
    assert ((a["commands"] >= lo - 1e-6) & (a["commands"] <= lo + sc + 1e-6)).all()
    assert (a["draw_ctr"] >= 30).all()                           # one per step and one per reset
    assert (a["obs_out"] <= 5.0).all() and (a["obs_out"] >= -5.0).all()
    assert np.array_equal(a["obs_out"], np.clip(a["obs_buf"], -5.0, 5.0))


def test_ring_histories_equal_the_shifting_layout():
    """The same seeds and actions with the histories as rings and in the reference's layout: every output and, read through the
    heads, both histories are identical bit for bit after every step and every reset."""
    ring, lin = make(5, hist_ring=True, episode_length=9.0), make(5, hist_ring=False, episode_length=9.0)
    rng = np.random.default_rng(1)
    for t in range(25):
        ia, ib = ring.reset_done(), lin.reset_done()
        assert np.array_equal(ia, ib)
        for k in (0, 1):
            assert np.array_equal(ring.history_linear()[k], lin.history_linear()[k]), (t, "after reset", k)
        assert np.array_equal(ring.a["obs_buf"], lin.a["obs_buf"]), t
        act = (rng.random((5, 12), dtype=np.float32) * 2 - 1) * 0.6
        ring.step(act); lin.step(act)
        for n in ("obs_buf", "obs_out", "rew_buf", "reset_buf", "amp_obs_buf", "commands", "qpos_noise", "action_log", "tau"):
            assert np.array_equal(ring.a[n], lin.a[n]), (t, n)
        for k in (0, 1):
            assert np.array_equal(ring.history_linear()[k], lin.history_linear()[k]), (t, k)
    assert ring.a["hist_head"].max() > 0 and lin.a["hist_head"].max() == 0


def test_device_draws_depend_on_seed_env_and_counter():
    a, b, c = make(4, seed=11), make(4, seed=11), make(4, seed=12)
    la, lb, lc = rollout(a, 6), rollout(b, 6), rollout(c, 6)
    for x, y in zip(la, lb):
        assert all(np.array_equal(u, v) for u, v in zip(x, y))                   # same seed: the same run
    assert not np.array_equal(a.a["qpos_bias"], c.a["qpos_bias"])                  # another seed: other draws
    assert len(np.unique(a.a["qpos_bias"][:, 0])) == 4                              # per env
    # per step: the encoder noise of two consecutive steps differs (the env's counter advanced)
    e = make(2, seed=5)
    e.reset_done()
    z = []
    for t in range(3):
        e.step(np.zeros((2, 12), np.float32))
        z.append((e.a["qpos_noise"] - e.sim.buf["dof_state"][..., 0]).copy())
    assert np.abs(z[0][:, 12:]).max() <= 0.00016 + 1e-7 and np.abs(z[0][:, 12:]).max() > 0
    assert not np.array_equal(z[0][:, 12:], z[1][:, 12:]) and not np.array_equal(z[1][:, 12:], z[2][:, 12:])


def test_caller_draws_are_used_verbatim():
    """device_draws = 0: the kernels take the caller's arrays (rows by env for dw_amp_reset_done)."""
    import ctypes as C
    from isaacgymdyros_amd import abi
    N = 4
    env = make(N, device_draws=False, episode_length=50.0)
    rng = np.random.default_rng(2)
    u = {n: rng.random(s, dtype=np.float32) for n, s in (("power_scale_u", (N, 12)), ("cmd_x_u", (N,)), ("cmd_y_u", (N,)), ("cmd_yaw_u", (N,)),
                                                       ("qpos_bias_u", (N, 12)), ("quat_bias_u", (N, 3)), ("damping_u", (N, 33)), ("armature_u", (N, 33)))}
    u["rootvel_noise"] = (rng.random((N, 6), dtype=np.float32) * 0.05 - 0.025).astype(np.float32)
    u["perturb_timing"] = rng.integers(0, 4000, N).astype(np.int64)
    u["delay_idx"] = rng.integers(2, 6, N).astype(np.int64)
    d = abi.DwAmpResetDraws()
    for n in abi.AMP_RESET_DRAW_NAMES:
        setattr(d, n, u[n].ctypes.data)
    env.a["randomize_buf"][:] = 1
    env.reset_done(C.byref(d))
    f32 = np.float32
    assert np.array_equal(env.a["power_scale"], f32(1.2 - 0.8) * u["power_scale_u"] + f32(0.8))
    assert np.array_equal(env.a["commands"][:, 0], f32(env.c.cmd_scale[0]) * u["cmd_x_u"] + f32(env.c.cmd_lo[0]))
    assert np.array_equal(env.a["delay_idx"], u["delay_idx"]) and np.array_equal(env.a["perturb_timing"], u["perturb_timing"])
    assert np.array_equal(env.sim.buf["dof_damping"], f32(0.1) + (f32(2.9 - 0.0) * u["damping_u"] + f32(0.0)))
    assert (env.a["randomize_buf"] == 0).all() and (env.a["reset_buf"] == 0).all()
    z0, z1 = (rng.standard_normal((N, 33)).astype(np.float32) * f32(0.00016 / 3.0) for _ in range(2))
    nz = u["rootvel_noise"]
    rd, ru = rng.integers(1, 250, N).astype(np.int64), rng.random((N, 3), dtype=np.float32)
    env.step(np.zeros((N, 12), np.float32), z=(z0, z1), rootvel_noise=nz, ramp=(rd, ru))
    q = env.sim.buf["dof_state"][..., 0]
    want = q + np.clip(z1, -0.00016, 0.00016).astype(np.float32)
    want[:, :12] += env.a["qpos_bias"]                          # (the observation function adds the bias in place)
    assert np.array_equal(env.a["qpos_noise"], want)


def test_reset_done_leaves_dof_properties_alone_without_task_randomize():
    """DwAmpConfig.randomize = 0: dw_amp_reset_done neither redraws damping / armature nor clears randomize_buf (the torch class and
    the reference run apply_randomizations, which is what resets the counter, under task.randomize only)."""
    N = 3
    env = make(N, randomize=False, episode_length=50.0)
    env.a["randomize_buf"][:] = 7
    damp, arm = env.sim.buf["dof_damping"].copy(), env.sim.buf["dof_armature"].copy()
    ids = env.reset_done()
    assert len(ids) == N and (env.a["reset_buf"] == 0).all()
    assert (env.a["randomize_buf"] == 7).all()
    assert np.array_equal(env.sim.buf["dof_damping"], damp) and np.array_equal(env.sim.buf["dof_armature"], arm)


def test_philox_restatement_known_answers():
    """oracle/amp_draws.py's Philox4x32-10 against the known-answer vectors of Random123 (kat_vectors: zero counter / key, and the
    digits-of-pi counter with key a4093822 299f31d0)."""
    from oracle.amp_draws import philox4x32_10
    c = np.array([[0, 0, 0, 0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xffffffff] * 4], dtype=np.uint32)
    assert philox4x32_10(c[0:1], 0, 0)[0].tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox4x32_10(c[1:2], 0xa4093822, 0x299f31d0)[0].tolist() == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert philox4x32_10(c[2:3], 0xffffffff, 0xffffffff)[0].tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def device_vs_caller_draws(dev, cal, steps, sync_physics=None, after=None):
    """Drives `dev` (device_draws = 1) and `cal` (device_draws = 0, every draw handed over from oracle/amp_draws.py, the numpy
    restatement of the kernels' generator) side by side.  Both are AmpEmul-like: .a tables, .c config, reset_done(draws), step(...).
    Returns the number of resets seen; `after(tag)` compares."""
    import ctypes as C
    from isaacgymdyros_amd import abi
    from oracle.amp_draws import AmpDraws
    N = dev.N
    gen = AmpDraws(dev.c.seed, N)
    ctr = np.zeros(N, dtype=np.uint64)          # the test's own count of the envs' draw counters
    rng = np.random.default_rng(11)
    resets = 0
    for t in range(steps):
        rmask = dev.arrays()["reset_buf"] != 0 if hasattr(dev, "arrays") else dev.a["reset_buf"] != 0
        resets += int(rmask.sum())
        u = gen.reset(ctr, int(dev.c.delay_idx_range[0]), int(dev.c.delay_idx_range[1]))
        dev.reset_done()
        cal.reset_done(u)
        ctr = ctr + rmask.astype(np.uint64)
        after((t, "reset"))
        act = ((rng.random((N, 12), dtype=np.float32) * 2 - 1) * 0.9).astype(np.float32)
        rd, ru = gen.ramp(ctr)
        z = [gen.encoder(ctr, k) for k in range(dev.K)]
        nz = gen.rootvel(ctr)
        dev.step(act, sync_physics=sync_physics)
        cal.step(act, z=tuple(z), rootvel_noise=nz, ramp=(rd, ru), sync_physics=sync_physics)
        ctr = ctr + np.uint64(1)
        after((t, "step"))
    return resets, ctr


def test_device_draws_equal_caller_draws_from_the_numpy_restatement():
    """The device-draws form of the fused kernels (what bench.py's amp_lower leg times) against the caller-draws form fed from an
    independent restatement of the generator (oracle/amp_draws.py): every integer and uniform-derived piece of state bit-identical, what
    passes through the encoder's Box-Muller to the rounding of log / cos.  On the host emulation here; tests/test_amp_gpu.py repeats it on
    the device."""
    N = 6
    kw = dict(seed=23, episode_length=9.0, hist_ring=True)
    dev, cal = make(N, device_draws=True, **kw), make(N, device_draws=False, **kw)
    exact = ["actions", "commands", "start_target_vel", "final_target_vel", "vel_change_duration", "cur_vel_change_duration", "epi_len", "power_scale",
             "delay_idx", "simul_len", "qpos_bias", "quat_bias", "progress_buf", "randomize_buf", "reset_buf", "terminate_buf", "perturb_timing", "hist_head",
             "action_history", "action_log", "tau"]
    close = {"qpos_noise": 3e-9, "qvel_noise": 3e-6, "obs_buf": 3e-5, "rew_buf": 3e-6}

    def after(tag):
        for n in exact:
            assert np.array_equal(dev.a[n], cal.a[n]), (tag, n)
        for n, tol in close.items():
            assert np.abs(dev.a[n].astype(np.float64) - cal.a[n]).max() <= tol, (tag, n)
        for k in ("dof_damping", "dof_armature", "root_states", "dof_state"):
            assert np.allclose(dev.sim.buf[k], cal.sim.buf[k], rtol=0, atol=1e-5), (tag, k)
        assert np.array_equal(dev.sim.buf["dof_damping"], cal.sim.buf["dof_damping"]), tag
    resets, ctr = device_vs_caller_draws(dev, cal, 30, after=after)
    assert resets >= 2 * N
    assert np.array_equal(dev.a["draw_ctr"].astype(np.uint64), ctr)          # one per step and one per reset, counted here independently
