"""Parity tests proper: the HIP library (through the C-ABI) against the reference goldens and the CPU oracle."""
import numpy as np
import pytest
import torch

import replay as R
from oracle import parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 2, 3], ids=["by_size", "two_waves", "hex"])
def wave_build(request):
    """The step / substep kernels are two builds of one source (dw_oct_kernels.hip): the one-wave-per-SIMD build keeps its per-joint
    state in registers and is what launches of N <= 8192 get; the two-waves build parks that state in HBM and is what the 16384-env
    headline runs.  0 = chosen by launch size (what a user gets: the hex instantiation up to 4096 envs), 2 = the two-waves build of
    the octet kernels forced (DwConfig.debug_wave_build), so that every small-N parity test below also checks the production-size
    code path bit for bit, 3 = the hex instantiation forced (16 lanes per env; the same source compiled with OCT_LPE = 16)."""
    return request.param


def _oracle_like(env, task_const):
    from oracle.oracle import OracleSim
    o = OracleSim(env.num_envs, task_const=task_const, cfg=env._ccfg, terrain=getattr(env, "terrain", None))
    for k, t in env._buf.items():
        o.buf[k][...] = t.cpu().numpy().reshape(o.buf[k].shape)
    return o


def test_task_logic_vs_reference_goldens(task_const, wave_build):
    """Physics frozen: the reference's torch task logic replayed through the HIP kernels.  Integer/flag fields and
    every float field without a transcendental are bit-identical to the reference's CPU torch run (torch_gpu_div=0
    selects torch's CPU division semantics); exp/sin/cos/asin/atan2-derived fields within abs 2e-6 + rel 4e-6
    (OCML vs SLEEF last-bit rounding)."""
    from hip_backend import HipBackend
    g = R.load("task_logic_frozen.npz")
    be = HipBackend(int(g["N"]), randomize=False, debug_freeze_physics=True, torch_gpu_div=False, debug_wave_build=wave_build)
    for t, ref, got in R.replay(g, be):
        exact = R.EXACT_LOGIC + ["qpos_noise", "qvel_noise", "root_states", "dof_state"]
        if "obs_history" in ref:
            exact = exact + ["action_history", "action_log", "actions_pre", "pre_joint_velocity_states",
                             "foot_force_pre", "action_torque_pre", "qpos_pre"]
        bad = P.compare(ref, got, exact=exact, atol=R.TRANSCENDENTAL)
        assert not bad, (t, bad)
    assert P.compare(ref, got, atol={"obs_history": (2e-6, 4e-6)}) == []


def test_whole_step_vs_oracle_goldens(task_const, wave_build):
    """Stated float tolerance on q/qd after N steps (contacts active, random torques): after 10 policy steps
    (20 substeps of 2 ms) |dq| <= 1e-4 rad, |dqd| <= 2e-2 rad/s (0.5 % of the 4.03 rad/s joint-speed limit), root pose <= 1e-4, reward <= 5e-3."""
    from hip_backend import HipBackend
    g = R.load("whole_step_oracle.npz")
    be = HipBackend(int(g["N"]), randomize=False, torch_gpu_div=False, debug_wave_build=wave_build)
    ref_rew, got_rew, ref_res, got_res = [], [], 0, 0
    for t, ref, got in R.replay(g, be):
        if t < 10:
            dq = np.abs(ref["dof_state"][:, :, 0] - got["dof_state"][:, :, 0]).max()
            dqd = np.abs(ref["dof_state"][:, :, 1] - got["dof_state"][:, :, 1]).max()
            assert dq < 1e-4 and dqd < 2e-2, (t, dq, dqd)
            assert np.abs(ref["root_states"][:, :7] - got["root_states"][:, :7]).max() < 1e-4, t
            assert np.abs(ref["rew_buf"] - got["rew_buf"]).max() < 5e-3, t     # ~1e-4 of reward per newton of sole load
            assert np.array_equal(ref["reset_buf"], got["reset_buf"]), t
        else:
            # past ~10 steps of contact the two fp32 trajectories separate chaotically (measured growth 1e-5 rad per 10
            # steps, then a contact flips); what must still agree is the statistics of the rollout
            ref_rew.append(ref["rew_buf"].mean()); got_rew.append(got["rew_buf"].mean())
            ref_res += int(ref["reset_buf"].sum()); got_res += int(got["reset_buf"].sum())
        assert np.isfinite(got["obs_buf"]).all() and np.isfinite(got["rew_buf"]).all(), t
    assert abs(np.mean(ref_rew) - np.mean(got_rew)) < 0.05, (np.mean(ref_rew), np.mean(got_rew))
    assert abs(ref_res - got_res) <= max(3, 0.3 * ref_res), (ref_res, got_res)


def test_physics_substep_vs_oracle(task_const, wave_build):
    """dw_simulate vs dwo_simulate, random in-flight states with randomised mass/damping/armature and a push:
    |dq| <= 1e-4 rad, root pose <= 1e-4, |dqd| and root velocity <= 1e-3 after 100 contact-free substeps."""
    from hip_backend import make_env
    rng = np.random.default_rng(1)
    N = 256
    env = make_env(N, self_collision=False, debug_wave_build=wave_build)      # random joint angles interpenetrate the legs; see the dedicated test
    b = env._buf
    root = np.zeros((N, 13), np.float32)
    root[:, 0:3] = rng.normal(size=(N, 3)) + np.array([0, 0, 3])
    q = rng.normal(size=(N, 4))
    root[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 7:13] = rng.normal(size=(N, 6)) * 0.5
    b["root_states"].copy_(torch.from_numpy(root))
    b["dof_state"][..., 0] = torch.from_numpy(rng.uniform(-1, 1, size=(N, 33)).astype(np.float32)).cuda()
    b["dof_state"][..., 1] = torch.from_numpy(rng.uniform(-1, 1, size=(N, 33)).astype(np.float32)).cuda()
    ora = _oracle_like(env, task_const)
    tau = rng.uniform(-50, 50, size=(N, 33)).astype(np.float32)
    push = rng.uniform(-100, 100, size=(N, 2)).astype(np.float32)
    tg, pg = torch.from_numpy(tau).cuda(), torch.from_numpy(push).cuda()
    for _ in range(100):
        env.simulate(tg, pg)
        ora.simulate(tau, push)
    torch.cuda.synchronize()
    assert np.abs(b["dof_state"][..., 0].cpu().numpy() - ora.buf["dof_state"][:, :, 0]).max() < 1e-4
    assert np.abs(b["dof_state"][..., 1].cpu().numpy() - ora.buf["dof_state"][:, :, 1]).max() < 1e-3
    droot = np.abs(b["root_states"].cpu().numpy() - ora.buf["root_states"])
    assert droot[:, :7].max() < 1e-4           # pose
    assert droot[:, 7:].max() < 1e-3           # velocities (|v| up to ~5 m/s after 0.2 s of random pushes)


def test_stance_contact_vs_oracle(task_const, wave_build):
    """Standing under full-strength PD: sole loads equal m*g within 2 % on both and agree with each other."""
    from hip_backend import make_env
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS, KP_RAW, KV_RAW
    env = make_env(64, randomize=False, debug_wave_build=wave_build)
    env._buf["root_states"][:, 0:2] = 0
    kp = torch.tensor(KP_RAW, device="cuda")
    kv = torch.tensor(KV_RAW, device="cuda")
    q0 = torch.tensor(INITIAL_DOF_POS, device="cuda")
    fz = []
    for i in range(600):
        env.simulate(kp * (q0 - env.dof_pos) - kv * env.dof_vel)
        if i >= 400:
            cf = env.contact_forces
            fz.append((cf[:, env.left_foot_idx, 2] + cf[:, env.right_foot_idx, 2]).mean().item())
    mg = env.model.nominal_total_mass * 9.81
    assert abs(np.mean(fz) - mg) < 0.02 * mg
    nonfoot = env.contact_forces[:, env.non_feet_idxs, :]
    assert float(nonfoot.abs().max()) == 0.0
    assert 0.90 < float(env.root_states[:, 2].mean()) < 0.94


def test_in_kernel_rng_matches_oracle_bitwise(task_const, wave_build):
    """Philox4x32-10 is integer work: with physics frozen and noise=None every uniform-derived field of the HIP
    step equals the oracle's bit for bit (reset draws, DR of damping/armature/friction, vel noise draw)."""
    from hip_backend import HipBackend
    from replay import OracleBackend
    g = R.load("task_logic_frozen.npz")
    N = int(g["N"])
    hb = HipBackend(N, randomize=True, debug_freeze_physics=True, debug_wave_build=wave_build)
    ob = OracleBackend(N, task_const, cfg=hb.env._ccfg)
    init = {k[5:]: v for k, v in g.items() if k.startswith("init_")}
    hb.load_buffers(init)
    ob.load_buffers(init)
    for t in range(8):
        for be in (hb, ob):
            be.write_state(g["inj_root"][t], g["inj_dof"][t], g["inj_cf"][t])
            be.step(g["actions"][t], None, t)
        a, b = P.snapshot_buffers(ob.read_buffers()), P.snapshot_buffers(hb.read_buffers())
        exact = ["reset_buf", "progress_buf", "delay_idx", "init_mocap_data_idx", "perturb_timing", "motor_constant_scale",
                 "target_vel", "mocap_data_idx", "action_torque", "target_data_qpos"]
        # obs_buf: the encoder draw uses the hardware log2 / cos (dw_task.h enc_normal): up to ~2e-4 relative on a 5e-5 rad
        # draw = 1e-8 rad in qpos_noise (measured 7.5e-9), x 1/dt = 5e-6 rad/s in qvel_noise, / std = 2e-5 in the observation
        bad = P.compare(a, b, exact=exact, atol={"qpos_bias": (1e-9, 1e-6), "obs_buf": (2e-5, 1e-5), "rew_buf": (2e-6, 4e-6)})
        assert np.abs(a["qpos_noise"] - b["qpos_noise"]).max() < 2e-8
        assert not bad, (t, bad)
        for k in ("dof_damping", "dof_armature", "friction_scale"):
            assert np.array_equal(ob.read_buffers()[k], hb.read_buffers()[k]), k


def test_reset_time_dr_vs_reference_on_gpu(wave_build):
    """The reference's apply_randomizations at reset (recorded over the fake gym with randomize = True) replayed through the
    HIP kernels: damping / armature to 2 ulp, the randomize_buf gate exact (tests/test_dr_reset.py has the details)."""
    from hip_backend import HipBackend
    from test_dr_reset import check_dr_replay
    g = R.load("dr_reset.npz")
    check_dr_replay(HipBackend(int(g["N"]), randomize=True, debug_freeze_physics=True, torch_gpu_div=False, debug_wave_build=wave_build))


@pytest.mark.parametrize("N,friction_dr", [(4096, False), (16384, False), (16384, True)])
def test_full_size_properties(N, friction_dr, wave_build):
    """BASELINE sizes (configs 2 and 5): size-independent properties of a 60-step random-action rollout with resets,
    mass/damping/armature DR, push perturbations forced on, and -- config 5 -- friction DR (divergent per-env
    contact sets)."""
    from hip_backend import make_env
    env = make_env(N, force_perturb_start=True, friction_dr=friction_dr, debug_wave_build=wave_build)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(42)
    resets = 0
    hist = []
    for t in range(60):
        a = torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
        obs, rew, done, extras = env.step(a)
        resets += int(done.sum())
        hist.append(obs["obs"][:, 333:370].clone())
        if t == 3:
            # pushes forced on (config 5): perturb_timing starts at 1, so every env that has not been reset since is inside
            # its push (durations are 25..249 policy steps) with a running count and a force of impulse / (duration * 0.004)
            fresh = env.epi_len >= 4
            assert int(fresh.sum()) > 0.5 * N
            assert bool((env.pert_on[fresh] == 1).all()) and bool((env.perturbation_count[fresh] == 3).all())
            mag = env.magnitude[fresh]
            assert float(mag.min()) >= 50 / (249 * 0.004) - 1e-3 and float(mag.max()) <= 249 / (25 * 0.004) + 1e-3
        if t >= 2:
            # obs_buf slot 8 at step t is slot 9 (the newest) at step t-2 unless the env was reset in between
            keep = (env.epi_len >= 3)
            assert torch.equal(obs["obs"][keep, 296:333], hist[t - 2][keep])
    torch.cuda.synchronize()
    assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all()
    assert resets > 0, "random actions must make some robots fall"
    qn = env.root_states[:, 3:7].norm(dim=1)
    assert float((qn - 1).abs().max()) < 1e-5
    assert float(rew.max()) <= 2.0 and float(rew.min()) >= -0.25
    assert int(env.nan_resets.sum()) == 0
    assert float(env.dof_vel.abs().max()) <= 4.03 + 1e-6
    assert extras["stacked_rewards"].shape == (N, 15) and len(extras["reward_names"]) == 15
    assert int(extras["time_outs"].sum()) == 0
    if friction_dr:
        fs = env._buf["friction_scale"]
        assert 0.7 <= float(fs.min()) and float(fs.max()) <= 1.3 and float(fs.std()) > 0.1
    # reward decomposition: total == sum of the 14 terms where the episode did not end on this step
    alive = done == 0
    assert torch.allclose(extras["stacked_rewards"][alive, :14].sum(1), rew[alive], atol=1e-5)


def test_push_moves_the_base():
    """A pushed env's base responds by impulse / effective mass: two runs that differ only in the perturbation switch,
    robots in free fall 3 m up (no contacts), zero actions.  Identical up to the step the push starts
    (perturb_timing = 1); after it the base velocity differs along the push direction by F*dt/m_eff with m_eff between
    the pelvis' own mass and the whole robot's (the push acts for the first 2 ms substep of the policy step only)."""
    from hip_backend import make_env
    N = 256
    envs = []
    for pert in (True, False):
        env = make_env(N, randomize=False, force_perturb_start=pert, seed=11)
        if not pert:
            env.perturb_start[:] = 0
        env._buf["root_states"][:, 2] = 3.0
        envs.append(env)
    a = torch.zeros(N, 13, device="cuda")
    A, B = envs
    # gravity stays on; both runs fall identically (3 m up, no contact within two steps), the difference isolates the push
    A.step(a); B.step(a)
    torch.cuda.synchronize()
    assert torch.equal(A.root_states, B.root_states)             # no push yet: bitwise the same kernel, same inputs
    A.step(a); B.step(a)
    torch.cuda.synchronize()
    assert bool((A.pert_on == 1).all()) and int(B.pert_on.sum()) == 0
    F = torch.stack([A.magnitude * torch.cos(A.phase), A.magnitude * torch.sin(A.phase)], 1)      # world x/y force [N]
    dv = (A.root_states[:, 7:9] - B.root_states[:, 7:9])
    J = F * 0.002                                                   # one substep of force
    along = (dv * J).sum(1) / J.norm(dim=1)                        # velocity change along the push
    m_total = float(A.total_mass.mean())
    m_pelvis = float(A.model.inert_mass[0])
    jn = J.norm(dim=1)
    assert bool((along > 0.9 * jn / m_total).all()), (float((along * m_total / jn).min()))
    assert bool((along < 1.1 * jn / m_pelvis).all()), (float((along * m_pelvis / jn).max()))
    # and it is a push, not a twist: the lateral part is small against the part along the force
    lateral = (dv - along[:, None] * J / jn[:, None]).norm(dim=1)
    assert float((lateral / along).max()) < 0.5


def test_determinism_and_reset_done():
    from hip_backend import make_env
    outs = []
    for _ in range(2):
        env = make_env(512)
        g = torch.Generator(device="cuda").manual_seed(3)
        for t in range(30):
            obs, rew, done, _ = env.step(torch.rand(512, 13, generator=g, device="cuda") * 2 - 1)
        outs.append((obs["obs"].clone(), rew.clone(), env.root_states.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    env.reset_buf[:] = 0
    env.reset_buf[5] = 1
    od, ids = env.reset_done()
    torch.cuda.synchronize()
    assert ids.tolist() == [5]
    assert float(env.root_states[5, 2]) == pytest.approx(0.93)
    assert int(env.progress_buf[5]) == 0 and float(env.epi_len[5]) == 0.0


def test_obs_reward_vs_torch_twin_on_gpu(task_const, model, wave_build):
    """The reference's observation/reward functions as an eager fp32 torch twin running ON THE GPU (torch-ROCm's own
    kernels: OCML transcendentals, torch's GPU reduction order, x/scalar as x*(1/s)) against the HIP kernels, same
    inputs, physics frozen, torch_gpu_div = 1 (the GPU flavour of division and of norm's summation order).  torch-GPU is
    not bit-identical to torch-CPU itself (transcendentals, `x / scalar`, the order in which norm() sums), so the CPU
    goldens pin the torch_gpu_div = 0 build and this test pins the default build: EVERY observation entry, every reward
    term and the total reward must be bit-identical to what torch computes on the same GPU."""
    from hip_backend import HipBackend
    from oracle import torch_twin as TW
    from isaacgymdyros_amd import abi
    g = R.load("task_logic_frozen.npz")
    N = int(g["N"])
    be = HipBackend(N, randomize=False, debug_freeze_physics=True, torch_gpu_div=True, debug_wave_build=wave_build)
    env = be.env
    be.load_buffers({k[5:]: v for k, v in g.items() if k.startswith("init_")})
    mean, var = env.obs_mean, env.obs_var
    nf = env.non_feet_idxs
    exact = total = 0
    # per output column: bit-identical count (37 observation entries, 14 reward terms, the total), for the attribution below
    col_exact = {"obs": torch.zeros(37, device="cuda"), "terms": torch.zeros(14, device="cuda"), "total": torch.zeros(1, device="cuda")}
    col_n = 0
    for t in range(int(g["steps"])):
        be.write_state(g["inj_root"][t], g["inj_dof"][t], g["inj_cf"][t])
        pre = {k: getattr(env, k).clone() for k in ("actions_pre", "pre_joint_velocity_states", "foot_force_pre")}
        root0, dof0, cf0 = env.root_states.clone(), env._buf["dof_state"].clone(), env.contact_forces.clone()
        be.step(g["actions"][t], g["noise"][t], t)
        nz = torch.from_numpy(g["noise"][t]).cuda()
        obs_t = TW.observation(env.root_states, env.quat_bias, env.qpos_noise, env.qpos_bias, env.qvel_noise, env.time,
                               env.init_mocap_data_idx.long(), env.target_vel,
                               nz[:, abi.K["DW_NZ_VEL"]:abi.K["DW_NZ_VEL"] + 6], mean, var)
        tot, st, r8, qe, col = TW.reward(root0, g_target_vel(env, g, t), env.target_data_qpos,
                                         env.target_data_force, dof0[..., 0], dof0[..., 1], pre["pre_joint_velocity_states"],
                                         env.actions, pre["actions_pre"], cf0, pre["foot_force_pre"][:, 0], pre["foot_force_pre"][:, 1],
                                         env.mocap_data_idx.long(), env.total_mass, nf, env.left_foot_idx, env.right_foot_idx)
        got_obs = env.obs_buf[:, 333:370]
        for a, b in ((obs_t, got_obs), (tot, env.rew_buf), (st, env._buf["stacked_rewards"][:, :14])):
            err = (a - b).abs()
            assert bool((err <= 2e-6 + 4e-6 * a.abs()).all()), (t, float(err.max()))
            exact += int((a == b).sum())
            total += a.numel()
        col_exact["obs"] += (obs_t == got_obs).float().sum(0)
        col_exact["terms"] += (st == env._buf["stacked_rewards"][:, :14]).float().sum(0)
        col_exact["total"] += (tot == env.rew_buf).float().sum()
        col_n += N
    frac = {k: (v / col_n).cpu().numpy() for k, v in col_exact.items()}
    print("bit-identical outputs HIP vs torch-GPU twin: %.2f %%" % (100.0 * exact / total))
    names = ["euler x", "euler y", "euler z"] + ["qpos %d" % i for i in range(12)] + ["qvel %d" % i for i in range(12)] + \
            ["sin phase", "cos phase", "target vx", "target vy"] + ["root vel %d" % i for i in range(6)]
    low = [(n, round(float(f), 4)) for n, f in zip(names, frac["obs"]) if f < 1.0]
    print("observation entries below 100 %:", low)
    print("reward terms:", [(n, round(float(f), 4)) for n, f in zip(env.extras["reward_names"][:14], frac["terms"])], "total", float(frac["total"][0]))
    # Measured on the MI355X (history: rounds 2-3 attributed the ~2 % of non-identical words to torch.norm and stopped there):
    #  * all 37 observation entries are bit-identical to torch-ROCm's eager result -- Euler angles (atan2), gait phase (sincos),
    #    normalisation included; exp is not a source of difference (OCML expf as hipcc links it equals torch.exp on 100 % of 3 M
    #    arguments, tools/probe_exp.py);
    #  * torch.norm over 33 / 12 / 3 / 2 elements on the GPU sums in the order of ATen's reduce kernel (a power-of-two number of
    #    threads per row, up to four accumulators per thread, a shuffle-down tree with ascending offsets), NOT in an order that
    #    depends on the row's alignment as round 3 believed: tools/probe_gpu_norm3.py predicts every bit of 1 M rows for each row
    #    length, any base alignment, any row count.  The kernels reproduce that order behind torch_gpu_div = 1 (dw_task.h
    #    norm_sel; the CPU order the reference goldens pin stays behind torch_gpu_div = 0), so every reward term is held to 100 %.
    assert float(frac["obs"].min()) == 1.0, low
    for n_, f_ in zip(env.extras["reward_names"][:14], frac["terms"]):
        assert f_ == 1.0, (n_, float(f_))
    assert float(frac["total"][0]) == 1.0
    assert exact == total


def g_target_vel(env, g, t):
    """target_vel as the reward saw it: the value BEFORE this step's reset (the golden records post-step values)."""
    prev = g["init_env_state"] if t == 0 else None
    if t == 0:
        import numpy as np
        es = torch.from_numpy(np.ascontiguousarray(prev))
        from isaacgymdyros_amd import abi
        return abi.es_view(es, "target_vel").cuda()
    return torch.from_numpy(g["step_target_vel"][t - 1]).cuda()


def _ppo():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "ppo_consumer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ppo_consumer_drives_the_env():
    """BASELINE config 3 in miniature: the DYROS PPO loop of examples/ppo_consumer.py (rl_games call pattern, separate
    optimisers, sigma schedule) consumes step/reset/extras unchanged; two epochs move the weights, losses stay finite."""
    mod = _ppo()
    stats = mod.train(num_envs=512, epochs=2, horizon=16, log=lambda *_: None)
    assert len(stats) == 2
    for s in stats:
        assert np.isfinite([s["mean_reward"], s["a_loss"], s["c_loss"], s["kl"], s["step_fps"]]).all()
        assert 0.0 <= s["mean_reward"] <= 2.0 and s["kl"] >= -0.01       # (rl_games' policy_kl carries a -13 * 5e-4 bias from its 1e-5 guards)
        assert len(s["reward_terms"]) == 15
    assert stats[0]["sigma"] > stats[1]["sigma"] > -2.9957                 # the action-noise schedule is running
    assert stats[0]["lr"] > stats[1]["lr"]


def test_ppo_rollout_step_captured_in_a_graph():
    """The rollout step of the PPO loop (policy inference, sampling, env step through dw_step_dev, bookkeeping) captured once in
    a hipGraph and replayed: the epoch statistics stay in the range of the eager loop's, every slot of the horizon is filled,
    and the env's device step counter has advanced once per replayed step (fresh noise per replay)."""
    mod = _ppo()
    eager = mod.train(num_envs=1024, epochs=2, horizon=32, log=lambda *_: None)
    graph = mod.train(num_envs=1024, epochs=2, horizon=32, log=lambda *_: None, graph_rollout=True)
    for e, g in zip(eager, graph):
        assert np.isfinite([g["mean_reward"], g["a_loss"], g["c_loss"], g["kl"], g["play_fps"]]).all()
        assert abs(g["mean_reward"] - e["mean_reward"]) < 0.15, (g["mean_reward"], e["mean_reward"])
        assert len(g["reward_terms"]) == 15 and abs(sum(list(g["reward_terms"].values())[:14]) - g["mean_reward"]) < 0.05
        assert g["mean_episode_length"] >= 0
    assert graph[0]["sigma"] > graph[1]["sigma"]                        # the in-place sigma schedule reaches the captured graph
    # ... and with the minibatch update captured as well (fused capturable Adam; learning rate as a device tensor the schedule writes)
    both = mod.train(num_envs=1024, epochs=2, horizon=32, log=lambda *_: None, graph_rollout=True, graph_update=True)
    for e, g in zip(eager, both):
        assert np.isfinite([g["mean_reward"], g["a_loss"], g["c_loss"], g["b_loss"], g["kl"], g["clip_frac"], g["total_fps"]]).all()
        assert abs(g["mean_reward"] - e["mean_reward"]) < 0.15
        assert abs(g["c_loss"] - e["c_loss"]) < 0.5 * abs(e["c_loss"]) + 1e-3, (g["c_loss"], e["c_loss"])       # the same optimisation problem, another Adam kernel
    assert both[0]["lr"] > both[1]["lr"]


def test_config3_16384_envs_with_the_ppo_loop_attached():
    """BASELINE config 3 at full size: 16384 envs, horizon 128, the DYROS PPO epoch (5 mini-epochs over minibatches of 4096).
    Finite losses, the per-term reward means logged from extras["stacked_rewards"], env stepping at a sane rate with a host
    synchronisation around every step; the line goes to profiles/."""
    import json
    import os
    mod = _ppo()
    stats = mod.train(num_envs=16384, epochs=1, horizon=128, log=lambda *_: None)
    s = stats[0]
    assert np.isfinite([s["mean_reward"], s["a_loss"], s["c_loss"], s["b_loss"], s["kl"], s["clip_frac"]]).all()
    assert len(s["reward_terms"]) == 15 and all(np.isfinite(v) for v in s["reward_terms"].values())
    assert abs(sum(list(s["reward_terms"].values())[:14]) - s["mean_reward"]) < 0.05      # the terms add up to the reward (alive envs)
    # (a floor, not a benchmark: this figure carries a host synchronisation per step and was seen between 18 M and 55 M on
    #  different boxes of the pool; bench.py is the measurement)
    assert s["step_fps"] > 8e6, s["step_fps"]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ppo_16384.json"), "w") as f:
        json.dump(dict(s, config="DyrosDynamicWalk num_envs=16384, PPO horizon 128, minibatch 4096, 5 mini-epochs, MLP 256-256"), f)


@pytest.mark.parametrize("N", [1, 3, 17, 100])
def test_small_and_odd_env_counts(N, task_const, wave_build):
    """Ragged sizes; compare with the oracle after a few steps.  All four sizes have N mod 16 in 1..8: the last octet workgroup's second
    wave has no envs and leaves the kernel before the per-substep s_barrier -- this test is the guard of the hardware rule that
    makes that legal (csrc/dw_oct_kernels.h, at the barrier); the lane kernels' last workgroup is partly empty as well."""
    from hip_backend import make_env
    env = make_env(N, debug_wave_build=wave_build)
    ora = _oracle_like(env, task_const)
    g = torch.Generator().manual_seed(N)
    for t in range(5):
        a = torch.rand(N, 13, generator=g) * 2 - 1
        obs, rew, done, _ = env.step(a.cuda())
        ora.step(a.numpy(), None, t)
    torch.cuda.synchronize()
    assert obs["obs"].shape == (N, 487) and torch.isfinite(obs["obs"]).all()
    assert np.abs(env.dof_pos.cpu().numpy() - ora.buf["dof_state"][:, :, 0]).max() < 1e-4
    assert np.array_equal(env.reset_buf.cpu().numpy(), ora.buf["reset_buf"])


def test_perturbation_gate_on_device():
    from hip_backend import make_env
    from isaacgymdyros_amd import abi
    N = 4096
    env = make_env(N, debug_freeze_physics=True)
    a = torch.zeros(N, 13, device="cuda")
    env.step(a)
    assert int(env.perturb_start.sum()) == 0
    env.epi_len_log[:] = 7000.0
    env.contact_reward_mean[:] = 0.18
    env.step(a)
    env.step(a)
    torch.cuda.synchronize()
    assert int(env.perturb_start.sum()) == N and int(env._buf["gate_acc"][abi.K["DW_GATE_LATCH"]]) == 1
    # integer bucket sums are order independent: exact expected totals of the statistics step
    acc = env._buf["gate_acc"].cpu().numpy()
    slot = 1
    tot_epi = acc[slot * 64: slot * 64 + 64: 2].sum()
    assert tot_epi == 7000 * N
    # the optional global gate of sharded runs (dist.sync_perturbation_gate), here on one rank: it reads the slot the newest step filled and
    # sets the latch by the same condition -- on a fresh env after ONE step with the statistics raised by hand, before the kernel's own gate
    # (which reads the PREVIOUS step's sums) has opened
    env2 = make_env(N, debug_freeze_physics=True)
    env2.epi_len_log[:] = 7000.0
    env2.contact_reward_mean[:] = 0.18
    env2.step(a)
    torch.cuda.synchronize()
    assert int(env2._buf["gate_acc"][abi.K["DW_GATE_LATCH"]]) == 0
    assert bool(env2.sync_perturbation_gate()) and int(env2._buf["gate_acc"][abi.K["DW_GATE_LATCH"]]) == 1
    env3 = make_env(N, debug_freeze_physics=True)
    env3.step(a)
    assert not bool(env3.sync_perturbation_gate()) and int(env3._buf["gate_acc"][abi.K["DW_GATE_LATCH"]]) == 0


def test_state_dict_roundtrip_resumes_bitwise():
    from hip_backend import make_env
    env = make_env(256)
    g = torch.Generator(device="cuda").manual_seed(9)
    acts = [torch.rand(256, 13, generator=g, device="cuda") * 2 - 1 for _ in range(20)]
    for a in acts[:10]:
        env.step(a)
    ck = env.state_dict()
    ref = [tuple(t.clone() for t in (env.step(a)[0]["obs"], env.rew_buf, env.reset_buf)) for a in acts[10:]]
    env2 = make_env(256)
    env2.load_state_dict(ck)
    for a, r in zip(acts[10:], ref):
        o = env2.step(a)
        assert torch.equal(o[0]["obs"], r[0]) and torch.equal(env2.rew_buf, r[1]) and torch.equal(env2.reset_buf, r[2])


def _crossed_legs(N, symmetric):
    """Joint angles of the self-collision sweeps: hips rolled inwards by 0.05..0.25 rad."""
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    q = torch.tensor(INITIAL_DOF_POS).repeat(N, 1)
    roll = torch.linspace(0.05, 0.25, N)
    q[:, 1] = -roll
    q[:, 7] = roll
    if not symmetric:
        q[:, 6] = 0.15            # right hip yawed and pitched: the two legs' capsule axes are skew lines at >= 8 cm
        q[:, 8] += 0.25
    return q


def test_self_collision_vs_oracle(task_const, model, wave_build):
    """SURVEY row f-1 on the device: legs rolled inwards in flight (centimetres of overlap, i.e. kilonewtons from the
    1e5 N/m penalty), the right leg yawed and pitched so that no two capsule axes are parallel or intersect (those cases
    are ill-conditioned by nature: see the symmetric test below).  After one substep the net contact forces of the colliding
    links agree with the oracle to 1e-3 relative and the joint state to 1e-5; the stiff explicit penalty then amplifies
    rounding differences, so four more substeps are only held to 2e-2 rad (in the task such a contact ends the episode at
    once)."""
    from hip_backend import make_env
    N = 64
    env = make_env(N, randomize=False, debug_wave_build=wave_build)
    b = env._buf
    b["root_states"][:, 0:2] = 0
    b["root_states"][:, 2] = 3.0
    b["dof_state"][..., 0] = _crossed_legs(N, symmetric=False).cuda()
    b["dof_state"][..., 1] = 0
    ora = _oracle_like(env, task_const)
    tau = torch.zeros(N, 33)
    env.simulate(tau.cuda())
    ora.simulate(tau.numpy())
    torch.cuda.synchronize()
    cg, co = env.contact_forces.cpu().numpy(), ora.buf["contact_forces"]
    assert (np.linalg.norm(co, axis=2) > 1.0).any(axis=1).sum() > N // 2          # most of the sweep collides
    assert np.abs(cg - co).max() <= 1e-3 * np.abs(co).max()
    assert np.array_equal(np.linalg.norm(cg, axis=2) > 1.0, np.linalg.norm(co, axis=2) > 1.0)
    assert np.abs(env.dof_pos.cpu().numpy() - ora.buf["dof_state"][:, :, 0]).max() < 1e-5
    for _ in range(4):
        env.simulate(tau.cuda())
        ora.simulate(tau.numpy())
    torch.cuda.synchronize()
    assert np.abs(env.dof_pos.cpu().numpy() - ora.buf["dof_state"][:, :, 0]).max() < 2e-2


def test_self_collision_random_poses_vs_oracle(task_const, model, wave_build):
    """Row f-1 detection on the device (axes built once per proxy, exchanged inside the octet by ds_bpermute, pairs tested in rounds
    by class against a half-precision threshold rounded up): random poses wide enough that pairs of every class touch; the set of
    loaded Gym bodies equals the oracle's exhaustive test and the forces agree to 1e-3 relative after one substep."""
    from hip_backend import make_env
    from test_kernel_emulation import _random_poses
    N = 200
    env = make_env(N, randomize=False, debug_wave_build=wave_build)
    b = env._buf
    b["root_states"][:, 0:2] = 0
    b["root_states"][:, 2] = 3.0
    b["dof_state"][..., 0] = torch.from_numpy(_random_poses(N, seed=5)).cuda()
    b["dof_state"][..., 1] = 0
    ora = _oracle_like(env, task_const)
    tau = torch.zeros(N, 33)
    env.simulate(tau.cuda())
    ora.simulate(tau.numpy())
    torch.cuda.synchronize()
    cg, co = env.contact_forces.cpu().numpy(), ora.buf["contact_forces"]
    lo = np.linalg.norm(co, axis=2) > 1.0
    assert lo.any(axis=1).sum() > N // 4 and lo.sum() > N // 2
    assert np.array_equal(np.linalg.norm(cg, axis=2) > 1.0, lo)
    assert np.abs(cg - co).max() <= 1e-3 * np.abs(co).max()


def test_self_collision_head_and_upper_arm_pairs_vs_oracle(task_const, model, wave_build):
    """The fourth tranche of pairs on the device (head against forearms / hands, upper arm against thigh and the other arm: pairs 32 ..
    46, the upper word of the 64-bit hit mask): arm poses over the whole joint range, the loaded Gym bodies equal the oracle's, the head
    among them, forces 1e-3 relative after one substep."""
    from hip_backend import make_env
    from test_kernel_emulation import _random_arm_poses
    N = 1024
    env = make_env(N, randomize=False, debug_wave_build=wave_build)
    b = env._buf
    b["root_states"][:, 0:2] = 0
    b["root_states"][:, 2] = 3.0
    b["dof_state"][..., 0] = torch.from_numpy(_random_arm_poses(N, seed=8)).cuda()
    b["dof_state"][..., 1] = 0
    ora = _oracle_like(env, task_const)
    tau = torch.zeros(N, 33)
    env.simulate(tau.cuda())
    ora.simulate(tau.numpy())
    torch.cuda.synchronize()
    cg, co = env.contact_forces.cpu().numpy(), ora.buf["contact_forces"]
    lo = np.linalg.norm(co, axis=2) > 1.0
    names = list(model.body_names)
    assert lo[:, names.index("Head_Link")].sum() >= 2
    assert np.array_equal(np.linalg.norm(cg, axis=2) > 1.0, lo)
    assert np.abs(cg - co).max() <= 1e-3 * np.abs(co).max()


def test_arms_into_torso_vs_oracle(task_const, model, wave_build):
    """Row f-1, second tranche, on the device: arm poses inside the joint limits that press upper arms, forearms and hands
    into the torso, a thigh or the other arm.  Same bars as the leg sweep: forces 1e-3 relative and the same set of loaded
    links after one substep, joint positions 1e-5; the whole step then sees the non-foot contact and ends the episode."""
    from hip_backend import make_env
    from test_oracle_physics import _arms_in
    N = 64
    env = make_env(N, randomize=False, debug_wave_build=wave_build)
    b = env._buf
    b["root_states"][:, 0:2] = 0
    b["root_states"][:, 2] = 3.0
    b["dof_state"][..., 0] = torch.from_numpy(_arms_in(N)).cuda()
    b["dof_state"][..., 1] = 0
    ora = _oracle_like(env, task_const)
    tau = torch.zeros(N, 33)
    env.simulate(tau.cuda())
    ora.simulate(tau.numpy())
    torch.cuda.synchronize()
    cg, co = env.contact_forces.cpu().numpy(), ora.buf["contact_forces"]
    loaded = np.linalg.norm(co, axis=2) > 1.0
    assert loaded[:, 19].sum() > N // 2 and loaded[:, [25, 27, 35, 37]].any(axis=1).sum() >= 4
    assert np.abs(cg - co).max() <= 1e-3 * np.abs(co).max()
    assert np.array_equal(np.linalg.norm(cg, axis=2) > 1.0, loaded)
    assert np.abs(env.dof_pos.cpu().numpy() - ora.buf["dof_state"][:, :, 0]).max() < 1e-5
    assert np.abs(env.dof_vel.cpu().numpy() - ora.buf["dof_state"][:, :, 1]).max() < 2e-2
    # through the task: a loaded torso / arm is a non-foot contact, the episode ends on the next step
    env2 = make_env(N, randomize=False, debug_wave_build=wave_build)
    env2.reset()
    env2._buf["dof_state"][..., 0] = torch.from_numpy(_arms_in(N)).cuda()
    ora2 = _oracle_like(env2, task_const)
    _, _, done, _ = env2.step(torch.zeros(N, env2.num_actions, device="cuda"))
    ora2.step(np.zeros((N, env2.num_actions), np.float32), None, 0)
    torch.cuda.synchronize()
    assert np.array_equal(done.cpu().numpy() != 0, ora2.buf["reset_buf"] != 0)
    assert int(done.sum()) >= N // 4


def test_self_collision_of_mirrored_legs_is_mirrored(wave_build):
    """The degenerate case: exactly mirror-symmetric legs make the two foot capsules (and the two ankle capsules) exactly
    parallel.  The written decision puts the contact of parallel capsules in the middle of their overlap, so a mirrored
    pose gets a mirrored response on every platform: left and right joint rates are mirror images and the base does not
    yaw.  (With the textbook closest-point rule the contact sat at whichever end rounding chose.)  Envs whose shank axes
    intersect (roll > ~0.17 rad: normal direction undefined) are left out."""
    from hip_backend import make_env
    N = 64
    env = make_env(N, randomize=False, debug_wave_build=wave_build)
    b = env._buf
    b["root_states"][:, 0:2] = 0
    b["root_states"][:, 2] = 3.0
    b["dof_state"][..., 0] = _crossed_legs(N, symmetric=True).cuda()
    b["dof_state"][..., 1] = 0
    env.simulate(torch.zeros(N, 33, device="cuda"))
    torch.cuda.synchronize()
    keep = torch.linspace(0.05, 0.25, N) < 0.16
    qd = env.dof_vel.cpu()[keep]
    cf = env.contact_forces.cpu()[keep]
    assert float(cf.norm(dim=2).max()) > 1000.0                     # feet and ankles are pressed into each other
    # TOCABI's leg joints: yaw, roll, pitch, pitch, pitch, roll; a mirrored motion flips the sign of yaw and roll rates
    sign = torch.tensor([-1.0, -1.0, 1.0, 1.0, 1.0, -1.0])
    assert float((qd[:, 0:6] - sign * qd[:, 6:12]).abs().max()) < 5e-2      # (the model is mirror-symmetric to ~1 % only; the end-point rule gave 0.9 rad/s)
    assert float(env.root_states.cpu()[keep][:, 12].abs().max()) < 2e-2        # no yaw rate of the base
    assert float(env.root_states.cpu()[keep][:, 8].abs().max()) < 2e-2         # no sideways velocity either


# ---------------------------------------------------------------------------------------------- terrain (row f-4)
@pytest.mark.gpu
def test_terrain_curriculum_vs_reference_golden_on_gpu(task_const):
    """The reference's curriculum fixture through the HIP library: levels, tile origins and spawn positions are
    bit-identical; the host class generates the same map as the reference did for that seed."""
    from hip_backend import HipBackend
    from oracle.make_goldens import TERRAIN_CASE
    g = R.load("terrain_logic_frozen.npz")
    hb = HipBackend(int(g["N"]), randomize=False, debug_freeze_physics=True, torch_gpu_div=False, terrain=dict(TERRAIN_CASE), seed=17)
    assert np.array_equal(hb.env.height_samples.cpu().numpy().ravel(), g["init_height_samples"].ravel())
    assert np.array_equal(hb.env.terrain_origins.cpu().numpy().ravel(), g["init_terrain_origins"].ravel())
    for t, ref, got in R.replay(g, hb):
        ref["stacked_rewards"] = ref["stacked_rewards"][:, :15]
        bad = P.compare(ref, got, exact=["reset_buf", "progress_buf", "target_vel", "root_states", "dof_state"],
                        atol=dict(R.TRANSCENDENTAL, obs_buf=(5e-6, 1e-5)))
        assert not bad, (t, bad)
        assert np.array_equal(g["step_terrain_levels"][t], got["terrain_levels"]), t
        assert np.array_equal(g["step_env_origins"][t], got["env_origins"]), t


@pytest.mark.gpu
def test_terrain_physics_vs_oracle_on_gpu(task_const, wave_build):
    """One substep on a generated map: HIP kernels (height-field variants) against the oracle.

    Round 1 saw joint rates differ by up to 5e-4 here where flat ground agrees to 1e-5.  Cause (tools/diag_terrain.py, run
    on the MI355X against the fp32 AND the fp64 oracle): not the height-field sampling or the contact frames -- the sets of
    loaded bodies are identical in all envs -- but conditioning.  The random poses of this test start up to centimetres
    inside rough terrain, the first substep answers with contact forces of up to 1e5 N (100 x the robot's weight), and every
    env above 2e-4 has a joint rate at the 4.03 rad/s clamp.  fp32 rounding scales with the force: the kernels are CLOSER
    to the fp64 oracle (7e-4) than the fp32 oracle itself is (1e-3).  So the bound scales with the load: 1.5e-4 rad/s plus
    3e-4 per 5 kN of peak contact force (an env standing on its feet, 1 kN, is held to 2.1e-4), 3e-5 without contact."""
    from hip_backend import make_env
    from isaacgymdyros_amd.terrain import Terrain, TerrainCfg
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    from oracle.oracle import OracleSim
    tdict = dict(mesh_type="heightfield", curriculum=True, num_rows=2, num_cols=4, border_size=2, max_init_terrain_level=1,
                 terrain_proportions=[0.2, 0.2, 0.3, 0.3, 0.0])
    N = 256
    env = make_env(N, randomize=False, terrain=tdict, seed=3, debug_wave_build=wave_build)
    t = Terrain(TerrainCfg(**tdict), N, seed=3)
    assert np.array_equal(env.height_samples.cpu().numpy(), t.heightsamples)
    A = OracleSim(N, terrain=t)
    rng = np.random.default_rng(5)
    org = t.env_origins.reshape(-1, 3)[rng.integers(0, 8, size=N)]
    A.buf["root_states"][:, 0:2] = org[:, 0:2] + rng.uniform(-3, 3, size=(N, 2))
    A.buf["root_states"][:, 2] = t.height_at(A.buf["root_states"][:, 0], A.buf["root_states"][:, 1]) + 0.93 + rng.uniform(-0.03, 0.05, size=N)
    A.buf["root_states"][:, 6] = 1.0
    A.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6)) * 0.3
    A.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS) + rng.normal(size=(N, 33)) * 0.05
    A.buf["dof_state"][:, :, 1] = rng.normal(size=(N, 33)) * 0.5
    env.root_states.copy_(torch.from_numpy(A.buf["root_states"]).cuda())
    env._buf["dof_state"].copy_(torch.from_numpy(A.buf["dof_state"]).cuda())
    tau = rng.uniform(-30, 30, size=(N, 33)).astype(np.float32)
    A.simulate(tau); env.simulate(torch.from_numpy(tau).cuda()); torch.cuda.synchronize()
    cfa, cfb = A.buf["contact_forces"], env.contact_forces.cpu().numpy()
    fmax = np.linalg.norm(cfa, axis=2).max(axis=1)                       # peak contact force per env [N]
    assert (fmax > 100.0).sum() > N // 2                                  # the terrain is being touched
    assert np.array_equal(np.linalg.norm(cfa, axis=2) > 0, np.linalg.norm(cfb, axis=2) > 0)      # same bodies loaded
    assert np.abs(cfa - cfb).max() <= 2e-3 * np.abs(cfa).max() + 0.05
    ds = env._buf["dof_state"].cpu().numpy()
    assert np.abs(A.buf["dof_state"][..., 0] - ds[..., 0]).max() < 1e-5           # positions after one substep
    err = np.abs(A.buf["dof_state"][..., 1] - ds[..., 1]).max(axis=1)
    assert (fmax < 5e3).sum() > N // 3                                    # a third of the envs carry realistic loads
    bound = 1.5e-4 + 3e-4 * fmax / 5e3                                    # measured on the MI355X: max err / bound = 0.87
    assert (err <= bound).all(), (err / bound).max()
    assert err[fmax == 0].max() < 3e-5                                    # no contact: the flat-ground figure
    assert np.abs(A.buf["root_states"] - env.root_states.cpu().numpy()).max() < 1e-3


@pytest.mark.gpu
def test_fallen_robots_on_high_rough_terrain_vs_oracle_on_gpu(wave_build):
    """The coarse bound of the height field (dw_physics.h terrain_bound, built at dw_bind) lets the kernels skip the fetches of bodies
    that cannot touch; the oracle samples under every primitive.  512 robots lying, kneeling and tumbling on the highest tiles of a
    generated map: the same bodies must be loaded, with the same forces (tests/test_terrain_physics.py runs the same scenario on the
    host emulation, where a bound lowered by 0.4 m was checked to fail)."""
    from hip_backend import make_env
    from isaacgymdyros_amd.terrain import Terrain, TerrainCfg
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    from oracle.oracle import OracleSim
    tdict = dict(mesh_type="heightfield", curriculum=True, num_rows=3, num_cols=5, border_size=2, max_init_terrain_level=2,
                 terrain_proportions=[0.1, 0.2, 0.35, 0.25, 0.1])
    N = 512
    env = make_env(N, randomize=False, terrain=tdict, seed=11, debug_wave_build=wave_build)
    t = Terrain(TerrainCfg(**tdict), N, seed=11)
    assert np.array_equal(env.height_samples.cpu().numpy(), t.heightsamples)
    A = OracleSim(N, terrain=t)
    rng = np.random.default_rng(2)
    org = t.env_origins.reshape(-1, 3)
    org = org[np.argsort(-org[:, 2])][:8]
    pick = org[rng.integers(0, len(org), size=N)]
    A.buf["root_states"][:, 0:2] = pick[:, 0:2] + rng.uniform(-3.5, 3.5, size=(N, 2))
    ground = t.height_at(A.buf["root_states"][:, 0], A.buf["root_states"][:, 1])
    A.buf["root_states"][:, 2] = ground + rng.uniform(0.12, 0.45, size=N)
    ax = rng.normal(size=(N, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0.6, 3.0, size=N)
    A.buf["root_states"][:, 3:6] = ax * np.sin(ang / 2)[:, None]
    A.buf["root_states"][:, 6] = np.cos(ang / 2)
    A.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6)) * 0.2
    A.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS) + rng.normal(size=(N, 33)) * 0.3
    A.buf["dof_state"][:, :, 1] = rng.normal(size=(N, 33)) * 0.5
    env.root_states.copy_(torch.from_numpy(A.buf["root_states"]).cuda())
    env._buf["dof_state"].copy_(torch.from_numpy(A.buf["dof_state"]).cuda())
    tau = np.zeros((N, 33), np.float32)
    A.simulate(tau); env.simulate(torch.from_numpy(tau).cuda()); torch.cuda.synchronize()
    cfa, cfb = A.buf["contact_forces"], env.contact_forces.cpu().numpy()
    ta, tb = np.linalg.norm(cfa, axis=2) > 1.0, np.linalg.norm(cfb, axis=2) > 1.0
    feet = np.zeros(38, bool); feet[[8, 16]] = True
    assert ta[:, ~feet].sum() >= 2 * N and float(ground.max()) > 0.3
    assert (ta != tb).sum() <= N // 100, int((ta != tb).sum())            # (a body within rounding of 1 N may flip)
    assert np.abs(cfa - cfb).max() <= 2e-3 * np.abs(cfa).max() + 0.05
    env.close()


@pytest.mark.gpu
def test_terrain_full_size_rollout_properties():
    """4096 envs on the default 10 x 20 curriculum map: finite, deterministic, robots spawn on their tiles, levels move."""
    from hip_backend import make_env
    tdict = dict(mesh_type="trimesh", curriculum=True)
    outs = []
    for rep in range(2):
        env = make_env(4096, terrain=tdict, force_perturb_start=True)
        g = torch.Generator(device="cuda").manual_seed(7)
        lv0 = env.terrain_levels.clone()
        for t in range(80):
            before = env.terrain_levels.clone()
            if t == 60:          # (a caller's edit of the table between steps is seen by the next step's logging columns)
                env.terrain_levels[::7] = (env.terrain_levels[::7] + 3) % 10
                before = env.terrain_levels.clone()
            if t in (70, 77):    # (a repeated / jumped step index -- a replay -- must not add into a logging slot nobody cleared: ADVICE r5)
                env._step_count += -1 if t == 70 else 5
            obs, rew, done, extras = env.step(torch.rand(4096, 13, generator=g, device="cuda") * 2 - 1)
            if t % 10 == 0 or t > 75:
                # the curriculum's logging columns as the reference forms them (tasks/dyros_dynamic_walk.py:417-421), from the levels the
                # step found (compute_reward runs before reset_idx)
                cols = []
                for i in range(20):
                    idx = (env.terrain_types == i).nonzero(as_tuple=False).squeeze(-1)
                    cols.append(torch.sum(before[idx]) / len(idx) * torch.ones_like(before).unsqueeze(-1))
                assert torch.equal(extras["stacked_rewards"][:, 15:], torch.cat(cols, 1)), t
                assert torch.equal(extras["stacked_rewards"][:, :15], env._buf["stacked_rewards"])
        torch.cuda.synchronize()
        outs.append((obs["obs"].clone(), env.root_states.clone(), env.terrain_levels.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all() and int(env.nan_resets.sum()) == 0
    assert extras["stacked_rewards"].shape == (4096, 15 + 20) and len(extras["reward_names"]) == 35
    lv = env.terrain_levels
    assert int(lv.min()) >= 0 and int(lv.max()) < 10 and int((lv != lv0).sum()) > 0          # random actions: robots fall early, levels go down
    d = (env.root_states[:, :2] - env.env_origins[:, :2]).norm(dim=1)
    fresh = env.progress_buf <= 1
    assert float(d[fresh].max()) < 1.5                                                          # spawned within 1 m (each axis) of the tile origin
    ground = torch.from_numpy(env.terrain.height_at(env.root_states[:, 0].cpu().numpy(), env.root_states[:, 1].cpu().numpy())).cuda()
    assert float((env.root_states[fresh, 2] - ground[fresh]).min()) > 0.5                       # above the terrain, not inside it


@pytest.mark.gpu
def test_terrain_checkpoint_and_reset_done():
    """On terrain: a checkpoint resumes bit for bit (levels and origins are part of the state), and the reset_done path
    (dw_reset_idx) runs the curriculum for the listed envs only."""
    from hip_backend import make_env
    tdict = dict(mesh_type="heightfield", curriculum=True, num_rows=4, num_cols=5, border_size=2, max_init_terrain_level=3)
    env = make_env(200, terrain=tdict)
    g = torch.Generator(device="cuda").manual_seed(9)
    acts = [torch.rand(200, 13, generator=g, device="cuda") * 2 - 1 for _ in range(40)]
    for a in acts[:25]:
        env.step(a)
    ck = env.state_dict()
    ref = [tuple(t.clone() for t in (env.step(a)[0]["obs"], env.rew_buf, env.terrain_levels, env.env_origins)) for a in acts[25:]]
    env2 = make_env(200, terrain=tdict)
    env2.load_state_dict(ck)
    for a, r in zip(acts[25:], ref):
        o = env2.step(a)
        assert torch.equal(o[0]["obs"], r[0]) and torch.equal(env2.rew_buf, r[1])
        assert torch.equal(env2.terrain_levels, r[2]) and torch.equal(env2.env_origins, r[3])
    # reset_done: env 7 has "walked" 6 m from its tile centre -> one level up (or a random level from the top one); env 8 stays
    lv = env2.terrain_levels.clone()
    env2.root_states[7, 0] = env2.env_origins[7, 0] + 6.0
    env2.reset_buf[:] = 0
    env2.reset_buf[7] = 1
    env2.reset_done()
    torch.cuda.synchronize()
    expect_up = int(lv[7]) + 1
    assert int(env2.terrain_levels[7]) == expect_up if expect_up < 4 else 0 <= int(env2.terrain_levels[7]) < 4
    assert torch.equal(env2.terrain_levels[8:], lv[8:]) and torch.equal(env2.terrain_levels[:7], lv[:7])
    org = env2.terrain_origins.view(4, 5, 3)[env2.terrain_levels[7], env2.terrain_types[7]]
    assert torch.equal(env2.env_origins[7], org)
    assert float((env2.root_states[7, :2] - org[:2]).abs().max()) <= 1.0 and float(env2.root_states[7, 2]) == pytest.approx(float(org[2]) + 0.93, abs=1e-6)


@pytest.mark.gpu
def test_env_from_yaml_config():
    """The constructor takes the nested dict a task YAML yields (config.load_task_yaml): round trip through YAML text."""
    import yaml
    from isaacgymdyros_amd.config import default_cfg, load_task_yaml
    from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
    text = yaml.safe_dump(default_cfg(64, "cuda:0")).replace("numEnvs: 64", "numEnvs: ${resolve_default:64,${...num_envs}}")
    env = DyrosDynamicWalk(load_task_yaml(text, num_envs=96), "cuda:0", 0, True)
    assert env.num_envs == 96
    obs, rew, done, extras = env.step(torch.zeros(96, 13, device="cuda"))
    torch.cuda.synchronize()
    assert obs["obs"].shape == (96, 487) and torch.isfinite(obs["obs"]).all()


@pytest.mark.gpu
def test_entry_points_do_not_depend_on_the_current_device():
    """The library makes the handle's device current for the call and restores the caller's (several GPUs per process,
    or torch switching devices in between).  On a one-GPU box: the current device is unchanged by a step."""
    from hip_backend import make_env
    env = make_env(64)
    before = torch.cuda.current_device()
    env.step(torch.zeros(64, 13, device="cuda"))
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before


@pytest.mark.gpu
def test_reset_idx_with_terrain_curriculum_vs_oracle_on_gpu(task_const):
    """dw_reset_idx (reset_done path) with the terrain curriculum: a moved env goes a level up, a stationary one a level
    down, an unlisted one is untouched -- every reset field bit-identical to the oracle's (the kernel used to
    read the base position from LDS it had not loaded)."""
    from hip_backend import make_env
    tdict = dict(mesh_type="heightfield", curriculum=True, num_rows=4, num_cols=5, border_size=2, max_init_terrain_level=3)
    N = 64
    env = make_env(N, terrain=tdict, seed=4)
    g = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(3):
        env.step(torch.rand(N, 13, generator=g, device="cuda") * 2 - 1)
    # the kernel before this test may have left anything in LDS: run a launch that fills the CU's LDS with other data
    env.terrain_levels[:] = 1
    org = env.terrain_origins.view(4, 5, 3)
    env.env_origins.copy_(org[1, env.terrain_types])
    env.root_states[:, :3] = env.env_origins + torch.tensor([0, 0, 0.93], device="cuda")
    env.root_states[3, 0] += 6.0
    from isaacgymdyros_amd import abi
    abi.es_view(env._buf["env_state"], "target_vel")[...] = torch.tensor([0.4, 0.0], device="cuda")
    abi.es_view(env._buf["env_state"], "epi_len")[...] = 10.0
    env.randomize_buf[:] = 5
    torch.cuda.synchronize()
    ora = _oracle_like(env, task_const)
    ids = torch.tensor([3, 5, 40], device="cuda", dtype=torch.int32)
    env._step_count = 17
    env.reset_idx(ids)
    torch.cuda.synchronize()
    ora.reset_idx([3, 5, 40], None, 17)
    assert int(ora.buf["terrain_levels"][3]) == 2 and int(ora.buf["terrain_levels"][5]) == 0 and int(ora.buf["terrain_levels"][6]) == 1
    for k in ("terrain_levels", "env_origins", "root_states", "dof_state", "env_state", "reset_buf", "progress_buf",
              "randomize_buf", "dof_damping", "dof_armature"):
        assert np.array_equal(ora.buf[k], env._buf[k].cpu().numpy()), k


@pytest.mark.gpu
def test_captured_step_replays_with_advancing_noise(wave_build):
    """dw_step_dev keeps the step counter in device memory: one captured launch, replayed 8 times, equals 8 eager dw_step
    calls bit for bit (in-kernel Philox draws included -- a captured dw_step would have replayed the noise of one step)."""
    from hip_backend import make_env
    N = 256
    a = make_env(N, seed=5, debug_wave_build=wave_build)
    b = make_env(N, seed=5, debug_wave_build=wave_build, device_step_counter=True)
    g = torch.Generator(device="cuda").manual_seed(11)
    acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(11)]
    for t in range(3):                                      # eager warm-up of both (module loading, allocator)
        a.step(acts[t]); b.step(acts[t])
    torch.cuda.synchronize()
    assert torch.equal(a._buf["env_state"], b._buf["env_state"]) and int(b._step_dev) == 3
    static_act = acts[3].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            stream = torch.cuda.current_stream().cuda_stream
            rc = b._api["step_dev"](b._h, static_act.data_ptr(), 0, b._step_dev.data_ptr(), stream)
            assert rc == 0
    torch.cuda.synchronize()
    assert int(b._step_dev) == 3                            # capture does not execute
    for t in range(3, 11):
        a.step(acts[t])
        static_act.copy_(acts[t])
        graph.replay()
    torch.cuda.synchronize()
    assert int(b._step_dev) == 11
    for k in ("env_state", "root_states", "dof_state", "obs_buf", "rew_buf", "reset_buf", "obs_history", "dof_damping"):
        assert torch.equal(a._buf[k], b._buf[k]), k
    # A checkpoint taken after replays carries the DEVICE counter (the host's count saw only the 3 eager steps): restored
    # into a fresh env, the next step draws the noise of step 11, not of step 3 again -- and equals the uninterrupted run.
    sd = b.state_dict()
    assert sd["_step_count"] == 11
    c = make_env(N, seed=5, debug_wave_build=wave_build)
    c.load_state_dict(sd)
    ids = torch.tensor([1, 7, 200], device="cuda")
    for e in (a, b, c):
        e.reset_idx(ids)                                    # keyed with the same step index on all three
    extra = torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
    a.step(extra); c.step(extra); b.step(extra)
    torch.cuda.synchronize()
    assert int(b._step_dev) == 12
    for k in ("env_state", "root_states", "dof_state", "obs_buf", "rew_buf", "reset_buf", "obs_history"):
        assert torch.equal(a._buf[k], c._buf[k]), k
        assert torch.equal(a._buf[k], b._buf[k]), k


@pytest.mark.gpu
def test_whole_step_vs_oracle_at_2048_envs_with_dr(task_const, wave_build):
    """a domain-randomised, in-contact rollout against the oracle above fixture size -- 2048 envs, mass scale
    per body, damping / armature per joint, friction per env (all re-drawn at reset by the in-kernel generator, identically in
    the oracle), random actions, 10 policy steps, the tolerances of the 8-env fixture (|dq| <= 1e-4 rad, |dqd| <= 2e-2 rad/s,
    root pose <= 1e-4, rewards 5e-3) for the 99th percentile of the envs, a looser bound for the discrete-event tail, reset flags
    identical up to isolated threshold flips.  The reset-time draws are bit-identical."""
    from hip_backend import make_env
    N = 2048
    env = make_env(N, debug_wave_build=wave_build, friction_dr=True, seed=21)
    rng = np.random.default_rng(3)
    env._buf["friction_scale"].copy_(torch.from_numpy(rng.uniform(0.7, 1.3, size=N).astype(np.float32)).cuda())
    env._buf["dof_damping"].copy_(torch.from_numpy((0.1 + rng.uniform(0, 2.9, size=(N, 33))).astype(np.float32)).cuda())
    ora = _oracle_like(env, task_const)
    assert float(ora.buf["mass_scale"].std()) > 0.05 and float(ora.buf["friction_scale"].std()) > 0.1      # DR really on
    g = torch.Generator().manual_seed(8)
    alive = np.ones(N, dtype=bool)
    w = {k: np.zeros(N) for k in ("dq", "dqd", "root", "rew")}            # per env: worst difference over the steps it was compared
    for t in range(10):
        a = torch.rand(N, 13, generator=g) * 2 - 1
        _, rew, done, _ = env.step(a.cuda())
        ora.step(a.numpy(), None, t)
        torch.cuda.synchronize()
        got_reset, ref_reset = env.reset_buf.cpu().numpy(), ora.buf["reset_buf"]
        # the reset decision is a threshold on contact forces / orientation: an env within rounding of it may flip; such flips
        # must stay isolated (a systematic difference would flip hundreds), and a flipped env leaves the comparison
        flip = got_reset != ref_reset
        assert int((flip & alive).sum()) <= 4, (t, int((flip & alive).sum()))
        alive &= ~flip
        cmp = alive & (ref_reset == 0)
        qa, qb = env._buf["dof_state"].cpu().numpy(), ora.buf["dof_state"]
        w["dq"][cmp] = np.maximum(w["dq"][cmp], np.abs(qa[cmp, :, 0] - qb[cmp, :, 0]).max(axis=1))
        w["dqd"][cmp] = np.maximum(w["dqd"][cmp], np.abs(qa[cmp, :, 1] - qb[cmp, :, 1]).max(axis=1))
        w["root"][cmp] = np.maximum(w["root"][cmp], np.abs(env.root_states.cpu().numpy()[cmp, :7] - ora.buf["root_states"][cmp, :7]).max(axis=1))
        w["rew"][cmp] = np.maximum(w["rew"][cmp], np.abs(rew.cpu().numpy()[cmp] - ora.buf["rew_buf"][cmp]))
    pct = {k: [float(np.percentile(v, p)) for p in (50, 99, 99.9, 100)] for k, v in w.items()}
    print("2048-env DR rollout, per-env worst over 10 steps, percentiles 50 / 99 / 99.9 / 100:", pct, "never flipped:", int(alive.sum()))
    # The tolerances of the 8-env fixture hold for 99 % of 2048 envs.  The tail is not rounding growth but discrete events: a
    # sole corner that enters the contact set one substep earlier on one side (gap within 1e-7 of the 2 mm contact offset)
    # changes that step's impulses by newtons, and a sole load crossing the 1 N threshold of the contact-phase reward moves the
    # reward by 0.2 -- such envs are bounded, not held to 1e-4.
    assert pct["dq"][1] < 1e-4 and pct["dqd"][1] < 2e-2 and pct["root"][1] < 1e-4, pct
    assert pct["rew"][0] < 1e-3 and pct["rew"][1] < 2e-2, pct          # (rewards: 5e-3 holds for ~98.5 % of the envs; the contact terms are thresholds)
    assert pct["dq"][3] < 1e-2 and pct["root"][3] < 5e-3 and pct["rew"][3] <= 0.45, pct
    assert int(alive.sum()) >= N - 16
    for k in ("dof_damping", "dof_armature", "friction_scale"):          # the reset-time draws themselves: bit-identical
        assert np.array_equal(env._buf[k].cpu().numpy()[alive], ora.buf[k][alive]), k


@pytest.mark.gpu
def test_sliding_sole_contacts_with_friction_dr_vs_oracle(task_const, wave_build):
    """VERDICT r2 3(a), BASELINE config 5's distinguishing physics: robots standing on their soles with a horizontal base
    velocity of 0.3 .. 1 m/s (the soles slide), friction_scale in [0.7, 1.3] per env.  HIP against the oracle after ONE substep: net sole
    forces within 5e-4 relative for the median env, 3e-3 for the 99th percentile, 1e-2 at worst (measured on the MI355X: 2.5e-4 /
    1.7e-3 / 6.1e-3 -- a sliding corner sits ON the cone, where five truncated Gauss-Seidel sweeps in two summation orders
    differ more than for sticking contacts, 1e-3 in test_whole_step_vs_oracle_goldens) -- with the tangential force on the cone
    of the env's own mu; and q within 1e-4 rad after 10 policy steps."""
    from hip_backend import make_env
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    N = 512
    env = make_env(N, debug_wave_build=wave_build, randomize=False, seed=9)
    rng = np.random.default_rng(12)
    mu = rng.uniform(0.7, 1.3, size=N).astype(np.float32)
    env._buf["friction_scale"].copy_(torch.from_numpy(mu).cuda())
    root = env.root_states.cpu().numpy().copy()
    root[:, 2] = 0.9285                                      # soles ~0.1 mm inside the plane
    ang = rng.uniform(0, 2 * np.pi, size=N)
    spd = rng.uniform(0.3, 1.0, size=N)
    root[:, 7], root[:, 8] = spd * np.cos(ang), spd * np.sin(ang)
    env.root_states.copy_(torch.from_numpy(root.astype(np.float32)).cuda())
    dof = env._buf["dof_state"].cpu().numpy().copy()
    dof[:, :, 0] = np.asarray(INITIAL_DOF_POS, dtype=np.float32)
    dof[:, :, 1] = 0.0
    env._buf["dof_state"].copy_(torch.from_numpy(dof).cuda())
    torch.cuda.synchronize()
    ora = _oracle_like(env, task_const)
    # hold the pose with the reference's PD gains so that the soles stay loaded
    tau = np.zeros((N, 33), dtype=np.float32)
    env.simulate(torch.from_numpy(tau).cuda())
    ora.simulate(tau)
    torch.cuda.synchronize()
    cf_g, cf_o = env._buf["contact_forces"].cpu().numpy(), ora.buf["contact_forces"]
    for foot in (env.left_foot_idx, env.right_foot_idx):
        fo, fg = cf_o[:, foot], cf_g[:, foot]
        loaded = fo[:, 2] > 50.0
        assert loaded.sum() > N // 2
        rel = np.abs(fg[loaded] - fo[loaded]).max(axis=1) / np.linalg.norm(fo[loaded], axis=1)
        print("sliding soles, body %d: relative force difference percentiles 50 / 99 / 100: %.2e %.2e %.2e (n = %d)" % (
            foot, np.percentile(rel, 50), np.percentile(rel, 99), rel.max(), int(loaded.sum())))
        assert np.percentile(rel, 50) < 5e-4 and np.percentile(rel, 99) < 3e-3 and rel.max() < 1e-2, (foot, rel.max())
        # sliding: the tangential force sits on the cone, |Ft| = mu |Fn|, per env
        ft = np.linalg.norm(fo[loaded, :2], axis=1)
        ratio = ft / fo[loaded, 2]
        on_cone = ratio > 0.95 * mu[loaded]
        assert on_cone.mean() > 0.5 and np.all(ratio <= mu[loaded] * 1.02 + 1e-3)
    g = torch.Generator().manual_seed(4)
    for t in range(10):
        a = torch.rand(N, 13, generator=g) * 0.2 - 0.1
        env.step(a.cuda())
        ora.step(a.numpy(), None, t)
    torch.cuda.synchronize()
    same = env.reset_buf.cpu().numpy() == ora.buf["reset_buf"]
    assert same.mean() > 0.99
    keep = same & (ora.buf["progress_buf"] == 10)                    # envs that went through all 10 steps on both sides
    assert keep.sum() > N // 2
    dq = np.abs(env.dof_pos.cpu().numpy()[keep] - ora.buf["dof_state"][keep, :, 0]).max()
    assert dq < 1e-4, dq


def test_physics_only_handle_as_the_engine_behind_the_references_own_task(task_const):
    """INTEGRATION.md section 2 (`Mi355Gym`): dw_create(task = NULL) + a bind of the physics tensors alone is the engine the
    reference's own task file would drive -- set_dof_actuation_force_tensor + simulate + refresh_* = dw_simulate.  The partial
    handle tracks the physics of a full one on the same state and torques, and refuses the task entry points."""
    import ctypes as C
    from isaacgymdyros_amd import _lib, abi
    from isaacgymdyros_amd.model import load_model
    from hip_backend import make_env
    N = 64
    full = make_env(N)
    lib, api = _lib.load()
    cfg = abi.DwConfig.from_buffer_copy(full._ccfg)             # (the same contact-model knobs as the full handle)
    cm = load_model().to_c()
    h = C.c_void_p()
    assert api["create"](C.byref(cfg), C.byref(cm), None, C.byref(h)) == 0, lib.dw_last_error()
    names = ("root_states", "dof_state", "contact_forces", "mass_scale", "dof_damping", "dof_armature", "friction_scale")
    buf = {k: full._buf[k].clone() for k in names}
    db = abi.DwBuffers()
    for k in names:
        setattr(db, k, buf[k].data_ptr())
    assert api["bind"](h, C.byref(db)) == 0, lib.dw_last_error()
    g = torch.Generator(device="cuda").manual_seed(9)
    for t in range(20):
        tau = (torch.rand(N, 33, generator=g, device="cuda") * 2 - 1) * 30
        assert api["simulate"](h, tau.data_ptr(), None, None) == 0, lib.dw_last_error()
        full.simulate(tau)
    torch.cuda.synchronize()
    # (not bit for bit: the contact solver warm-starts from the impulses it keeps in the task record, env_state, which a
    #  physics-only handle does not have -- it starts every solve from zero and converges to the same contact within the tolerance)
    assert float((buf["root_states"][:, :3] - full._buf["root_states"][:, :3]).abs().max()) < 2e-3
    assert float((buf["dof_state"][..., 0] - full._buf["dof_state"][..., 0]).abs().max()) < 5e-3
    fz_a, fz_b = buf["contact_forces"][:, [8, 16], 2].sum(1), full._buf["contact_forces"][:, [8, 16], 2].sum(1)
    assert float((fz_a - fz_b).abs().mean()) < 0.05 * float(fz_b.abs().mean()) + 1.0
    assert float(buf["contact_forces"].abs().max()) > 10.0            # the feet did touch down in those 20 substeps
    # no task constants, no task buffers: the task entry points refuse
    a = torch.zeros(N, 13, device="cuda")
    assert api["step"](h, a.data_ptr(), None, 0, None) != 0 and b"dw_step" in lib.dw_last_error()
    ids = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert api["reset_idx"](h, ids.data_ptr(), 1, None, 0, None) != 0
    assert api["destroy"](h) == 0
    full.close()


def test_step_returns_a_fresh_observation_tensor_without_a_copy():
    """The reference's contract (`torch.clamp(self.obs_buf, ...)`, tasks/base/vec_task.py:338: a new tensor every step) met by
    letting the kernel write each step's observations into a newly allocated tensor (dw_step_obs): tensors returned by earlier
    steps stay what they were, the numbers equal those of the bound-buffer path (alias_obs), env.obs_buf is the newest
    observation, and reset_idx leaves it alone (the reference rebuilds a reset env's observations in the next step)."""
    from hip_backend import make_env
    N = 200
    a_env, f_env = make_env(N, alias_obs=True), make_env(N)
    assert f_env._fresh_obs and not a_env._fresh_obs
    g = torch.Generator(device="cuda").manual_seed(12)
    kept = []
    for t in range(12):
        a = torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
        oa = a_env.step(a)[0]["obs"]
        of = f_env.step(a)[0]["obs"]
        assert torch.equal(oa, of), t                              # same kernel, same numbers
        assert of is f_env.obs_buf and of.data_ptr() != f_env._bound_obs.data_ptr()
        kept.append((of, of.clone()))
    assert len({o.data_ptr() for o, _ in kept}) == len(kept)      # twelve live tensors, twelve storages
    for o, c in kept:
        assert torch.equal(o, c)                                   # later steps did not touch what was returned earlier
    held = f_env.obs_buf
    snap = held.clone()
    ids = torch.tensor([3, 77, 150], device="cuda")
    f_env.reset_idx(ids)
    a_env.reset_idx(ids)
    torch.cuda.synchronize()
    assert torch.equal(held, snap) and f_env.obs_buf is held              # reset_idx does not write observations ...
    assert torch.equal(f_env.obs_buf, a_env.obs_buf)                      # ... on either path
    for t in range(3):                                                    # and the steps after the reset agree again, reset envs included
        a = torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
        assert torch.equal(a_env.step(a)[0]["obs"], f_env.step(a)[0]["obs"]), t
    sd = f_env.state_dict()
    assert torch.equal(sd["obs_buf"], f_env.obs_buf)
    a_env.close(); f_env.close()
