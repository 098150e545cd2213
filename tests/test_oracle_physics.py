"""Known-answer tests that pin the oracle's physics (parity against PhysX itself is unpinned: the engine is a
closed binary absent from the reference checkout)."""
import numpy as np
import pytest

from dense_dynamics_np import DenseDynamics, body_twist_from_root
from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS, KP_RAW, KV_RAW
from oracle.oracle import OracleSim

G = np.array([0, 0, -9.81])


def _randomise(sim, rng, z=3.0):
    N = sim.N
    sim.buf["root_states"][:, 0:3] = rng.normal(size=(N, 3)) + np.array([0, 0, z])
    q = rng.normal(size=(N, 4))
    sim.buf["root_states"][:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    sim.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6))
    sim.buf["dof_state"][:, :, 0] = rng.uniform(-1, 1, size=(N, 33))
    sim.buf["dof_state"][:, :, 1] = rng.uniform(-2, 2, size=(N, 33))


@pytest.mark.parametrize("double,tol", [(True, 5e-8), (False, 5e-6)])
def test_aba_matches_dense_jacobian_formulation(model, double, tol):
    """Recursive body-frame ABA (oracle) vs world-frame M = sum J'IJ + dense solve (numpy fp64), random states and
    randomised mass / damping / armature."""
    rng = np.random.default_rng(0)
    sim = OracleSim(6, double=double, self_collision=0)      # random joint angles interpenetrate the legs
    _randomise(sim, rng)
    sim.buf["mass_scale"][:] = rng.uniform(0.8, 1.2, size=(6, 38))
    sim.buf["dof_damping"][:] = 0.1 + rng.uniform(0, 2.9, size=(6, 33))
    sim.buf["dof_armature"][:] *= rng.uniform(0.8, 1.2, size=(6, 33)).astype(np.float32)
    dd = DenseDynamics(model)
    for e in range(6):
        tau = rng.uniform(-100, 100, size=33).astype(np.float32)
        qdd, a0 = sim.forward_dynamics(e, tau)
        root = sim.buf["root_states"][e].astype(float)
        nu_b = body_twist_from_root(root, model.inert_com[0], True)
        a0r, qddr, _ = dd.accelerations(
            root[:3], root[3:7], sim.buf["dof_state"][e, :, 0].astype(float), nu_b,
            sim.buf["dof_state"][e, :, 1].astype(float), sim.buf["mass_scale"][e].astype(float),
            sim.buf["dof_armature"][e].astype(float), sim.buf["dof_damping"][e].astype(float),
            0.002, np.array([0, 0, float(np.float32(-9.81))]), tau.astype(float))
        assert np.abs(qdd - qddr).max() <= tol * np.abs(qddr).max()
        assert np.abs(a0 - a0r).max() <= tol * np.abs(a0r).max()


def test_free_fall_is_rigid(model):
    """Uniform gravity exerts no joint torque: with qd = 0 and tau = 0 the robot falls as one rigid body."""
    sim = OracleSim(2)
    sim.buf["root_states"][:, 2] = 5.0
    sim.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS, np.float32)
    q0 = sim.buf["dof_state"][:, :, 0].copy()
    n = 100
    for _ in range(n):
        sim.simulate(np.zeros((2, 33), np.float32))
    assert np.abs(sim.buf["dof_state"][:, :, 0] - q0).max() < 2e-5
    assert np.allclose(sim.buf["root_states"][:, 9], -9.81 * 0.002 * n, rtol=1e-4)
    assert np.abs(sim.buf["root_states"][:, 10:13]).max() < 1e-4
    assert np.abs(sim.buf["contact_forces"]).max() == 0


def test_momentum_and_energy_in_flight(model):
    """No gravity, no contact, no joint damping, zero torque: linear and angular momentum about the world origin
    are invariants of the exact dynamics and total energy too; semi-implicit Euler keeps them to O(dt)."""
    rng = np.random.default_rng(3)
    sim = OracleSim(1, double=True, gravity=(0.0, 0.0, 0.0), self_collision=0)
    _randomise(sim, rng, z=10.0)
    sim.buf["root_states"][:, 7:13] *= 0.5
    sim.buf["dof_state"][:, :, 1] *= 0.25
    sim.buf["dof_damping"][:] = 0.0
    dd = DenseDynamics(model)

    def invariants():
        root = sim.buf["root_states"][0].astype(float)
        nu_b = body_twist_from_root(root, model.inert_com[0], True)
        return dd.energy_momentum(root[:3], root[3:7], sim.buf["dof_state"][0, :, 0].astype(float), nu_b,
                                  sim.buf["dof_state"][0, :, 1].astype(float), np.ones(38),
                                  sim.buf["dof_armature"][0].astype(float), np.zeros(3))
    ke0, _, mom0 = invariants()
    for _ in range(50):
        sim.simulate(np.zeros((1, 33), np.float32))
    ke1, _, mom1 = invariants()
    assert np.abs(mom1[3:] - mom0[3:]).max() < 1e-3 * np.abs(mom0[3:]).max()      # linear momentum
    # armature (reflected rotor inertia) is not part of the body momenta, so angular momentum is only
    # approximately carried by the links; energy includes the rotor term and is the sharper check
    assert abs(ke1 - ke0) < 2e-2 * ke0


def test_static_stance_carries_the_weight(model):
    """Full-strength PD to the initial pose on flat ground: the two soles carry m*g, nothing else touches."""
    sim = OracleSim(1)
    kp, kv = np.asarray(KP_RAW), np.asarray(KV_RAW)
    q0 = np.asarray(INITIAL_DOF_POS)
    sim.buf["dof_state"][:, :, 0] = q0
    fz = []
    for i in range(1000):
        q, qd = sim.buf["dof_state"][:, :, 0], sim.buf["dof_state"][:, :, 1]
        sim.simulate((kp * (q0 - q) - kv * qd).astype(np.float32))
        if i >= 500:
            cf = sim.buf["contact_forces"][0]
            fz.append(cf[model.left_foot_idx, 2] + cf[model.right_foot_idx, 2])
            nonfoot = np.delete(cf, [model.left_foot_idx, model.right_foot_idx], axis=0)
            assert np.abs(nonfoot).max() == 0
    mg = model.nominal_total_mass * 9.81
    assert abs(np.mean(fz) - mg) < 0.02 * mg
    assert 0.90 < sim.buf["root_states"][0, 2] < 0.94
    assert abs(sim.buf["root_states"][0, 6]) > 0.999


def test_non_foot_contact_is_reported(model):
    """A pelvis below the ground plane produces a net contact force on a non-foot body (termination input)."""
    sim = OracleSim(1)
    sim.buf["root_states"][0, 2] = -0.02                # pelvis box/cylinders reach 5 mm below the body origin
    sim.buf["dof_state"][:, :, 0] = 0
    sim.simulate(np.zeros((1, 33), np.float32))
    cf = sim.buf["contact_forces"][0]
    assert cf[0, 2] > 1.0
    nonfoot = np.delete(cf, [model.left_foot_idx, model.right_foot_idx], axis=0)
    assert np.linalg.norm(nonfoot, axis=1).max() > 1.0


def test_joint_velocity_clamp_and_limits(model):
    sim = OracleSim(1)
    sim.buf["root_states"][0, 2] = 5.0
    tau = np.zeros((1, 33), np.float32)
    tau[0, 3] = 5000.0
    for _ in range(20):
        sim.simulate(tau)
    assert sim.buf["dof_state"][0, 3, 1] == pytest.approx(4.03, abs=1e-6)
    sim.buf["dof_state"][0, 12, 0] = 2.0935
    sim.buf["dof_state"][0, 12, 1] = 4.0
    sim.simulate(np.zeros((1, 33), np.float32))
    assert sim.buf["dof_state"][0, 12, 0] <= 2.094 + 1e-6


def test_fp32_vs_fp64_divergence_budget(model):
    """Same inputs through the fp32 and fp64 builds: |dq| <= 1e-4 rad after 100 contact-free substeps."""
    rng = np.random.default_rng(5)
    a, b = OracleSim(4, self_collision=0), OracleSim(4, double=True, self_collision=0)
    _randomise(a, rng)
    a.buf["dof_state"][:, :, 1] *= 0.5
    for k in ("root_states", "dof_state"):
        b.buf[k][:] = a.buf[k]
    tau = rng.uniform(-30, 30, size=(4, 33)).astype(np.float32)
    for _ in range(100):
        a.simulate(tau)
        b.simulate(tau)
    assert np.abs(a.buf["dof_state"][:, :, 0] - b.buf["dof_state"][:, :, 0]).max() < 1e-4


def test_crossed_legs_report_self_collision(model):
    """the reference collides every primitive with every other (filter 0) and ends the episode on any
    non-foot contact above 1 N.  Leg-vs-leg capsule proxies: nothing at the rest pose, both legs' links loaded with
    equal and opposite forces once the hips roll inwards."""
    sim = OracleSim(2)
    sim.buf["root_states"][:, 2] = 3.0
    sim.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS, np.float32)
    sim.buf["dof_state"][1, 1, 0] = -0.2
    sim.buf["dof_state"][1, 7, 0] = 0.2
    sim.simulate(np.zeros((2, 33), np.float32))
    cf = sim.buf["contact_forces"]
    assert np.abs(cf[0]).max() == 0
    hit = np.linalg.norm(cf[1], axis=1) > 1.0
    assert hit[[3, 4, 11, 12]].all()                       # thighs and shanks of both legs
    assert not hit[[model.left_foot_idx, model.right_foot_idx]].any()
    assert np.abs(cf[1].sum(axis=0)).max() < 1e-2          # internal forces cancel
    off = OracleSim(1, self_collision=0)
    off.buf["root_states"][:, 2] = 3.0
    off.buf["dof_state"][:, :, 0] = sim.buf["dof_state"][1:2, :, 0] * 0 + np.asarray(INITIAL_DOF_POS, np.float32)
    off.buf["dof_state"][0, 1, 0] = -0.2
    off.buf["dof_state"][0, 7, 0] = 0.2
    off.simulate(np.zeros((1, 33), np.float32))
    assert np.abs(off.buf["contact_forces"]).max() == 0


_ARMS_IN = {}


def _arms_in(N):
    """Arm poses INSIDE the joint limits that press upper arm, forearm or hand into the torso or a thigh (a few reach the
    other arm): seeded excursions of up to 1.2 rad on all 16 arm joints, the right arm 0.8 as far as the left so that no two
    capsule axes are parallel; three quarters of the returned envs collide, the rest are clear."""
    if N in _ARMS_IN:
        return _ARMS_IN[N].copy()
    from isaacgymdyros_amd.model import load_model
    m = load_model()
    lo, hi = np.asarray(m.d["dof_lower"], np.float32), np.asarray(m.d["dof_upper"], np.float32)
    pool = 32 * N
    rng = np.random.default_rng(17)
    q = np.tile(np.asarray(INITIAL_DOF_POS, np.float32), (pool, 1))
    d = rng.uniform(-1.2, 1.2, size=(pool, 8)).astype(np.float32)
    q[:, 15:23] += d
    q[:, 25:33] -= 0.8 * d * rng.choice([1.0, -0.5], size=(pool, 1)).astype(np.float32)
    q = np.clip(q, lo, hi)
    sim = OracleSim(pool)
    sim.buf["root_states"][:, 2] = 3.0
    sim.buf["dof_state"][:, :, 0] = q
    sim.simulate(np.zeros((pool, 33), np.float32))
    hit = (np.linalg.norm(sim.buf["contact_forces"], axis=2) > 1.0).any(axis=1)
    nh = (3 * N) // 4
    sel = np.concatenate([np.nonzero(hit)[0][:nh], np.nonzero(~hit)[0][:N - nh]])
    assert len(sel) == N
    _ARMS_IN[N] = q[np.sort(sel)]
    return _ARMS_IN[N].copy()


def _brute_capsule_gap(a0, a1, b0, b1, n=400):
    """Distance between two segments by exhaustive sampling, refined twice: independent of the closed-form rule."""
    sa = np.linspace(0, 1, n)[:, None, None]
    sb = np.linspace(0, 1, n)[None, :, None]
    lo_a, hi_a, lo_b, hi_b = 0.0, 1.0, 0.0, 1.0
    for _ in range(3):
        sa = np.linspace(lo_a, hi_a, n)
        sb = np.linspace(lo_b, hi_b, n)
        d = np.linalg.norm((a0 + sa[:, None, None] * (a1 - a0)) - (b0 + sb[None, :, None] * (b1 - b0)), axis=2)
        i, k = np.unravel_index(d.argmin(), d.shape)
        wa, wb = (hi_a - lo_a) * 2 / n, (hi_b - lo_b) * 2 / n
        lo_a, hi_a = max(0.0, sa[i] - wa), min(1.0, sa[i] + wa)
        lo_b, hi_b = max(0.0, sb[k] - wb), min(1.0, sb[k] + wb)
    return d.min(), a0 + sa[i] * (a1 - a0), b0 + sb[k] * (b1 - b0)


def test_arm_into_torso_known_answer(model):
    """Second tranche of row f-1 (forearm / hand against torso and thigh, arm against arm).  At rest in flight the first
    substep sees zero velocities, so each touching pair loads its two bodies with exactly k * depth along the line between
    the closest points -- recomputed here from the compiled proxies with numpy kinematics and brute-force distances."""
    from isaacgymdyros_amd import abi
    N = 12
    sim = OracleSim(N, double=True)
    sim.buf["root_states"][:, 0:2] = 0
    sim.buf["root_states"][:, 2] = 3.0
    sim.buf["dof_state"][:, :, 0] = _arms_in(N)
    q = sim.buf["dof_state"][:, :, 0].astype(float).copy()
    sim.simulate(np.zeros((N, 33), np.float32))
    cf = sim.buf["contact_forces"].astype(float)
    k = float(sim.cfg.penalty_stiffness)
    dd = DenseDynamics(model)
    prox, pairs = model.d["sc_proxies"], model.d["sc_pairs"]
    touching = 0
    for e in range(N):
        Rw, pw = dd.kinematics([0, 0, 3.0], [0, 0, 0, 1], q[e])
        want = np.zeros_like(cf[e])
        for ia, ib in pairs:
            A, B = prox[ia], prox[ib]
            a0, a1 = (pw[A["moving"]] + Rw[A["moving"]] @ np.array(A[p]) for p in ("p0", "p1"))
            b0, b1 = (pw[B["moving"]] + Rw[B["moving"]] @ np.array(B[p]) for p in ("p0", "p1"))
            gap, ca, cb = _brute_capsule_gap(a0, a1, b0, b1)
            depth = A["radius"] + B["radius"] - gap
            if depth > 0:
                touching += 1
                want[A["gym"]] += k * depth * (ca - cb) / gap
                want[B["gym"]] -= k * depth * (ca - cb) / gap
        assert np.abs(cf[e] - want).max() <= 2e-3 * max(1.0, np.abs(want).max()), e
        assert np.abs(cf[e].sum(axis=0)).max() < 1e-6 * max(1.0, np.abs(cf[e]).max())
    assert touching >= (3 * N) // 4                         # the poses do reach the torso


def test_reset_poses_are_clear_of_self_collision(model):
    """The proxies must not fire where the reference's robot stands freely: nothing at all at the rest pose and +-0.03 rad
    around it (wider than any reset draw); at +-0.1 rad the feet may brush each other (first tranche, real geometry), but no
    arm or torso proxy reports anything."""
    rng = np.random.default_rng(5)
    N = 128
    for amp in (0.03, 0.1):
        sim = OracleSim(N)
        sim.buf["root_states"][:, 2] = 3.0
        q = np.tile(np.asarray(INITIAL_DOF_POS, np.float32), (N, 1))
        q[1:] += rng.uniform(-amp, amp, size=(N - 1, 33)).astype(np.float32)
        sim.buf["dof_state"][:, :, 0] = q
        sim.simulate(np.zeros((N, 33), np.float32))
        cf = sim.buf["contact_forces"]
        assert np.abs(cf[:, 17:]).max() == 0, amp
        if amp == 0.03:
            assert np.abs(cf).max() == 0


# ------------------------------------------------------------------------------------------------ pendulum closed forms
# The oracle is built for the TOCABI tree, so the pendulums are cut out of it: every joint but the one or two under test is locked by an
# armature of 1e12 kg m^2 (qdd = u / D -> 0), and the base is made an inertial frame by a mass scale of 1e9 (measured: the closed forms
# are then met to 1e-7; with 1e7 / 1e9 the base's reaction shows at 4e-6).  Parameters of the closed forms come straight from the model JSON (masses, centres of mass, inertias, joint offsets), by the
# parallel-axis theorem -- none of the oracle's own code.
L_KNEE, L_ANKLE_PITCH = 3, 4          # dof indices (moving bodies 4, 5): both hinges about +y, so the shank and the foot assembly move in the x-z plane


def _ry2(th, v):
    """Rotation about +y by th of a vector (x, z) of the x-z plane."""
    x, z = v
    return np.array([x * np.cos(th) + z * np.sin(th), -x * np.sin(th) + z * np.cos(th)])


def _planar_link(model, mvs, origin_mv):
    """Mass, centre of mass (x, z) and I_yy about that centre of the rigid assembly of moving bodies `mvs` (their joints at 0), in the frame
    of moving body origin_mv.  The leg's frames are parallel at q = 0 (mv_rot0 = identity there)."""
    off = {origin_mv: np.zeros(3)}
    for mv in sorted(mvs):
        if mv != origin_mv:
            assert np.allclose(np.array(model.mv_rot0[mv]), np.eye(3))
            off[mv] = off[model.mv_parent[mv]] + np.array(model.mv_pos[mv])
    m, mc, recs = 0.0, np.zeros(3), []
    for i, mv in enumerate(model.inert_mv):
        if mv in mvs:
            c = off[mv] + np.array(model.inert_com[i])
            recs.append((model.inert_mass[i], c, np.array(model.inert_I[i])))
            m += model.inert_mass[i]; mc += model.inert_mass[i] * c
    com = mc / m
    Iyy = sum(I[1][1] + mi * ((c[0] - com[0]) ** 2 + (c[2] - com[2]) ** 2) for mi, c, I in recs)
    return m, np.array([com[0], com[2]]), Iyy


BASE_SCALE = 1e9


def _cut_out(sim, free_dofs, base_scale=BASE_SCALE):
    sim.buf["dof_armature"][:] = 1e12
    for d in free_dofs:
        sim.buf["dof_armature"][:, d] = 0.02          # (a value of its own: armature is part of the closed form's inertia)
    sim.buf["dof_damping"][:] = 0.0
    sim.buf["mass_scale"][:, sim.model.inert_gym[0]] = base_scale
    sim.buf["root_states"][:, 2] = 5.0


def test_single_pendulum_period_is_the_elliptic_integral(model):
    """The left shank with its locked foot assembly, hanging from the knee of a base that is pushed with F = M g_eff (gravity off): in the base's
    frame a physical pendulum in a field g_eff along -x.  Closed form: T = 4 sqrt(I / (m g d)) K(sin^2(A / 2)), I = sum of I_yy + m r^2
    about the knee axis + armature.  Checks joint-space inertia incl. armature, the gravity-equivalent generalised force and the integrator
    (semi-implicit Euler's period error is O(dt^2))."""
    from scipy.special import ellipk
    sim = OracleSim(1, double=True, gravity=(0.0, 0.0, 0.0), self_collision=0)
    _cut_out(sim, [L_KNEE])
    m, com, Icom = _planar_link(model, {4, 5, 6}, 4)
    d = float(np.hypot(*com))
    I = Icom + m * d * d + 0.02
    g_eff = 4.0
    M_total = float(sum(mi * (BASE_SCALE if model.inert_gym[i] == model.inert_gym[0] else 1.0) for i, mi in enumerate(model.inert_mass)))
    push = np.array([[M_total * g_eff, 0.0]])
    # potential V = m g x_w(theta), x_w = X cos + Z sin: stable where the centre of mass points along -x
    th_eq = float(np.arctan2(-com[1], -com[0]))          # x_w = d cos(theta - atan2(Z, X)) minimal at theta = atan2(Z, X) + pi
    th_eq = (th_eq + np.pi) % (2 * np.pi) - np.pi
    A = 0.6
    sim.buf["dof_state"][0, L_KNEE, 0] = th_eq + A
    T_closed = 4.0 * np.sqrt(I / (m * g_eff * d)) * ellipk(np.sin(A / 2.0) ** 2)
    dt, ts, th = 0.002, [], []
    n = int(2.6 * T_closed / dt)
    tau = np.zeros((1, 33), np.float32)
    for k in range(n):
        sim.simulate(tau, push)
        ts.append((k + 1) * dt); th.append(float(sim.buf["dof_state"][0, L_KNEE, 0]) - th_eq)
    th = np.array(th)
    # downward zero crossings, linearly interpolated: two of them are one period apart
    idx = [k for k in range(1, n) if th[k - 1] > 0 >= th[k]]
    assert len(idx) >= 2
    cross = [ts[k - 1] + (ts[k] - ts[k - 1]) * th[k - 1] / (th[k - 1] - th[k]) for k in idx[:2]]
    T_sim = cross[1] - cross[0]
    assert abs(T_sim - T_closed) <= 2e-3 * T_closed, (T_sim, T_closed)
    assert abs(th.max() - A) <= 2e-3 and abs(th.min() + A) <= 5e-3          # (energy: the amplitude is kept over 2.6 periods)
    assert np.abs(np.delete(sim.buf["dof_state"][0, :, 0], L_KNEE)).max() < 1e-6          # the locked joints stay locked
    # the small-angle limit of the same formula, T0 = 2 pi sqrt(I / (m g d)), from a 0.02 rad swing
    sim2 = OracleSim(1, double=True, gravity=(0.0, 0.0, 0.0), self_collision=0)
    _cut_out(sim2, [L_KNEE])
    sim2.buf["dof_state"][0, L_KNEE, 0] = th_eq + 0.02
    th2 = []
    T0 = 2 * np.pi * np.sqrt(I / (m * g_eff * d))
    for k in range(int(1.6 * T0 / dt)):
        sim2.simulate(tau, push)
        th2.append(float(sim2.buf["dof_state"][0, L_KNEE, 0]) - th_eq)
    i2 = [k for k in range(1, len(th2)) if th2[k - 1] > 0 >= th2[k]] + [k for k in range(1, len(th2)) if th2[k - 1] < 0 <= th2[k]]
    i2.sort()
    c2 = [(k - 1 + th2[k - 1] / (th2[k - 1] - th2[k])) * dt for k in i2[:2]]
    assert abs(2.0 * (c2[1] - c2[0]) - T0) <= 2e-3 * T0


def test_double_pendulum_accelerations_match_the_closed_form(model):
    """Shank (knee) + foot assembly (ankle pitch): the compound planar double pendulum.  With absolute angles a = q1, b = q1 + q2, r = the ankle
    in the shank's frame, c1 / c2 the centres of mass and h(q2) = r . R(q2) c2:
        M_aa = I1 + m1 |c1|^2 + m2 |r|^2,  M_bb = I2 + m2 |c2|^2,  M_ab = m2 h,
        M_aa a'' + M_ab b'' + m2 h' b'^2 = tau1 - tau2,   M_ab a'' + M_bb b'' - m2 h' a'^2 = tau2
    (Lagrange, no gravity), plus armature and the implicit joint damping of DESIGN.md section 3 on the relative coordinates.  Against the oracle's
    forward dynamics at random states and torques: mass matrix, centrifugal coupling, armature and damping in one go."""
    rng = np.random.default_rng(5)
    sim = OracleSim(4, double=True, gravity=(0.0, 0.0, 0.0), self_collision=0)
    _cut_out(sim, [L_KNEE, L_ANKLE_PITCH])
    m1, c1, I1 = _planar_link(model, {4}, 4)
    m2, c2, I2 = _planar_link(model, {5, 6}, 5)
    r = np.array([model.mv_pos[5][0], model.mv_pos[5][2]])
    damp = np.array([0.7, 1.9])
    sim.buf["dof_damping"][:, L_KNEE], sim.buf["dof_damping"][:, L_ANKLE_PITCH] = damp
    dt = 0.002
    for e in range(4):
        q = rng.uniform(-1.5, 1.5, size=2); qd = rng.uniform(-3, 3, size=2); tau2 = rng.uniform(-30, 30, size=2)
        sim.buf["dof_state"][e, [L_KNEE, L_ANKLE_PITCH], 0] = q
        sim.buf["dof_state"][e, [L_KNEE, L_ANKLE_PITCH], 1] = qd
        tau = np.zeros(33, np.float32); tau[[L_KNEE, L_ANKLE_PITCH]] = tau2
        qdd, a0 = sim.forward_dynamics(e, tau)
        # (what the oracle was handed: the buffers are float32)
        tau2 = tau[[L_KNEE, L_ANKLE_PITCH]].astype(float)
        q = sim.buf["dof_state"][e, [L_KNEE, L_ANKLE_PITCH], 0].astype(float); qd = sim.buf["dof_state"][e, [L_KNEE, L_ANKLE_PITCH], 1].astype(float)
        damp = sim.buf["dof_damping"][e, [L_KNEE, L_ANKLE_PITCH]].astype(float); arm = sim.buf["dof_armature"][e, [L_KNEE, L_ANKLE_PITCH]].astype(float)
        h = float(r @ _ry2(q[1], c2))
        hp = float(r @ np.array([-c2[0] * np.sin(q[1]) + c2[1] * np.cos(q[1]), -c2[0] * np.cos(q[1]) - c2[1] * np.sin(q[1])]))
        Maa, Mbb, Mab = I1 + m1 * c1 @ c1 + m2 * r @ r, I2 + m2 * c2 @ c2, m2 * h
        ad, bd = qd[0], qd[0] + qd[1]
        ca, cb = m2 * hp * bd * bd, -m2 * hp * ad * ad
        # relative coordinates: q'' = J^-1 (a'', b''), J = [[1, 0], [1, 1]]
        Mrel = np.array([[Maa + 2 * Mab + Mbb, Mab + Mbb], [Mab + Mbb, Mbb]]) + np.diag(arm + dt * damp)
        rhs = tau2 - damp * qd - np.array([ca + cb, cb])
        want = np.linalg.solve(Mrel, rhs)
        got = qdd[[L_KNEE, L_ANKLE_PITCH]]
        assert np.abs(got - want).max() <= 5e-7 * max(1.0, np.abs(want).max()), (e, got, want)
        assert np.abs(np.delete(qdd, [L_KNEE, L_ANKLE_PITCH])).max() < 1e-5 and np.abs(a0).max() < 1e-4          # locked joints, inertial base


def test_double_support_load_agrees_with_the_mocap_force_columns(model):
    """The one reference-held number the contact model can be held to: the mocap table's foot-force targets (columns 34, 35 of
    assets/DeepMimic/processed_data_tocabi_walk.txt, loaded at tasks/dyros_dynamic_walk.py:112, compared with the measured sole loads after
    scaling by total_mass / 104.48 at :917).  Over the double-support rows of the gait (:877-879) they are a standing robot's load per foot;
    the oracle's static stance under PD must carry the same -- plausibility of contact normal force and mass model, not a trajectory pin."""
    from isaacgymdyros_amd.task_constants import load_task_constants
    tc = load_task_constants()
    mocap = np.asarray(tc["mocap"], dtype=np.float64).reshape(-1, 36)
    target = -mocap[:, 34:36]
    # the file's own consistency: in every row the two targets add up to the weight of the 104.48 kg robot; in single support one foot has it
    # all, over the double-support windows (:877-879) it passes from one foot to the other, and at their middle each foot has half
    assert np.abs(target.sum(1) - 104.48 * 9.81).max() <= 0.01 * 104.48 * 9.81
    assert np.abs(target[600:1200] - np.array([0.0, 1025.0])).max() < 1.0 and np.abs(target[2400:3000] - np.array([1025.0, 0.0])).max() < 1.0
    per_foot_ref = float(target[1650:1950].mean())
    assert abs(per_foot_ref - 0.5 * 104.48 * 9.81) <= 0.01 * per_foot_ref and np.abs(target[1650:1950].mean(0) - per_foot_ref).max() < 0.01 * per_foot_ref
    sim = OracleSim(1, double=False, self_collision=0)
    kp, kv = np.asarray(KP_RAW, np.float32), np.asarray(KV_RAW, np.float32)          # (full-strength PD, as test_static_stance_carries_the_weight)
    q0 = np.asarray(INITIAL_DOF_POS, np.float32)
    sim.buf["dof_state"][0, :, 0] = q0
    loads = []
    for k in range(1500):
        q, qd = sim.buf["dof_state"][0, :, 0], sim.buf["dof_state"][0, :, 1]
        sim.simulate((kp * (q0 - q) - kv * qd)[None, :].astype(np.float32))
        if k >= 1000:
            loads.append([sim.buf["contact_forces"][0, model.left_foot_idx, 2], sim.buf["contact_forces"][0, model.right_foot_idx, 2]])
    loads = np.array(loads).mean(0)
    scale = float(sim.buf["total_mass"][0]) / 104.48          # weight_scale of :917
    assert abs(loads.sum() - 2 * scale * per_foot_ref) <= 0.02 * 2 * per_foot_ref, (loads, per_foot_ref)
    assert np.abs(loads - scale * per_foot_ref).max() <= 0.10 * per_foot_ref, (loads, per_foot_ref)          # (a symmetric stance: each foot half, as the table's mid-DSP rows)
