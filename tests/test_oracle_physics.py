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
