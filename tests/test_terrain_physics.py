"""Height-field contact (row f-4): known answers for the oracle, and the kernel body (host emulation) against the
oracle on terrain.  The physics of the reference engine is unpinned (closed binary); these tests pin the written
decision: bilinear ground under every contact point, penetration and forces along the local normal."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emul_backend import EmulSim
from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS, KP_RAW, KV_RAW
from isaacgymdyros_amd.terrain import Terrain, TerrainCfg
from oracle.oracle import OracleSim


class _Field:
    """A hand-made height field with the attributes OracleSim needs."""

    def __init__(self, samples, hscale=0.1, vscale=0.005, border=2.0):
        self.heightsamples = np.ascontiguousarray(samples, dtype=np.int16)
        self.tot_rows, self.tot_cols = self.heightsamples.shape
        self.env_length = 8.0
        self.env_origins = np.zeros((1, 1, 3))
        self.cfg = TerrainCfg(mesh_type="heightfield", horizontal_scale=hscale, vertical_scale=vscale, border_size=border,
                              curriculum=False, num_rows=1, num_cols=1)


def _stand(sim, steps, z0):
    kp, kv = np.asarray(KP_RAW), np.asarray(KV_RAW)
    q0 = np.asarray(INITIAL_DOF_POS)
    if steps > 1:
        sim.buf["dof_state"][:, :, 0] = q0
        sim.buf["root_states"][:, 2] = z0
    for _ in range(steps):
        q, qd = sim.buf["dof_state"][:, :, 0], sim.buf["dof_state"][:, :, 1]
        sim.simulate((kp * (q0 - q) - kv * qd).astype(np.float32))


def test_flat_field_equals_the_plane(model):
    """A height field of zeros must reproduce the ground plane (other code path, same mechanics)."""
    plane, field = OracleSim(1), OracleSim(1, terrain=_Field(np.zeros((60, 60))), terrain_curriculum=0)
    _stand(plane, 300, 0.93)
    _stand(field, 300, 0.93)
    assert np.abs(plane.buf["root_states"] - field.buf["root_states"]).max() < 2e-5
    assert np.abs(plane.buf["dof_state"] - field.buf["dof_state"]).max() < 2e-4
    fz = lambda s: s.buf["contact_forces"][0, [model.left_foot_idx, model.right_foot_idx], 2].sum()
    assert abs(fz(plane) - fz(field)) < 0.01 * fz(plane)


def test_platform_carries_the_weight_at_its_height(model):
    """Standing on a raised platform (0.25 m) that only exists under the robot: the soles carry m g, the base rests
    0.25 m higher than on the plane -- i.e. world (x, y) -> sample (row, col) and the vertical scale are right."""
    hs = np.zeros((60, 60), np.int16)
    hs[12:28, 12:28] = 50                   # rows/cols 12..27 <-> x, y in [-0.8, 0.7] m at border 2 m, 0.1 m spacing; 50 * 0.005 = 0.25 m
    sim = OracleSim(1, terrain=_Field(hs), terrain_curriculum=0)
    _stand(sim, 1000, 0.93 + 0.25)
    cf = sim.buf["contact_forces"][0]
    mg = model.nominal_total_mass * 9.81
    assert abs(cf[model.left_foot_idx, 2] + cf[model.right_foot_idx, 2] - mg) < 0.03 * mg
    assert 0.90 + 0.25 < sim.buf["root_states"][0, 2] < 0.94 + 0.25
    off = OracleSim(1, terrain=_Field(hs), terrain_curriculum=0)      # same pose 3 m away: nothing underneath at that height
    off.buf["root_states"][:, 0] = 3.0
    _stand(off, 20, 0.93 + 0.25)
    assert np.abs(off.buf["contact_forces"]).max() == 0


def test_forces_follow_the_slope_normal(model):
    """A robot set down square to a uniform 3 % slope stays there: the ground reaction balances gravity (vertical in
    total, so it is made of a normal part along (-s, 0, 1)/|.| plus uphill friction), no sliding with mu = 1."""
    rows = np.arange(80).reshape(-1, 1) * np.ones((1, 80))
    s = 0.03
    hs = np.rint(rows * 0.1 * s / 0.005)                   # height = s * (x + border)
    sim = OracleSim(1, terrain=_Field(hs, border=4.0), terrain_curriculum=0)
    th = np.arctan(s)
    sim.buf["root_states"][:, 3:7] = [0, np.sin(-th / 2), 0, np.cos(-th / 2)]          # pitched back by the slope angle
    sim.buf["root_states"][:, 0] = -0.93 * np.sin(th)
    _stand(sim, 1200, 0.93 * np.cos(th) + s * 4.0 + 0.002)
    F = np.zeros(3)
    for _ in range(300):                                                            # average over the residual rocking
        _stand(sim, 1, sim.buf["root_states"][0, 2])
        cf = sim.buf["contact_forces"][0]
        F += (cf[model.left_foot_idx] + cf[model.right_foot_idx]) / 300
    mg = model.nominal_total_mass * 9.81
    assert abs(F[2] - mg) < 0.02 * mg
    assert abs(F[0]) < 0.02 * mg and abs(F[1]) < 0.01 * mg      # no net horizontal force at rest
    assert np.abs(sim.buf["root_states"][0, 7:10]).max() < 0.12  # residual rocking only, not sliding downhill
    assert abs(sim.buf["root_states"][0, 0] + 0.93 * np.sin(th)) < 0.05
    assert -0.05 < sim.buf["root_states"][0, 4] < 0.0            # still pitched back by about the slope angle


def test_kernel_body_matches_oracle_on_generated_terrain():
    """Same inputs through the oracle and through the kernel source (host emulation of either kernel generation) on a
    generated map: random poses near the ground so that sole corners and primitives touch rough terrain."""
    t = Terrain(TerrainCfg(mesh_type="heightfield", curriculum=True, num_rows=2, num_cols=4, border_size=2,
                           terrain_proportions=[0.2, 0.2, 0.3, 0.3, 0.0]), 8, seed=3)
    rng = np.random.default_rng(5)
    N = 12
    A, B = OracleSim(N, terrain=t), EmulSim(N, terrain=t)
    org = t.env_origins.reshape(-1, 3)[rng.integers(0, 8, size=N)]
    A.buf["root_states"][:, 0:2] = org[:, 0:2] + rng.uniform(-3, 3, size=(N, 2))
    ground = t.height_at(A.buf["root_states"][:, 0], A.buf["root_states"][:, 1])
    A.buf["root_states"][:, 2] = ground + 0.93 + rng.uniform(-0.03, 0.05, size=N)
    A.buf["root_states"][:, 6] = 1.0
    A.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6)) * 0.3
    A.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS) + rng.normal(size=(N, 33)) * 0.05
    A.buf["dof_state"][:, :, 1] = rng.normal(size=(N, 33)) * 0.5
    for k in ("root_states", "dof_state"):
        B.buf[k][:] = A.buf[k]
    tau = rng.uniform(-30, 30, size=(N, 33)).astype(np.float32)
    A.simulate(tau); B.simulate(tau)
    cfa, cfb = A.buf["contact_forces"], B.buf["contact_forces"]
    assert np.abs(cfa).max() > 100.0                                   # the terrain is being touched
    assert np.abs(cfa - cfb).max() <= 2e-3 * np.abs(cfa).max() + 0.05
    assert np.abs(A.buf["dof_state"] - B.buf["dof_state"]).max() < 2e-4
    assert np.abs(A.buf["root_states"] - B.buf["root_states"]).max() < 2e-4
    for _ in range(20):
        A.simulate(tau); B.simulate(tau)
    assert np.abs(A.buf["dof_state"][:, :, 0] - B.buf["dof_state"][:, :, 0]).max() < 2e-2
    assert np.isfinite(B.buf["root_states"]).all()


def test_fallen_robots_on_high_rough_terrain_touch_like_the_oracle():
    """The kernels skip the height-field fetches of a body that is higher above the coarse bound of the field around its robot
    (dw_physics.h terrain_bound, a table built at bind) than its bounding radius; the oracle samples under every primitive.  Robots
    lying, kneeling and tumbling on the highest and roughest tiles of a generated map -- torso, arms, knees and head on the ground,
    at heights far from zero and next to steps and slopes -- must report the same contact forces body by body: a bound that is too
    low, or indexed wrongly, would lose contacts here."""
    t = Terrain(TerrainCfg(mesh_type="heightfield", curriculum=True, num_rows=3, num_cols=5, border_size=2,
                           terrain_proportions=[0.1, 0.2, 0.35, 0.25, 0.1]), 15, seed=11)
    rng = np.random.default_rng(2)
    N = 16
    A, B = OracleSim(N, terrain=t), EmulSim(N, terrain=t)
    org = t.env_origins.reshape(-1, 3)
    org = org[np.argsort(-org[:, 2])][:8]                                 # the highest tile origins (top rows of the curriculum)
    pick = org[rng.integers(0, len(org), size=N)]
    A.buf["root_states"][:, 0:2] = pick[:, 0:2] + rng.uniform(-3.5, 3.5, size=(N, 2))
    ground = t.height_at(A.buf["root_states"][:, 0], A.buf["root_states"][:, 1])
    A.buf["root_states"][:, 2] = ground + rng.uniform(0.12, 0.45, size=N)      # base a hand's width to knee height above the field
    ax = rng.normal(size=(N, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0.6, 3.0, size=N)                                    # far from upright
    A.buf["root_states"][:, 3:6] = ax * np.sin(ang / 2)[:, None]
    A.buf["root_states"][:, 6] = np.cos(ang / 2)
    A.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6)) * 0.2
    A.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS) + rng.normal(size=(N, 33)) * 0.3
    A.buf["dof_state"][:, :, 1] = rng.normal(size=(N, 33)) * 0.5
    for k in ("root_states", "dof_state"):
        B.buf[k][:] = A.buf[k]
    assert float(ground.max()) > 0.3 and float(np.abs(ground).max()) > 0.3           # not the plane
    tau = np.zeros((N, 33), np.float32)
    A.simulate(tau); B.simulate(tau)
    cfa, cfb = A.buf["contact_forces"], B.buf["contact_forces"]
    touching = np.linalg.norm(cfa, axis=2) > 1.0
    feet = np.zeros(38, bool); feet[[8, 16]] = True
    assert touching[:, ~feet].sum() >= 2 * N                               # non-sole bodies are on the ground, many of them
    assert np.array_equal(touching, np.linalg.norm(cfb, axis=2) > 1.0) or (touching != (np.linalg.norm(cfb, axis=2) > 1.0)).sum() <= 1
    assert np.abs(cfa - cfb).max() <= 2e-3 * np.abs(cfa).max() + 0.05
    for _ in range(5):
        A.simulate(tau); B.simulate(tau)
    ta, tb = np.linalg.norm(A.buf["contact_forces"], axis=2) > 1.0, np.linalg.norm(B.buf["contact_forces"], axis=2) > 1.0
    assert (ta != tb).sum() <= 3 and ta[:, ~feet].sum() >= N
    assert np.abs(A.buf["root_states"][:, :3] - B.buf["root_states"][:, :3]).max() < 5e-3
