"""Row f-3: the CPU oracle of the sibling TOCABI tasks' env-side functions (oracle/dw_amp.c) against the reference's own
TorchScript functions (fixture tests/golden/amp_lower_ref.npz, minted by oracle/make_amp_goldens.py from
tasks/amp/tocabi_amp_lower_base.py:918-1069 and tasks/tocabi_new_walk.py:384-496)."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests import amp_calls

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "amp_lower_ref.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(G)


def oracle_outputs(g, early=True):
    lib, api = oracle.load()

    def chk(rc):
        assert rc == 0, lib.dwo_last_error()
    return amp_calls.run_all(api, g, lambda a: np.ascontiguousarray(a), lambda s, d: np.zeros(s, d), chk, early)


def ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_amp_observations_bitwise(g):
    o = oracle_outputs(g)
    d = ulps(o["obs"], g["ref_obs"])
    # 35 of the 36 entries are the reference's bit patterns (roll and yaw, the rotated velocity with torch.cross's contracted
    # product, the sums with the biases).  The pitch angle is atan2(-m20, cy): torch's CPU kernel is SLEEF's vectorised
    # atan2f (1.0 ulp), glibc's atan2f rounds differently in one case of five
    cols = [c for c in range(36) if c != 1]
    assert d[:, cols].max() == 0, np.argwhere(d[:, cols] > 0)[:5]
    assert d[:, 1].max() <= 2 and (d[:, 1] == 0).mean() > 0.7


def test_amp_reward_terms(g):
    o = oracle_outputs(g)
    # torch's vectorised exp (SLEEF) and glibc's expf may differ in the last places: <= 2 ulp per term, rarely
    d = ulps(o["reward_values"], g["ref_reward_values"])
    assert d.max() <= 2, (d.max(), np.argwhere(d > 2)[:5])
    assert (d == 0).mean() > 0.98
    assert np.abs(o["reward"] - g["ref_reward"]).max() <= 6e-8
    # the branch structure: which envs are over the contact-force threshold
    assert np.array_equal(o["reward_values"][:, 3], g["ref_reward_values"][:, 3])


def test_amp_reset_exact(g):
    for early in (True, False):
        o = oracle_outputs(g, early)
        assert np.array_equal(o["reset"], g["ref_reset_early%d" % early])
        assert np.array_equal(o["terminated"], g["ref_terminated_early%d" % early])
    assert 0.2 < g["ref_terminated_early1"].mean() < 0.95          # the fixture exercises both outcomes


def test_newwalk_reward(g):
    o = oracle_outputs(g)
    assert np.array_equal(o["nw_reset"], g["ref_nw_reset"])
    dead = g["ref_nw_total"] == -1.0
    assert 0.1 < dead.mean() < 0.9
    assert np.array_equal(o["nw_total"] == -1.0, dead)
    # tan / exp of libm against torch's vectorised kernels: a few ulp on the tan terms near +-1
    assert np.abs(o["nw_reward8"] - g["ref_nw_reward8"]).max() <= 5e-7
    assert np.abs(o["nw_total"] - g["ref_nw_total"]).max() <= 2e-7


def test_newwalk_class_cannot_normalise_its_observation(g):
    """tasks/tocabi_new_walk.py:558-567 subtracts a 37-entry mean from a [N,30] observation: torch refuses the shapes, so the class
    never steps and only its reward function is a meaningful parity target."""
    assert "must match" in str(g["nw_broadcast_error"]) and "30" in str(g["nw_broadcast_error"]) and "37" in str(g["nw_broadcast_error"])


def test_oracle_termination_on_the_reference_class_run():
    """The class-level fixture (tests/golden/amp_class_ref.npz: the reference's TocabiAMPLowerBase stepping over the oracle's
    physics) seen from the CPU side: the oracle's termination function, fed the recorded physics state of every step, returns
    the reset and terminate flags the reference class produced."""
    import ctypes as C
    c = np.load(os.path.join(os.path.dirname(G), "amp_class_ref.npz"))
    lib, api = oracle.load()
    N, steps = int(c["num_envs"]), int(c["steps"])
    ids = np.array([8, 16], np.int32)
    seen = 0
    for t in range(steps):
        k = 2 * t + 1                                   # the state after the step's second simulate()
        pos = np.zeros((N, 38, 3), np.float32)
        rot = np.zeros((N, 38, 4), np.float32)
        pos[:, 0] = c["sim_root"][k][:, 0:3]
        pos[:, 8], pos[:, 16] = c["sim_feet"][k][:, 0], c["sim_feet"][k][:, 1]
        rot[:, 0] = c["sim_root"][k][:, 3:7]
        prog = np.ascontiguousarray(c["ref_progress_buf"][t])
        cf = np.ascontiguousarray(c["sim_contact"][k])
        rs, term = np.zeros(N, np.int64), np.zeros(N, np.int64)
        p = lambda a: C.c_void_p(a.ctypes.data)          # noqa: E731
        assert api["amp_reset"](N, p(prog), p(cf), p(ids), 2, p(pos), p(rot), float(c["episode_length"]), 1, 0.6, p(rs), p(term), None) == 0
        assert np.array_equal(rs, c["ref_reset_buf"][t]), t
        assert np.array_equal(term, c["ref__terminate_buf"][t]), t
        seen += int(term.sum())
    assert seen > 0 and int(c["ref_timeout_buf"].sum()) > 0          # falls and time-outs both occur in the run
