"""SURVEY row f-1 with evidence: the offline reachability sweep of TOCABI's 61 exact collision primitives against the 16
capsule proxies / 47 proxy pairs the kernels collide (tools/reach/).  The reference collides every primitive with every other
(tasks/dyros_dynamic_walk.py:354, filter 0) and ends the episode on any non-foot contact above 1 N (:590).  CPU only, fp64.

What is held here: the distance routine of the sweep against closed forms and brute force; the nominal pose is touch-free; the task's leg
excursions lie inside the swept envelope; a small sweep reproduces its committed result exactly (so a change of the model, the proxies or the
pair list shows up); and the conclusions DESIGN.md section 3 draws from the committed 10^6-sample sweeps (profiles/r06_reach_*.json)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "reach"))
import reach_sweep as rs  # noqa: E402


class Placed(C.Structure):
    _fields_ = [("type", C.c_int), ("c", C.c_double * 3), ("R", C.c_double * 9), ("size", C.c_double * 3), ("brad", C.c_double)]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("reach") / "libreach.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-DREACH_NO_MAIN", "-Wno-unused-result", "-o", so, os.path.join(ROOT, "tools", "reach", "reach_sweep.c"), "-lm"])
    l = C.CDLL(so)
    l.prim_distance.restype = C.c_double
    l.prim_distance.argtypes = [C.POINTER(Placed), C.POINTER(Placed), C.c_double]
    return l


def _placed(kind, c, R, size):
    p = Placed()
    p.type = kind
    p.c[:] = list(c); p.R[:] = list(np.asarray(R, float).ravel()); p.size[:] = list(size)
    return p


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _points(kind, c, R, size, rng, n):
    """Points of the solid: its volume plus (half of them) its surface -- the closest pair of two disjoint solids lies on their surfaces."""
    if kind == 0:
        l = rng.uniform(-1, 1, size=(n, 3)) * np.asarray(size)
        k = rng.integers(0, 3, size=n // 2)
        l[np.arange(n // 2), k] = np.sign(rng.normal(size=n // 2)) * np.asarray(size)[k]
    else:
        th, rr, zz = rng.uniform(0, 2 * np.pi, n), size[0] * np.sqrt(rng.uniform(0, 1, n)), rng.uniform(-size[1], size[1], n)
        rr[: n // 4] = size[0]; zz[n // 4: n // 2] = np.sign(rng.normal(size=n // 2 - n // 4)) * size[1]
        l = np.stack([rr * np.cos(th), rr * np.sin(th), zz], 1)
    return np.asarray(c) + l @ np.asarray(R).T


def test_distance_routine_against_closed_forms_and_brute_force(lib):
    I = np.eye(3)
    d = lambda a, b: lib.prim_distance(C.byref(a), C.byref(b), 0.0)
    # boxes, face to face and corner to corner; parallel and skew cylinders; a cylinder's cap over a box
    assert d(_placed(0, (0, 0, 0), I, (0.1, 0.2, 0.3)), _placed(0, (0.5, 0.1, -0.1), I, (0.1, 0.2, 0.3))) == pytest.approx(0.3, abs=1e-9)
    assert d(_placed(0, (0, 0, 0), I, (0.1, 0.1, 0.1)), _placed(0, (0.5, 0.5, 0.5), I, (0.1, 0.1, 0.1))) == pytest.approx(np.sqrt(3) * 0.3, abs=1e-9)
    assert d(_placed(1, (0, 0, 0), I, (0.05, 0.3, 0)), _placed(1, (0.4, 0, 0.1), I, (0.07, 0.2, 0))) == pytest.approx(0.4 - 0.12, abs=1e-9)
    Rx = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])          # local z -> world -y: axes skew, closest points inside both axes
    assert d(_placed(1, (0, 0, 0), I, (0.05, 0.3, 0)), _placed(1, (0.4, 0, 0.0), Rx, (0.07, 0.3, 0))) == pytest.approx(0.4 - 0.12, abs=1e-9)
    assert d(_placed(1, (0, 0, 0.5), I, (0.05, 0.1, 0)), _placed(0, (0.02, 0.01, 0), I, (0.3, 0.3, 0.1))) == pytest.approx(0.3, abs=1e-9)
    assert d(_placed(0, (0, 0, 0), I, (0.1, 0.1, 0.1)), _placed(1, (0.12, 0, 0), I, (0.05, 0.1, 0))) == 0.0          # overlapping
    # random pairs against brute force over sampled points: the routine's distance is a lower bound of every sampled pair's, and close to the
    # least of them; intersecting pairs (a sampled point of one inside the other) give 0
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    for k in range(40):
        ka, kb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        sa, sb = rng.uniform(0.03, 0.2, 3), rng.uniform(0.03, 0.2, 3)
        Ra, Rb = _rot(rng), _rot(rng)
        ca, cb = np.zeros(3), rng.normal(size=3) * rng.uniform(0.1, 0.45)
        got = d(_placed(ka, ca, Ra, sa), _placed(kb, cb, Rb, sb))
        pa, pb = _points(ka, ca, Ra, sa, rng, 40000), _points(kb, cb, Rb, sb, rng, 40000)
        brute = float(cKDTree(pa).query(pb)[0].min())
        assert got <= brute + 1e-9, (k, got, brute)
        assert brute - got <= 6e-3, (k, got, brute)          # (the sampling's resolution: 4e4 points on ~0.1 m^2 of surface)


def test_nominal_pose_is_touch_free_and_mocap_legs_stay_inside_the_envelope():
    r = rs.run(4, ranges="envelope", envelope=(0.0, 0.0), corner=0.0)
    assert r["samples_with_a_touch"] == 0 and all(p["touch"] == 0 for p in r["pairs"])
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS, load_task_constants
    mocap = np.asarray(load_task_constants()["mocap"], dtype=float).reshape(-1, 36)
    legs = mocap[:, 1:13]          # (column 0 is time; 33 joint targets follow: tasks/dyros_dynamic_walk.py:455-457)
    assert np.abs(legs - np.asarray(INITIAL_DOF_POS)[:12]).max() <= 0.5          # the mocap gait's leg excursions: inside the +- 0.5 rad sweep


def test_small_sweep_reproduces_its_committed_result():
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "reach_sweep_small.json")))
    got = rs.run(20000, seed=7, corner=0.1, ranges="envelope", envelope=(0.5, 0.05))
    assert got == want, "the sweep changed: model, proxies, pair list or the routine -- re-run tools/reach/reach_sweep.py and DESIGN.md section 3"


def test_conclusions_drawn_from_the_committed_sweeps():
    """profiles/r06_reach_*.json: what DESIGN.md section 3 says about them."""
    ld = lambda n: json.load(open(os.path.join(ROOT, "profiles", n)))
    full, wide, tight = ld("r06_reach_mjcf.json"), ld("r06_reach_envelope_1p0_0p2.json"), ld("r06_reach_envelope_0p5_0p05.json")
    for r in (full, wide, tight):
        assert r["samples"] >= 1000000
    # the MJCF ranges (+- 3.14 on every leg joint) are no operating envelope: a uniformly drawn configuration self-intersects
    assert full["samples_with_a_touch"] > 0.99 * full["samples"]
    # legs within +- 0.5 rad of the initial pose, upper body held (+- 0.05): every touching link pair but one is covered by a proxy pair
    unc = {(p["a"], p["b"]) for p in tight["pairs"] if p["covered_by_pair"] < 0 and p["touch"] > 0}
    assert unc == {("L_HipRoll_Link", "R_HipRoll_Link")}
    assert tight["samples_with_an_uncovered_touch"] <= 0.02 * tight["samples"]
    cov = [p for p in tight["pairs"] if p["covered_by_pair"] >= 0 and p["touch"] > 0]
    assert len(cov) >= 13 and all(p["a"].startswith("L_") and p["b"].startswith("R_") for p in cov)          # leg against leg: the 16 pairs of round 1
    # false negatives exist and are bounded: a capsule inside a box misses the box's corners -- at most 32 mm of overlap before the proxy fires
    assert max(p["max_miss_depth"] for p in cov) <= 0.032
    # the wide envelope (legs +- 1.0, upper body +- 0.2): the classes DESIGN.md lists as absent -- hip links, pelvis, elbow / wrist-1 links, waist
    unc_w = {(p["a"], p["b"]) for p in wide["pairs"] if p["covered_by_pair"] < 0 and p["touch"] > 1000}
    absent = ("HipRoll", "HipCenter", "base_link", "Elbow", "Wrist1", "Waist", "Neck", "Shoulder")
    same_leg = lambda a, b: a[:2] == b[:2] and all(any(k in x for k in ("Hip", "Thigh", "Knee", "Ankle")) for x in (a, b))
    assert all(any(k in a or k in b for k in absent) or same_leg(a, b) for a, b in unc_w), sorted(unc_w)
