"""Row f-3, the AMP subclass' layer (reference tasks/tocabi_amp_lower.py): the discriminator observation function
(oracle/dw_amp.c::dwo_amp_disc_observations against the reference's build_amp_observations) and the motion library
(isaacgymdyros_amd/motion_lib.py against the reference's TocabiLowerMotionLib on synthetic tables).  Fixture:
tests/golden/amp_disc_ref.npz, minted by oracle/make_amp_disc_goldens.py."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from isaacgymdyros_amd import motion_lib as ML
from oracle import oracle
from tests import amp_motion_synth as SY

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "amp_disc_ref.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(G)


def ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def disc_obs(api, chk, root, dof_pos, dof_vel, row, elem, local, key):
    N, nk = root.shape[0], key.shape[1]
    out = np.zeros((N, 28 + 3 * nk), np.float32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    chk(api["amp_disc_observations"](N, p(root), p(dof_pos), p(dof_vel), row, elem, int(local), p(key), nk, p(out), None))
    return out


def test_disc_observations_vs_reference(g):
    lib, api = oracle.load()

    def chk(rc):
        assert rc == 0, lib.dwo_last_error()
    assert int(g["num_amp_obs_per_step"]) == 34
    root, key = np.ascontiguousarray(g["root_states"]), np.ascontiguousarray(g["key_pos"])
    dp, dv = np.ascontiguousarray(g["dof_pos"]), np.ascontiguousarray(g["dof_vel"])
    for local in (0, 1):
        o = disc_obs(api, chk, root, dp, dv, 33, 1, local, key)
        ref = g["ref_obs_local%d" % local]
        d = ulps(o, ref)
        # root height, the 24 dof entries: copies.  Euler angles: atan2f of glibc against torch's SLEEF kernel (<= 2 ulp, as in
        # test_amp_oracle).  Key-body positions in the heading frame: downstream of atan2 / sin / cos of the heading
        copies = [0] + list(range(4, 28))
        assert d[:, copies].max() == 0
        assert d[:, 1:4].max() <= 2
        if local:
            assert d[:, 28:].max() == 0
        else:
            assert np.abs(o[:, 28:] - ref[:, 28:]).max() <= 1e-6, np.abs(o[:, 28:] - ref[:, 28:]).max()
            assert (d[:, 28:] == 0).mean() > 0.5
    # the interleaved layout of dof_state [N,33,2] and the 12-wide tensors of the motion library give the same rows
    ds = np.ascontiguousarray(np.stack([dp, dv], axis=-1))
    flat = ds.reshape(-1)
    o_il = disc_obs(api, chk, root, flat, flat[1:], 66, 2, 0, key)
    assert np.array_equal(o_il, disc_obs(api, chk, root, dp, dv, 33, 1, 0, key))
    o12 = disc_obs(api, chk, root, np.ascontiguousarray(dp[:, :12]), np.ascontiguousarray(dv[:, :12]), 12, 1, 0, key)
    assert np.array_equal(o12, o_il)
    assert np.abs(o12 - g["ref_obs_12wide"]).max() <= 1e-6


@pytest.fixture(scope="module")
def lib_and_dir(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("motions"))
    yml = SY.write(tmp)
    return ML.TocabiLowerMotionLib(yml, 33, "cpu"), tmp


def test_motion_tables_load_as_in_the_reference(g, lib_and_dir):
    lib, _ = lib_and_dir
    assert lib.num_motions() == 4
    for mine, ref in ((lib._motion_lengths, "ml_lengths"), (lib._motion_weights, "ml_weights"), (lib._motion_fps, "ml_fps"),
                      (lib._motion_dt, "ml_dt"), (lib._motion_num_frames, "ml_num_frames")):
        assert np.array_equal(np.asarray(mine), g[ref]), ref
    assert lib._motion_dt[0] < 0 and lib._motion_dt[1] > 0           # motion 0 is played backwards
    for m in range(4):
        assert np.array_equal(lib.get_motion(m)[0], g["ml_first_rows"][m])
        assert np.array_equal(lib.get_motion(m)[-1], g["ml_last_rows"][m])
    assert lib.get_total_length() == pytest.approx(float(g["ml_lengths"].sum()))


def test_motion_sampling_and_state_bitwise(g, lib_and_dir):
    lib, _ = lib_and_dir
    np.random.seed(1234)
    ids = lib.sample_motions(512)
    times = lib.sample_time(ids)
    tt = lib.sample_time(ids, truncate_time=0.004)
    assert np.array_equal(ids, g["ml_ids"]) and np.array_equal(times, g["ml_times"]) and np.array_equal(tt, g["ml_times_trunc"])
    assert np.array_equal(lib._motion_lengths, g["ml_lengths"])      # (truncation must not eat into the stored lengths)
    qi, qt = g["ml_query_ids"], g["ml_query_times"]
    f0, f1, bl = lib.frame_blend(qi, qt)
    assert np.array_equal(f0, g["ml_frame0"]) and np.array_equal(f1, g["ml_frame1"]) and np.array_equal(bl, g["ml_blend"])
    out = lib.get_motion_state(qi, qt)
    names = ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel", "key_pos")
    for name, t in zip(names, out):
        assert t.dtype == torch.float32
        ref = g["ml_" + name]
        assert t.shape == ref.shape, name
        assert np.array_equal(t.numpy(), ref, equal_nan=True), (name, np.nanmax(np.abs(t.numpy() - ref)))
    # the fixture walks through slerp's three branches
    q0 = torch.tensor(np.stack([lib.get_motion(m)[i] for m, i in zip(qi, f0)])[:, 28:32], dtype=torch.float)
    q1 = torch.tensor(np.stack([lib.get_motion(m)[i] for m, i in zip(qi, f1)])[:, 28:32], dtype=torch.float)
    c = (q0 * q1).sum(-1).abs()
    s = torch.sqrt(1 - c * c)
    assert (c >= 1).any() and ((s < 0.001) & (c < 1)).any() and (s > 0.01).any()


def test_missing_table_is_reported(tmp_path):
    p = tmp_path / "m.yaml"
    p.write_text("motions:\n  - {file: nowhere.txt, weight: 1.0}\n")
    with pytest.raises(FileNotFoundError, match="nowhere.txt"):
        ML.TocabiLowerMotionLib(str(p), 33, "cpu")


def test_single_table_file(tmp_path):
    np.savetxt(tmp_path / "one.txt", SY.table(2)[:300])
    lib = ML.TocabiLowerMotionLib(str(tmp_path / "one.txt"), 33, "cpu")
    assert lib.num_motions() == 1 and lib._motion_num_frames[0] == 300
    assert lib._motion_dt[0] == pytest.approx(0.0005) and lib._motion_weights[0] == 1.0
    rp, rr, rv, ra, dp, dv, kp = lib.get_motion_state(np.array([0, 0]), np.array([0.0, 0.01]))
    assert dp.shape == (2, 12) and kp.shape == (2, 2, 3) and torch.isfinite(rr).all()
    assert np.allclose(rv[0].numpy(), lib.get_motion(0)[0, 32:35].astype(np.float32))        # dt = frame period: velocities as stored
