"""Reference-motion table of the TOCABI AMP task: what `TocabiAMPLower` draws its reference state
initialisation and the discriminator's demonstration observations from.

Mirrors `TocabiLowerMotionLib` of the reference (paths relative to python/IsaacGymEnvs/isaacgymenvs):
  tasks/amp/utils_amp/tocabi_lower_motion_lib.py:38-47    constructor (12 leg dofs = num_dofs - 21)
                                                :61-154   get_motion_state: frame pair + blend, linear root / key-body positions,
                                                          slerp of the root rotation, velocities rescaled by 0.0005 / dt
                                                :156-233  _load_motions: text tables (column layout below), the step_time windows,
                                                          play_speed (negative = played backwards)
                                                :235-264  _fetch_motion_files: yaml list with weight / step_time / play_speed
  tasks/amp/utils_amp/motion_lib.py:60-79                 sample_motions, sample_time (numpy's global generator, as the reference)
                                   :233-241               _calc_frame_blend
  utils/torch_jit_utils.py:298-330                        slerp

Columns of a motion table (one row per 0.5 ms frame): 0 time | 1..12 leg dof positions | 13..24 leg dof velocities | 25..27 root
position | 28..31 root rotation xyzw | 32..34 root linear velocity | 35..37 root angular velocity | 38..43 the two foot positions.

Layout here: all motions live in ONE float64 table, motion m in rows start[m] .. start[m] + frames[m] - 1, so a query of any mix
of motions is a single gather (the reference loops over the distinct motion ids of a query).  Host-side numpy, like the
reference's: this is the data loader, called at resets and once per discriminator batch, not on the step path; the blends are
torch on the task's device.  Pinned against the reference class on synthetic tables (tests/golden/amp_disc_ref.npz,
tests/test_amp_motion.py): frame indices and every returned tensor bit for bit on the CPU.

The reference's motion tables themselves (assets/amp/tocabi_motions/*.txt) are not in the checkout -- only the yaml that lists
them -- so a user supplies them (`cfg.env.motion_file`, a path to a .yaml or a single .txt).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import yaml

FRAME_DT = 0.0005                      # the tables' frame period: velocities are stored per this step (:99-100,109)
STEP_WINDOWS = {0.6: (4400, 9201), 0.9: (5600, 12801), "yaw": (20000, 54201)}          # :184-189
COL_QPOS, COL_QVEL, COL_RPOS, COL_RROT, COL_RVEL, COL_RANG, COL_KEY = slice(1, 13), slice(13, 25), slice(25, 28), slice(28, 32), slice(32, 35), slice(35, 38), slice(38, 44)


def slerp(q0: torch.Tensor, q1: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """utils/torch_jit_utils.py:298-330, operation for operation (w term first in the dot product; the near-parallel and the
    identical cases fall back to the mean / to q0)."""
    c = q0[..., 3] * q1[..., 3] + q0[..., 0] * q1[..., 0] + q0[..., 1] * q1[..., 1] + q0[..., 2] * q1[..., 2]
    q1 = torch.where((c < 0).unsqueeze(-1), -q1, q1)
    c = torch.abs(c).unsqueeze(-1)
    half = torch.acos(c)
    s = torch.sqrt(1.0 - c * c)
    ra = torch.sin((1 - t) * half) / s
    rb = torch.sin(t * half) / s
    q = ra * q0 + rb * q1
    q = torch.where(torch.abs(s) < 0.001, 0.5 * q0 + 0.5 * q1, q)
    return torch.where(torch.abs(c) >= 1, q0, q)


class TocabiLowerMotionLib:

    def __init__(self, motion_file: str, num_dofs: int, device):
        self._num_dof = num_dofs - 21
        self._device = device
        files, weights, step_times, speeds = self._read_list(motion_file)
        tables, self._motion_files = [], list(files)
        dts, fps, frames, lengths = [], [], [], []
        for path, st, sp in zip(files, step_times, speeds):
            if not os.path.exists(path):
                raise FileNotFoundError("motion table %r not found (the reference's assets/amp/tocabi_motions/*.txt are not part of "
                                        "its checkout; point cfg.env.motion_file at your own)" % path)
            m = np.loadtxt(path)
            if st in STEP_WINDOWS:
                a, b = STEP_WINDOWS[st]
                m = m[a:b, :]
            if sp is not None and sp < 0.0:
                m, sp = np.flip(m, axis=0), -sp           # played backwards: the time column now decreases, so dt < 0 below
            dt = (m[1, 0] - m[0, 0]) if sp is None else (m[1, 0] - m[0, 0]) / sp
            f = 1.0 / dt if sp is None else np.abs(1.0 / dt)
            tables.append(np.ascontiguousarray(m))
            dts.append(dt); fps.append(f); frames.append(m.shape[0]); lengths.append(1.0 / f * (m.shape[0] - 1))
        self._table = np.concatenate(tables, axis=0)
        self._start = np.concatenate(([0], np.cumsum(frames)[:-1])).astype(np.int64)
        self._motion_lengths = np.array(lengths)
        self._motion_weights = np.array(weights, dtype=np.float64)
        self._motion_weights /= np.sum(self._motion_weights)
        self._motion_fps = np.array(fps)
        self._motion_dt = np.array(dts)
        self._motion_num_frames = np.array(frames)
        self.motion_ids = torch.arange(len(tables), dtype=torch.long, device=device)

    # ------------------------------------------------------------------ the list of tables
    @staticmethod
    def _read_list(motion_file: str) -> Tuple[List[str], List[float], List[object], List[Optional[float]]]:
        if os.path.splitext(motion_file)[1] != ".yaml":
            return [motion_file], [1.0], [None], [None]
        with open(os.path.join(os.getcwd(), motion_file), "r") as fh:
            entries = yaml.load(fh, Loader=yaml.SafeLoader)["motions"]
        base = os.path.dirname(motion_file)
        for en in entries:
            assert en["weight"] >= 0
        return ([os.path.join(base, en["file"]) for en in entries], [en["weight"] for en in entries],
                [en.get("step_time", None) for en in entries], [en.get("play_speed", None) for en in entries])

    # ------------------------------------------------------------------ sizes
    def num_motions(self) -> int:
        return len(self._motion_lengths)

    def get_total_length(self):
        return sum(self._motion_lengths)

    def get_motion(self, motion_id: int) -> np.ndarray:
        a = int(self._start[motion_id])
        return self._table[a:a + int(self._motion_num_frames[motion_id])]

    def get_motion_length(self, motion_ids):
        return self._motion_lengths[motion_ids]

    # ------------------------------------------------------------------ sampling (numpy's global generator, as the reference)
    def sample_motions(self, n: int) -> np.ndarray:
        return np.random.choice(self.num_motions(), size=n, replace=True, p=self._motion_weights)

    def sample_time(self, motion_ids: np.ndarray, truncate_time: Optional[float] = None) -> np.ndarray:
        phase = np.random.uniform(low=0.0, high=1.0, size=motion_ids.shape)
        length = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            length = length - truncate_time
        return phase * length

    # ------------------------------------------------------------------ state at (motion, time)
    def frame_blend(self, motion_ids: np.ndarray, motion_times: np.ndarray):
        """(frame0, frame1, blend) within each motion (motion_lib.py:233-241 with |dt|, as the subclass calls it)."""
        length, frames = self._motion_lengths[motion_ids], self._motion_num_frames[motion_ids]
        dt = np.abs(self._motion_dt[motion_ids])
        phase = np.clip(motion_times / length, 0.0, 1.0)
        i0 = (phase * (frames - 1)).astype(int)
        i1 = np.minimum(i0 + 1, frames - 1)
        return i0, i1, (motion_times - i0 * dt) / dt

    def get_motion_state(self, motion_ids: Sequence[int], motion_times: Sequence[float]):
        """-> root_pos [n,3], root_rot [n,4], root_vel [n,3], root_ang_vel [n,3], dof_pos [n,12], dof_vel [n,12], key_pos [n,2,3]
        (float32 on the device)."""
        motion_ids = np.asarray(motion_ids)
        motion_times = np.asarray(motion_times)
        i0, i1, blend = self.frame_blend(motion_ids, motion_times)
        r0, r1 = self._table[self._start[motion_ids] + i0], self._table[self._start[motion_ids] + i1]
        dt = self._motion_dt[motion_ids][:, np.newaxis]          # signed: a motion played backwards has its velocities reversed

        def dev(a):
            return torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=self._device)
        # (the reference forms v * 0.0005 / dt left to right in float64, then rounds to float32)
        root_vel, root_ang = dev(r0[:, COL_RVEL] * FRAME_DT / dt), dev(r0[:, COL_RANG] * FRAME_DT / dt)
        dof_pos, dof_vel = dev(r0[:, COL_QPOS]), dev(r0[:, COL_QVEL] * FRAME_DT / dt)
        b = dev(blend[:, np.newaxis])
        root_pos = (1.0 - b) * dev(r0[:, COL_RPOS]) + b * dev(r1[:, COL_RPOS])
        root_rot = slerp(dev(r0[:, COL_RROT]), dev(r1[:, COL_RROT]), b)
        be = b.unsqueeze(-1)
        key_pos = (1.0 - be) * dev(r0[:, COL_KEY].reshape(-1, 2, 3)) + be * dev(r1[:, COL_KEY].reshape(-1, 2, 3))
        return root_pos, root_rot, root_vel, root_ang, dof_pos, dof_vel, key_pos
