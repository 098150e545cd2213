"""Synthetic motion tables for the row f-3 motion library (the reference's own tables are not in its checkout).  Used by the
script that mints tests/golden/amp_disc_ref.npz from the reference class and by the test that replays it: every value is an
exact function of integers (one IEEE division / square root at the end), so both sides see the same bits on any machine."""
import os

import numpy as np

ROWS, COLS = 13000, 52                  # enough rows for the step_time 0.6 and 0.9 windows of the loader


def table(salt: int) -> np.ndarray:
    i = np.arange(ROWS, dtype=np.int64)[:, None]
    c = np.arange(COLS, dtype=np.int64)[None, :]
    # slow saw-tooth per column (period ~ 1000 frames) plus a column offset: smooth enough to look like motion, exact in float64
    v = (((i * (3 + (c + salt) % 7) + 131 * c + 977 * salt) % 2000) - 1000).astype(np.float64) / 1000.0
    m = v.copy()
    m[:, 0] = np.arange(ROWS, dtype=np.float64) * 0.0005 + 1.0
    m[:, 27] = 1.2 + 0.1 * v[:, 27]          # root height: the robot starts above the ground
    # root rotation: a unit quaternion that turns by up to ~0.2 rad between frames (exercises slerp's main branch) on some
    # stretches and stays put on others (its `identical` and `nearly parallel` branches)
    ang = ((i[:, 0] * (5 + salt)) % 3000).astype(np.float64) / 10.0
    hold = (i[:, 0] // 500) % 3 == 0
    ang = np.where(hold, np.floor(ang / 100.0) * 100.0, ang)
    tiny = (i[:, 0] // 500) % 3 == 1
    ang = np.where(tiny, np.floor(ang / 100.0) * 100.0 + (i[:, 0] % 500) * 1e-4, ang)
    ax = np.array([1.0 + (salt % 3), 2.0, 3.0 - (salt % 2)])
    ax = ax / np.sqrt(ax @ ax)
    m[:, 28:31] = ax[None, :] * np.sin(ang / 2.0)[:, None]
    m[:, 31] = np.cos(ang / 2.0)
    n = np.sqrt((m[:, 28:32] ** 2).sum(axis=1, keepdims=True))
    m[:, 28:32] /= n
    return m


def write(dirname: str):
    """Three tables and a yaml that exercises every loader branch: both step_time windows, no window, forward / backward /
    default play speed.  Returns the yaml's path."""
    for k in range(3):
        np.savetxt(os.path.join(dirname, "synth%d.txt" % k), table(k))
    with open(os.path.join(dirname, "motions.yaml"), "w") as fh:
        fh.write("motions:\n"
                 "  - {file: synth0.txt, weight: 0.5, step_time: 0.9, play_speed: -0.7}\n"
                 "  - {file: synth1.txt, weight: 0.25, step_time: 0.6}\n"
                 "  - {file: synth2.txt, weight: 1.0, play_speed: 1.5}\n"
                 "  - {file: synth1.txt, weight: 0.25, step_time: 0.9, play_speed: 0.5}\n")
    return os.path.join(dirname, "motions.yaml")
