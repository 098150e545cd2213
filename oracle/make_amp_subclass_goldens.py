#!/usr/bin/env python3
"""Mint tests/golden/amp_subclass_ref.npz: the reference's `TocabiAMPLower` SUBCLASS (tasks/tocabi_amp_lower.py: reference state
initialisation from the motion library, the discriminator's observation history, demonstration observations) stepping over a fake
gym.  TEST INFRASTRUCTURE ONLY; runs only where /root/reference is mounted.

Same harness as oracle/make_amp_class_goldens.py (the base class' fixture): the class is imported from where it lies and driven
as the AMP learner drives it -- `reset_done()`, `step(actions)`, now and then `fetch_amp_obs_demo(n)` -- over a fake `gym` whose
`simulate` is the CPU oracle's physics.  `stateInit: Hybrid` (so that both the default and the motion-library starts occur),
`numAMPObsSteps: 3`, the motion library on the synthetic tables of tests/amp_motion_synth.py (the reference's own tables are not in
its checkout).  Committed: the actions, every torch draw (incl. the Bernoulli draws that pick the kind of start), the physics state
after every `simulate`, and the reference's outputs: the ids it reset and which of them started from the motion library, the
root / dof state right after `reset_done()`, `_amp_obs_buf` after every `reset_done()` and every `step()`, the demonstrations.
numpy's global generator (the motion library's sampling) is seeded right before the loop; the replay seeds it the same way.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_harness as RH                                 # noqa: E402
from oracle.oracle import OracleSim                                  # noqa: E402
from oracle import make_amp_class_goldens as MC                      # noqa: E402
import amp_motion_synth as SY                                         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "amp_subclass_ref.npz")
N, STEPS, DEMO_EVERY, DEMO_N, NP_SEED = 16, 40, 8, 6, 123


def main():
    osim = OracleSim(N)
    fake = MC.AmpFakeGym(osim)
    RH.load_reference(lambda: fake)
    MC.load_amp_module()
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules.setdefault("termcolor", tc)          # (the image lacks it; the reference's logger imports it for colours)
    sys.path.insert(0, RH.IGE)                       # (motion_lib.py imports `tasks.amp.humanoid_amp_base` relative to isaacgymenvs/)
    if not hasattr(np, "int"):
        np.int = int                                 # (motion_lib.py:237, removed from numpy 1.24)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        sub = RH._load("isaacgymenvs.tasks.tocabi_amp_lower", os.path.join(RH.IGE, "tasks", "tocabi_amp_lower.py"))
    base = sys.modules["isaacgymenvs.tasks.amp.tocabi_amp_lower_base"]
    RH._loaded["gymapi"].acquire_gym = lambda: fake
    cfg = MC.amp_cfg(N)
    cfg["env"].update(stateInit="Hybrid", hybridInitProb=0.5, numAMPObsSteps=3)
    tmp = tempfile.mkdtemp()
    yml = SY.write(tmp)
    # the class builds its path as <tasks dir>/../../assets/amp/tocabi_motions/ + motion_file: climb out of the reference tree
    assets = os.path.realpath(os.path.join(RH.IGE, "tasks", "../../assets/amp/tocabi_motions"))
    cfg["env"]["motion_file"] = os.path.relpath(yml, assets)
    torch.manual_seed(7)
    np.random.seed(7)
    cwd = os.getcwd()
    os.chdir(RH.IGE)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            env = sub.TocabiAMPLower(cfg, "cpu", 0, True)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(12)
    rec = RH.RngRecorder()
    jit_rand_float = base.torch_rand_float
    base.torch_rand_float = lambda lower, upper, shape, device: (upper - lower) * torch.rand(*shape, device=device) + lower
    orig_bernoulli = torch.bernoulli

    def bernoulli(p, *a, **k):
        out = orig_bernoulli(p, *a, **k)
        rec.log.append(("bernoulli", out.detach().clone()))
        return out
    per = {k: [] for k in ("reset_ids", "ref_ids", "root_after_reset", "dof_pos_after_reset", "dof_vel_after_reset", "amp_after_reset",
                           "amp_after_step", "reset_buf", "progress_buf")}
    actions, demos = [], []
    pad = lambda a: np.pad(np.asarray(a, np.int64), (0, N - len(a)), constant_values=-1)
    # which of the reset envs start from the motion library: the ids every call of _reset_ref_state_init receives (the class'
    # own _reset_ref_env_ids / _reset_default_env_ids are never cleared between resets -- a quirk the replay has to share)
    ref_now = []
    inner = env._reset_ref_state_init

    def spy(ids):
        ref_now.append(ids.clone().numpy())
        return inner(ids)
    env._reset_ref_state_init = spy
    np.random.seed(NP_SEED)
    with rec:
        torch.bernoulli = bernoulli
        try:
            for t in range(STEPS):
                ref_now.clear()
                _, ids = env.reset_done()
                per["reset_ids"].append(pad(ids.numpy()))
                per["ref_ids"].append(pad(np.concatenate(ref_now) if ref_now else []))
                per["root_after_reset"].append(env._root_states.clone().numpy())
                per["dof_pos_after_reset"].append(env._dof_pos.clone().numpy())
                per["dof_vel_after_reset"].append(env._dof_vel.clone().numpy())
                per["amp_after_reset"].append(env._amp_obs_buf.clone().numpy())
                if t % DEMO_EVERY == DEMO_EVERY - 1:
                    demos.append(env.fetch_amp_obs_demo(DEMO_N).clone().numpy())
                a = (rng.uniform(-1, 1, size=(N, 12)) * (0.2 if t < 10 else 1.0)).astype(np.float32)
                actions.append(a)
                _, _, _, extras = env.step(torch.from_numpy(a))
                assert extras["amp_obs"].shape == (N, 3 * 34)
                per["amp_after_step"].append(env._amp_obs_buf.clone().numpy())
                for k in ("reset_buf", "progress_buf"):
                    per[k].append(getattr(env, k).clone().numpy())
        finally:
            torch.bernoulli = orig_bernoulli
    base.torch_rand_float = jit_rand_float
    flat, kinds, shapes = [], [], []
    for kind, tns in rec.log:
        kinds.append(kind)
        shapes.append(np.array(list(tns.shape) + [0] * (3 - tns.dim()), np.int64))
        flat.append(tns.float().numpy().ravel())
    out = {"actions": np.stack(actions), "draw_kind": np.array(kinds), "draw_shape": np.stack(shapes),
           "draw_offset": np.cumsum([0] + [len(x) for x in flat]).astype(np.int64), "draw_data": np.concatenate(flat),
           "num_envs": np.array(N), "steps": np.array(STEPS), "episode_length": np.array(cfg["env"]["episodeLength"]),
           "demo_every": np.array(DEMO_EVERY), "demo_n": np.array(DEMO_N), "np_seed": np.array(NP_SEED),
           "ref_demos": np.stack(demos), "total_mass": env.total_mass.numpy()}
    for k, v in per.items():
        out["ref_" + k] = np.stack(v)
    for name in ("root", "dof", "contact", "feet", "tau"):
        out["sim_" + name] = np.stack([s[name] for s in fake.after_sim])
    np.savez_compressed(OUT, **out)
    nref = [int((r >= 0).sum()) for r in per["ref_ids"]]
    print("wrote", OUT, "draws", len(rec.log), "of which bernoulli", kinds.count("bernoulli"), "simulates", len(fake.after_sim),
          "resets", [int((r >= 0).sum()) for r in per["reset_ids"]], "reference starts", nref)


if __name__ == "__main__":
    main()
