"""Diagnostic: self-collision scenario, quad pipeline vs oracle, per-joint differences (run on the GPU box)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hip_backend import make_env
from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS, load_task_constants
from oracle.oracle import OracleSim
tc = load_task_constants()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for pipe in (2, 1):
    env = make_env(N, randomize=False, pipeline=pipe)
    b = env._buf
    b["root_states"][:, 0:2] = 0; b["root_states"][:, 2] = 3.0
    q = torch.tensor(INITIAL_DOF_POS).repeat(N, 1)
    roll = torch.linspace(0.05, 0.25, N)
    q[:, 1] = -roll; q[:, 7] = roll
    b["dof_state"][..., 0] = q.cuda(); b["dof_state"][..., 1] = 0
    ora = OracleSim(N, task_const=tc, cfg=env._ccfg)
    for k, t in env._buf.items():
        ora.buf[k][...] = t.cpu().numpy()
    tau = torch.zeros(N, 33)
    env.simulate(tau.cuda()); ora.simulate(tau.numpy()); torch.cuda.synchronize()
    dq = np.abs(env.dof_vel.cpu().numpy() - ora.buf["dof_state"][:, :, 1])
    print("pipeline", pipe, "max dqd", dq.max(), "root dv", np.abs(env.root_states.cpu().numpy() - ora.buf["root_states"]).max())
    np.set_printoptions(linewidth=250, precision=2, suppress=False)
    print("per-joint max over envs:", dq.max(axis=0))
    print("per-env max:", dq.max(axis=1))
    e = int(dq.max(axis=1).argmax())
    print("env", e, "gpu qd", env.dof_vel[e].cpu().numpy()); print("ora qd", ora.buf["dof_state"][e, :, 1])
