"""How a change of the self-collision pair set moves the episode statistics: the same random policy on
two compiled models, the shipped one and the one given in DW_MODEL_B (a tocabi_model.json), same seeds.
usage: DW_MODEL_B=path/to/other_model.json python tools/sc_pairs_effect.py [N] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd import model as M, dyros_dynamic_walk as D
from isaacgymdyros_amd.config import default_cfg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
res = {}
for tag, path in (("shipped", M.MODEL_JSON), ("B", os.environ["DW_MODEL_B"])):
    D.load_model = lambda p=path: M.load_model(p)
    env = D.DyrosDynamicWalk(default_cfg(N, "cuda:0"), "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(11)
    done, rew, sc_term = 0, 0.0, 0
    for t in range(STEPS):
        a = torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
        o, r, d, ex = env.step(a)
        done += int(d.sum()); rew += float(r.mean())
    el = env.epi_len_log.cpu().numpy().ravel()
    res[tag] = (len(env.model.sc_pairs), done, el.mean(), el.std(), rew / STEPS)
    env.close()
for tag, (npairs, done, m, s, r) in res.items():
    print("%-8s pairs %2d  episodes finished %6d  last episode length %.2f +- %.2f  mean reward/step %.5f" % (tag, npairs, done, m, s, r))
