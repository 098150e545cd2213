"""Config 3 on ONE GPU with the fused update: the rollout as replayed hipGraphs (8 steps per graph) against plain launches.
usage: python tools/ppo_rollout_graph_ab.py [N] [epochs]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for r in range(2):
    for name, gr in (("rollout in hipGraphs", True), ("rollout as plain launches", False)):
        st = ppo.train(N, epochs=E, device="cuda:0", log=lambda s: None, graph_rollout=gr, fused_update=True)
        fps = sorted(s["total_fps"] for s in st[1:]); pl = sorted(s["play_fps"] for s in st[1:])
        print("%-28s total_fps median %.2f M, play_fps median %.2f M, mean reward %.4f" % (name, fps[len(fps) // 2] / 1e6, pl[len(pl) // 2] / 1e6, st[-1]["mean_reward"]), flush=True)
