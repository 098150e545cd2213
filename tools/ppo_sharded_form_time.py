"""Config 3 on ONE GPU with the update in its SHARDED form (FusedPpoUpdate(collective=True): dwp_grad_bucket + the one-slab statistics / Adam; the
all-reduce itself has nothing to do on one rank) next to the plain four launches: what the fifth launch and the split graphs cost an update.
usage: [DW_PPO_GRAPH_COLLECTIVE=1] python tools/ppo_sharded_form_time.py [N] [epochs]"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for name, kw in (("four launches", {}), ("sharded form (bucket + 1-slab tail), graph collective = %s" % ppo.GRAPH_COLLECTIVE, dict(fused_collective=True))):
    st = ppo.train(N, epochs=E, device="cuda:0", log=lambda s: None, graph_rollout=True, fused_update=True, **kw)
    fps = sorted(s["total_fps"] for s in st[1:])
    print("%-70s total_fps median %.2f M (epochs 2..%d: %s), mean reward %.4f" % (name, fps[len(fps) // 2] / 1e6, E, " ".join("%.2f" % (f / 1e6) for f in fps), st[-1]["mean_reward"]), flush=True)
