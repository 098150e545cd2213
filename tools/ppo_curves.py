"""Learning curves of the PPO consumer, autograd update (hipGraph) against the fused path, same seed: mean reward, episode length, losses per epoch."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = int(sys.argv[2]) if len(sys.argv) > 2 else 40
runs = {}
for name, kw in (("fused", dict(fused_update=True)), ("autograd", dict(graph_update=True))):
    runs[name] = ppo.train(N, epochs=E, device="cuda:0", log=lambda s: None, graph_rollout=True, max_epochs=E, **kw)
    print(name, "done", flush=True)
print("%5s | %-44s | %-44s" % ("epoch", "fused: reward  ep.len  a_loss   c_loss   kl", "autograd: reward  ep.len  a_loss   c_loss   kl"))
for i in range(E):
    f, a = runs["fused"][i], runs["autograd"][i]
    fmt = lambda s: "%8.4f %7.1f %8.4f %8.3f %8.5f" % (s["mean_reward"], s["mean_episode_length"], s["a_loss"], s["c_loss"], s["kl"])
    if i < 5 or i % 5 == 4:
        print("%5d | %-44s | %-44s" % (i + 1, fmt(f), fmt(a)))
import statistics
for k in ("mean_reward", "mean_episode_length"):
    print("last 10 epochs, mean %s: fused %.4f, autograd %.4f" % (k, statistics.mean(s[k] for s in runs["fused"][-10:]), statistics.mean(s[k] for s in runs["autograd"][-10:])))
print("total_fps (median): fused %.3g, autograd %.3g" % (statistics.median(s["total_fps"] for s in runs["fused"][1:]), statistics.median(s["total_fps"] for s in runs["autograd"][1:])))
