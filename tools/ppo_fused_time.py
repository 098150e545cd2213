"""Config 3 with the autograd update in a hipGraph vs the fused update: total_fps, play_fps, mean_reward, losses (one GPU)."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
import torch
if os.environ.get("PPO_TUNE"):
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.set_filename('/tmp/dw_tunableop.csv')
    torch.cuda.tunable.set_max_tuning_duration(30)
    torch.cuda.tunable.set_max_tuning_iterations(20)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for name, kw in ((("fused_update", dict(fused_update=True)),) if os.environ.get("PPO_ONLY_FUSED") else (("fused_update", dict(fused_update=True)), ("graph_update", dict(graph_update=True)))):
    st = ppo.train(N, epochs=E, device="cuda:0", log=lambda s: None, graph_rollout=True, **kw)
    print(name, json.dumps([{k: (round(s[k], 5) if isinstance(s[k], float) else s[k]) for k in ("total_fps", "play_fps", "mean_reward", "a_loss", "c_loss", "kl", "mean_episode_length")} for s in st]), flush=True)
