import importlib.util, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
kw = dict(fused_update=True) if os.environ.get("PPO_FUSED") else dict(graph_update=True)
st = ppo.train(int(os.environ.get("PPO_ENVS", "4096")), epochs=2, device="cuda:0", log=print, graph_rollout=True, **kw)
