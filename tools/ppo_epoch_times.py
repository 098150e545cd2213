import importlib.util, os, sys, json
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
os.environ["DW_PPO_TIMES"] = "1"
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
st = ppo.train(16384, epochs=4, device="cuda:0", log=lambda s: None, graph_rollout=True, fused_update=True)
for s in st: print({k: round(s[k], 3) for k in ("play_ms", "prep_ms", "update_ms")}, s["total_fps"])
