"""Config 3 on ONE GPU, the forms of the fused update's tail next to each other, alternating in one process:
  * dwp_grad_stats | dwp_adam_finish with the rollout policy's fp32 operand-order copy rewritten by every update (round 5's form),
  * the same with that copy made once per epoch (FusedPpoUpdate(policy_copy_per_update=False) + sync_policy_copy()),
  * dwp_stats_adam_finish (ONE launch, the blocks wait for each other's share of the norm), copy once per epoch.
usage: python tools/ppo_tail_form_time.py [N] [epochs] [rounds]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for r in range(R):
    for name, merged, per_update in (("two launches, policy copy per update", False, True), ("two launches, policy copy per epoch", False, False),
                                     ("one launch (stats_adam_finish), copy per epoch", True, False)):
        ppo.MERGED_TAIL, ppo.POLICY_COPY_PER_UPDATE = merged, per_update
        st = ppo.train(N, epochs=E, device="cuda:0", log=lambda s: None, graph_rollout=True, fused_update=True)
        fps = sorted(s["total_fps"] for s in st[1:])
        print("%-48s total_fps median %.2f M (epochs 2..%d: %s), mean reward %.4f" % (name, fps[len(fps) // 2] / 1e6, E, " ".join("%.2f" % (f / 1e6) for f in fps), st[-1]["mean_reward"]), flush=True)
