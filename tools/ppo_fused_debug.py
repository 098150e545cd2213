"""Debug print of the fused PPO update's intermediates (tests/test_ppo_gpu.py's first case)."""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isaacgymdyros_amd import ppo_update as U
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
c = dict(ppo.TRAIN_CFG["config"])
torch.manual_seed(3)
dev = "cuda:0"
net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
if os.environ.get("GAIN"):
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.requires_grad and p_.dim() == 2: torch.nn.init.orthogonal_(p_, gain=float(os.environ["GAIN"]))
            elif p_.requires_grad: p_.uniform_(-0.1, 0.1)
B, nmb = 4096, 3
fused = U.FusedPpoUpdate(net, c, B, nmb, dev)
fused.set_learning_rates(3e-4, 5e-4)
g = torch.Generator(device=dev).manual_seed(11)
n = B * nmb
obs = torch.randn(n, U.IN, generator=g, device=dev)
with torch.no_grad():
    mu0, logstd, _ = net(obs)
    sigma = torch.exp(logstd)
    act = mu0 + sigma * torch.randn(n, U.ACT, generator=g, device=dev)
    mu_old = mu0 + 0.1 * sigma * torch.randn(n, U.ACT, generator=g, device=dev)
    nlp_old = ppo.neglogp(act, mu_old, sigma, logstd)
adv = torch.randn(n, generator=g, device=dev); ret = torch.randn(n, generator=g, device=dev)
fused.bind_batch(obs, act, nlp_old, mu_old, adv, ret)
fused.update(); torch.cuda.synchronize()
f = lambda t: (float(t.float().abs().max()), bool(torch.isfinite(t.float()).all()))
print("state", fused.state.tolist())
for k in ("x16", "h1", "h2", "out", "dout", "dh2", "dh1"):
    print(k, f(getattr(fused, k)))
for k, v in fused.gviews.items():
    print("g", k, f(v[0]), f(v[1]))
print("p16", f(fused.p16), "p", f(fused.p))
bad = (fused.p16 != fused.p.half()).nonzero().flatten()
print("p16 mismatches", bad.numel(), bad[:8].tolist(), fused.p16[bad[:8]].tolist(), fused.p[bad[:8]].tolist(), fused.p.half()[bad[:8]].tolist())
