"""The DYROS PPO pieces of examples/ppo_consumer.py against closed forms and the reference's YAML (VERDICT r1
item 8).  CPU only; the rollout against the real env is a GPU test (tests/test_hip_gpu.py)."""
import importlib.util
import json
import math
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _mod():
    spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gae_closed_form_and_done_masking():
    m = _mod()
    H, N, g, lam = 6, 3, 0.99, 0.95
    # constant reward 1, zero values, no terminations: A_t = sum_k (g lam)^k
    z = torch.zeros(H, N, 1)
    adv = m.discount_values(torch.zeros(N), torch.zeros(N, 1), torch.zeros(H, N), z, torch.ones(H, N, 1), g, lam)
    for t in range(H):
        assert adv[t, 0, 0].item() == pytest.approx(sum((g * lam) ** k for k in range(H - t)), rel=1e-6)
    # a termination recorded at t = 3 (mb_fdones[3] = the done flag BEFORE step 3) cuts the sum for t < 3
    d = torch.zeros(H, N); d[3, 1] = 1.0
    adv2 = m.discount_values(torch.zeros(N), torch.zeros(N, 1), d, z, torch.ones(H, N, 1), g, lam)
    assert adv2[2, 1, 0].item() == pytest.approx(1.0) and adv2[1, 1, 0].item() == pytest.approx(1 + g * lam)
    assert torch.equal(adv2[:, 0], adv[:, 0]) and torch.equal(adv2[3:, 1], adv[3:, 1])
    # bootstrap from the last value and from stored values
    v = torch.full((H, N, 1), 2.0)
    adv3 = m.discount_values(torch.zeros(N), torch.full((N, 1), 2.0), torch.zeros(H, N), v, torch.zeros(H, N, 1), g, 0.0)
    assert torch.allclose(adv3, torch.full((H, N, 1), g * 2.0 - 2.0))


def test_losses_match_hand_computed_values():
    m = _mod()
    old, new, adv = torch.tensor([1.0, 1.0, 1.0, 1.0]), torch.tensor([1.0, 0.5, 1.5, 0.9]), torch.tensor([2.0, 2.0, -1.0, -3.0])
    loss, frac = m.actor_loss(old, new, adv, 0.2)
    ratio = torch.exp(old - new)
    exp = torch.stack([torch.max(-a * r, -a * min(max(r, 0.8), 1.2)) for a, r in zip(adv, ratio)])
    assert torch.allclose(loss, exp)
    assert loss[1].item() == pytest.approx(-2.0 * 1.2)                       # positive advantage, ratio clipped at 1 + e
    assert loss[2].item() == pytest.approx(1.0 * 0.8)                        # negative advantage, ratio clipped at 1 - e
    assert frac.item() == pytest.approx(0.5)
    assert torch.equal(m.critic_loss(torch.zeros(2), torch.tensor([1.0, 3.0]), 0.2, torch.tensor([2.0, 2.0]), False), torch.tensor([1.0, 1.0]))
    cl = m.critic_loss(torch.tensor([0.0]), torch.tensor([1.0]), 0.2, torch.tensor([2.0]), True)
    assert cl.item() == pytest.approx(max((1.0 - 2.0) ** 2, (0.2 - 2.0) ** 2))
    # bound loss as the reference writes it (a2c_continuous_seperate.py:233-241): clamp_MAX, so it penalises mu INSIDE the
    # bound and is 0 outside -- harmless upstream because bounds_loss_coef = 0 (yaml:90); restated literally, not "fixed"
    mu = torch.tensor([[0.0, 1.2, -1.5], [1.1, -1.1, 2.0]])
    exp = torch.tensor([1.21 + 6.76 + 0.01, 4.84 + 0.81])
    lit = (torch.clamp_max(mu - 1.1, 0.0) ** 2 + torch.clamp_max(-mu + 1.1, 0.0) ** 2).sum(-1)
    assert torch.allclose(m.bound_loss(mu), lit) and torch.allclose(lit, exp)
    x, mean, ls = torch.tensor([[0.3, -0.2]]), torch.tensor([[0.1, 0.0]]), torch.tensor([[-1.0, -2.0]])
    ref = -torch.distributions.Normal(mean, ls.exp()).log_prob(x).sum(-1)
    assert torch.allclose(m.neglogp(x, mean, ls.exp(), ls), ref, atol=1e-6)


def test_sigma_and_lr_schedules():
    m = _mod()
    net = m.DyrosActorCritic(487, 13)
    assert not net.sigma.requires_grad and all(p.requires_grad for p in net.actor_parameters() + net.critic_parameters())
    ids = {id(p) for p in net.actor_parameters()} & {id(p) for p in net.critic_parameters()}
    assert not ids                                                           # separate trunks: separate optimisers
    net.update_action_noise(1.0)
    assert net.sigma[0].item() == pytest.approx(-2.302585)
    net.update_action_noise(0.75)
    assert net.sigma[0].item() == pytest.approx(0.5 * -2.302585 + 0.5 * -2.9957)
    for pr in (0.5, 0.2, 0.0):
        net.update_action_noise(pr)
        assert net.sigma[0].item() == pytest.approx(-2.9957)
    mu, logstd, value = net(torch.zeros(4, 487))
    assert mu.shape == (4, 13) and logstd.shape == (4, 13) and value.shape == (4, 1)
    assert float(mu.abs().max()) < 0.5                                       # orthogonal init, gain 0.01: a near-zero start
    lr = m.LinearLR(1e-5, 3e-6, 5000)
    assert lr(0) == pytest.approx(1e-5) and lr(5000) == pytest.approx(3e-6) and lr(2500) == pytest.approx(6.5e-6) and lr(9999) == pytest.approx(3e-6)


def test_built_in_train_cfg_equals_the_reference_yaml():
    path = "/root/reference/python/IsaacGymEnvs/isaacgymenvs/cfg/train/DyrosDynamicWalkPPO.yaml"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    m = _mod()
    ref = m.load_train_yaml(path)
    assert ref["network"] == m.TRAIN_CFG["network"]
    for k, v in ref["config"].items():
        assert m.TRAIN_CFG["config"][k] == v, (k, m.TRAIN_CFG["config"][k], v)


def test_gradient_allreduce_averages_over_two_ranks(tmp_path):
    """Sharded training: the N>1 path of train() is one all-reduce of the flat gradient bucket; here on gloo, world 2."""
    worker = tmp_path / "w.py"
    worker.write_text('''
import importlib.util, json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from isaacgymdyros_amd import dist as dwdist
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(%r, "examples", "ppo_consumer.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
rank, _, world = dwdist.init_from_env("gloo")
torch.manual_seed(0)
net = m.DyrosActorCritic(8, 3, dict(m.TRAIN_CFG["network"], mlp_units=[16]))
x = torch.full((4, 8), float(rank + 1))
mu, _, v = net(x)
(mu.sum() + v.sum()).backward()
ps = net.actor_parameters() + net.critic_parameters()
local = [p.grad.clone() for p in ps]
m.allreduce_grads(ps, world)
others = [torch.zeros_like(g) for g in local]
for i, g in enumerate(local):
    t = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(t, g)
    others[i] = sum(t) / world
ok = all(torch.allclose(p.grad, o, atol=1e-7) for p, o in zip(ps, others))
json.dump({"ok": bool(ok), "world": world}, open(os.path.join(sys.argv[1], "r%%d.json" %% rank), "w"))
dist.barrier(); dist.destroy_process_group()
''' % (ROOT, ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29733", str(worker), str(tmp_path)], check=True, timeout=300, env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for r in range(2):
        o = json.load(open(tmp_path / ("r%d.json" % r)))
        assert o["ok"] and o["world"] == 2


def test_sharded_train_keeps_ranks_in_step_on_gloo(tmp_path):
    """train() with world = 2 (gloo, CPU, a plumbing env that runs no kernel): gradients are averaged BEFORE they are
    unscaled / clipped / applied, so both ranks end with identical weights although their envs and their
    exploration noise differ."""
    worker = tmp_path / "w.py"
    worker.write_text('''
import importlib.util, json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from isaacgymdyros_amd import dist as dwdist
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(%r, "examples", "ppo_consumer.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
rank, _, world = dwdist.init_from_env("gloo")

class Env:                      # VecTask surface train() touches; dynamics = noise that depends on the rank
    num_envs, num_obs, num_acts = 32, 12, 3
    def __init__(self):
        self.g = torch.Generator().manual_seed(100 + rank)
        self.extras = {}
        self.episodes_finished = torch.zeros(32)
        self.epi_len_log = torch.zeros(32)
        self.first_actions = None
    def reset(self):
        return {"obs": torch.zeros(32, 12)}
    def step(self, a):
        if self.first_actions is None: self.first_actions = a.clone()
        o = torch.randn(32, 12, generator=self.g) + a.sum(1, keepdim=True)
        return {"obs": o}, o[:, 0].tanh(), (torch.rand(32, generator=self.g) < 0.05).long(), {}
cfg = {"network": dict(m.TRAIN_CFG["network"], mlp_units=[16]), "config": dict(m.TRAIN_CFG["config"], horizon_length=8, minibatch_size=64, mini_epochs=2)}
nets = []
orig = m.DyrosActorCritic
class Spy(orig):
    def __init__(self, *a, **k):
        super().__init__(*a, **k); nets.append(self)
m.DyrosActorCritic = Spy
env = Env()
st = m.train(epochs=2, device="cpu", cfg=cfg, env=env, rank=rank, world=world, log=lambda s: None)
flat = torch.cat([p.detach().reshape(-1) for p in nets[0].parameters()])
t = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(t, flat)
fa = [torch.zeros_like(env.first_actions) for _ in range(world)]
dist.all_gather(fa, env.first_actions)
json.dump({"same_weights": bool(torch.equal(t[0], t[1])), "moved": bool((flat - flat.mean()).abs().sum() > 0), "finite": bool(torch.isfinite(flat).all()),
           "same_noise": bool(torch.equal(fa[0], fa[1])), "epochs": len(st)}, open(os.path.join(sys.argv[1], "t%%d.json" %% rank), "w"))
dist.barrier(); dist.destroy_process_group()
''' % (ROOT, ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29741", str(worker), str(tmp_path)], timeout=600, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for k in range(2):
        o = json.load(open(tmp_path / ("t%d.json" % k)))
        assert o["same_weights"] and o["finite"] and o["epochs"] == 2, o
        assert not o["same_noise"], "every rank drew the same exploration noise"


def test_fused_update_refuses_what_it_was_not_written_for():
    """isaacgymdyros_amd/ppo_update.py: configurations and network shapes other than DyrosDynamicWalkPPO.yaml's are a ValueError on the
    host before the library is touched; the layout constants mirror include/dyros_ppo.h."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _mod()
    c = dict(ppo.TRAIN_CFG["config"])
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"])
    for bad in ({"clip_value": True}, {"entropy_coef": 0.01}, {"bounds_loss_coef": 0.1}, {"truncate_grads": False}, {"mixed_precision": False}):
        with pytest.raises(ValueError):
            U.FusedPpoUpdate(net, dict(c, **bad), 4096, 4, "cpu")
    wide = ppo.DyrosActorCritic(U.IN, U.ACT, dict(ppo.TRAIN_CFG["network"], mlp_units=[512, 256]))
    with pytest.raises(ValueError):
        U.FusedPpoUpdate(wide, c, 4096, 4, "cpu")
    assert (U.IN, U.INP, U.HID, U.OUTP, U.ACT) == (487, 512, 256, 16, 13) and U.NP == 2 * (256 * 512 + 256 * 256 + 16 * 256 + 256 + 256 + 16)
    with pytest.raises(ValueError):
        ppo.train(8, epochs=1, horizon=4, device="cpu", fused_update=True, env=type("E", (), {"num_envs": 8, "num_obs": U.IN, "num_acts": U.ACT})())


def test_operand_order_layouts_are_bijections():
    """include/dyros_ppo.h / csrc/dw_ppo.hip frag_pos, frag32_pos: the positions of a weight matrix's elements in the order the matrix instructions take
    them are a permutation of its row-major positions (restated here: a changed formula on one side shows as a size mismatch or a collision), and the
    copies' sizes are the header's."""
    import numpy as np
    from isaacgymdyros_amd import ppo_update as U

    def frag_pos(nt, row, k):          # v_mfma_f32_16x16x32_f16: eight halves of k = 32 kk + 8 g + j per lane (g * 16 + row % 16)
        return ((((k >> 5) * nt + (row >> 4)) * 64 + ((k & 31) >> 3) * 16 + (row & 15)) << 3) + (k & 7)

    def frag32_pos(nt, row, k):        # v_mfma_f32_16x16x4_f32: four words of k = 16 kg + 4 j + g per lane
        return ((((k >> 4) * nt + (row >> 4)) * 64 + (k & 3) * 16 + (row & 15)) << 2) + ((k >> 2) & 3)
    for fn in (frag_pos, frag32_pos):
        for rows, K in ((U.HID, U.INP), (U.HID, U.HID), (U.OUTP, U.HID)):
            r, k = np.meshgrid(np.arange(rows), np.arange(K), indexing="ij")
            pos = fn(rows // 16, r, k).ravel()
            assert pos.min() == 0 and pos.max() == rows * K - 1 and len(np.unique(pos)) == rows * K, (fn.__name__, rows, K)
    # the heads' input-gradient operand: rows = the 256 inputs, k = the 16 outputs padded to 32 (the upper half of every fragment stays zero)
    r, k = np.meshgrid(np.arange(U.HID), np.arange(U.OUTP), indexing="ij")
    pos = frag_pos(U.HID // 16, r, k).ravel()
    assert len(np.unique(pos)) == U.HID * U.OUTP and pos.max() < U.HID * 32
    nw = 2 * (U.HID * U.INP + U.HID * U.HID + U.OUTP * U.HID)
    assert U.K["DWP_P32F_WORDS"] == nw and U.K["DWP_P16F_WORDS"] == nw + 2 * U.HID * U.HID + 2 * U.HID * 32


def _c_kinds(args: str):
    """'const float *p, int32_t B, float x, void *stream' -> ['ptr', 'int', 'float', 'ptr']"""
    kinds = []
    for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
        kinds.append("ptr" if "*" in a else ("float" if a.split()[0] == "float" else "int"))
    return kinds


def test_ppo_ctypes_prototypes_match_the_header():
    """ctypes passes arguments by position and checks nothing: a C signature that gained a parameter while the binding kept the old list hands a
    kernel the NEXT argument as its pointer (the r5m4 memory fault).  Every dwp_* prototype of include/dyros_ppo.h against
    the argtypes ppo_update.declare() sets: same count, and pointer / int32 / float in the same places."""
    import ctypes as C
    import re
    from isaacgymdyros_amd import build, ppo_update as U
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dyros_ppo.h")).read(), flags=re.S)
    protos = {m.group(2): (m.group(1).strip(), m.group(3)) for m in re.finditer(r"\b(int|const char \*)\s*(dwp_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", src)}
    assert sorted(protos) == sorted("dwp_" + n for n in U.EXPORTS)
    api = U.declare(C.CDLL(build.build()))
    for name, (ret, args) in protos.items():
        f = api[name[4:]]
        got = ["ptr" if (t is C.c_void_p or t is C.c_char_p or hasattr(t, "contents") or issubclass(t, C._Pointer)) else ("float" if t is C.c_float else "int") for t in f.argtypes]
        assert got == _c_kinds(args), (name, got, _c_kinds(args))
        assert (f.restype is C.c_char_p) == (ret != "int"), name


def test_kernel_pointer_arguments_are_checked_before_a_launch():
    """ppo_update._req: what RolloutRecorder.pre / post, FusedPpoUpdate.policy and bind_batch run on every tensor whose data_ptr() goes to a
    kernel.  A bool mask where the kernel reads int64 would be read 8 x out of bounds: refused on the host, with ValueError."""
    from isaacgymdyros_amd import ppo_update as U
    ok = torch.zeros(8, dtype=torch.int64)
    for bad, kw in ((torch.zeros(8, dtype=torch.bool), dict(numel=8)),          # TocabiAMPLower's timeout_buf as time_outs
                    (torch.zeros(8, dtype=torch.int32), dict(numel=8)),
                    (ok[::2], dict(numel=4)),                                      # not contiguous
                    (ok, dict(numel=16)),                                          # too short for the launch
                    (torch.zeros(4, 2, dtype=torch.int64), dict(shape=(8, 1))),
                    ([0] * 8, dict(numel=8))):                                     # not a tensor
        with pytest.raises(ValueError):
            U._req("time_outs", bad, torch.int64, **kw)
    with pytest.raises(ValueError, match="GPU"):
        U._req("time_outs", ok, torch.int64, 8)          # right type and size, but host memory: no kernel may see its address


def test_sharded_fused_update_averages_gradients_with_one_collective_on_gloo(tmp_path):
    """The N > 1 path of the fused update (FusedPpoUpdate(world=2)): dwp_mlp | dwp_wgrad | dwp_grad_bucket | ONE all-reduce of the
    [weights | biases] bucket | dwp_stats_adam_finish (or dwp_grad_stats | dwp_adam_finish), the gradients averaged BEFORE the statistics and the step
    (a2c_continuous_seperate.py:171-180).  CPU, gloo, world 2; the library is a stand-in that works on the raw addresses it is handed (the
    kernels themselves are GPU tests): each rank's 'weight gradient' depends on its rank, and both ranks must apply the same average."""
    worker = tmp_path / "w.py"
    worker.write_text('''
import ctypes, importlib.util, json, os, sys, types
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from isaacgymdyros_amd import dist as dwdist, ppo_update as U
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(%r, "examples", "ppo_consumer.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
rank, _, world = dwdist.init_from_env("gloo")
NWT, NBT, SL = U.NWT, U.NBT, U.K["DWP_WGRAD_SLABS"]
view = lambda ptr, n: np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))
calls = []
def wgrad(xf, h1f, h2f, doutf, dz2f, dz1f, st, g32, B, s):
    calls.append("wgrad")
    g = view(g32, SL * NWT).reshape(SL, NWT)
    for k in range(SL):
        g[k] = (rank + 1) * (k + 1) * (1.0 + (np.arange(NWT) %% 7))          # this rank's partial gradients
    return 0
def mlp(args, s):
    calls.append("mlp")
    a = ctypes.cast(args, ctypes.POINTER(U.DwpMlp)).contents
    view(a.pbuf, U.K["DWP_PBUF_BUCKETS"] * 2 * U.K["DWP_PBUF_WORDS"])[:] = 0.25 * (rank + 1)          # bias-gradient buckets
    return 0
def grad_bucket(g32, pbuf, bucket, inv_world, s):
    calls.append("grad_bucket")
    b = view(bucket, NWT + NBT)
    b[:NWT] = view(g32, SL * NWT).reshape(SL, NWT).sum(0) * inv_world
    pb = view(pbuf, U.K["DWP_PBUF_BUCKETS"] * 2 * U.K["DWP_PBUF_WORDS"])
    b[NWT:] = U.K["DWP_PBUF_BUCKETS"] * pb[0] * inv_world
    pb[:] = 0.0
    return 0
seen = {}
def grad_stats(g16, gb, st, part, pbuf, g32, slabs, s):
    calls.append("grad_stats"); seen["stats"] = (g16, gb, pbuf, g32, slabs)
    return 0
def adam_finish(p, p16, mm, v, gb, st, part, max_norm, p16t, g32, slabs, p32f, B, nmb, gi, pbuf, s):
    calls.append("adam_finish"); seen["adam"] = (gb, g32, slabs)
    view(p, NWT + NBT)[:] -= 0.125 * np.concatenate([view(g32, NWT), view(gb, NBT)])
    return 0
def stats_adam_finish(p, p16, mm, v, gb, st, part, max_norm, p16t, g32, slabs, p32f, B, nmb, gi, pbuf, pbuf_bias, s):
    calls.append("stats_adam_finish"); seen["stats"] = (None, gb, pbuf_bias, g32, slabs); seen["adam"] = (gb, g32, slabs)
    view(p, NWT + NBT)[:] -= 0.125 * np.concatenate([view(g32, NWT), view(gb, NBT)])
    return 0
noop = lambda *a: 0
stub = dict(mlp=mlp, wgrad=wgrad, grad_bucket=grad_bucket, grad_stats=grad_stats, adam_finish=adam_finish, stats_adam_finish=stats_adam_finish, retile=noop, retile32=noop,
            last_error=lambda: b"stub")
U._lib.load = lambda: (None, None)
U.declare = lambda lib: stub
torch.cuda.current_stream = lambda d=None: types.SimpleNamespace(cuda_stream=0)
n_coll = [0]
real = dist.all_reduce
def counted(t, *a, **k):
    n_coll[0] += 1; seen["numel"] = t.numel()
    return real(t, *a, **k)
dist.all_reduce = counted
torch.manual_seed(0)
net = m.DyrosActorCritic(U.IN, U.ACT, m.TRAIN_CFG["network"])
f = U.FusedPpoUpdate(net, dict(m.TRAIN_CFG["config"]), 64, 2, "cpu", rowmajor=False, world=world, merged_tail=sys.argv[2] == "merged")
z = torch.zeros(128)
f.src = (torch.zeros(128, U.INP, dtype=torch.float16), torch.zeros(128, U.ACT), z, torch.zeros(128, U.ACT), z, z)
p0 = f.p.clone()
f.update(); f.update()
mine = f.p.clone()
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
gavg = sum((r + 1) * 10.0 for r in range(world)) / world * (1.0 + (torch.arange(NWT) %% 7).float())          # sum over slabs k + 1 = 10
bavg = sum(0.25 * (r + 1) for r in range(world)) / world * U.K["DWP_PBUF_BUCKETS"]
ok_w = bool(torch.allclose(p0[:NWT] - mine[:NWT], 2 * 0.125 * gavg, rtol=1e-6))
ok_b = bool(torch.allclose(p0[NWT:] - mine[NWT:], torch.full((NBT,), 2 * 0.125 * bavg), rtol=1e-6))
gb_expected = f.bucket.data_ptr() + 4 * NWT
json.dump({"same": bool(torch.equal(both[0], both[1])), "ok_w": ok_w, "ok_b": ok_b, "collectives": n_coll[0], "numel": seen["numel"],
           "order": calls[:len(calls) // 2], "stats_args_ok": seen["stats"] == (None, gb_expected, None, f.bucket.data_ptr(), 1),
           "adam_args_ok": seen["adam"] == (gb_expected, f.bucket.data_ptr(), 1)}, open(os.path.join(sys.argv[1], "f%%d.json" %% rank), "w"))
dist.barrier(); dist.destroy_process_group()
''' % (ROOT, ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    from isaacgymdyros_amd import ppo_update as U
    for form, tail in (("merged", ["stats_adam_finish"]), ("two", ["grad_stats", "adam_finish"])):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29747", str(worker), str(tmp_path), form], timeout=600, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        for k in range(2):
            o = json.load(open(tmp_path / ("f%d.json" % k)))
            assert o["same"] and o["ok_w"] and o["ok_b"], o
            assert o["collectives"] == 2 and o["numel"] == U.NWT + U.NBT, o          # ONE all-reduce per update, of the whole bucket
            assert o["order"] == ["mlp", "wgrad", "grad_bucket"] + tail and o["stats_args_ok"] and o["adam_args_ok"], (form, o)


def test_fused_update_refuses_a_world_it_cannot_serve():
    from isaacgymdyros_amd import ppo_update as U
    ppo = _mod()
    c = dict(ppo.TRAIN_CFG["config"])
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"])
    with pytest.raises(ValueError):
        U.FusedPpoUpdate(net, c, 4096, 4, "cpu", world=0)
    with pytest.raises(ValueError):          # the library-GEMM form has no sharded path
        U.FusedPpoUpdate(net, c, 4096, 4, "cpu", mfma=False, world=2)
    with pytest.raises(ValueError):          # no process group
        U.FusedPpoUpdate(net, c, 4096, 4, "cpu", world=2)
