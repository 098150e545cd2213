#!/usr/bin/env python3
"""The consumer side of the drop-in boundary: the DYROS PPO of the reference's rl_games fork, restated in plain torch
(BASELINE config 3: 16384 envs with the PPO training loop attached).

rl_games 1.1.4 is not installed here and is not part of the simulation step, so this file restates what the reference's
`learning/` code adds to it, with the YAML it is configured by (paths relative to python/IsaacGymEnvs/isaacgymenvs):

  network            separate actor / critic MLPs 256-256 relu, orthogonal init gain 0.01, linear mu / value heads, a FIXED
                     (non-trainable) log-sigma                           cfg/train/DyrosDynamicWalkPPO.yaml:10-38
  sigma schedule     log-sigma from -2.3026 to -2.9957 over the first half of training
                                                                          learning/rl_games_custom/models_dyros.py:64-70
  neglogp, forward   learning/rl_games_custom/models_dyros.py:27-62
  losses             clipped surrogate on exp(old_neglogp - neglogp), squared value error, bound loss on |mu| > 1.1
                     learning/rl_games_custom/common_losses.py:4-30, a2c_continuous_seperate.py:233-241
  optimisers         Adam(actor, lr schedule) and Adam(critic, 5e-4) stepped from ONE backward of
                     a_loss + 0.5 c_loss critic_coef - entropy entropy_coef + b_loss bounds_loss_coef; gradient-norm
                     clipping on the ACTOR parameters only                a2c_continuous_seperate.py:50-54,150-190
  rollout            play_steps with value bootstrap on time-outs, GAE, per-term reward logging from
                     extras["stacked_rewards"]                             a2c_common_dyros.py:485-500,629-703,1019-1030
  schedule           linear learning rate 1e-5 -> 3e-6 over max_epochs (rl_games common/schedulers.py LinearScheduler, a
                     dependency absent from the checkout: lr = min + (start - min) * max(0, max_steps - epoch) / max_steps)
  sharded training   one process per GPU, env shards, gradients averaged with an RCCL all-reduce (the reference:
                     Horovod, utils/rlgames_utils.py:71-81, a2c_continuous_seperate.py:171-180)
Reports step_fps / step-and-inference fps / total fps as a2c_common_dyros.py:1005-1008 does.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# cfg/train/DyrosDynamicWalkPPO.yaml, the values this loop reads (line numbers of that file)
TRAIN_CFG = {
    "network": {"separate": True, "mlp_units": [256, 256], "activation": "relu", "init_gain": 0.01,        # :12, :27-33
                "sigma_init": -2.302585, "sigma_last": -2.9957, "fixed_sigma": True},                      # :19-25
    "config": {
        "mixed_precision": True, "normalize_input": False, "normalize_value": False, "value_bootstrap": True,   # :57-60
        "clip_actions": True, "reward_scale": 1.0, "normalize_advantage": True, "gamma": 0.99, "tau": 0.95,     # :61-67
        "learning_rate": 1e-5, "learning_rate_min": 3e-6, "lr_schedule": "linear", "kl_threshold": 0.008,       # :68-71
        "max_epochs": 5000, "grad_norm": 0.5, "entropy_coef": 0.0, "truncate_grads": True, "e_clip": 0.2,       # :73-83
        "horizon_length": 128, "minibatch_size": 4096, "mini_epochs": 5, "critic_coef": 0.5, "clip_value": False,   # :84-88
        "bounds_loss_coef": 0.0, "num_rewards": 14, "separate_opt": True, "critic_lr": 5e-4,                    # :90, :94, :96; a2c_continuous_seperate.py:53
    },
}


def load_train_yaml(path: str, **root_overrides) -> dict:
    """TRAIN_CFG from the reference's own YAML (Hydra interpolations resolved like isaacgymdyros_amd.config.load_task_yaml)."""
    import yaml
    from isaacgymdyros_amd.config import ROOT_DEFAULTS, _resolve
    root = dict(ROOT_DEFAULTS, checkpoint="", experiment="", max_iterations="", **root_overrides)
    root["task"] = {"env": {"numEnvs": root.get("num_envs") or 4096}}

    def walk(x):
        if isinstance(x, dict):
            return {k: walk(v) for k, v in x.items()}
        if isinstance(x, list):
            return [walk(v) for v in x]
        if isinstance(x, str) and "task.env.numEnvs" in x:
            return root["task"]["env"]["numEnvs"]
        if isinstance(x, str) and (".name}" in x):
            return x
        return _resolve(x, root)
    p = walk(yaml.safe_load(open(path)))["params"]
    net, c = p["network"], p["config"]
    sp = net["space"]["continuous"]
    return {
        "network": {"separate": net["separate"], "mlp_units": net["mlp"]["units"], "activation": net["mlp"]["activation"],
                    "init_gain": net["mlp"]["initializer"]["gain"], "sigma_init": sp["sigma_init"]["val"],
                    "sigma_last": sp["sigma_last"]["val"], "fixed_sigma": sp["fixed_sigma"]},
        "config": {
            "mixed_precision": c["mixed_precision"], "normalize_input": c["normalize_input"], "normalize_value": c["normalize_value"],
            "value_bootstrap": c["value_bootstrap"], "clip_actions": c["clip_actions"], "reward_scale": c["reward_shaper"]["scale_value"],
            "normalize_advantage": c["normalize_advantage"], "gamma": c["gamma"], "tau": c["tau"],
            "learning_rate": float(c["learning_rate"]), "learning_rate_min": float(c["learning_rate_min"]), "lr_schedule": c["lr_schedule"],
            "kl_threshold": c["kl_threshold"], "max_epochs": c["max_epochs"], "grad_norm": c["grad_norm"],
            "entropy_coef": c["entropy_coef"], "truncate_grads": c["truncate_grads"], "e_clip": c["e_clip"],
            "horizon_length": c["horizon_length"], "minibatch_size": c["minibatch_size"], "mini_epochs": c["mini_epochs"],
            "critic_coef": c["critic_coef"], "clip_value": c["clip_value"], "bounds_loss_coef": float(c["bounds_loss_coef"]),
            "num_rewards": c["num_rewards"], "separate_opt": c["separate_opt"], "critic_lr": 5e-4,
        },
    }


# ------------------------------------------------------------------------------------------------ model
def _mlp(inp, units, gain):
    layers, d = [], inp
    for u in units:
        lin = nn.Linear(d, u)
        nn.init.orthogonal_(lin.weight, gain=gain)           # orthogonal_initializer, gain 0.01 (yaml:31-33)
        nn.init.zeros_(lin.bias)
        layers += [lin, nn.ReLU()]
        d = u
    return nn.Sequential(*layers), d


class DyrosActorCritic(nn.Module):
    """network_builder_dyros.py `actor_critic_dyros` with `separate: True`: two MLP trunks, mu and value heads, a fixed
    log-sigma parameter (requires_grad False, :104) moved by update_action_noise."""

    def __init__(self, num_obs, num_act, net_cfg=None):
        super().__init__()
        nc = net_cfg or TRAIN_CFG["network"]
        self.actor_mlp, d = _mlp(num_obs, nc["mlp_units"], nc["init_gain"])
        self.critic_mlp, _ = _mlp(num_obs, nc["mlp_units"], nc["init_gain"])
        self.mu, self.value = nn.Linear(d, num_act), nn.Linear(d, 1)
        self.sigma_init, self.sigma_last = float(nc["sigma_init"]), float(nc["sigma_last"])
        self.sigma = nn.Parameter(torch.full((num_act,), self.sigma_init), requires_grad=False)

    def forward(self, obs):
        mu = self.mu(self.actor_mlp(obs))
        value = self.value(self.critic_mlp(obs))
        return mu, mu * 0.0 + self.sigma, value

    def update_action_noise(self, progress_remaining: float):
        """models_dyros.py:64-70: sigma_init -> sigma_last over the first half of training, constant afterwards."""
        b = 2 * progress_remaining - 1 if progress_remaining > 0.5 else 0.0
        self.sigma[:] = self.sigma_init * b + self.sigma_last * (1 - b)

    def actor_parameters(self):
        return list(self.actor_mlp.parameters()) + list(self.mu.parameters())

    def critic_parameters(self):
        return list(self.critic_mlp.parameters()) + list(self.value.parameters())


def neglogp(x, mean, std, logstd):
    """models_dyros.py:59-62"""
    return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] + logstd.sum(dim=-1)


# ------------------------------------------------------------------------------------------------ losses, returns, schedules
def actor_loss(old_neglogp, new_neglogp, advantage, e_clip):
    """common_losses.py:17-26 (ppo branch): returns (per-sample loss, clip fraction)."""
    ratio = torch.exp(old_neglogp - new_neglogp)
    surr1 = advantage * ratio
    surr2 = advantage * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)
    return torch.max(-surr1, -surr2), torch.mean((torch.abs(ratio - 1) > e_clip).float())


def critic_loss(value_preds, values, e_clip, returns, clip_value):
    """common_losses.py:4-14"""
    if clip_value:
        clipped = value_preds + (values - value_preds).clamp(-e_clip, e_clip)
        return torch.max((values - returns) ** 2, (clipped - returns) ** 2)
    return (returns - values) ** 2


def bound_loss(mu, soft_bound=1.1):
    """a2c_continuous_seperate.py:233-241"""
    hi = torch.clamp_max(mu - soft_bound, 0.0) ** 2
    lo = torch.clamp_max(-mu + soft_bound, 0.0) ** 2
    return (lo + hi).sum(dim=-1)


def discount_values(fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma, tau):
    """GAE exactly as a2c_common_dyros.py:485-500.  Shapes: [H, N, 1] for the mb_* value tensors, [H, N] for mb_fdones."""
    H = mb_rewards.shape[0]
    lastgaelam = 0
    advs = torch.zeros_like(mb_rewards)
    for t in reversed(range(H)):
        if t == H - 1:
            nextnonterminal, nextvalues = 1.0 - fdones, last_values
        else:
            nextnonterminal, nextvalues = 1.0 - mb_fdones[t + 1], mb_values[t + 1]
        nextnonterminal = nextnonterminal.unsqueeze(1)
        delta = mb_rewards[t] + gamma * nextvalues * nextnonterminal - mb_values[t]
        advs[t] = lastgaelam = delta + gamma * tau * nextnonterminal * lastgaelam
    return advs


def policy_kl(mu, sigma, old_mu, old_sigma):
    """rl_games torch_ext.policy_kl (mean over the batch)."""
    c1 = torch.log(sigma / old_sigma + 1e-5)
    c2 = (old_sigma ** 2 + (old_mu - mu) ** 2) / (2.0 * (sigma ** 2 + 1e-5))
    return (c1 + c2 - 0.5).sum(dim=-1).mean()


class LinearLR:
    """rl_games common/schedulers.py LinearScheduler (schedule_type legacy: by epoch)."""

    def __init__(self, start_lr, min_lr, max_steps):
        self.start_lr, self.min_lr, self.max_steps = float(start_lr), float(min_lr), int(max_steps)

    def __call__(self, epoch):
        mul = max(0, self.max_steps - epoch) / self.max_steps
        return self.min_lr + (self.start_lr - self.min_lr) * mul


def allreduce_grads(params, world: int, group=None):
    """Average the gradients over the ranks with ONE collective on a flat bucket (RCCL over xGMI on the GPUs, gloo in the
    CPU test): what Horovod's optimizer.synchronize() does in the reference (a2c_continuous_seperate.py:171-173)."""
    if world <= 1:
        return
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat /= world
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n


def _sync(device):
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ the loop
ROLL_CHAIN = int(os.environ.get("DW_PPO_ROLL_CHAIN", "8"))          # rollout steps per replayed graph
UPD_CHAIN = int(os.environ.get("DW_PPO_CHAIN", "16"))          # fused updates per replayed graph
# sharded fused update: the all-reduce captured inside the graph of updates (default; an RCCL all-reduce is accepted by a hipGraph capture on this
# stack, profiles/r06_graph_collective_probe.txt) -- if the capture is refused, or with DW_PPO_GRAPH_COLLECTIVE=0, the update is two graphs with
# the collective enqueued between them (measured on one rank at 16384 envs: 14.51 M captured, 12.05 M split, 14.68 M unsharded)
GRAPH_COLLECTIVE = os.environ.get("DW_PPO_GRAPH_COLLECTIVE", "1") == "1"
# the update's statistics + Adam + scaler as ONE launch with a grid barrier (dwp_stats_adam_finish) -- three launches per update; 0: as two
MERGED_TAIL = os.environ.get("DW_PPO_MERGED_TAIL", "0") == "1"
# the rollout policy's fp32 operand-order copy of the weights: rewritten by every update's Adam launch (1), or once per epoch after the updates (0)
POLICY_COPY_PER_UPDATE = os.environ.get("DW_PPO_POLICY_COPY_PER_UPDATE", "0") == "1"


def train(num_envs=16384, epochs=2, horizon=None, device="cuda:0", log=print, cfg=None, max_epochs=None, env=None,
          rank=0, world=1, seed=42, graph_rollout=False, graph_update=False, fused_update=False, fused_collective=None, global_gate=False):
    """`epochs` PPO epochs of the DYROS configuration on `num_envs` envs of this rank.  Returns one stats dict per epoch.
    graph_rollout: one rollout step (policy inference, sampling, env step, bookkeeping) is captured once in a hipGraph and
    replayed `horizon` times per epoch -- possible because dw_step_dev keeps the step counter in device memory, so a replayed
    launch draws fresh noise (include/dyros_walk.h).  The eager loop pays ~40 kernel launches and two host syncs per step.
    graph_update: one minibatch update (forward, the four losses, backward, unscale, clip, both optimiser steps, scaler update) is
    captured once and replayed 5 x 512 times per epoch; needs the fused, capturable Adam (its update is what GradScaler can skip
    on the device instead of asking the host), one rank.
    fused_update: the same update as FOUR launches on the matrix cores (include/dyros_ppo.h, isaacgymdyros_amd/ppo_update.py), captured once
    and replayed; GPU only.  Sharded (world > 1) every update carries ONE all-reduce of the ranks' 1.61 MB gradient bucket between the
    weight-gradient launch and the statistics (FusedPpoUpdate.allreduce: where the reference's Horovod optimizer.synchronize() stands,
    a2c_continuous_seperate.py:171-180), captured inside the chain of updates (one replay per UPD_CHAIN updates); if the capture of the
    collective is refused, or with DW_PPO_GRAPH_COLLECTIVE=0, the update is two replayed graphs with the collective enqueued between them.
    fused_collective=True runs that sharded form of the update on ONE rank too (tests: same bits as the plain four launches).
    global_gate (sharded runs): once per horizon the ranks' push-perturbation gates latch on the means over ALL envs
    (env.sync_perturbation_gate: one all-reduce of 3 doubles) instead of per rank as in the reference's Horovod layout."""
    from isaacgymdyros_amd.config import default_cfg
    from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
    cfg = cfg or TRAIN_CFG
    c = cfg["config"]
    H = int(horizon or c["horizon_length"])
    max_epochs = int(max_epochs or c["max_epochs"])
    own_env = env is None
    if own_env:
        ecfg = default_cfg(num_envs, device)
        ecfg["seed"] = seed + rank
        if graph_rollout:
            ecfg["sim"]["mi355"]["device_step_counter"] = True
            ecfg["sim"]["mi355"]["alias_obs"] = True          # (the loop copies what it keeps)
        env = DyrosDynamicWalk(ecfg, device, 0, True)
    N = env.num_envs
    torch.manual_seed(seed)                          # same initial weights on every rank
    net = DyrosActorCritic(env.num_obs, env.num_acts, cfg["network"]).to(device)
    torch.manual_seed(seed + 7919 * rank)            # ... but its own exploration noise (Normal.sample draws from the global generator)
    graph_update = bool(graph_update) and str(device).startswith("cuda") and world == 1
    fused, upd_chain, upd_tail = None, None, None
    if fused_update:
        if not str(device).startswith("cuda"):
            raise ValueError("fused_update needs a GPU")
        from isaacgymdyros_amd.ppo_update import FusedPpoUpdate
        _b = int(horizon or c["horizon_length"]) * env.num_envs
        _m = min(int(c["minibatch_size"]), _b)
        # (re-points the module's parameters at its master buffer: before any capture.  Sharded: every rank starts from the same weights --
        #  the seed above -- and applies the same averaged gradient, so the ranks stay in step without a broadcast)
        fused = FusedPpoUpdate(net, c, _m, _b // _m, device, rowmajor=False, merged_tail=MERGED_TAIL, policy_copy_per_update=POLICY_COPY_PER_UPDATE,
                               world=world, collective=fused_collective)
        graph_update = False
    if graph_update:        # (learning rates as device tensors: the schedule writes them in place and the captured step reads them)
        opt_a = torch.optim.Adam(net.actor_parameters(), lr=torch.tensor(float(c["learning_rate"]), device=device), eps=1e-8, fused=True, capturable=True)
        opt_c = torch.optim.Adam(net.critic_parameters(), lr=torch.tensor(float(c["critic_lr"]), device=device), eps=1e-8, fused=True, capturable=True)
    else:
        opt_a = torch.optim.Adam(net.actor_parameters(), lr=c["learning_rate"], eps=1e-8)
        opt_c = torch.optim.Adam(net.critic_parameters(), lr=c["critic_lr"], eps=1e-8)
    upd_graph, upd_static, upd_out = None, None, {}
    sched = LinearLR(c["learning_rate"], c["learning_rate_min"], max_epochs)
    amp = bool(c["mixed_precision"]) and str(device).startswith("cuda")
    scaler = torch.amp.GradScaler("cuda", enabled=amp)
    names = list(env.extras.get("reward_names", []))
    obs = env.reset()["obs"].clone()
    dones = torch.zeros(N, device=device)
    if str(device).startswith("cuda"):
        # one untimed policy + env step: the first use of every torch kernel in the loop loads its code object, which on a
        # fresh process costs tens of milliseconds and would otherwise be booked as env-step time of the first epoch
        with torch.no_grad():
            mu, logstd, _ = net(obs)
            a = torch.distributions.Normal(mu, torch.exp(logstd)).sample()
            obs = env.step(torch.clamp(a, -1.0, 1.0))[0]["obs"].clone()
        torch.cuda.synchronize()
    # (fused update behind a captured rollout: the observations are recorded straight into the update's env-major flat batch, below)
    direct_obs = fused is not None and graph_rollout
    mb = dict(obs=None if direct_obs else torch.zeros(H, N, env.num_obs, device=device), act=torch.zeros(H, N, env.num_acts, device=device),
              neglogp=torch.zeros(H, N, device=device), val=torch.zeros(H, N, 1, device=device), rew=torch.zeros(H, N, 1, device=device),
              done=torch.zeros(H, N, device=device), mu=torch.zeros(H, N, env.num_acts, device=device))
    batch = H * N
    mbs = min(int(c["minibatch_size"]), batch)
    assert batch % mbs == 0, "horizon * num_envs must be a multiple of minibatch_size (a2c_common_dyros.py:192)"
    stats = []
    graph, recorder = None, None
    if graph_rollout:
        if not str(device).startswith("cuda") or getattr(env, "_step_dev", None) is None:
            raise ValueError("graph_rollout needs a GPU env with cfg sim.mi355.device_step_counter = True")
        g_obs, g_dones = obs.clone(), dones.clone()
        g_n = torch.zeros(1, dtype=torch.long, device=device)
        g_terms = torch.zeros(len(names) or 15, device=device)

        if fused is not None:          # (the step's bookkeeping in two launches instead of ~30: isaacgymdyros_amd/ppo_update.py::RolloutRecorder)
            from isaacgymdyros_amd.ppo_update import RolloutRecorder
            # (static homes of the epoch's flat arrays: a captured update replays their addresses)
            # (the observations as fp16 rows of 512: what the update's first layer reads -- the cast is done once, by the recorder)
            from isaacgymdyros_amd.ppo_update import INP as _INP
            fused.bind_batch(torch.zeros(batch, _INP, device=device, dtype=torch.float16),
                             *[torch.empty(batch, *sh, device=device) for sh in ((env.num_acts,), (), (env.num_acts,), (), ())])
            recorder = RolloutRecorder(mb, g_n, net.sigma, c["reward_scale"], c["gamma"], c["value_bootstrap"], obs_env_major=fused.src[0], num_obs=env.num_obs)
            if env.obs_dict["obs"].data_ptr() == env.obs_buf.data_ptr():
                # (alias_obs: the env's own buffer is the policy's input -- the recorder has copied it into the batch before env.step
                #  overwrites it, so the step's 32 MB copy into a second home falls away)
                g_obs = env.obs_buf

        pol = (torch.empty(N, env.num_acts, device=device), torch.empty(N, 1, device=device)) if fused is not None else None

        def rollout_step():
            if pol is not None:          # (the fp32 forward of both nets in one launch on the matrix cores: FusedPpoUpdate.policy)
                mu, value = fused.policy(g_obs, *pol)
            else:
                mu, logstd, value = net(g_obs)
            if recorder is not None:
                act = recorder.pre(mu, value, torch.randn_like(mu), g_obs, g_dones)
                o, r, d, infos = env.step(act)
                st = infos.get("stacked_rewards")
                recorder.post(r, value, infos.get("time_outs"), st if st is not None and st.is_contiguous() else None, d, o["obs"], g_terms, g_dones, g_obs)
                g_n.add_(1)
                return
            sigma = torch.exp(logstd)
            a = mu + sigma * torch.randn_like(mu)           # (Normal(mu, sigma).sample() and torch.normal check sigma >= 0 on the host: a sync, not capturable)
            for k, v in (("obs", g_obs), ("act", a), ("mu", mu), ("neglogp", neglogp(a, mu, sigma, logstd)), ("val", value), ("done", g_dones)):
                mb[k].index_copy_(0, g_n, v.unsqueeze(0))
            o, r, d, infos = env.step(torch.clamp(a, -1.0, 1.0))
            r = r.unsqueeze(1) * c["reward_scale"]
            if c["value_bootstrap"] and "time_outs" in infos:
                r = r + c["gamma"] * value * infos["time_outs"].unsqueeze(1).float()
            mb["rew"].index_copy_(0, g_n, r.unsqueeze(0))
            if "stacked_rewards" in infos:
                g_terms.add_(infos["stacked_rewards"][:, :g_terms.numel()].mean(0))
            g_dones.copy_(d.float())
            g_obs.copy_(o["obs"])
            g_n.add_(1)

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):                                                  # (warm-up on the side stream, as capture requires)
                g_n.zero_()
                rollout_step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        g_n.zero_()
        roll_chain = ROLL_CHAIN if H % ROLL_CHAIN == 0 else 1          # (steps per replayed graph: the step counter and the buffers' row index live on the device)
        with torch.no_grad(), torch.cuda.graph(graph, stream=side):
            for _ in range(roll_chain):
                rollout_step()
        obs, dones = g_obs, g_dones
    for ep in range(1, epochs + 1):
        net.update_action_noise((max_epochs - ep) / max_epochs)                 # a2c_common_dyros.py:985
        lr = sched(ep)
        if fused is not None:
            fused.set_learning_rates(lr, float(c["critic_lr"]))
        for g in opt_a.param_groups:                                           # update_lr touches the actor only (:293-295)
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr
        t0 = time.perf_counter()
        step_time = 0.0
        terms = torch.zeros(len(names) or 15, device=device)
        with torch.no_grad():                                                   # a2c_common_dyros.py:842
            if graph is not None:
                g_n.zero_(); g_terms.zero_()
                for n in range(H // roll_chain):
                    graph.replay()
                terms = g_terms.clone()
                step_time = float("nan")                                        # (the env step is not separable inside the graph)
            for n in range(H if graph is None else 0):
                mu, logstd, value = net(obs)
                sigma = torch.exp(logstd)
                a = torch.distributions.Normal(mu, sigma).sample()
                mb["obs"][n], mb["act"][n], mb["mu"][n] = obs, a, mu
                mb["neglogp"][n], mb["val"][n], mb["done"][n] = neglogp(a, mu, sigma, logstd), value, dones
                _sync(device); ts = time.perf_counter()
                o, r, d, infos = env.step(torch.clamp(a, -1.0, 1.0))            # clip_actions (:467-478)
                _sync(device); step_time += time.perf_counter() - ts
                r = r.unsqueeze(1) * c["reward_scale"]
                if c["value_bootstrap"] and "time_outs" in infos:               # :656-659
                    r = r + c["gamma"] * value * infos["time_outs"].unsqueeze(1).float()
                mb["rew"][n] = r
                if "stacked_rewards" in infos:                                  # :667-669 -> per-term means for the logger
                    terms += infos["stacked_rewards"][:, :terms.numel()].mean(0)
                dones = d.float()
                obs = o["obs"].clone()
            last_values = net(obs)[2]
            if fused is not None:          # (the same recursion in one launch instead of 128 x 8: isaacgymdyros_amd/ppo_update.py::gae)
                from isaacgymdyros_amd.ppo_update import gae as _gae
                advs = _gae(dones, last_values, mb["done"], mb["val"], mb["rew"], c["gamma"], c["tau"])
            else:
                advs = discount_values(dones, last_values, mb["done"], mb["val"], mb["rew"], c["gamma"], c["tau"])
            returns = advs + mb["val"]
        if global_gate and world > 1 and hasattr(env, "sync_perturbation_gate"):
            env.sync_perturbation_gate()
        if graph is not None and recorder is not None:
            recorder.rows()                                                     # (the device row counter must stand at H: a host read, next to the sync below)
        _sync(device)                                                           # (the rollout's device work is part of play_time in both modes)
        play_time = time.perf_counter() - t0
        # swap_and_flatten01: env-major flat batch, minibatches are contiguous slices (no shuffling in rl_games' dataset)
        flat = lambda x: x.transpose(0, 1).reshape(batch, *x.shape[2:])        # noqa: E731
        if fused is not None and fused.src is not None:
            # (the observations are in their static home, env-major, already -- or get there in one strided copy of 4 GB instead of two)
            if not direct_obs:
                fused.src[0].view(N, H, -1).copy_(mb["obs"].transpose(0, 1))
            B = {k: (fused.src[0] if k == "obs" else flat(v)) for k, v in mb.items()}
        else:
            B = {k: flat(v) for k, v in mb.items()}
        ret, val = flat(returns), B["val"]
        adv = (ret - val).sum(dim=1)
        if c["normalize_advantage"]:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)                       # :945
        a_l = c_l = b_l = cf = kl = torch.zeros((), device=device)

        def minibatch_update(obs_, act_, nlp_old, mu_old, adv_, ret_, val_):
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                mu, logstd, value = net(obs_)
                sigma = torch.exp(logstd)
                nlp = neglogp(act_, mu, sigma, logstd)
                a_loss, cf_ = actor_loss(nlp_old, nlp, adv_, c["e_clip"])
                c_loss = critic_loss(val_, value, c["e_clip"], ret_, c["clip_value"])
                b_loss = bound_loss(mu)
                # Normal(mu, sigma).entropy() written out (torch/distributions/normal.py): the constructor's argument check is a host sync
                entropy = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(sigma)).sum(dim=-1)
                al, cl, bl = a_loss.mean(), c_loss.mean(), b_loss.mean()
                loss = al + 0.5 * cl * c["critic_coef"] - entropy.mean() * c["entropy_coef"] + bl * c["bounds_loss_coef"]
            for p in net.parameters():
                p.grad = None
            scaler.scale(loss).backward()
            # synchronise first, unscale after (a2c_continuous_seperate.py:171-175): the still-scaled gradients are averaged
            # (the loss scale is the same on every rank), so an overflow on one rank reaches every rank through the sum,
            # every rank's unscale_ finds it, every rank skips the step and backs its scale off alike
            allreduce_grads(net.actor_parameters() + net.critic_parameters(), world)
            scaler.unscale_(opt_a); scaler.unscale_(opt_c)
            if c["truncate_grads"]:
                nn.utils.clip_grad_norm_(net.actor_parameters(), c["grad_norm"])       # the actor only (:178)
            scaler.step(opt_a); scaler.step(opt_c); scaler.update()
            with torch.no_grad():
                kl_ = policy_kl(mu.detach().float(), sigma.detach().float(), mu_old, torch.exp(net.sigma).expand_as(mu))
            return al.detach(), cl.detach(), bl.detach(), cf_.detach(), kl_

        if fused is not None:
            if fused.src is None:          # (static homes of the epoch's flat arrays: a captured update replays their addresses)
                fused.bind_batch(*[torch.empty(batch, *sh, device=device) for sh in ((env.num_obs,), (env.num_acts,), (), (env.num_acts,), (), ())])
            for d_, x in zip(fused.src, (B["obs"], B["act"], B["neglogp"], B["mu"], adv, ret)):
                if d_.data_ptr() != x.data_ptr():
                    d_.copy_(x.reshape(d_.shape))
            fused.rewind()
            if os.environ.get("DW_PPO_TIMES"):
                _sync(device); t_prep = time.perf_counter() - t0 - play_time
            n_upd = int(c["mini_epochs"]) * (batch // mbs)
            done_upd = 0
            if upd_graph is None:
                # (two updates run eagerly on a side stream: the warm-up a capture requires -- GEMM workspaces --, spent on real work)
                side_u = torch.cuda.Stream(device=device)
                side_u.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side_u), torch.no_grad():
                    for _ in range(min(2, n_upd)):
                        fused.update()
                        done_upd += 1
                torch.cuda.current_stream(device).wait_stream(side_u)
                torch.cuda.synchronize()
                split = fused.collective and not GRAPH_COLLECTIVE
                if done_upd < n_upd and not split:
                    try:
                        upd_graph = torch.cuda.CUDAGraph()
                        with torch.no_grad(), torch.cuda.graph(upd_graph, stream=side_u):
                            fused.update()
                        # (the minibatch index lives on the device, so a graph may hold any number of updates: UPD_CHAIN of them back to back
                        #  have no gap between graph launches inside)
                        if n_upd >= 2 * UPD_CHAIN:
                            upd_chain = torch.cuda.CUDAGraph()
                            with torch.no_grad(), torch.cuda.graph(upd_chain, stream=side_u):
                                for _ in range(UPD_CHAIN):
                                    fused.update()
                    except RuntimeError as err:          # (a collective the capture refuses: fall back to the split form below)
                        if fused.world <= 1:
                            raise
                        log("fused update: the capture of the all-reduce was refused (%s); two graphs per update instead" % str(err).split("\n")[0])
                        torch.cuda.synchronize()
                        upd_graph = upd_chain = None
                        split = True
                if done_upd < n_upd and split:
                    # sharded, the collective outside the graphs: head (dwp_mlp, dwp_wgrad, dwp_grad_bucket) | all-reduce | tail (statistics, Adam)
                    upd_graph, upd_tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.no_grad(), torch.cuda.graph(upd_graph, stream=side_u):
                        fused.update_head()
                    with torch.no_grad(), torch.cuda.graph(upd_tail, stream=side_u):
                        fused.update_tail()
            left = n_upd - done_upd
            while upd_chain is not None and left >= UPD_CHAIN:
                upd_chain.replay()
                left -= UPD_CHAIN
            for _ in range(left):
                upd_graph.replay()
                if upd_tail is not None:
                    fused.allreduce()
                    upd_tail.replay()
            if not fused.policy_copy_per_update:
                fused.sync_policy_copy()          # (the next rollout's dwp_policy reads the weights in its own operand order: one launch per epoch)
            lg = fused.logged()
            a_l, c_l, b_l, cf, kl = lg[0], lg[1], lg[2], lg[3], lg[4]
        srcs = (B["obs"], B["act"], B["neglogp"], B["mu"], adv, ret, val)
        if graph_update and upd_static is None:
            upd_static = [torch.empty_like(x[:mbs]) for x in srcs]
        it = 0
        for _ in range(int(c["mini_epochs"]) if fused is None else 0):
            for i in range(batch // mbs):
                sl = slice(i * mbs, (i + 1) * mbs)
                if not graph_update:
                    a_l, c_l, b_l, cf, kl = minibatch_update(*[x[sl] for x in srcs])
                    continue
                for d_, x in zip(upd_static, srcs):
                    d_.copy_(x[sl])
                if upd_graph is None and it < 3:
                    # (the first three minibatches run eagerly on a side stream: the warm-up a capture requires, spent on real work)
                    side_u = torch.cuda.Stream(device=device)
                    side_u.wait_stream(torch.cuda.current_stream(device))
                    with torch.cuda.stream(side_u):
                        outs = minibatch_update(*upd_static)
                    torch.cuda.current_stream(device).wait_stream(side_u)
                    a_l, c_l, b_l, cf, kl = outs
                else:
                    if upd_graph is None:
                        torch.cuda.synchronize()
                        for p in net.parameters():
                            p.grad = None
                        upd_graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(upd_graph):
                            upd_out["v"] = minibatch_update(*upd_static)
                    upd_graph.replay()
                    a_l, c_l, b_l, cf, kl = upd_out["v"]
                it += 1
        _sync(device)
        total = time.perf_counter() - t0
        fin = env.episodes_finished > 0
        s = dict(epoch=ep, step_fps=H * N / step_time, play_fps=H * N / play_time, total_fps=H * N / total,
                 mean_reward=float(mb["rew"].mean()), a_loss=float(a_l), c_loss=float(c_l), b_loss=float(b_l), clip_frac=float(cf),
                 kl=float(kl), lr=lr, sigma=float(net.sigma[0]),
                 reward_terms={(names[i] if i < len(names) else "term%d" % i): float(terms[i] / H) for i in range(terms.numel())},
                 mean_episode_length=float(env.epi_len_log[fin].mean()) if int(fin.sum()) else 0.0)
        if os.environ.get("DW_PPO_TIMES") and fused is not None:
            s.update(play_ms=1e3 * play_time, prep_ms=1e3 * t_prep, update_ms=1e3 * (total - play_time - t_prep))
        stats.append(s)
        if rank == 0:
            log("epoch %(epoch)d: fps step %(step_fps).3g  step+inference %(play_fps).3g  total %(total_fps).3g  mean reward %(mean_reward).3f  "
                "a_loss %(a_loss).3g  c_loss %(c_loss).3g  kl %(kl).2g  lr %(lr).2g  log-sigma %(sigma).3f  episode length %(mean_episode_length).1f" % s)
    if own_env:
        env.close()
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=16384, help="envs per GPU")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--fused", action="store_true", help="rollout step and minibatch update replayed from hipGraphs, the update, the rollout's "
                    "forward, its bookkeeping and GAE as the HIP kernels of include/dyros_ppo.h (14.7 M frames/s at 16384 envs on one GPU instead of 0.35 M "
                    "eager); under torchrun every update carries one RCCL all-reduce of the gradient bucket")
    a = ap.parse_args()
    from isaacgymdyros_amd import dist as dwdist
    rank, local_rank, world = dwdist.init_from_env("nccl")
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    train(a.num_envs, a.epochs, a.horizon, device=dev, rank=rank, world=world, graph_rollout=a.fused, fused_update=a.fused)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
