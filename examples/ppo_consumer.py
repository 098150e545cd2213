#!/usr/bin/env python3
"""Minimal PPO loop that consumes the env exactly the way the reference's rl_games fork does (BASELINE config 3).

Call pattern mirrored from learning/rl_games_custom/a2c_common_dyros.py: env_reset (:480-483), then per epoch
play_steps (:629-703: policy forward, env_step (:467-478), time-out bootstrap (:656-659), done bookkeeping), GAE
(:485-500), and mini-epochs of the clipped PPO loss (common_losses.py:4-30) with SEPARATE actor / critic optimisers
(a2c_continuous_seperate.py:50-54).  Network and hyper-parameters from cfg/train/DyrosDynamicWalkPPO.yaml
(MLP 256-256, horizon 128, minibatch = whole batch, 5 mini-epochs).  rl_games itself is not a dependency; this
is the consumer side of the drop-in boundary, not part of the simulation step.  Reports step_fps / total_fps as
a2c_common_dyros.py:1005-1008 does.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgymdyros_amd.config import default_cfg                      # noqa: E402
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk     # noqa: E402


def mlp(i, o):
    return nn.Sequential(nn.Linear(i, 256), nn.ELU(), nn.Linear(256, 256), nn.ELU(), nn.Linear(256, o))


class ActorCritic(nn.Module):
    def __init__(self, num_obs, num_act):
        super().__init__()
        self.actor, self.critic = mlp(num_obs, num_act), mlp(num_obs, 1)
        self.log_std = nn.Parameter(torch.zeros(num_act))

    def dist(self, obs):
        return torch.distributions.Normal(self.actor(obs), self.log_std.exp())


def train(num_envs=16384, epochs=2, horizon=128, mini_epochs=5, gamma=0.99, lam=0.95, clip=0.2, lr=1e-4, device="cuda:0",
          log=print):
    env = DyrosDynamicWalk(default_cfg(num_envs, device), device, 0, True)
    net = ActorCritic(env.num_obs, env.num_acts).to(device)
    opt_a = torch.optim.Adam(list(net.actor.parameters()) + [net.log_std], lr=lr)
    opt_c = torch.optim.Adam(net.critic.parameters(), lr=lr)
    N, H = num_envs, horizon
    obs = env.reset()["obs"].clone()
    buf = dict(obs=torch.zeros(H, N, env.num_obs, device=device), act=torch.zeros(H, N, env.num_acts, device=device),
               logp=torch.zeros(H, N, device=device), val=torch.zeros(H, N, device=device),
               rew=torch.zeros(H, N, device=device), done=torch.zeros(H, N, device=device))
    stats = []
    for ep in range(epochs):
        t0 = time.perf_counter()
        step_time = 0.0
        with torch.no_grad():                                   # a2c_common_dyros.py:842
            for n in range(H):
                d = net.dist(obs)
                a = d.sample()
                buf["obs"][n], buf["act"][n] = obs, a
                buf["logp"][n], buf["val"][n] = d.log_prob(a).sum(-1), net.critic(obs).squeeze(-1)
                torch.cuda.synchronize(); ts = time.perf_counter()
                o, r, dones, infos = env.step(torch.clamp(a, -1.0, 1.0))
                torch.cuda.synchronize(); step_time += time.perf_counter() - ts
                r = r.clone()
                if "time_outs" in infos:                         # value bootstrap on time-outs (:656-659)
                    r += gamma * buf["val"][n] * infos["time_outs"].float()
                buf["rew"][n], buf["done"][n] = r, dones.float()
                obs = o["obs"].clone()
            last_val = net.critic(obs).squeeze(-1)
            adv = torch.zeros(H, N, device=device)
            gae = torch.zeros(N, device=device)
            for n in reversed(range(H)):                         # :485-500
                nv = last_val if n == H - 1 else buf["val"][n + 1]
                nonterm = 1.0 - buf["done"][n]
                delta = buf["rew"][n] + gamma * nv * nonterm - buf["val"][n]
                gae = delta + gamma * lam * nonterm * gae
                adv[n] = gae
            ret = adv + buf["val"]
        play_time = time.perf_counter() - t0
        B = {k: v.reshape(H * N, *v.shape[2:]) for k, v in buf.items()}
        A, Rt = adv.reshape(-1), ret.reshape(-1)
        A = (A - A.mean()) / (A.std() + 1e-8)
        for _ in range(mini_epochs):
            d = net.dist(B["obs"])
            ratio = (d.log_prob(B["act"]).sum(-1) - B["logp"]).exp()
            a_loss = torch.max(-A * ratio, -A * ratio.clamp(1 - clip, 1 + clip)).mean()
            opt_a.zero_grad(); a_loss.backward(); opt_a.step()
            c_loss = ((net.critic(B["obs"]).squeeze(-1) - Rt) ** 2).mean()
            opt_c.zero_grad(); c_loss.backward(); opt_c.step()
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        s = dict(epoch=ep, step_fps=H * N / step_time, play_fps=H * N / play_time, total_fps=H * N / total,
                 mean_reward=float(buf["rew"].mean()), a_loss=float(a_loss.detach()), c_loss=float(c_loss.detach()),
                 mean_episode_length=float(env.epi_len_log[env.episodes_finished > 0].mean()) if int((env.episodes_finished > 0).sum()) else 0.0)
        stats.append(s)
        log("epoch %(epoch)d: fps step %(step_fps).3g  step+inference %(play_fps).3g  total %(total_fps).3g  "
            "mean reward %(mean_reward).3f  episode length %(mean_episode_length).1f" % s)
    env.close()
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=16384)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--horizon", type=int, default=128)
    a = ap.parse_args()
    train(a.num_envs, a.epochs, a.horizon)
