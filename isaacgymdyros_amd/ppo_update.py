"""The fused minibatch update of the on-GPU PPO consumer (include/dyros_ppo.h).

Reference: `calc_gradients` of learning/rl_games_custom/a2c_continuous_seperate.py:108-193 for the DyrosDynamicWalk configuration
(cfg/train/DyrosDynamicWalkPPO.yaml: two separate [256, 256] relu MLPs, mixed_precision, separate_opt, truncate_grads with
grad_norm 0.5 on the actor, e_clip 0.2, clip_value False, entropy_coef 0, bounds_loss_coef 0).  Two forms of one update:

  * the default (`mfma=True`), FOUR launches on the matrix cores: dwp_mlp | dwp_wgrad | dwp_grad_stats | dwp_adam_finish.  Sharded over
    several ranks (`world > 1`): dwp_mlp | dwp_wgrad | dwp_grad_bucket | ONE all-reduce of the 1.61 MB gradient bucket | dwp_grad_stats |
    dwp_adam_finish -- the still-scaled gradients are averaged before unscale / clip / step, as the reference's Horovod
    `optimizer.synchronize()` does (a2c_continuous_seperate.py:171-180).  `merged_tail=True` runs the last two launches as one
    (dwp_stats_adam_finish: the same time, off by default); `policy_copy_per_update=False` leaves the rollout policy's fp32 copy of the
    weights to one `sync_policy_copy()` per epoch (what examples/ppo_consumer.py does).
  * the library-GEMM form (`mfma=False`), 17 launches:
    stage (1 launch) | 3 batched GEMMs + 2 bias-relu | loss (1) | 5 batched GEMMs + 2 relu-backward | grad stats, Adam, finish (3)

instead of the ~190 of torch's autograd under autocast.  Arithmetic types: fp16 operands and outputs with fp32 accumulation in the products
(what autocast gives nn.Linear), fp32 in the losses, fp32 master parameters and Adam moments, dynamic loss scaling as
torch.amp.GradScaler does it.  Weight gradients: fp16 in the 17-launch form (what a backward under autocast produces); in the
four-launch form they stay fp32 sums of fp16 products -- a deviation towards more bits: found_inf then fires on an overflow of the fp16
output / activation gradients only, so near the fp16 range the loss scale backs off later than the reference's would, and the actor's
clip norm is taken from unrounded gradients (include/dyros_ppo.h).  Actor and critic have the same shapes, so in the 17-launch form
each layer is ONE batched GEMM (batch index 0 = actor, 1 = critic; torch.baddbmm / torch.bmm: hipBLASLt / rocBLAS); the kernels are
csrc/dw_ppo.hip.  Nothing here has a CPU form: the class raises without the HIP library.

The network's own parameters (`DyrosActorCritic` of examples/ppo_consumer.py, any module with actor_mlp / critic_mlp / mu / value)
are re-pointed at views of the flat fp32 master buffer, so the module used for the rollout sees every update."""
from __future__ import annotations

import ctypes as C
import re
import os

import torch

from . import _lib


def _constants() -> dict:
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dyros_ppo.h")).read()
    return {k: int(v) for k, v in re.findall(r"#define\s+(DWP_[A-Z0-9_]+)\s+(\d+)", src)}


K = _constants()
EXPORTS = ["abi_version", "last_error", "sizeof_mlp", "stage_obs", "bias_relu", "loss", "relu_bwd", "grad_stats", "grad_bucket", "adam", "adam_finish", "stats_adam_finish", "finish", "retile", "mlp", "wgrad", "gae", "rollout_pre", "rollout_post", "policy", "retile32"]
IN, INP, HID, OUTP, ACT = K["DWP_IN"], K["DWP_INP"], K["DWP_HID"], K["DWP_OUTP"], K["DWP_ACT"]
NW1, NW2, NW3 = 2 * HID * INP, 2 * HID * HID, 2 * OUTP * HID
NWT = NW1 + NW2 + NW3
NB1, NB2, NB3 = 2 * HID, 2 * HID, 2 * OUTP
NBT = NB1 + NB2 + NB3
NP = NWT + NBT


def _req(name: str, t, dtype, numel: int = None, shape: tuple = None):
    """A per-call tensor argument whose data_ptr() goes to a kernel: a contiguous device tensor of exactly this dtype and size, or ValueError
    (a wrong dtype is read with the wrong stride -- a bool mask read as int64 is 8 x out of bounds)."""
    if not torch.is_tensor(t):
        raise ValueError("%s: a tensor is required, got %r" % (name, type(t).__name__))
    if t.dtype != dtype:
        raise ValueError("%s: dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise ValueError("%s: must be contiguous (shape %r, strides %r)" % (name, tuple(t.shape), t.stride()))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s: shape %r, expected %r" % (name, tuple(t.shape), tuple(shape)))
    if numel is not None and t.numel() != numel:
        raise ValueError("%s: %d elements (shape %r), expected %d" % (name, t.numel(), tuple(t.shape), numel))
    if not t.is_cuda:
        raise ValueError("%s: must live on the GPU (it is on %s)" % (name, t.device))
    return t


class DwpMlp(C.Structure):          # include/dyros_ppo.h
    _fields_ = [(n, C.c_void_p) for n in ("obs", "state", "act", "old_nlp", "old_mu", "adv", "ret", "logstd", "obs16", "p16", "p16t", "pbuf",
                                          "x16", "h1", "h2", "out16", "dout16", "dz2", "dz1", "xf", "h1f", "h2f", "doutf", "dz2f", "dz1f")] + [("B", C.c_int32), ("e_clip", C.c_float), ("critic_coef", C.c_float)]


def declare(lib: C.CDLL) -> dict:
    P = C.c_void_p

    def fn(name, restype, *argtypes):
        f = getattr(lib, "dwp_" + name)
        f.restype, f.argtypes = restype, list(argtypes)
        return f
    api = {"abi_version": fn("abi_version", C.c_int), "last_error": fn("last_error", C.c_char_p), "sizeof_mlp": fn("sizeof_mlp", C.c_int)}
    if api["sizeof_mlp"]() != C.sizeof(DwpMlp) or api["abi_version"]() != K["DWP_ABI_VERSION"]:
        raise RuntimeError("libdyroswalk_hip.so and isaacgymdyros_amd/ppo_update.py disagree about include/dyros_ppo.h (DwpMlp is %d bytes here, ABI %d): rebuild"
                           % (C.sizeof(DwpMlp), K["DWP_ABI_VERSION"]))
    api["stage_obs"] = fn("stage_obs", C.c_int, P, P, C.c_int32, P, P)
    api["bias_relu"] = fn("bias_relu", C.c_int, P, P, C.c_int32, P)
    api["loss"] = fn("loss", C.c_int, P, P, P, P, P, P, P, P, P, P, C.c_int32, C.c_float, C.c_float, P, P)
    api["relu_bwd"] = fn("relu_bwd", C.c_int, P, P, P, C.c_int32, P)
    api["grad_stats"] = fn("grad_stats", C.c_int, P, P, P, P, P, P, C.c_int32, P)
    api["grad_bucket"] = fn("grad_bucket", C.c_int, P, P, P, C.c_float, P)
    api["adam"] = fn("adam", C.c_int, P, P, P, P, P, P, P, P, C.c_float, P, P, C.c_int32, P, P)
    api["policy"] = fn("policy", C.c_int, P, P, P, C.c_int32, P, P, P)
    api["retile32"] = fn("retile32", C.c_int, P, P, P)
    api["stats_adam_finish"] = fn("stats_adam_finish", C.c_int, P, P, P, P, P, P, P, C.c_float, P, P, C.c_int32, P, C.c_int32, C.c_int32, C.c_int32, P, P, P)
    api["adam_finish"] = fn("adam_finish", C.c_int, P, P, P, P, P, P, P, C.c_float, P, P, C.c_int32, P, C.c_int32, C.c_int32, C.c_int32, P, P)
    api["finish"] = fn("finish", C.c_int, P, P, C.c_int32, C.c_int32, C.c_int32, P, P)
    api["retile"] = fn("retile", C.c_int, P, P, P)
    api["rollout_pre"] = fn("rollout_pre", C.c_int, P, P, P, P, P, P, P, C.c_int32, C.c_int32, P, P, P, P, P, P, P, C.c_int32, C.c_int32, C.c_int32, P)
    api["rollout_post"] = fn("rollout_post", C.c_int, P, P, P, P, C.c_int32, P, P, P, C.c_int32, C.c_int32, C.c_float, C.c_float, P, P, C.c_int32, P, P, C.c_int32, P)
    api["gae"] = fn("gae", C.c_int, P, P, P, P, P, C.c_float, C.c_float, C.c_int32, C.c_int32, P, P)
    api["mlp"] = fn("mlp", C.c_int, C.POINTER(DwpMlp), P)
    api["wgrad"] = fn("wgrad", C.c_int, P, P, P, P, P, P, P, P, C.c_int32, P)
    if api["abi_version"]() != K["DWP_ABI_VERSION"]:
        raise RuntimeError("libdyroswalk_hip.so: dwp ABI %d, header %d" % (api["abi_version"](), K["DWP_ABI_VERSION"]))
    return api


def gae(fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma: float, tau: float):
    """`discount_values` of the reference (a2c_common_dyros.py:485-500) in one launch (dwp_gae).  fp32 CUDA tensors: fdones [N], last_values [N, 1],
    mb_fdones [H, N], mb_values / mb_rewards [H, N, 1]; returns advs [H, N, 1]."""
    api = declare(_lib.load()[0])
    H, N = int(mb_rewards.shape[0]), int(mb_rewards.shape[1])
    ts = [t.contiguous() for t in (fdones, last_values, mb_fdones, mb_values, mb_rewards)]
    for t, n in zip(ts, (N, N, H * N, H * N, H * N)):
        if t.dtype != torch.float32 or t.numel() != n or not t.is_cuda:
            raise ValueError("gae: expected fp32 GPU tensors of [N] / [N, 1] / [H, N] / [H, N, 1]")
    advs = torch.empty_like(ts[4])
    rc = api["gae"](*[t.data_ptr() for t in ts], float(gamma), float(tau), H, N, advs.data_ptr(), torch.cuda.current_stream(advs.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(api["last_error"]().decode())
    return advs


class RolloutRecorder:
    """The bookkeeping of one rollout step around env.step in two launches (dwp_rollout_pre / _post): `mb` is the consumer's dict of rollout
    buffers (obs [H, N, num_obs], act, mu [H, N, 13], neglogp, done [H, N], val, rew [H, N, 1]), n a device int64 [1] the caller advances."""

    def __init__(self, mb: dict, n: torch.Tensor, logstd: torch.Tensor, reward_scale: float, gamma: float, bootstrap: bool, obs_env_major=None, num_obs=None):
        """obs_env_major: a [N * H, num_obs] (or [N, H, num_obs]) tensor that takes the observations instead of mb["obs"], env-major: the flat batch of
        the update, written as the rollout goes (mb["obs"] is then not touched and may be absent).  A float16 [N * H, 512] tensor (zeros) takes them as
        fp16 rows, the form the update's first layer reads (num_obs = the row length of the env's observations must be given then)."""
        self.api = declare(_lib.load()[0])
        self.mb, self.n, self.logstd = mb, n, logstd
        self.scale, self.gamma, self.bootstrap = float(reward_scale), float(gamma), bool(bootstrap)
        if not (torch.is_tensor(mb.get("act")) and mb["act"].dim() == 3 and mb["act"].shape[2] == ACT):
            raise ValueError("RolloutRecorder: mb['act'] must be [H, N, %d]" % ACT)
        self.H, self.N = (int(x) for x in mb["act"].shape[:2])
        self.obs_dst, self.env_major = (obs_env_major, self.H) if obs_env_major is not None else (mb["obs"], 0)
        self.half = self.obs_dst.dtype == torch.float16          # (fp16 rows of INP: the update's own input format, FusedPpoUpdate.bind_batch)
        self.nobs = int(num_obs if num_obs is not None else self.obs_dst.shape[-1])
        if not (self.obs_dst.is_contiguous() and self.obs_dst.is_cuda):
            raise ValueError("RolloutRecorder: the observation buffer must be a contiguous device tensor")
        if self.half and not (self.env_major and num_obs is not None and self.obs_dst.numel() == self.H * self.N * INP):
            raise ValueError("RolloutRecorder: a float16 observation buffer is the env-major [N * H, %d] batch and needs num_obs" % INP)
        if not self.half and not (self.obs_dst.numel() == self.H * self.N * self.nobs and self.obs_dst.dtype == torch.float32):
            raise ValueError("RolloutRecorder: the observation buffer must hold H * N rows of num_obs float32")
        if (self.N * self.nobs) % 4:
            raise ValueError("RolloutRecorder: N * num_obs must be a multiple of 4 (the kernels move 16-byte pieces)")
        H, N = self.H, self.N
        for k, shape in (("act", (H, N, ACT)), ("mu", (H, N, ACT)), ("neglogp", (H, N)), ("done", (H, N)), ("val", (H, N, 1)), ("rew", (H, N, 1))):
            _req("RolloutRecorder: mb[%r]" % k, mb.get(k), torch.float32, H * N * (ACT if k in ("act", "mu") else 1))
            if tuple(mb[k].shape[:2]) != (H, N):
                raise ValueError("RolloutRecorder: mb[%r] is %r, expected [%d, %d, ...]" % (k, tuple(mb[k].shape), H, N))
        _req("RolloutRecorder: n (the device row counter)", n, torch.int64, 1)
        _req("RolloutRecorder: logstd", logstd, torch.float32, ACT)
        self.act = torch.empty(self.N, ACT, device=mb["act"].device)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.api["last_error"]().decode())

    def rows(self) -> int:
        """The device row counter, read back (a host sync: call it at the end of an epoch, not per step).  More than H means steps were replayed
        without rewinding n: the kernels dropped their rows (include/dyros_ppo.h), and this raises."""
        r = int(self.n.item())
        if r < 0 or r > self.H:
            raise RuntimeError("RolloutRecorder: the row counter stands at %d of %d rows: n was not rewound before the rollout (the steps past the "
                               "buffers recorded nothing)" % (r, self.H))
        return r

    def pre(self, mu, value, noise, obs, dones):
        """Returns the clipped action for env.step.  mu, noise [N, 13], value [N] / [N, 1], obs [N, num_obs], dones [N]: contiguous float32 GPU tensors."""
        N = self.N
        _req("RolloutRecorder.pre: mu", mu, torch.float32, shape=(N, ACT)); _req("RolloutRecorder.pre: noise", noise, torch.float32, shape=(N, ACT))
        _req("RolloutRecorder.pre: value", value, torch.float32, N); _req("RolloutRecorder.pre: dones", dones, torch.float32, N)
        _req("RolloutRecorder.pre: obs", obs, torch.float32, N * self.nobs)
        mb, s = self.mb, torch.cuda.current_stream(obs.device).cuda_stream
        self._chk(self.api["rollout_pre"](mu.data_ptr(), value.data_ptr(), noise.data_ptr(), obs.data_ptr(), dones.data_ptr(), self.logstd.data_ptr(), self.n.data_ptr(),
                                          self.N, self.nobs, self.obs_dst.data_ptr(), mb["act"].data_ptr(), mb["mu"].data_ptr(), mb["neglogp"].data_ptr(),
                                          mb["val"].data_ptr(), mb["done"].data_ptr(), self.act.data_ptr(), self.env_major, int(self.half), self.H, s))
        return self.act

    def post(self, rew, value, time_outs, stacked, done_buf, new_obs, terms, g_dones, g_obs):
        """rew, value [N] float32; time_outs (or None), done_buf [N] INT64 (VecTask's reset_buf / extras['time_outs']: a bool mask such as
        TocabiAMPLower's timeout_buf must be converted by the caller); stacked [N, cols] float32 or None; terms: float32, at most
        min(cols, DWP_ROLL_TERMS_MAX) words -- the first columns of stacked are summed into it; new_obs, g_obs [N, num_obs]; g_dones [N]."""
        N = self.N
        _req("RolloutRecorder.post: rew", rew, torch.float32, N); _req("RolloutRecorder.post: value", value, torch.float32, N)
        _req("RolloutRecorder.post: done_buf", done_buf, torch.int64, N)
        _req("RolloutRecorder.post: new_obs", new_obs, torch.float32, N * self.nobs); _req("RolloutRecorder.post: g_obs", g_obs, torch.float32, N * self.nobs)
        _req("RolloutRecorder.post: g_dones", g_dones, torch.float32, N)
        to = None
        if self.bootstrap and time_outs is not None:
            to = _req("RolloutRecorder.post: time_outs", time_outs, torch.int64, N).data_ptr()
        nterms = 0
        if stacked is not None:
            if stacked.dim() != 2:
                raise ValueError("RolloutRecorder.post: stacked must be [N, columns]")
            _req("RolloutRecorder.post: stacked", stacked, torch.float32, N * int(stacked.shape[1]))
            nterms = int(_req("RolloutRecorder.post: terms", terms, torch.float32).numel())
            if not 1 <= nterms <= min(int(stacked.shape[1]), K["DWP_ROLL_TERMS_MAX"]):
                raise ValueError("RolloutRecorder.post: terms has %d words; it takes the first columns of stacked [N, %d], at most %d"
                                 % (nterms, int(stacked.shape[1]), K["DWP_ROLL_TERMS_MAX"]))
        s = torch.cuda.current_stream(rew.device).cuda_stream
        self._chk(self.api["rollout_post"](rew.data_ptr(), value.data_ptr(), to, stacked.data_ptr() if stacked is not None else None,
                                           int(stacked.shape[1]) if stacked is not None else 0, done_buf.data_ptr(), new_obs.data_ptr(), self.n.data_ptr(), self.N, self.nobs,
                                           self.scale, self.gamma, self.mb["rew"].data_ptr(), terms.data_ptr() if stacked is not None else None,
                                           nterms, g_dones.data_ptr(), g_obs.data_ptr(), self.H, s))


class FusedPpoUpdate:
    """Owns the flat parameter / moment / gradient buffers and the activations of one minibatch size; `update()` enqueues one update
    of the minibatch whose index lives in the device state (it advances by itself: the call has no argument that changes, so it can
    be captured in a hipGraph once and replayed)."""

    def __init__(self, net, cfg: dict, minibatch: int, num_minibatches: int, device, mfma: bool = True, rowmajor: bool = True, split_tail: bool = False,
                 merged_tail: bool = False, policy_copy_per_update: bool = True, world: int = 1, group=None, collective: bool = None, fp16_grads: bool = False):
        """mfma: forward, loss and input gradients in ONE launch on the matrix cores (dwp_mlp + dwp_wgrad; the minibatch must be a multiple of 32, else the library-GEMM form runs) instead
        of eight library GEMM launches with six kernels between them.  rowmajor (mfma only): dwp_mlp also writes its activations and their
        gradients as plain [2, B, 256] / [B, 512] matrices (x16, h1, h2, dh2, dh1: what the tests read); a trainer passes False.
        merged_tail (mfma only): dwp_grad_stats and dwp_adam_finish as ONE launch whose blocks wait for each other's share of the norm,
        dwp_stats_adam_finish -- three launches per update (plus dwp_grad_bucket and the all-reduce when sharded).  Measured: exactly as long as the
        two launches it replaces (profiles/r06_ppo_tail_forms.txt), so it is off by default; the actor's step may differ from theirs in the last place
        (the norm's partial sums are taken in another order).  barrier_timed_out() reports the one way the merged launch can fail.
        policy_copy_per_update (mfma only): every update's Adam launch also rewrites the fp32 operand-order copy dwp_policy reads (401 408 scattered
        words, 1.1 us of the launch).  False: the updates leave that copy alone and `sync_policy_copy()` -- one launch -- brings it up to date; a
        trainer calls it once after an epoch's updates.  `policy()` refuses to run on a copy it knows to be stale.
        split_tail (mfma only): the last launch, dwp_adam_finish, as dwp_adam + dwp_finish (five launches: what a test compares the merged one with).
        world, group (mfma only): the ranks that train together (torch.distributed, backend nccl = RCCL) -- every update then averages the
        ranks' gradients with ONE all-reduce of the 1.61 MB bucket between dwp_wgrad and dwp_grad_stats (`update()` =
        `update_head()`, `allreduce()`, `update_tail()`).  collective: run that bucket path although world == 1 (tests: bit-identical to the
        plain four launches; default: only when world > 1).  The learning rates start at cfg's learning_rate / critic_lr.
        fp16_grads (mfma only): the summed weight gradients are rounded through fp16 before unscale / clip / Adam -- the arithmetic type a backward
        under autocast gives them (inf beyond 65 504), so found_inf and the loss scale move exactly as the reference's GradScaler would; default
        off: the fp32 sums as they are (include/dyros_ppo.h DWP_S_G16)."""
        c = cfg
        if bool(c.get("clip_value")) or float(c.get("entropy_coef", 0.0)) != 0.0 or float(c.get("bounds_loss_coef", 0.0)) != 0.0:
            raise ValueError("the fused update is written for clip_value False, entropy_coef 0, bounds_loss_coef 0 (DyrosDynamicWalkPPO.yaml)")
        if not bool(c.get("truncate_grads", True)) or not bool(c.get("mixed_precision", True)):
            raise ValueError("the fused update is written for truncate_grads and mixed_precision (DyrosDynamicWalkPPO.yaml)")
        lins = [[m for m in trunk if isinstance(m, torch.nn.Linear)] for trunk in (net.actor_mlp, net.critic_mlp)]
        shapes = [tuple(l.weight.shape) for l in lins[0]] + [tuple(net.mu.weight.shape), tuple(net.value.weight.shape)]
        if shapes != [(HID, IN), (HID, HID), (ACT, HID), (1, HID)] or [tuple(l.weight.shape) for l in lins[1]] != shapes[:2]:
            raise ValueError("the fused update is built for %d -> [%d, %d] -> %d / 1 networks, got %r" % (IN, HID, HID, ACT, shapes))
        self.lib = _lib.load()[0]
        self.api = declare(self.lib)
        self.dev = torch.device(device)
        self.B, self.nmb = int(minibatch), int(num_minibatches)
        self.split_tail = bool(split_tail)
        self.merged_tail = bool(merged_tail) and not self.split_tail
        self.policy_copy_per_update = bool(policy_copy_per_update)
        self._policy_copy_stale = False
        self.world, self.group = int(world), group
        if self.world < 1:
            raise ValueError("FusedPpoUpdate: world must be at least 1")
        self.collective = bool(collective) if collective is not None else self.world > 1
        if self.collective and not (bool(mfma) and self.B % 32 == 0) or (self.collective and split_tail):
            raise ValueError("FusedPpoUpdate: the sharded update is the four-launch form (mfma, minibatch a multiple of 32, no split_tail)")
        if self.world > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise ValueError("FusedPpoUpdate: world = %d needs an initialised torch.distributed process group" % self.world)
            if dist.get_world_size(group) != self.world:
                raise ValueError("FusedPpoUpdate: world = %d, but the process group has %d ranks" % (self.world, dist.get_world_size(group)))
        self.e_clip, self.critic_coef, self.max_norm = float(c["e_clip"]), float(c["critic_coef"]), float(c["grad_norm"])
        f32 = dict(device=self.dev, dtype=torch.float32)
        f16 = dict(device=self.dev, dtype=torch.float16)
        self.p = torch.zeros(NP, **f32)
        self.m, self.v = torch.zeros(NP, **f32), torch.zeros(NP, **f32)
        self.p16 = torch.zeros(NP, **f16)
        self.g16 = torch.zeros(NWT, **f16)
        self.gb = torch.zeros(NB1 + NB2 + NB3, **f32)
        self.state = torch.zeros(K["DWP_S_WORDS"], **f32)
        self.part = torch.zeros(K["DWP_PARTS"], **f32)
        self.state[K["DWP_S_SCALE"]] = 65536.0
        self.state[K["DWP_S_G16"]] = 1.0 if fp16_grads else 0.0
        # (the learning rates live in the device state; a caller's schedule overwrites them: set_learning_rates.  Left at zero, update() would
        #  advance Adam's moments and step counts while no parameter moves, without any error)
        self._lr_set = False
        if c.get("learning_rate") is not None and c.get("critic_lr") is not None:
            self.state[K["DWP_S_LR"]] = float(c["learning_rate"]); self.state[K["DWP_S_LR"] + 1] = float(c["critic_lr"])
            self._lr_set = True
        o = 0
        self.views, self.views16, self._gviews16, self._gshape = {}, {}, {}, {}
        for name, shape in (("W1", (2, HID, INP)), ("W2", (2, HID, HID)), ("W3", (2, OUTP, HID)), ("b1", (2, HID)), ("b2", (2, HID)), ("b3", (2, OUTP))):
            n = 1
            for s in shape:
                n *= s
            self.views[name] = self.p[o:o + n].view(shape)
            self.views16[name] = self.p16[o:o + n].view(shape)
            if name.startswith("W"):
                self._gviews16[name] = self.g16[o:o + n].view(shape)
                self._gshape[name] = (o, n, shape)
            o += n
        # adopt the module's initial values, then make the module's parameters views of the master buffer
        with torch.no_grad():
            for k, (trunk, head, rows) in enumerate(((lins[0], net.mu, ACT), (lins[1], net.value, 1))):
                for name, lin in (("1", trunk[0]), ("2", trunk[1])):
                    cols = lin.weight.shape[1]          # (W1: 487 of its 512 padded columns; the padding stays zero -- its gradient is zero)
                    self.views["W" + name][k, :, :cols].copy_(lin.weight)
                    self.views["b" + name][k].copy_(lin.bias)
                    lin.weight.data, lin.bias.data = self.views["W" + name][k, :, :cols], self.views["b" + name][k]
                self.views["W3"][k, :rows].copy_(head.weight)
                self.views["b3"][k, :rows].copy_(head.bias)
                head.weight.data, head.bias.data = self.views["W3"][k, :rows], self.views["b3"][k, :rows]
            self.p16.copy_(self.p)
        B = self.B
        self.x16 = torch.zeros(B, INP, **f16)
        self.h1, self.h2 = torch.zeros(2, B, HID, **f16), torch.zeros(2, B, HID, **f16)
        self.out = torch.zeros(2, B, OUTP, **f16)
        self.dout = torch.zeros(2, B, OUTP, **f16)
        self.dh2, self.dh1 = torch.zeros(2, B, HID, **f16), torch.zeros(2, B, HID, **f16)
        self.logstd = net.sigma
        self.src = None
        self.mfma = bool(mfma) and B % 32 == 0
        self.rowmajor = bool(rowmajor) or not self.mfma
        self.p16t = torch.zeros(K["DWP_P16F_WORDS"], **f16)          # the weights once more, in the order dwp_mlp's matrix instructions take them
        self._chk(self.api["retile"](self.p16.data_ptr(), self.p16t.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))
        self.p32f = torch.zeros(K["DWP_P32F_WORDS"], **f32)          # the fp32 weights in dwp_policy's operand order
        self._chk(self.api["retile32"](self.p.data_ptr(), self.p32f.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))
        self.pbuf = torch.zeros(K["DWP_PBUF_BUCKETS"], 2, K["DWP_PBUF_WORDS"], **f32)          # dwp_mlp: accumulators of bias gradients and logged sums
        self._mlp_args = None
        if self.mfma:          # dwp_mlp's outputs once more as operands of dwp_wgrad, and dwp_wgrad's partial gradients
            self.xf = torch.zeros(B * INP, **f16)
            self.h1f, self.h2f, self.dz2f, self.dz1f = (torch.zeros(2 * B * HID, **f16) for _ in range(4))
            self.doutf = torch.zeros(2 * B * OUTP, **f16)
            self.g32 = torch.zeros(K["DWP_WGRAD_SLABS"], NWT, **f32)          # dwp_wgrad's partial gradients, one copy per slab of samples
        # sharded training: this rank's gradient [weights | biases] / world in one piece, the operand of the update's one all-reduce
        self.bucket = torch.zeros(NWT + NBT, **f32) if self.collective else None

    @property
    def gviews(self):
        """The weight gradients of the last update (still multiplied by its loss scale), by layer: fp16 [2, out, in] from the library GEMMs,
        or fp32 (the sum of dwp_wgrad's partial gradients)."""
        if not self.mfma:
            return self._gviews16
        g = self.g32.sum(0)
        return {name: g[o:o + n].view(shape) for name, (o, n, shape) in self._gshape.items()}

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.api["last_error"]().decode())

    def policy(self, obs: torch.Tensor, mu: torch.Tensor = None, value: torch.Tensor = None):
        """The rollout's forward in fp32 (dwp_policy): (mu [N, 13], value [N, 1]) for obs [N, 487].  mu / value: output
        tensors to reuse (a captured rollout step passes the same ones every time)."""
        if not (torch.is_tensor(obs) and obs.dim() == 2):
            raise ValueError("FusedPpoUpdate.policy: obs must be [N, %d]" % IN)
        N = int(obs.shape[0])
        _req("FusedPpoUpdate.policy: obs", obs, torch.float32, shape=(N, IN))
        if mu is None:
            mu, value = torch.empty(N, ACT, device=self.dev), torch.empty(N, 1, device=self.dev)
        _req("FusedPpoUpdate.policy: mu", mu, torch.float32, shape=(N, ACT)); _req("FusedPpoUpdate.policy: value", value, torch.float32, N)
        if self._policy_copy_stale:
            raise RuntimeError("FusedPpoUpdate.policy: updates ran with policy_copy_per_update=False and sync_policy_copy() was not called since")
        self._chk(self.api["policy"](obs.data_ptr(), self.p.data_ptr(), self.p32f.data_ptr(), N, mu.data_ptr(), value.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))
        return mu, value

    def sync_policy_copy(self):
        """policy_copy_per_update=False: bring dwp_policy's fp32 operand-order copy of the weights up to date with the masters (one launch, dwp_retile32;
        on the current stream).  Call it after the last update and before the next policy() -- also after REPLAYING a captured update, which this
        object cannot see."""
        self._chk(self.api["retile32"](self.p.data_ptr(), self.p32f.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))
        self._policy_copy_stale = False

    def refresh_copies(self):
        """After the module's parameters were written from outside (a checkpoint loaded into the network: they are views of the master buffer):
        the fp16 copies the next forward reads, row-major and in fragment order."""
        with torch.no_grad():
            self.p16.copy_(self.p)
        self._chk(self.api["retile"](self.p16.data_ptr(), self.p16t.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))
        self._chk(self.api["retile32"](self.p.data_ptr(), self.p32f.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream))

    def state_dict(self) -> dict:
        """Optimiser side of a checkpoint (the parameters themselves are the network's): Adam moments, step counts, loss scale and its growth
        tracker (torch.optim.Adam.state_dict + GradScaler.state_dict in one)."""
        st = self.state
        return {"m": self.m.clone(), "v": self.v.clone(), "scale": float(st[K["DWP_S_SCALE"]]), "growth": float(st[K["DWP_S_GROWTH"]]),
                "steps": st[K["DWP_S_STEP"]:K["DWP_S_STEP"] + 2].tolist()}

    def load_state_dict(self, d: dict):
        if tuple(d["m"].shape) != (NP,) or tuple(d["v"].shape) != (NP,):
            raise ValueError("fused PPO update: moments of %r / %r, expected (%d,)" % (tuple(d["m"].shape), tuple(d["v"].shape), NP))
        with torch.no_grad():
            self.m.copy_(d["m"]); self.v.copy_(d["v"])
            self.state[K["DWP_S_SCALE"]] = float(d["scale"]); self.state[K["DWP_S_GROWTH"]] = float(d["growth"])
            self.state[K["DWP_S_STEP"]:K["DWP_S_STEP"] + 2] = torch.tensor([float(x) for x in d["steps"]], device=self.dev)
        self.refresh_copies()

    def set_learning_rates(self, lr_actor: float, lr_critic: float):
        self.state[K["DWP_S_LR"]:K["DWP_S_LR"] + 2] = torch.tensor([lr_actor, lr_critic], device=self.dev)
        self._lr_set = True

    def bind_batch(self, obs, act, neglogp, mu, adv, ret):
        """The epoch's flat arrays (env-major, `batch` rows; fp32, contiguous).  Their ADDRESSES are what a captured update replays:
        keep the tensors and copy_ each epoch's data into them.  obs: fp32 [batch, 487], or (dwp_mlp form only) fp16 [batch, 512] with
        zero padding -- what RolloutRecorder(obs_env_major=...) fills."""
        rows = self.B * self.nmb
        # (dwp_mlp / dwp_loss index the action arrays as [row][13] and the others as [row]: shapes are checked, not only row counts)
        for name, t, width in (("act", act, ACT), ("neglogp", neglogp, 1), ("mu", mu, ACT), ("adv", adv, 1), ("ret", ret, 1)):
            _req("bind_batch: " + name, t, torch.float32, rows * width)
            if t.shape[0] != rows or (width > 1 and tuple(t.shape) != (rows, width)):
                raise ValueError("bind_batch: %s is %r, expected %d rows%s" % (name, tuple(t.shape), rows, " of %d" % width if width > 1 else ""))
        if not (torch.is_tensor(obs) and obs.is_cuda):
            raise ValueError("bind_batch: obs must be a GPU tensor")
        ok32 = obs.dtype == torch.float32 and tuple(obs.shape) == (rows, IN)
        ok16 = self.mfma and obs.dtype == torch.float16 and tuple(obs.shape) == (rows, INP)
        if not (obs.is_contiguous() and (ok32 or ok16)):
            raise ValueError("bind_batch: obs must be contiguous float32 [%d, %d]%s" % (rows, IN, " or float16 [%d, %d]" % (rows, INP) if self.mfma else ""))
        self._mlp_args = None
        self.src = (obs, act, neglogp, mu, adv, ret)

    def rewind(self):
        """Start the next pass at minibatch 0 (a new epoch)."""
        self.state[K["DWP_S_MB"]] = 0.0

    def _ready(self):
        if self.src is None:
            raise RuntimeError("FusedPpoUpdate.update: bind_batch() first")
        if not self._lr_set:
            raise RuntimeError("FusedPpoUpdate.update: the learning rates are unset (cfg had no learning_rate / critic_lr): set_learning_rates() first")

    def update_head(self):
        """Four-launch form, first part: dwp_mlp, dwp_wgrad and -- sharded -- dwp_grad_bucket: this rank's gradient, ready for the all-reduce."""
        self._ready()
        api, st, B = self.api, self.state.data_ptr(), self.B
        obs, act, nlp, mu_old, adv, ret = self.src
        s = torch.cuda.current_stream(self.dev).cuda_stream
        if self._mlp_args is None:
            a = DwpMlp()
            for k_, t_ in (("obs16" if obs.dtype == torch.float16 else "obs", obs), ("state", self.state), ("act", act), ("old_nlp", nlp), ("old_mu", mu_old), ("adv", adv), ("ret", ret), ("logstd", self.logstd),
                           ("p16", self.p16), ("p16t", self.p16t), ("pbuf", self.pbuf), ("x16", self.x16), ("h1", self.h1), ("h2", self.h2),
                           ("out16", self.out), ("dout16", self.dout), ("dz2", self.dh2), ("dz1", self.dh1), ("xf", self.xf), ("h1f", self.h1f), ("h2f", self.h2f),
                           ("doutf", self.doutf), ("dz2f", self.dz2f), ("dz1f", self.dz1f)):
                setattr(a, k_, t_.data_ptr() if self.rowmajor or k_ not in ("x16", "h1", "h2", "dz2", "dz1") else None)
            a.B, a.e_clip, a.critic_coef = B, self.e_clip, self.critic_coef
            self._mlp_args = a
        self._chk(api["mlp"](C.byref(self._mlp_args), s))
        self._chk(api["wgrad"](self.xf.data_ptr(), self.h1f.data_ptr(), self.h2f.data_ptr(), self.doutf.data_ptr(), self.dz2f.data_ptr(), self.dz1f.data_ptr(), st,
                               self.g32.data_ptr(), B, s))
        if self.collective:
            self._chk(api["grad_bucket"](self.g32.data_ptr(), self.pbuf.data_ptr(), self.bucket.data_ptr(), 1.0 / self.world, s))

    def allreduce(self):
        """ONE collective per update: the sum over the ranks of bucket = gradient / world (1.61 MB; RCCL on the GPUs), on the current stream --
        where the reference's Horovod optimizer.synchronize() stands, before unscale_ / clip / step (a2c_continuous_seperate.py:171-180)."""
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.bucket, group=self.group)

    def update_tail(self):
        """Second part: statistics, clip, Adam and the scaler on the (averaged) gradient -- dwp_stats_adam_finish, or dwp_grad_stats and dwp_adam_finish."""
        api, st, B = self.api, self.state.data_ptr(), self.B
        s = torch.cuda.current_stream(self.dev).cuda_stream
        p32f = self.p32f.data_ptr() if self.policy_copy_per_update else None
        self._policy_copy_stale = not self.policy_copy_per_update
        if self.merged_tail:
            g, gb, nsl = (self.bucket.data_ptr(), self.bucket.data_ptr() + 4 * NWT, 1) if self.collective else (self.g32.data_ptr(), self.gb.data_ptr(), K["DWP_WGRAD_SLABS"])
            self._chk(api["stats_adam_finish"](self.p.data_ptr(), self.p16.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), gb, st, self.part.data_ptr(), self.max_norm,
                                               self.p16t.data_ptr(), g, nsl, p32f, B, self.nmb, 2000, self.pbuf.data_ptr(),
                                               None if self.collective else self.pbuf.data_ptr(), s))
            return
        if self.collective:
            gb = self.bucket.data_ptr() + 4 * NWT
            self._chk(api["grad_stats"](None, gb, st, self.part.data_ptr(), None, self.bucket.data_ptr(), 1, s))
            self._chk(api["adam_finish"](self.p.data_ptr(), self.p16.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), gb, st, self.part.data_ptr(),
                                         self.max_norm, self.p16t.data_ptr(), self.bucket.data_ptr(), 1, p32f, B, self.nmb, 2000, self.pbuf.data_ptr(), s))
            return
        nsl = K["DWP_WGRAD_SLABS"]
        self._chk(api["grad_stats"](None, self.gb.data_ptr(), st, self.part.data_ptr(), self.pbuf.data_ptr(), self.g32.data_ptr(), nsl, s))
        if self.split_tail:
            self._chk(api["adam"](self.p.data_ptr(), self.p16.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), None, self.gb.data_ptr(), st,
                                  self.part.data_ptr(), self.max_norm, self.p16t.data_ptr(), self.g32.data_ptr(), nsl, p32f, s))
            self._chk(api["finish"](st, self.gb.data_ptr(), B, self.nmb, 2000, self.pbuf.data_ptr(), s))
            return
        self._chk(api["adam_finish"](self.p.data_ptr(), self.p16.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.gb.data_ptr(), st, self.part.data_ptr(),
                                     self.max_norm, self.p16t.data_ptr(), self.g32.data_ptr(), nsl, p32f, B, self.nmb, 2000, self.pbuf.data_ptr(), s))

    def barrier_timed_out(self) -> bool:
        """True once a block of dwp_stats_adam_finish gave up waiting at its grid barrier (the launch did not have the device to itself for tens of
        ms): that update was published as skipped (DWP_S_OUT[7] = 2) but some blocks may have stepped -- reload a checkpoint.  Synchronises."""
        return bool(self.part[642].item() != 0.0)

    def update(self):
        """Enqueue one minibatch update on the current stream."""
        if self.mfma:
            self.update_head()
            self.allreduce()
            self.update_tail()
            return
        self._ready()
        api, st, B = self.api, self.state.data_ptr(), self.B
        obs, act, nlp, mu_old, adv, ret = self.src
        s = torch.cuda.current_stream(self.dev).cuda_stream
        W, W16, G = self.views, self.views16, self._gviews16
        self._chk(api["stage_obs"](obs.data_ptr(), st, B, self.x16.data_ptr(), s))
        # forward: Linear + relu twice, the two heads (fp16 in, fp32 accumulate, fp16 out: nn.Linear under autocast)
        torch.bmm(self.x16.unsqueeze(0).expand(2, B, INP), W16["W1"].transpose(1, 2), out=self.h1)
        self._chk(api["bias_relu"](self.h1.data_ptr(), W16["b1"].data_ptr(), B, s))
        torch.bmm(self.h1, W16["W2"].transpose(1, 2), out=self.h2)
        self._chk(api["bias_relu"](self.h2.data_ptr(), W16["b2"].data_ptr(), B, s))
        torch.bmm(self.h2, W16["W3"].transpose(1, 2), out=self.out)
        self._chk(api["loss"](self.out.data_ptr(), W16["b3"].data_ptr(), act.data_ptr(), nlp.data_ptr(), mu_old.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                              self.logstd.data_ptr(), st, self.gb.data_ptr(), B, self.e_clip, self.critic_coef, self.dout.data_ptr(), s))
        # backward: weight gradients dY' X, input gradients dY W, relu masks (with the bias gradients) in between
        torch.bmm(self.dout.transpose(1, 2), self.h2, out=G["W3"])
        torch.bmm(self.dout, W16["W3"], out=self.dh2)
        self._chk(api["relu_bwd"](self.h2.data_ptr(), self.dh2.data_ptr(), self.gb.data_ptr() + 4 * NB1, B, s))
        torch.bmm(self.dh2.transpose(1, 2), self.h1, out=G["W2"])
        torch.bmm(self.dh2, W16["W2"], out=self.dh1)
        self._chk(api["relu_bwd"](self.h1.data_ptr(), self.dh1.data_ptr(), self.gb.data_ptr(), B, s))
        torch.bmm(self.dh1.transpose(1, 2), self.x16.unsqueeze(0).expand(2, B, INP), out=G["W1"])
        # unscale + clip + Adam + scaler
        self._chk(api["grad_stats"](self.g16.data_ptr(), self.gb.data_ptr(), st, self.part.data_ptr(), None, None, 0, s))
        self._chk(api["adam"](self.p.data_ptr(), self.p16.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.g16.data_ptr(), self.gb.data_ptr(), st, self.part.data_ptr(), self.max_norm, self.p16t.data_ptr(), None, 0, self.p32f.data_ptr(), s))
        self._chk(api["finish"](st, self.gb.data_ptr(), B, self.nmb, 2000, None, s))

    def logged(self):
        """(a_loss, c_loss, b_loss, clip fraction, kl, actor grad norm, loss scale, skipped) of the last update: a device tensor view."""
        return self.state[K["DWP_S_OUT"]:K["DWP_S_OUT"] + 8]
