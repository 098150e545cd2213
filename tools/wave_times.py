"""Distribution of the wave lifetimes of one step-kernel launch (library built with -DDQ_WAVE_TIME: tools/abl_build.sh DQ_WAVE_TIME for
the octet kernels, tools/ab_lib.sh wt -DDQ_WAVE_TIME + DW_PIPE=2 for the quad kernels; every wave leaves its cycle count in stacked_rewards[first env of the wave, 14])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg, with_terrain
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = default_cfg(N, "cuda:0")
if os.environ.get("DW_TERRAIN"): cfg = with_terrain(cfg, mesh_type="trimesh", curriculum=True)
PIPE = int(os.environ.get("DW_PIPE", "3"))
cfg["sim"]["mi355"]["pipeline"] = PIPE
EPW = 8 if PIPE == 3 else 16          # envs per wave
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(300):
    env.step(acts[i % 8])
allc, withr, without, nres, phys, post = [], [], [], [], [], []
for i in range(20):
    env.step(acts[i % 8])
    torch.cuda.synchronize()
    c = env._buf["stacked_rewards"].view(-1, EPW, 15)[:, 0, 14].cpu().numpy().astype(np.float64)
    phys.append(env._buf["stacked_rewards"].view(-1, EPW, 15)[:, 0, 13].cpu().numpy().astype(np.float64))
    post.append(env._buf["stacked_rewards"].view(-1, EPW, 15)[:, 1, 1:13].cpu().numpy().astype(np.float64))
    rn = env.reset_buf.view(-1, EPW).cpu().numpy().sum(axis=1)
    r = rn > 0
    allc.append(c); withr.append(c[r]); without.append(c[~r]); nres.append(rn)
c = np.concatenate(allc)
print("waves %d x %d launches: cycles min %.0f  p50 %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f" % (len(allc[0]), len(allc), c.min(), np.percentile(c, 50), c.mean(), np.percentile(c, 90), np.percentile(c, 99), c.max()))
print("per launch: mean of max %.0f, mean of mean %.0f" % (np.mean([x.max() for x in allc]), np.mean([x.mean() for x in allc])))
a, b = np.concatenate(withr), np.concatenate(without)
print("waves with a reset this step: %d, mean %.0f; without: %d, mean %.0f" % (len(a), a.mean() if len(a) else 0, len(b), b.mean()))
rn = np.concatenate(nres)
ph = np.concatenate(phys)
for k in range(0, 5):
    m = rn == k
    if m.any(): print("  %d resets in the wave: %5d waves, mean %.0f (up to the end of the physics %.0f, after it %.0f), p95 %.0f, max %.0f" % (k, m.sum(), c[m].mean(), ph[m].mean(), (c[m] - ph[m]).mean(), np.percentile(c[m], 95), c[m].max()))
top = np.argsort(c)[-20:]
print("  resets in the 20 slowest waves:", rn[top].tolist())
if PIPE == 3:
    po = np.concatenate(post)
    names = ["stage..Q3", "r:terrain", "r:draws", "r:DR", "r:joints", "r:zero", "r:scalars", "taps+Q4 obs", "Q5 obs_buf", "Q6", "write back"]
    for k in range(0, 3):
        m = rn == k
        if m.any(): print("  post phases, %d resets: " % k + ", ".join("%s %.0f" % (names[i], po[m][:, i].mean()) for i in range(len(names) if k else 1)))
    # placement: HW_ID bits -- wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]
    sr = env._buf["stacked_rewards"].view(-1, EPW, 15)
    hw = sr[:, 2, 1].cpu().numpy().astype(np.int64); xcc = sr[:, 2, 2].cpu().numpy().astype(np.int64)
    cyc = sr[:, 0, 14].cpu().numpy().astype(np.float64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simdid = cuid * 4 + simd
    import collections
    per = collections.Counter(simdid.tolist())
    print("  last launch: distinct CUs %d, distinct SIMDs %d, waves per SIMD histogram %s" % (len(set(cuid.tolist())), len(per), dict(collections.Counter(per.values()))))
    # is a wave slow because its SIMD partner is slow?  correlation of a wave's cycles with its partner's
    bys = collections.defaultdict(list)
    for i, sdx in enumerate(simdid.tolist()): bys[sdx].append(i)
    pairs = [(cyc[v[0]], cyc[v[1]]) for v in bys.values() if len(v) == 2]
    if len(pairs) > 10:
        pa = np.array(pairs)
        print("  SIMD partners: corr of lifetimes %.3f; mean |difference| %.0f cycles" % (np.corrcoef(pa[:, 0], pa[:, 1])[0, 1], np.abs(pa[:, 0] - pa[:, 1]).mean()))
    bycu = collections.defaultdict(list)
    for i, c_ in enumerate(cuid.tolist()): bycu[c_].append(cyc[i])
    cum = np.array([np.mean(v) for v in bycu.values()]); cux = np.array([np.max(v) for v in bycu.values()])
    print("  per CU: mean of wave lifetimes min %.0f p50 %.0f max %.0f; max-in-CU p50 %.0f" % (cum.min(), np.median(cum), cum.max(), np.median(cux)))
    byx = collections.defaultdict(list)
    for i, x_ in enumerate(xcc.tolist()): byx[x_].append(cyc[i])
    print("  per XCD mean lifetime:", {k: int(np.mean(v)) for k, v in sorted(byx.items())})
    # what makes a wave slow?  least-squares fit of its lifetime on what its 8 envs did this step
    cf = env._buf["contact_forces"].view(-1, EPW, 38, 3)
    mag = cf.norm(dim=-1)
    feet = (mag[:, :, [8, 16]] > 0).any(dim=-1).float().sum(dim=1).cpu().numpy()
    other = mag.clone(); other[:, :, [8, 16]] = 0
    nonfoot = (other > 0).any(dim=-1).float().sum(dim=1).cpu().numpy()
    nres = env.reset_buf.view(-1, EPW).float().sum(dim=1).cpu().numpy()
    anyres = (nres > 0).astype(np.float64)
    A = np.stack([np.ones_like(cyc), feet, nonfoot, anyres, (feet > 0).astype(np.float64), (nonfoot > 0).astype(np.float64)], axis=1)
    coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
    resid = cyc - A @ coef
    print("  lifetime ~ %.0f + %.0f * envs with foot contact + %.0f * envs with body contact + %.0f * [any reset] + %.0f * [any foot contact] + %.0f * [any body contact]; residual sd %.0f (lifetime sd %.0f)" % (*coef, resid.std(), cyc.std()))
    print("  envs per wave with foot contact: mean %.2f; with body contact: mean %.2f; waves with no foot contact at all: %d" % (feet.mean(), nonfoot.mean(), int((feet == 0).sum())))
    t0 = sr[:, 2, 3].cpu().numpy().astype(np.int64); t1 = sr[:, 2, 4].cpu().numpy().astype(np.int64)
    base = t0.min()
    t0 = ((t0 - base) & 0xffffff) * 10.0; t1 = ((t1 - base) & 0xffffff) * 10.0          # ns after the first wave's start
    print("  wave START (ns after the first): p50 %.0f p90 %.0f max %.0f;  wave END: min %.0f p50 %.0f p90 %.0f max %.0f;  lifetime ns p50 %.0f" % (
        np.median(t0), np.percentile(t0, 90), t0.max(), t1.min(), np.median(t1), np.percentile(t1, 90), t1.max(), np.median(t1 - t0)))
    print("  by wave-index quartile: start ns %s  end ns %s  lifetime cycles %s" % ([int(np.mean(q)) for q in np.array_split(t0, 4)], [int(np.mean(q)) for q in np.array_split(t1, 4)], [int(np.mean(q)) for q in np.array_split(cyc, 4)]))
    # the two waves of a SIMD: does the one dispatched first (lower global wave index) live shorter?
    firsts, seconds = [], []
    for v in bys.values():
        if len(v) == 2:
            a_, b_ = sorted(v)
            firsts.append(cyc[a_]); seconds.append(cyc[b_])
    if firsts:
        firsts, seconds = np.array(firsts), np.array(seconds)
        print("  on a SIMD: the wave with the lower index lives %.0f cycles, the other %.0f (first shorter in %.0f %% of the SIMDs); index distance p50 %d" % (
            firsts.mean(), seconds.mean(), 100.0 * (firsts < seconds).mean(), int(np.median([abs(v[0] - v[1]) for v in bys.values() if len(v) == 2]))))
    wid = hw & 15
    print("  wave slot ids (HW_ID.wave_id) seen:", dict(collections.Counter(wid.tolist())))
    pairs_w = [(wid[sorted(v)[0]], wid[sorted(v)[1]]) for v in bys.values() if len(v) == 2]
    print("  (slot of the lower-index wave, slot of the other) per SIMD:", dict(collections.Counter(pairs_w)))
