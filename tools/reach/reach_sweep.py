"""Driver of tools/reach/reach_sweep.c: writes the compiled model as the C program's text input, runs the sweep and
turns its output into JSON + a markdown table.  Analysis / test infrastructure: nothing in the product loads it.

usage: python tools/reach/reach_sweep.py [--samples 1000000] [--seed 1] [--corner 0.2] [--ranges mjcf|envelope] [--out profiles/r06_reach_sweep.json]
  --ranges mjcf      joint angles uniform in the MJCF ranges (assets/tocabi_model.json dof_lower / dof_upper), the whole configuration space
  --ranges envelope  legs uniform within +- `--leg` rad (default 1.0) of the task's initial pose, waist / arms / neck within +- `--upper` (default 0.2: they
                     are held at that pose by the task's PD), clipped to the MJCF ranges: the neighbourhood the task works in (the mocap's leg
                     excursions are inside it: tests/test_reach_sweep.py)"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "_build", "reach_sweep")


def build():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    src = os.path.join(HERE, "reach_sweep.c")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-Wno-unused-result", "-o", BIN, src, "-lm"])
    return BIN


def write_model(path, lo=None, hi=None, contact_offset=0.002):
    from isaacgymdyros_amd.model import load_model
    m = load_model()
    lo = list(m.dof_lower) if lo is None else lo
    hi = list(m.dof_upper) if hi is None else hi
    with open(path, "w") as f:
        nb = len(m.mv_parent)
        f.write("%d %d %d %d %d %.9g\n" % (nb, len(m.geoms), len(m.sc_proxies), len(m.sc_pairs), len(m.body_names), contact_offset))
        for i in range(nb):
            rng = (0.0, 0.0) if i == 0 else (lo[i - 1], hi[i - 1])
            vals = [m.mv_parent[i]] + list(m.mv_pos[i]) + [x for r in m.mv_rot0[i] for x in r] + list(m.mv_axis[i]) + list(rng)
            f.write(" ".join("%d" % v if k == 0 else "%.17g" % v for k, v in enumerate(vals)) + "\n")
        for g in m.geoms:
            vals = [g["moving"], g["gym"], g["type"]] + list(g["pos"]) + [x for r in g["rot"] for x in r] + list(g["size"])
            f.write(" ".join("%d" % v if k < 3 else "%.17g" % v for k, v in enumerate(vals)) + "\n")
        for p in m.sc_proxies:
            vals = [p["moving"], p["gym"]] + list(p["p0"]) + list(p["p1"]) + [p["radius"]]
            f.write(" ".join("%d" % v if k < 2 else "%.17g" % v for k, v in enumerate(vals)) + "\n")
        for a, b in m.sc_pairs:
            f.write("%d %d\n" % (a, b))
    return m


def envelope_ranges(m, leg, upper):
    """legs (dofs 0..11: what the policy drives) within +- leg of the initial pose, waist / arms / neck (dofs 12..32: held at that pose by the
    task's PD, tasks/dyros_dynamic_walk.py:504-509) within +- upper; clipped to the MJCF ranges"""
    from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
    hw = [leg if d < 12 else upper for d in range(len(INITIAL_DOF_POS))]
    lo = [max(l, q - w) for l, q, w in zip(m.dof_lower, INITIAL_DOF_POS, hw)]
    hi = [min(h, q + w) for h, q, w in zip(m.dof_upper, INITIAL_DOF_POS, hw)]
    return lo, hi


def run(samples, seed=1, corner=0.2, ranges="mjcf", envelope=(1.0, 0.2), threads=None):
    from isaacgymdyros_amd.model import load_model
    m = load_model()
    lo, hi = (None, None) if ranges == "mjcf" else envelope_ranges(m, *envelope)
    with tempfile.TemporaryDirectory() as td:
        mp = os.path.join(td, "model.txt")
        write_model(mp, lo, hi)
        env = dict(os.environ)
        if threads:
            env["OMP_NUM_THREADS"] = str(threads)
        out = subprocess.run([build(), mp, str(samples), str(seed), str(corner)], check=True, capture_output=True, text=True, env=env).stdout
    lines = out.strip().split("\n")
    head = lines[0].split()
    res = dict(samples=int(head[1]), samples_with_a_touch=int(head[3]), samples_with_an_uncovered_touch=int(head[5]), candidate_primitive_pairs=int(head[7]),
               ranges=ranges, envelope_leg_upper=list(envelope) if ranges != "mjcf" else None, seed=seed, corner_fraction=corner, pairs=[])
    names = [m.body_names[g] for g in m.mv_gym]          # a link (moving body) by the name of its first gym body
    for l in lines[1:]:
        t = l.split()
        a, b, cv = int(t[1]), int(t[2]), int(t[4])
        d = dict(zip(t[5::2], t[6::2]))
        res["pairs"].append(dict(a=names[a], b=names[b], covered_by_pair=cv, within_offset=int(d["near"]), touch=int(d["touch"]), proxy_hit=int(d["hit"]), proxy_miss=int(d["miss"]),
                                 proxy_false_positive=int(d["falsepos"]), min_distance=float(d["min_dist"]), max_miss_depth=float(d["max_miss_depth"])))
    return res


def table(res, limit=None):
    n = res["samples"]
    rows = ["| link pair | proxy pair | touch (share of samples) | proxy fires | touches the proxy misses (deepest overlap) | proxy fires with primitives > 2 mm apart |", "|---|---|---|---|---|---|"]
    ps = sorted(res["pairs"], key=lambda p: -p["touch"])
    for p in ps[:limit]:
        rows.append("| %s x %s | %s | %d (%.3f %%) | %d | %s | %s |" % (
            p["a"].replace("_Link", ""), p["b"].replace("_Link", ""), "#%d" % p["covered_by_pair"] if p["covered_by_pair"] >= 0 else "**none**", p["touch"], 100.0 * p["touch"] / n,
            p["proxy_hit"], ("%d (%.1f mm)" % (p["proxy_miss"], 1e3 * p["max_miss_depth"])) if p["covered_by_pair"] >= 0 else "-",
            ("%d (%.0f %% of its hits)" % (p["proxy_false_positive"], 100.0 * p["proxy_false_positive"] / max(1, p["proxy_hit"]))) if p["covered_by_pair"] >= 0 else "-"))
    return "\n".join(rows)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--corner", type=float, default=0.2)
    ap.add_argument("--ranges", default="mjcf", choices=["mjcf", "envelope"])
    ap.add_argument("--leg", type=float, default=1.0, help="envelope: half width of the leg joints' interval around the initial pose [rad]")
    ap.add_argument("--upper", type=float, default=0.2, help="envelope: half width for waist / arms / neck [rad]")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a.samples, a.seed, a.corner, a.ranges, (a.leg, a.upper))
    if a.out:          # (one pair per line)
        pairs = r.pop("pairs")
        with open(a.out, "w") as f:
            f.write(json.dumps(r)[:-1] + ', "pairs": [\n' + ",\n".join(json.dumps(p) for p in pairs) + "\n]}\n")
        r["pairs"] = pairs
    print("samples %d, with a touch %d (%.2f %%), with a touch no proxy pair covers %d (%.2f %%)" % (
        r["samples"], r["samples_with_a_touch"], 100.0 * r["samples_with_a_touch"] / r["samples"], r["samples_with_an_uncovered_touch"],
        100.0 * r["samples_with_an_uncovered_touch"] / r["samples"]))
    print(table(r))
