"""Register / LDS / scratch budget of the shipped kernels, read from the compiler (hipcc -Rpass-analysis=kernel-resource-usage,
the flags of isaacgymdyros_amd/build.py; cross-compiles without a GPU).  The octet kernels run two waves per SIMD only if they
fit 256 registers and 40 KB of LDS per workgroup, and a spill inside a memory phase costs a full round trip per reload (r03:
nine reloads in the encoder epilogue were 30 % of the step) -- so the budget is an assertion, not a comment: a change that
spills fails here instead of showing up as a few per cent in a bench line."""
import json
import os
import re
import subprocess

import pytest

from isaacgymdyros_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resource_usage(src, extra):
    cmd = [build.hipcc()] + build.FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, os.path.join(build.CSRC, src)]
    err = subprocess.run(cmd, cwd=build.CSRC, capture_output=True, text=True, check=True).stderr
    out, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            k = re.search(r"(dw_k_[a-z_]+?)ILb([01])E(?:Li(n?\d)E)?(?:Li(n?\d)E)?", name)
            # (the octet kernels exist in two builds, dw_oct_kernels.hip: "<..>" is the two-waves-per-SIMD one, "<..,1>" the spread one; the
            #  step kernels also per torch flavour of the post phase's norms -- the GPU flavour, the default, is the one named plainly -- and as
            #  the build that reads every switch at run time, which tests with an injected noise record or frozen physics get)
            cur = name
            if k:
                g = [x for x in k.groups()[2:] if x is not None]
                is_step = "_step_" in k.group(1)
                flav = g[-1] if is_step and g else None
                wpe = g[0] if ("_oct" in k.group(1) and g) else None
                cur = "%s<%s%s>" % (k.group(1), "true" if k.group(2) == "1" else "false", ",1" if wpe == "1" else "")
                if flav == "0":
                    cur += "[cpu flavour]"
                elif flav == "n1":
                    cur += "[test build]"
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def usage():
    u = {}
    from concurrent.futures import ThreadPoolExecutor
    todo = [(src, extra) for src, extra in build.SOURCES if src != "dw_hip.hip"]
    with ThreadPoolExecutor(max_workers=len(todo)) as pool:          # (one hipcc per translation unit, side by side)
        for r in pool.map(lambda a: resource_usage(*a), todo):
            u.update(r)
    # (the tracked record under profiles/ is refreshed on request only: DW_WRITE_PROFILES=1 python -m pytest tests/test_kernel_resources.py)
    if os.environ.get("DW_WRITE_PROFILES") == "1":
        json.dump(u, open(os.path.join(ROOT, "profiles", "r06_kernel_resources.json"), "w"), indent=1, sort_keys=True)
    return u


def test_octet_kernels_fit_two_waves_per_simd_without_scratch(usage):
    for k in ("dw_k_step_oct<false>", "dw_k_simulate_oct<false>", "dw_k_simulate_oct<true>"):
        r = usage[k]
        assert r["ScratchSize"] == 0, (k, r)
        assert r["Occupancy"] == 2, (k, r)
        assert r["VGPRs"] + r["AGPRs"] <= 256, (k, r)
        assert r["LDS Size"] <= 40960, (k, r)


def test_spread_build_of_the_octet_kernels(usage):
    """Launches with no more waves than SIMDs use the build declared for one wave per SIMD: the same code and LDS, and with the
    whole register file to itself nothing may spill (the height-field step kernel takes AGPRs for what its two-wave build spills)."""
    for k in ("dw_k_step_oct<false%s>", "dw_k_simulate_oct<false%s>", "dw_k_simulate_oct<true%s>", "dw_k_step_oct<true%s>"):
        a, b = usage[k % ""], usage[k % ",1"]
        assert b["Occupancy"] == 1 and b["ScratchSize"] == 0, (k, b)
        assert b["LDS Size"] == a["LDS Size"] and b["VGPRs"] + b["AGPRs"] <= 512, (k, a, b)


def test_hex_instantiation_budget(usage):
    """The 16-lanes-per-env instantiation (dw_hex_kernels.hip; launches of at most 4096 envs): one wave per SIMD by construction, so
    the whole register file and no scratch, and LDS small enough that a CU could hold more than the four workgroups it gets."""
    for k in ("dw_k_step_hex<false>", "dw_k_step_hex<true>", "dw_k_simulate_hex<false>", "dw_k_simulate_hex<true>", "dw_k_step_hex<false>[cpu flavour]"):
        r = usage[k]
        assert r["ScratchSize"] == 0 and r["Occupancy"] == 1, (k, r)
        assert r["VGPRs"] + r["AGPRs"] <= 512 and r["LDS Size"] <= 40960, (k, r)


def test_terrain_step_kernel_scratch_is_bounded(usage):
    """The height-field variant of the step kernel carries 36 more words of contact frames through the solve; what it spills is
    recorded and may not grow (work list: DESIGN.md section 7)."""
    # (100 B with the default machine scheduler, 152 B with the iterative-ilp strategy the octet unit is built with since the end of
    #  round 3: the strategy is worth 2.2 % on the flat step kernel and leaves this kernel's time where it was, 0.216 ms at 16384 envs)
    # (round 6, row-split recursion of the inward pass: 160 -> 168 B; same-box A/B of the height-field step 0.1612 -> 0.1614 ms,
    #  gpurun_out/r6a_abt.txt -- within the noise of the pool)
    r = usage["dw_k_step_oct<true>"]
    assert r["Occupancy"] == 2 and r["ScratchSize"] <= 168, r


def test_small_kernels_do_not_spill(usage):
    """The fused TocabiAMPLower kernels, the stateless row f-3 functions and dw_k_reset's siblings: no scratch, and the LDS of the
    four-wave step-end kernel leaves room for eight workgroups per CU."""
    seen = 0
    for k, r in usage.items():
        if "dw_k_amp_step_oct" in k:
            # (the one-launch step of round 6, off by default: the compiler hoists the task tables' ~150 invariant fields out of the regions and
            #  spills them around the octet substep's 256 registers -- 528 B, reloaded once per region; with the tables as by-value kernel
            #  arguments the whole 1 140 B went to scratch.  Part of why it measured slower than the five launches,
            #  isaacgymdyros_amd/tocabi_amp_lower.py amp_one_launch; bounded here)
            assert r["ScratchSize"] <= 560 and r["Occupancy"] == 2 and r["LDS Size"] <= 40960, (k, r)
        elif "dw_k_amp" in k or "dw_k_newwalk" in k or "dw_k_body_positions" in k:
            seen += 1
            assert r["ScratchSize"] == 0, (k, r)
    assert seen >= 10
    end = [r for k, r in usage.items() if "dw_k_amp_step_end" in k][0]
    assert end["LDS Size"] <= 20480, end


def test_ppo_kernels_budget(usage):
    """csrc/dw_ppo.hip: no scratch anywhere; the matrix-core kernels' LDS leaves one workgroup of four waves per CU (dwp_mlp: the input rows and three
    fp16 images, dwp_policy: the fp32 input rows and one image) and dwp_wgrad keeps at least two waves per SIMD for its latency."""
    ppo = {k: r for k, r in usage.items() if any(x in k for x in ("k_mlp", "k_wgrad", "k_policy", "k_adam", "k_grad_stats", "k_finish", "k_gae", "k_roll_pre", "k_roll_post",
                                                                   "k_loss", "k_relu_bwd", "k_bias_relu", "k_stage_obs", "k_retile"))}
    assert len(ppo) >= 14, sorted(ppo)
    for k, r in ppo.items():
        assert r["ScratchSize"] == 0, (k, r)
    mlp = [r for k, r in ppo.items() if "k_mlp" in k][0]
    pol = [r for k, r in ppo.items() if "k_policy" in k][0]
    wg = [r for k, r in ppo.items() if "k_wgrad" in k][0]
    assert 80 * 1024 <= mlp["LDS Size"] <= 96 * 1024 and pol["LDS Size"] <= 104 * 1024
    assert wg["LDS Size"] == 0 and wg["Occupancy"] >= 2
