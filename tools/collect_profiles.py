"""Copies the judged summaries of `tools/round_artifacts.sh <tag>` from gpurun_out/<tag>/ into profiles/ and
regenerates profiles/pmc_traffic.json (the HBM bytes per launch that bench.py reports as roofline.traffic).
usage: python tools/collect_profiles.py <tag>"""
import glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
traffic_only = "--traffic-only" in sys.argv      # on the GPU box, between the PMC passes and the bench run
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
if not traffic_only:
  shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % tag))
  shutil.copy(os.path.join(src, "bench_torchrun1.json"), os.path.join(dst, "%s_bench_torchrun_1rank.json" % tag))
  stats = glob.glob(os.path.join(src, "prof", "**", "*kernel_stats.csv"), recursive=True)[0]
  with open(stats) as f:
    lines = f.readlines()
  with open(os.path.join(dst, "%s_kernel_stats_bench_16384.csv" % tag), "w") as f:
    f.writelines(lines[:12])                     # header + the ten largest rows; the rest are torch fill/copy kernels
summ = json.loads(open(os.path.join(src, "pmc", "summary.txt")).read())
name = "%s_pmc_summary_step_16384.json" % tag
json.dump(summ, open(os.path.join(dst, name), "w"), indent=1)
kname = next(n for n in ("dw_k_step_oct", "dw_k_step_quad", "dw_k_step") if n in summ)
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
k = summ[kname]
fetch, write = k["FETCH_SIZE"], k["WRITE_SIZE"]
traffic = {"16384": {
    "kernel": kname,
    "kernel_source_hash": kernel_source_hash(),          # bench.py quotes this record only for exactly these sources
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "lane_slots_per_env_step": k["SQ_INSTS_VALU"] * 64.0 / 16384.0,          # VALU instructions x 64 lanes, per env
    "fetch_size_kb": fetch, "write_size_kb": write,
    "correction": "FETCH_SIZE x2 (gfx950), units KB -> bytes x1024; dword-per-lane loads are uncalibrated, so this is an "
                  "upper bound; uncorrected sum = %d" % int((fetch + write) * 1024.0),
    "source": "profiles/" + name}}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
w = k["SQ_WAVES"]
print("per wave: VALU %.1fk LDS %.1fk SALU %.1fk; wait_any/wave_cycles %.2f; bank conflict/idx_active %.3f; VALU busy %.2f" % (
    k["SQ_INSTS_VALU"] / w / 1e3, k["SQ_INSTS_LDS"] / w / 1e3, k["SQ_INSTS_SALU"] / w / 1e3,
    k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"],
    k["SQ_ACTIVE_INST_VALU"] * 4 / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)))
print("traffic MB/launch: %.1f (algorithmic %.1f)" % (traffic["16384"]["hbm_bytes_per_launch"] / 1e6, 6816 * 16384 / 1e6))
