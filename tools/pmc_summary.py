"""Aggregates rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel means."""
import csv, glob, os, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()     # template kernels: "void dw_k_step_quad<false>(...)"
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not k.startswith("dw_k"):
        continue
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["dispatches"] = max(len(v) for v in d.values())
print(json.dumps(res, indent=1, sort_keys=True))
