#!/usr/bin/env python3
"""Turn the rocprofv3 outputs collected by tools/collect_profiles.sh into the committed summaries under profiles/.
Usage: python tools/summarize_profiles.py <tag> <round-prefix e.g. r01>"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
src = glob.glob(f"gpurun_out/{tag}_stats/*/*_kernel_stats.csv")[0]
shutil.copy(src, f"profiles/{rnd}_kernel_stats.csv")
rows = list(csv.DictReader(open(src)))
with open(f"profiles/{rnd}_kernel_stats.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats ({rnd})\n\n")
    f.write("Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 "
            "--prof none --no-cpu-baseline`\n(65536 rows per step; calls = warm-up + timed steps of the end-to-end leg and of the model-step-only leg; hg38-1Mb, table front end, d=64, L=5; raw CSV next to this file)\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:26]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |\n")


def load(d, name):
    f = glob.glob(f"gpurun_out/{tag}_{d}/*/*_counter_collection.csv")[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            out[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return out


def cls(n):
    if "gemm_tn_kernel" in n:
        return "gemm_tn"
    if "gemm_lds_kernel<false" in n:
        return "gemm_nt"
    if "gemm_lds_kernel<true" in n:
        return "gemm_nn"
    if "fused_fwd_kernel" in n:
        return "fused_fwd"
    if "fused_bwd_kernel" in n:
        return "fused_bwd"
    if "front_fwd_kernel" in n:
        return "front_fwd"
    if "front_bwd_kernel" in n:
        return "front_bwd"
    if "tg_" in n:
        return "embed_scatter"
    if "gather_rows" in n:
        return "gather_rows"
    for k in ("attn_fwd", "attn_bwd", "embed_fwd", "embed_scatter", "ln3_fwd", "ln3_bwd", "head_fwd", "head_bwd", "adamw_kernel", "neg_sample", "adj_encode"):
        if k in n:
            return k.replace("_kernel", "")
    return None


fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
ids = sorted(fe)
start = max(i for i in ids if "neg_sample" in fe[i][0])     # the last full step
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i in ids:
    if i < start:
        continue
    c = cls(fe[i][0])
    if c:
        a = agg[c]
        a[0] += 1
        a[1] += fe[i][1] * 1024 * 2          # KB -> B; x2: gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md §HBM)
        a[2] += wr.get(i, (None, 0))[1] * 1024
out = {c: {"launches_per_step": v[0], "hbm_read_bytes_per_launch": v[1] / v[0], "hbm_write_bytes_per_launch": v[2] / v[0],
           "hbm_bytes_per_launch": (v[1] + v[2]) / v[0]} for c, v in agg.items()}
import hashlib
import os
h = hashlib.sha256()
for fsrc in sorted(glob.glob(os.path.join("matcha_amd", "csrc", "*.h*"))):
    h.update(open(fsrc, "rb").read())
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of bench.py --steps 2 --warmup 1, 65536 rows/step); "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM; bytes per launch averaged over the launches of the class in one step",
           "csrc_sha16": h.hexdigest()[:16],          # bench.py quotes these figures only for the kernel sources they were measured on
           "workload": ["hg38_1mb", 64, "table", 65536],
           "classes": out}, open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1)
for c, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"]):
    print(f"{c:14s} x{v['launches_per_step']:2d}  {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")


# ---- matrix-core utilisation (two more PMC passes, when collected) --------------------------------------------------
if glob.glob(f"gpurun_out/{tag}_mfma1/*/*_counter_collection.csv"):
    names1, names2 = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], ["SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"]
    res = collections.OrderedDict()
    for d, names in (("mfma1", names1), ("mfma2", names2)):
        for name in names:
            vals = load(d, name)
            ids = sorted(vals)
            start = max(i for i in ids if "neg_sample" in vals[i][0])
            for i in ids:
                c = cls(vals[i][0])
                if i >= start and c in ("fused_fwd", "fused_bwd", "front_fwd", "front_bwd"):
                    res.setdefault(c, collections.OrderedDict())
                    res[c][name] = res[c].get(name, 0.0) + vals[i][1]
    for c, v in res.items():
        v["MfmaUtil_percent"] = round(100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 1)
    json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 --prof none "
                          "--no-cpu-baseline  (second pass: --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES); tools/collect_profiles.sh",
               "note": "sums over the launches of one step; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so MfmaUtil = 100 * MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); "
                       "SQ_VALU_MFMA_COEXEC_CYCLES counts cycles in which vector and matrix instructions execute together",
               "kernels": res}, open(f"profiles/{rnd}_pmc_mfma.json", "w"), indent=1)
    for c, v in res.items():
        print(f"{c:10s} MfmaUtil {v['MfmaUtil_percent']} %  coexec {v.get('SQ_VALU_MFMA_COEXEC_CYCLES')}")
