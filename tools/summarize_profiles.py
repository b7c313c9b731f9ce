#!/usr/bin/env python3
"""Turn the rocprofv3 outputs collected by tools/collect_profiles.sh into the committed summaries under profiles/.
Usage: python tools/summarize_profiles.py <tag> <round-prefix e.g. r01>"""
import collections
import csv
import glob as _glob


class glob:      # gpurun merges a call's files into existing directories: several runs' CSVs may sit side by side -- newest first
    @staticmethod
    def glob(pattern):
        import os as _os
        return sorted(_glob.glob(pattern), key=lambda f: -_os.path.getmtime(f))

import json
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
src = glob.glob(f"gpurun_out/{tag}_stats/*/*_kernel_stats.csv")[0]
shutil.copy(src, f"profiles/{rnd}_kernel_stats.csv")
rows = list(csv.DictReader(open(src)))
with open(f"profiles/{rnd}_kernel_stats.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats ({rnd})\n\n")
    f.write("Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 "
            "--prof none --no-cpu-baseline --no-extras`\n(65536 rows per step; calls = 5 warm-up + 7 windows x 20 timed steps; hg38-1Mb, table front end, d=64, L=5; raw CSV next to this file)\n\n")
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
    if "fused_fwd_kernel" in n or "fused_fwd32_kernel" in n:
        return "fused_fwd"
    if "fused_bwd_kernel" in n or "fused_bwd8_kernel" in n or "fused_bwdm_kernel" in n or "fused_bwdh_kernel" in n:
        return "fused_bwd"
    if "front_fwd_kernel" in n or "front_fwd2_kernel" in n or "front_fwd3_kernel" in n:
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
start = max(i for i in ids if "neg_sample" in fe[i][0])     # the last end-to-end step: from its negative sampling ...
plans = [i for i in ids if i > start and "row_count_kernel" in fe[i][0]]
stop = plans[1] if len(plans) > 1 else ids[-1] + 1            # ... to the plan of the step behind it (the model-step-only leg follows)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i in ids:
    if i < start or i >= stop:
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
for fsrc in sorted(_glob.glob(os.path.join("matcha_amd", "csrc", "*.h*"))):
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
            plans = [i for i in ids if i > start and "row_count_kernel" in vals[i][0]]
            stop = plans[1] if len(plans) > 1 else ids[-1] + 1
            for i in ids:
                c = cls(vals[i][0])
                if start <= i < stop and c in ("fused_fwd", "fused_bwd", "front_fwd", "front_bwd"):
                    res.setdefault(c, collections.OrderedDict())
                    res[c][name] = res[c].get(name, 0.0) + vals[i][1]
    # third pass (round 6): LDS occupancy and bank conflicts
    if glob.glob(f"gpurun_out/{tag}_mfma3/*/*_counter_collection.csv"):
        for name in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
            vals = load("mfma3", name)
            ids = sorted(vals)
            start = max(i for i in ids if "neg_sample" in vals[i][0])
            plans = [i for i in ids if i > start and "row_count_kernel" in vals[i][0]]
            stop = plans[1] if len(plans) > 1 else ids[-1] + 1
            for i in ids:
                c = cls(vals[i][0])
                if start <= i < stop and c in res:
                    res[c][name] = res[c].get(name, 0.0) + vals[i][1]
    for c, v in res.items():
        v["MfmaUtil_percent"] = round(100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 1)
        # pipe occupancy as shares of the kernel's SIMD time (cycles = GRBM_GUI_ACTIVE / 8 XCDs; 1024 SIMDs, 256 CUs): a vector instruction
        # holds its SIMD's issue port for 4 cycles (64 lanes on 16-wide ALUs); co-execution as a share of the matrix pipe's busy time
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        d_ = {"mfma_busy": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)}
        if "SQ_ACTIVE_INST_VALU" in v:
            d_["valu_busy"] = round(4.0 * v["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4)
            co = v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0)
            d_["coexec_of_mfma_busy"] = round(co / max(v["SQ_VALU_MFMA_BUSY_CYCLES"], 1.0), 4)
            d_["neither_pipe_issuing"] = round(max(0.0, 1.0 - d_["mfma_busy"] - d_["valu_busy"] + co / (cyc * 1024)), 4)
        if "SQ_LDS_IDX_ACTIVE" in v:
            d_["lds_busy"] = round(v["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 4)
            d_["lds_bank_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
        v["derived"] = d_
    json.dump({"csrc_sha16": h.hexdigest()[:16],
               "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 --prof none "
                          "--no-cpu-baseline  (second pass: --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES); tools/collect_profiles.sh",
               "note": "sums over the launches of one step; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so MfmaUtil = 100 * MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); "
                       "SQ_VALU_MFMA_COEXEC_CYCLES counts cycles in which vector and matrix instructions execute together",
               "kernels": res}, open(f"profiles/{rnd}_pmc_mfma.json", "w"), indent=1)
    for c, v in res.items():
        print(f"{c:10s} MfmaUtil {v['MfmaUtil_percent']} %  coexec {v.get('SQ_VALU_MFMA_COEXEC_CYCLES')}")


# ---- the embedding gather alone + the other configurations (tools/collect_profiles.sh, second half) -------------------------
def stats_md(subdir, out_md, title, cmd, n=18, must=()):
    g = glob.glob(f"gpurun_out/{tag}_{subdir}/*/*_kernel_stats.csv")
    if not g:
        return None
    shutil.copy(g[0], out_md.replace(".md", ".csv"))
    rws = list(csv.DictReader(open(g[0])))
    with open(out_md, "w") as f:
        f.write(f"# {title}\n\nCommand (on the MI355X box): `{cmd}`\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in [r for i, r in enumerate(rws) if i < n or any(m in r["Name"] for m in must)]:
            f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    return rws


stats_md("d128_stats", f"profiles/{rnd}_d128_kernel_stats.md", f"rocprofv3 --kernel-trace --stats, BASELINE configs[3] shape on one GPU ({rnd})",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 3 --prof none --no-cpu-baseline --no-extras --layout hg38_100kb --dim 128")
stats_md("c5_stats", f"profiles/{rnd}_c5_kernel_stats.md", f"rocprofv3 --kernel-trace --stats, BASELINE configs[4] (C5: 1 M nodes, d = 256, 10 M known hyperedges) on one GPU ({rnd})",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 3 --prof none --no-cpu-baseline --no-extras --layout c5 --dim 256 --ks 2,3,4,5,6,7,8 --rows 16384 --edges 10000000")
grows = stats_md("gather_stats", f"profiles/{rnd}_gather_kernel_stats.md", f"rocprofv3 --kernel-trace --stats of the embedding gather alone ({rnd})",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- python tools/gather_bench.py  (gather_rows_kernel<1>: d = 64, three tables; <4>: d = 256)")
if grows and glob.glob(f"gpurun_out/{tag}_gather_fetch/*/*_counter_collection.csv"):
    def load2(d, name):
        f = glob.glob(f"gpurun_out/{tag}_{d}/*/*_counter_collection.csv")[0]
        out = []
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "gather_rows_kernel" in r["Kernel_Name"]:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
        return sorted(out)
    fe2, wr2 = load2("gather_fetch", "FETCH_SIZE"), load2("gather_write", "WRITE_SIZE")
    # tools/gather_bench.py launches 13 gathers per table (3 warm-up + 10 timed), six cases in this order
    cases = [("C2: 3 067 nodes x 64 (L2-resident)", 64, 1 << 20), ("1 M nodes x 64 (244 MiB)", 64, 1 << 24), ("16 M nodes x 64 (4 GiB)", 64, 1 << 24),
             ("C5: 1 M nodes x 256 (1 GiB)", 256, 1 << 22), ("16 M nodes x 64 (4 GiB), step-sized launch", 64, 65536 * 5),
             ("C5: 1 M nodes x 256 (1 GiB), step-sized launch", 256, 16384 * 8)]
    tr = glob.glob(f"gpurun_out/{tag}_gather_stats/*/*_kernel_trace.csv")
    durs = []
    if tr:
        durs = [(int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(tr[0]))
                if "gather_rows_kernel" in r["Kernel_Name"]]
        durs = [d_ for _, d_ in sorted(durs)]
    res = []
    for ci, (name, d, T) in enumerate(cases):
        f_ = [v for _, _, v in fe2[13 * ci + 3:13 * ci + 13]]
        w_ = [v for _, _, v in wr2[13 * ci + 3:13 * ci + 13]]
        if not f_ or not w_:
            continue
        rd = sum(f_) / len(f_) * 1024 * 2             # KB -> B; x2: gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md §HBM)
        wrb = sum(w_) / len(w_) * 1024
        us = durs[13 * ci + 3:13 * ci + 13]
        avg_us = sum(us) / len(us) if us else None
        res.append({"table": name, "d": d, "rows_per_launch": T, "rocprof_avg_us": None if avg_us is None else round(avg_us, 1),
                    "read_gbs_of_4d_plus_8": None if avg_us is None else round(T * (4 * d + 8) / avg_us / 1e3, 1),
                    "frac_of_8tbs_read_roof": None if avg_us is None else round(T * (4 * d + 8) / avg_us / 1e3 / 8000.0, 4),
                    "algorithmic_read_bytes": T * (4 * d + 8), "algorithmic_write_bytes": T * 4 * d,
                    "pmc_fetch_bytes_per_launch": rd, "pmc_write_bytes_per_launch": wrb,
                    "fetch_over_algorithmic_read": round(rd / (T * (4 * d + 8)), 3), "write_over_algorithmic_write": round(wrb / (T * 4 * d), 3)})
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of tools/gather_bench.py); FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM "
                       "(it counts Infinity-Cache hits too: requests that leave the XCD's L2); averages over the 10 timed launches per table",
               "cases": res}, open(f"profiles/{rnd}_gather_pmc.json", "w"), indent=1)
    with open(f"profiles/{rnd}_gather_kernel_stats.md", "a") as f:
        f.write("\nPer table (the kernel-trace durations of the 10 timed launches of each case, in launch order; HBM bytes from the two PMC passes):\n\n"
                "| table | d | rows per launch | avg us | read GB/s (4d+8) | frac of 8 TB/s | FETCH / algorithmic read | WRITE / algorithmic write |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in res:
            f.write(f"| {r['table']} | {r['d']} | {r['rows_per_launch']} | {r['rocprof_avg_us']} | {r['read_gbs_of_4d_plus_8']} | {r['frac_of_8tbs_read_roof']} | "
                    f"{r['fetch_over_algorithmic_read']} | {r['write_over_algorithmic_write']} |\n")
    for r in res:
        print(r["table"], "fetch/alg", r["fetch_over_algorithmic_read"], "write/alg", r["write_over_algorithmic_write"])


# ---- the gather the model step executes on HBM-resident tables (tools/front_gather_bench.py = bench.py roofline_gather_in_step) ----
frows = stats_md("front_stats", f"profiles/{rnd}_gather_in_step_kernel_stats.md", f"rocprofv3 --kernel-trace --stats of the in-step gather on HBM-resident tables ({rnd})",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- python tools/front_gather_bench.py  (front_fwd_kernel: d = 64, 16 M x 64 table, "
                 "327 681 tokens per launch; embed_fwd_kernel<4>: d = 256, C5 table 1 M x 256, 131 073 tokens per launch)", n=8, must=("front_fwd_kernel", "front_fwd2_kernel", "front_fwd3_kernel", "embed_fwd_kernel"))
ff = glob.glob(f"gpurun_out/{tag}_front_fetch/*/*_counter_collection.csv")
if frows and ff:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(ff[0])):
        if r["Counter_Name"] == "FETCH_SIZE":
            for key in ("front_fwd3_kernel", "front_fwd2_kernel", "front_fwd_kernel", "embed_fwd_kernel"):
                if key in r["Kernel_Name"]:
                    acc[key].append(float(r["Counter_Value"]) * 1024 * 2)
    # round 4: embed_fwd rebuilds the attribute row from the node id (attr_mode 1: nothing read); front_fwd reads it as one 128-byte unit
    # (rows padded to 32 floats), of which 4 * 24 bytes are algorithmic
    cases = {"front_fwd3_kernel": ("front_fwd3 (d = 64, wave-independent, attribute rows rebuilt from the id) on a 16 M x 64 table (4 GiB)", 327681, 8 + 256),
             
             "front_fwd_kernel": ("front_fwd (d = 64) on a 16 M x 64 table (4 GiB)", 327681, 8 + 256 + 4 * 24),
             "embed_fwd_kernel": ("embed_fwd (d = 256) on the C5 table 1 M x 256 (1 GiB)", 131073, 8 + 1024)}
    res = []
    for key, vals in acc.items():
        name, tokens, rb = cases[key]
        vals = vals[2:]                                   # two warm-up forwards
        avg = sum(vals) / max(len(vals), 1)
        us = [float(r["AverageNs"]) / 1e3 for r in frows if key in r["Name"]]
        res.append({"kernel": key, "table": name, "tokens_per_launch": tokens, "algorithmic_read_bytes_per_token": rb, "pmc_fetch_bytes_per_launch": avg,
                    "fetch_over_algorithmic_read": round(avg / (tokens * rb), 3), "rocprof_avg_us": round(us[0], 1) if us else None,
                    "frac_of_8tbs_read_roof": round(tokens * rb / us[0] / 1e3 / 8000.0, 4) if us else None})
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE pass of tools/front_gather_bench.py (FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); uniform random ids, "
                       "inference forwards; the kernels also write 4 d bytes per token (X / x0 rows)", "cases": res},
              open(f"profiles/{rnd}_gather_in_step_pmc.json", "w"), indent=1)
    for r in res:
        print(r["kernel"], "fetch/alg", r["fetch_over_algorithmic_read"], "frac", r["frac_of_8tbs_read_roof"])


# ---- the adj front end (the reference's own mode, Modules.py:176-201): tools/collect_adj_profiles.sh -------------------------------
arows = stats_md("adj_stats", f"profiles/{rnd}_adj_kernel_stats.md", f"rocprofv3 --kernel-trace --stats, adj front end, 65 536 rows per step ({rnd})",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --prof none --no-cpu-baseline --no-extras --front-end adj", n=24)
stats_md("adj384_stats", f"profiles/{rnd}_adj384_kernel_stats.md", f"rocprofv3 --kernel-trace --stats, adj front end at the reference's batch of 384 rows ({rnd})",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --rows 384 --prof none --no-cpu-baseline --no-extras --front-end adj", n=30)
stats_md("table384_stats", f"profiles/{rnd}_table384_kernel_stats.md", f"rocprofv3 --kernel-trace --stats, table front end at the reference's batch of 384 rows ({rnd})",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --rows 384 --prof none --no-cpu-baseline --no-extras", n=30)
af = glob.glob(f"gpurun_out/{tag}_adj_fetch/*/*_counter_collection.csv")
if arows and af:
    def load3(d, name):
        f = glob.glob(f"gpurun_out/{tag}_{d}/*/*_counter_collection.csv")
        out = collections.defaultdict(list)
        if not f:
            return out
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == name:
                for key in ("adj_fused_fwd_kernel", "adj_recon_kernel", "adj_fused_bwd_kernel", "front_bwd_kernel", "adj_scan_kernel", "adj_scatter_kernel", "adj_hist_kernel"):
                    if key in r["Kernel_Name"]:
                        out[key].append(float(r["Counter_Value"]))
        return out
    fe3, wr3 = load3("adj_fetch", "FETCH_SIZE"), load3("adj_write", "WRITE_SIZE")
    mb3, ga3 = load3("adj_mfma1", "SQ_VALU_MFMA_BUSY_CYCLES"), load3("adj_mfma1", "GRBM_GUI_ACTIVE")
    res = {}
    for key in fe3:
        us = [float(r["AverageNs"]) / 1e3 for r in arows if key in r["Name"]]
        last = lambda v: v[-1] if v else None            # the last launch of the pass: a steady-state step
        e = {"rocprof_avg_us": round(us[0], 1) if us else None,
             "hbm_fetch_bytes_per_launch": None if not fe3[key] else last(fe3[key]) * 1024 * 2,      # x2: gfx950 FETCH_SIZE tallies 128-B requests at 64 B
             "hbm_write_bytes_per_launch": None if not wr3[key] else last(wr3[key]) * 1024}
        if mb3[key] and ga3[key]:
            e["MfmaUtil_percent"] = round(100.0 * last(mb3[key]) / (last(ga3[key]) / 8 * 1024), 1)
        res[key] = e
    json.dump({"csrc_sha16": h.hexdigest()[:16],
               "note": "adj front end at 65 536 rows per step (hg38 1 Mb, k in {2..5}, d = 64): separate rocprofv3 passes --pmc FETCH_SIZE / --pmc WRITE_SIZE / "
                       "--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of bench.py --steps 2 --warmup 1 --front-end adj (tools/collect_adj_profiles.sh); values of "
                       "the last launch of each kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM; MfmaUtil = 100 * MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs)",
               "kernels": res}, open(f"profiles/{rnd}_adj_pmc.json", "w"), indent=1)
    for k_, v in res.items():
        print(k_, v)
