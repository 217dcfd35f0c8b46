#!/usr/bin/env python3
"""Condense a gpurun_out/<dir> produced by tools/gpu_profiles.sh into profiles/<round>/ (tracked):
kernel stats of the bench command, per-kernel averages of the PMC passes, bench JSON lines, test logs."""
import json
import os
import shutil
import sys

import pandas as pd

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
lines = ["# rocprofv3 evidence (MI355X, ROCm 7.2)", "",
         "Commands: `tools/gpu_profiles.sh` (bench.py under `rocprofv3 --kernel-trace --stats`, then separate "
         "`--pmc` passes as MI355X_MICROARCH.md prescribes).  `FETCH_SIZE`/`WRITE_SIZE` are in KiB; on gfx950 "
         "FETCH_SIZE counts 64 B per 128-B request, so read bytes = FETCH_SIZE x 1024 x 2.", ""]
for f in ("pytest_gpu.log", "smoke.log", "coop_ab.md", "rt_ab.md", "valu_rates.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for f in sorted(os.listdir(src)):
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
ks = os.path.join(src, "stats", "bench_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(dst, "bench_kernel_stats.csv"))
    df = pd.read_csv(ks)
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py` (kernel_stats)", "",
              "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    for _, r in df.iterrows():
        lines.append(f"| `{r['Name'].split('(')[0][:90]}` | {r['Calls']} | {r['AverageNs']:.0f} | {r['MinNs']} | "
                     f"{r['MaxNs']} | {r['Percentage']} |")
    kt = pd.read_csv(os.path.join(src, "stats", "bench_kernel_trace.csv"))
    kt["dur"] = kt["End_Timestamp"] - kt["Start_Timestamp"]
    kt["kernel"] = kt["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:90]
    g = kt.groupby(["kernel", "Grid_Size_X"])["dur"].agg(["count", "mean", "min", "max"]).reset_index()
    lines += ["", "per (kernel, grid) from the kernel trace of the same run:", "",
              "| kernel | rows (grid) | launches | mean us | min us | max us |", "|---|---|---|---|---|---|"]
    for _, r in g.iterrows():
        lines.append(f"| `{r['kernel']}` | {r['Grid_Size_X']} | {r['count']} | {r['mean']/1e3:.2f} | {r['min']/1e3:.2f} | "
                     f"{r['max']/1e3:.2f} |")
    # the HBM-sized legs: launches in time order, and the mean of the last 30 = the launches bench.py's HIP events
    # bracket (the ones before are construction, warm-up and the untimed first replay of the timed graph)
    big = kt[kt["Grid_Size_X"] >= (1 << 22)].sort_values("Start_Timestamp")
    lines += ["", "HBM-sized legs, launch by launch (us, time order) - the first launches after idle run at boost clocks, "
              "the power limiter overshoots, then the rate settles; `roofline.achieved` is measured on the last "
              "`--roofline-steps` launches:", ""]
    for (kname, grid), grp in big.groupby(["kernel", "Grid_Size_X"], sort=False):
        d = (grp["dur"] / 1e3).round().astype(int).tolist()
        tail = d[-30:] if len(d) >= 43 else d[-10:]
        lines.append(f"* `{kname}`, {grid} rows: {d} -> mean of the timed launches {sum(tail) / len(tail):.1f} us")
rows = []
for d in sorted(os.listdir(src)):
    cc = os.path.join(src, d, "bench_counter_collection.csv")
    if d.startswith("pmc_") and os.path.exists(cc):
        df = pd.read_csv(cc)
        df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:90]
        g = df.groupby(["kernel", "Grid_Size", "Counter_Name", "VGPR_Count", "Accum_VGPR_Count", "Scratch_Size",
                        "LDS_Block_Size"])["Counter_Value"].mean().reset_index()
        g.to_csv(os.path.join(dst, f"{d}_mean_per_launch.csv"), index=False)
        rows.append(g)
if rows:
    allc = pd.concat(rows)
    lines += ["", "## PMC passes (mean per launch)", "", "| kernel | rows | counter | mean per launch |", "|---|---|---|---|"]
    for _, r in allc.iterrows():
        lines.append(f"| `{r['kernel']}` | {r['Grid_Size']} | {r['Counter_Name']} | {r['Counter_Value']:.6g} |")
    lines += ["", "## Derived", ""]
    piv = allc.pivot_table(index=["kernel", "Grid_Size"], columns="Counter_Name", values="Counter_Value")
    traffic = {}
    for (k, gsz), r in piv.iterrows():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r and r["FETCH_SIZE"] == r["FETCH_SIZE"]:
            traffic[f"{k.replace(' ', '')}:{int(gsz)}"] = {
                "read_bytes": float(r["FETCH_SIZE"] * 1024 * 2), "write_bytes": float(r["WRITE_SIZE"] * 1024),
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)"}
    commit = os.environ.get("ABRK_PROFILE_COMMIT", "unknown")
    traffic["_commit"] = commit
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    # executed instructions per row, issue utilisation and effective clock per (kernel, rows): what bench.py's
    # `roofline.valu` block is computed from
    durs = {}
    ktp = os.path.join(src, "pmc_grbm", "bench_kernel_trace.csv")
    if not os.path.exists(ktp):
        import glob as _g
        cand = _g.glob(os.path.join(src, "pmc_grbm", "**", "*kernel_trace.csv"), recursive=True)
        ktp = cand[0] if cand else None
    if ktp:
        kt2 = pd.read_csv(ktp)
        kt2["dur"] = kt2["End_Timestamp"] - kt2["Start_Timestamp"]
        kt2["kernel"] = kt2["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:90]
        for (k, gsz), grp in kt2.groupby(["kernel", "Grid_Size_X"]):
            durs[(k, int(gsz))] = float(grp["dur"].mean())
    counters = {"_commit": commit}
    # grid-stride kernels (Sliding) launch fewer lanes than rows: the rows of each leg are in the bench line of the pass
    rows_of = {}
    try:
        line = [l for l in open(os.path.join(src, "pmc_sq.log")) if l.startswith("{")][-1]
        bj = json.loads(line)
        legs = [bj.get("roofline"), bj.get("roofline_full_outputs")] + list((bj.get("also") or {}).values())
        for leg in legs:
            if leg:
                rows_of[leg["kernel"].replace(" ", "")] = int(leg["batch"])
    except (OSError, IndexError, ValueError, KeyError):
        pass
    for (k, gsz), r in piv.iterrows():
        if "SQ_WAVES" not in r or r["SQ_WAVES"] != r["SQ_WAVES"] or gsz < 4096:
            continue
        w = r["SQ_WAVES"]
        rows = rows_of.get(k.replace(" ", ""), 0)
        if rows > gsz >= (1 << 20):  # more rows than lanes: per-row figures, keyed by the rows
            w = rows / 64.0
            gsz = rows
        e = {"valu_per_row": float(r["SQ_INSTS_VALU"] / w), "salu_per_row": float(r["SQ_INSTS_SALU"] / w),
             "wave_cycles_per_wave": float(4 * r["SQ_WAVE_CYCLES"] / w),
             "busy_cycles": float(r["SQ_BUSY_CYCLES"]) if "SQ_BUSY_CYCLES" in r else None}
        if "GRBM_GUI_ACTIVE" in r and r["GRBM_GUI_ACTIVE"] == r["GRBM_GUI_ACTIVE"] and (k, int(gsz)) in durs:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: shader clock = counter / 8 / kernel duration (of these short
            # PMC passes - the first launches after idle run above the sustained clock)
            gui = float(r["GRBM_GUI_ACTIVE"]) / 8.0
            e["clock_ghz"] = round(gui / durs[(k, int(gsz))], 3)
            # issue utilisation: executed VALU issue cycles (4 per fp64 wave-instruction, 2 per fp32) of all waves over
            # the SIMD-cycles the kernel was resident (1024 SIMDs x active cycles)
            cyc = 4 if "double" in k else 2
            e["issue_util"] = round(float(r["SQ_INSTS_VALU"]) * cyc / (1024 * gui), 4)
        counters[f"{k.replace(' ', '')}:{int(gsz)}"] = e
    json.dump(counters, open(os.path.join(dst, "counters.json"), "w"), indent=1)
    for (k, gsz), r in piv.iterrows():
        if gsz < 100000:
            continue
        bits = [f"* `{k}`, {gsz} rows:"]
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r and r["FETCH_SIZE"] == r["FETCH_SIZE"]:
            rd, wr = r["FETCH_SIZE"] * 1024 * 2, r["WRITE_SIZE"] * 1024
            bits.append(f"HBM traffic = {rd/1e6:.1f} MB read (FETCH_SIZE x2) + {wr/1e6:.1f} MB written = "
                        f"{(rd+wr)/gsz:.1f} B/row")
        if "SQ_WAVES" in r and r["SQ_WAVES"] == r["SQ_WAVES"]:
            w = r["SQ_WAVES"]
            bits.append(f"{r['SQ_INSTS_VALU']/w:.0f} VALU + {r['SQ_INSTS_SALU']/w:.0f} SALU instr/wave; "
                        f"wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f}, VALU-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f}, "
                        f"wait-any {4*r['SQ_WAIT_ANY']/w:.0f}, issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f} cycles")
        lines.append(" ".join(bits))
open(os.path.join(dst, "SUMMARY.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
