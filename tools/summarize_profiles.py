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


def norm(k):
    return k.split("(")[0].replace("void abrk::", "").replace(" ", "")


def legs_of(path):
    """every roofline-like block ({kernel, grid_threads, batch, ...}) of the bench JSON line(s) in a log file"""
    out = []
    try:
        lines = [l for l in open(path) if l.startswith("{")]
    except OSError:
        return out

    def walk(o):
        if isinstance(o, dict):
            if "kernel" in o and "batch" in o and "grid_threads" in o:
                out.append(o)
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)

    for l in lines:
        try:
            walk(json.loads(l))
        except ValueError:
            pass
    return out


# (kernel, Grid_Size as rocprofv3 prints it) -> rows of the launch.  The grid-stride kernels (Sliding, AvoidObstacles,
# the six-row OSC law's first pass) cap their grid, so rows != grid for them: bench.py states both for every leg.
ROWS = {}
for f in sorted(os.listdir(src)):
    if f.endswith((".log", ".json")):
        for leg in legs_of(os.path.join(src, f)):
            ROWS[(norm(leg["kernel"]), int(leg["grid_threads"]))] = int(leg["batch"])


def osc_args(kn):
    """(arm + type prefix, [KM, USE_C, FEAT, PASS, NOTS, EEF]) of a normalised osc_kernel name, or None"""
    if not kn.startswith("osc_kernel<") or not kn.endswith(">"):
        return None
    parts = kn[:-1].split(",")
    if len(parts) < 7:
        return None
    tail = parts[-6:]
    if tail[-1] not in ("true", "false") or tail[-2] not in ("true", "false") or not tail[-3].isdigit():
        return None  # a name of an earlier round (no EEF argument)
    return ",".join(parts[:-6]), tail


def rows_for(kernel, grid):
    """rows a launch of `kernel` with `grid` threads processed (the grid itself for one-lane-per-row kernels)"""
    kn = norm(kernel)
    if (kn, int(grid)) in ROWS:
        return ROWS[(kn, int(grid))]
    # the six-row law's second pass (PASS = 0 at a fixed 2048-block grid) belongs to the leg of its first pass
    # (PASS = 1, with or without the training-signal output)
    # (since round 5 the second pass carries the first pass's NOTS flag: <.., 1, true> goes with <.., 0, true>)
    # (round 6: osc_kernel<arm, T, KM, USE_C, FEAT, PASS, NOTS, EEF> - the first pass may be the EEF instantiation)
    a = osc_args(kn)
    if a and a[1][-3] == "0" and int(grid) == 8 * 256 * 64:
        big = [b for (k, g), b in ROWS.items() if osc_args(k) and osc_args(k)[0] == a[0] and osc_args(k)[1][:-3] == a[1][:-3]
               and osc_args(k)[1][-3] == "1"]
        if big:
            return max(big)
    return int(grid)
lines = ["# rocprofv3 evidence (MI355X, ROCm 7.2)", "",
         f"Commands: `{os.environ.get('ABRK_PROFILE_SCRIPT', 'tools/gpu_profiles_r6.sh')}` (bench.py under `rocprofv3 --kernel-trace --stats`, then separate "
         "`--pmc` passes as MI355X_MICROARCH.md prescribes).  `FETCH_SIZE`/`WRITE_SIZE` are in KiB; on gfx950 "
         "FETCH_SIZE counts 64 B per 128-B request, so read bytes = FETCH_SIZE x 1024 x 2.", ""]
for f in ("pytest_gpu.log", "smoke.log", "coop_ab.md", "rt_ab.md", "valu_rates.txt", "host.txt", "osc6_step_trace.txt"):
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
    lines += ["", "per (kernel, grid) from the kernel trace of the same run (rows: what the launch processed - the "
              "grid-stride kernels launch fewer threads than rows; the bench line of the run states both):", "",
              "| kernel | grid threads | rows | launches | mean us | min us | max us |", "|---|---|---|---|---|---|---|"]
    for _, r in g.iterrows():
        lines.append(f"| `{r['kernel']}` | {r['Grid_Size_X']} | {rows_for(r['kernel'], r['Grid_Size_X'])} | {r['count']} | "
                     f"{r['mean']/1e3:.2f} | {r['min']/1e3:.2f} | {r['max']/1e3:.2f} |")
    # the figure every bench JSON's `roofline.frac` must reproduce: algorithmic bytes x rows / trace mean / 8 TB/s
    legs = {}
    for f in sorted(os.listdir(src)):
        # (the bench lines of the PMC passes are left out: counter collection serialises the launches and their short
        #  runs are not the sustained protocol)
        if f.endswith((".log", ".json")) and not f.startswith("pmc_"):
            for leg in legs_of(os.path.join(src, f)):
                if leg["batch"] >= (1 << 20) and "bytes_per_eval" in leg:
                    legs.setdefault((norm(leg["kernel"]), int(leg["grid_threads"])), []).append((f, leg))
    if legs:
        lines += ["", "HBM-sized legs: the kernel-trace mean of ALL launches of the leg's (kernel, grid) in the profiled run "
                  "against the `roofline` blocks of the bench JSONs (frac = algorithmic bytes / time / 8 TB/s):", "",
                  "| kernel | rows | B/row | launches in trace | trace mean us | frac from trace | bench JSONs: us (frac) |",
                  "|---|---|---|---|---|---|---|"]
        for (kn, grid), ll in sorted(legs.items()):
            sel = kt[(kt["kernel"].str.replace(" ", "") == kn[:90]) & (kt["Grid_Size_X"] == grid)]
            if sel.empty:
                continue
            leg = ll[0][1]
            mean_us = sel["dur"].mean() / 1e3
            extra = 0.0
            oa = osc_args(kn)
            if oa and oa[1][-3] == "1":  # + the dense second pass of the same calls (PASS = 0, same NOTS, same EEF)
                second = ",".join([oa[0]] + oa[1][:3] + ["0", oa[1][4], oa[1][5]]) + ">"
                s2 = kt[(kt["kernel"].str.replace(" ", "") == second[:90]) & (kt["Grid_Size_X"] == 8 * 256 * 64)]
                if not s2.empty:
                    extra = s2["dur"].mean() / 1e3
            frac = leg["batch"] * leg["bytes_per_eval"] / ((mean_us + extra) * 1e-6) / 8e12
            js = "; ".join(f"{f}: {l['us_per_launch']:.1f} ({l['frac']:.3f})" for f, l in ll)
            lines.append(f"| `{kn[:100]}` | {leg['batch']} | {leg['bytes_per_eval']} | {len(sel)} | {mean_us:.1f}"
                         f"{f' + {extra:.1f} (second pass)' if extra else ''} | {frac:.3f} | {js} |")
    # the HBM-sized legs: launches in time order, and the mean of the last 30 = the launches bench.py's HIP events
    # bracket (the ones before are construction, warm-up and the untimed first replay of the timed graph)
    big = kt[kt["Grid_Size_X"] >= (1 << 22)].sort_values("Start_Timestamp")
    lines += ["", "HBM-sized legs, launch by launch (us, time order) - the first launches after idle run at boost clocks, "
              "the power limiter overshoots, then the rate settles; `roofline.achieved` is measured on the last "
              "`--roofline-steps` launches:", ""]
    for (kname, grid), grp in big.groupby(["kernel", "Grid_Size_X"], sort=False):
        d = (grp["dur"] / 1e3).round().astype(int).tolist()
        t_end = ((grp["End_Timestamp"] - grp["Start_Timestamp"].iloc[0]) / 1e9).tolist()
        last = [x for x, te in zip(d, t_end) if te > t_end[-1] - 1.0]
        shown = d if len(d) <= 60 else d[:40] + ["..."] + d[-10:]
        lines.append(f"* `{kname}`, {rows_for(kname, grid)} rows, {len(d)} launches over {t_end[-1]:.2f} s: {shown} -> mean of "
                     f"all {sum(d) / len(d):.1f} us, of the last second {sum(last) / len(last):.1f} us")
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
            traffic[f"{k.replace(' ', '')}:{rows_for(k, gsz)}"] = {
                "grid_threads": int(gsz),
                "read_bytes": float(r["FETCH_SIZE"] * 1024 * 2), "write_bytes": float(r["WRITE_SIZE"] * 1024),
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)"}
    commit = os.environ.get("ABRK_PROFILE_COMMIT", "unknown")
    traffic["_commit"] = commit
    try:  # the hash of the kernel / host sources of the tree the profile was taken on (bench.py kernel_sources_hash)
        import importlib.util as _iu

        _sp = _iu.spec_from_file_location("abrk_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        _b = _iu.module_from_spec(_sp)
        _sp.loader.exec_module(_b)
        traffic["_sources"] = _b.kernel_sources_hash()
    except Exception:  # noqa: BLE001
        pass
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
    counters = {"_commit": commit, "_sources": traffic.get("_sources")}
    for (k, gsz), r in piv.iterrows():
        if "SQ_WAVES" not in r or r["SQ_WAVES"] != r["SQ_WAVES"] or gsz < 4096:
            continue
        w = r["SQ_WAVES"]
        rows = rows_for(k, gsz)
        grid0 = int(gsz)
        if rows != gsz:  # grid-stride kernel: per-row figures, keyed by the rows
            w = rows / 64.0
            gsz = rows
        e = {"valu_per_row": float(r["SQ_INSTS_VALU"] / w), "salu_per_row": float(r["SQ_INSTS_SALU"] / w),
             "wave_cycles_per_wave": float(4 * r["SQ_WAVE_CYCLES"] / w),
             "busy_cycles": float(r["SQ_BUSY_CYCLES"]) if "SQ_BUSY_CYCLES" in r else None}
        # (legs under 50 us are launch-bound: GRBM_GUI_ACTIVE counts the lead-in of a cold dispatch too, and divided by
        #  the kernel's own duration it printed 4 - 5.5 "GHz" in rounds 4 and 5 - no clock / utilisation for them)
        if ("GRBM_GUI_ACTIVE" in r and r["GRBM_GUI_ACTIVE"] == r["GRBM_GUI_ACTIVE"] and (k, grid0) in durs
                and durs[(k, grid0)] >= 50e3):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: shader clock = counter / 8 / kernel duration (of these short
            # PMC passes - the first launches after idle run above the sustained clock)
            gui = float(r["GRBM_GUI_ACTIVE"]) / 8.0
            e["clock_ghz"] = round(gui / durs[(k, grid0)], 3)
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
                        f"{(rd+wr)/rows_for(k, gsz):.1f} B/row ({rows_for(k, gsz)} rows)")
        if "SQ_WAVES" in r and r["SQ_WAVES"] == r["SQ_WAVES"]:
            w = r["SQ_WAVES"]
            bits.append(f"{r['SQ_INSTS_VALU']/w:.0f} VALU + {r['SQ_INSTS_SALU']/w:.0f} SALU instr/wave; "
                        f"wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f}, VALU-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f}, "
                        f"wait-any {4*r['SQ_WAIT_ANY']/w:.0f}, issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f} cycles")
        lines.append(" ".join(bits))
open(os.path.join(dst, "SUMMARY.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
