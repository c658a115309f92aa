"""HBM traffic of the DOMINANT kernel from the separate rocprofv3 --pmc passes (tools/run_profile.sh), with the gfx950
corrections of MI355X_MICROARCH.md (section HBM): bytes_read = FETCH_SIZE * 1024 * 2 (FETCH_SIZE tallies the 128-B
requests of a wide coalesced stream at 64 B), bytes_written = WRITE_SIZE * 1024.

Dominant = the (kernel symbol, grid) group with the largest total duration in the plain --kernel-trace run - the top row
of a rocprofv3 kernel summary once launches of one symbol on different grids are kept apart (several network ops can be
the same instantiation on the same shape: rec ops 13/15/17/19 are four launches of one dwpw_kernel<5,1,1,16,...> grid).
bench.py picks its `roofline` kernel by the same rule from its HIP-event survey; when its JSON line is given (3rd
argument) the group's label is taken from there and the two average durations are compared.

    python tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/<name>.json [bench_line.json [symbol-substring,...]]

A 4th argument names further kernels by substrings of their symbols (e.g. "dw_lds_kernel,dw_conv_kernel"): for each (symbol,
grid) group that matches, the same per-launch figures go into `other_kernels` (dispatch counts, average duration, HBM bytes
read and written per launch) - how the depthwise kernels' halo traffic is checked against their tensors' sizes.
"""
import csv
import glob
import json
import os
import sys


def pmc_rows(d, counter):
    out = []
    for fn in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"]),
                            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return sorted(out)


def trace_rows(root):
    tr = []
    for fn in glob.glob(os.path.join(root, "trace", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(fn)):
            grid = int(r.get("Grid_Size") or 0) or int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
            tr.append((int(r["Start_Timestamp"]), r["Kernel_Name"], grid, 0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return sorted(tr)


def main():
    root, dst = sys.argv[1], sys.argv[2]
    bench = None
    if len(sys.argv) > 3:
        bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["roofline"]
    tr = trace_rows(root)
    groups = {}
    for r in tr:
        g = groups.setdefault((r[1], r[2]), [0, 0])
        g[0] += r[4]
        g[1] += 1
    total = sum(g[0] for g in groups.values())
    ranked = sorted(groups.items(), key=lambda kv: -kv[1][0])
    (sym, grid), (dur, n) = ranked[0]
    sel = lambda rs: [r for r in rs if r[1] == sym and r[2] == grid]
    fe = sel(pmc_rows(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"))
    wr = sel(pmc_rows(os.path.join(root, "pmc_write"), "WRITE_SIZE"))
    mf = sel(pmc_rows(os.path.join(root, "pmc_sq"), "SQ_VALU_MFMA_BUSY_CYCLES"))
    gui = sel(pmc_rows(os.path.join(root, "pmc_sq"), "GRBM_GUI_ACTIVE"))
    rd = sum(r[3] for r in fe) / len(fe) * 1024 * 2
    wb = sum(r[3] for r in wr) / len(wr) * 1024
    out = {
        "kernel": bench["kernel"] if bench else "%s grid %d" % (sym, grid),
        "launch": "%s, grid %d threads (%d workgroups of 256)" % (sym.split("(")[0], grid, grid // 256),
        "kernel_group": [sym.split("(")[0], grid],
        "share_of_kernel_time_in_trace": dur / total,
        "next_groups": [{"launch": "%s grid %d" % (k[0].split("(")[0], k[1]), "share": v[0] / total, "dispatches": v[1]} for k, v in ranked[1:4]],
        "dispatches_in_kernel_trace": n,
        "avg_duration_us_kernel_trace": dur / n / 1e3,
        "dispatches_averaged": len(fe),
        "hbm_read_bytes_per_launch": rd,
        "hbm_write_bytes_per_launch": wb,
        "traffic_bytes_per_launch": rd + wb,
        "mfma_busy_fraction": (sum(r[3] for r in mf) / (sum(r[3] for r in gui) / 8.0 * 1024.0)) if mf and gui else None,
        "avg_duration_us_under_pmc": sum(r[4] for r in fe) / len(fe) / 1e3,
        "corrections": "FETCH_SIZE*1024*2, WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section); separate --pmc passes",
    }
    issue = {}
    for cname in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
        rs = sel(pmc_rows(os.path.join(root, "pmc_issue"), cname))
        if rs:
            issue[cname] = sum(r[3] for r in rs) / len(rs)
    if issue:
        out["issue_counters_per_launch"] = issue
        if issue.get("SQ_INSTS_MFMA"):
            out["valu_instructions_per_mfma"] = issue.get("SQ_INSTS_VALU", 0.0) / issue["SQ_INSTS_MFMA"]
            # SQ_INSTS_VALU counts the matrix instructions too (checked against the disassembly: DESIGN.md section 6)
            out["non_mfma_valu_instructions_per_mfma"] = out["valu_instructions_per_mfma"] - 1.0
            out["lds_instructions_per_mfma"] = issue.get("SQ_INSTS_LDS", 0.0) / issue["SQ_INSTS_MFMA"]
    if bench:
        alg = bench["algorithmic_bytes_per_launch"]
        out["algorithmic_bytes_per_launch"] = alg
        out["traffic_over_algorithmic"] = (rd + wb) / alg
        out["bench_avg_launch_ms"] = bench["avg_launch_ms"]
        out["durations_agree_within_10pct"] = abs(bench["avg_launch_ms"] * 1e3 - out["avg_duration_us_kernel_trace"]) <= 0.10 * out["avg_duration_us_kernel_trace"]
        if not out["durations_agree_within_10pct"]:
            # the trace's top (symbol, grid) group is not provably the bench's roofline kernel: do not lend it that name
            # (bench.py attributes `traffic` by name and skips entries whose durations disagree)
            out["kernel"] = "%s grid %d (NOT matched to the bench line's kernel %r: average durations differ by more than 10 %%)" % (sym.split("(")[0], grid, bench["kernel"])
        out["bench_frac"] = bench["frac"]
        if bench["bound"] == "mfma":
            out["frac_from_kernel_trace"] = bench["algorithmic_flops_per_launch"] / (out["avg_duration_us_kernel_trace"] * 1e-6) / 1e12 / bench["peak"]
        else:
            out["frac_from_kernel_trace"] = alg / (out["avg_duration_us_kernel_trace"] * 1e-6) / 1e9 / bench["peak"]
    if len(sys.argv) > 4:
        others = []
        fe_all, wr_all = pmc_rows(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"), pmc_rows(os.path.join(root, "pmc_write"), "WRITE_SIZE")
        for sub in sys.argv[4].split(","):
            for (ksym, kgrid), (kdur, kn) in ranked:
                if sub not in ksym:
                    continue
                f_ = [r for r in fe_all if r[1] == ksym and r[2] == kgrid]
                w_ = [r for r in wr_all if r[1] == ksym and r[2] == kgrid]
                if not f_ or not w_:
                    continue
                others.append({"launch": "%s grid %d" % (ksym.split("(")[0], kgrid), "dispatches_in_kernel_trace": kn,
                               "share_of_kernel_time_in_trace": kdur / total, "avg_duration_us_kernel_trace": kdur / kn / 1e3,
                               "hbm_read_bytes_per_launch": sum(r[3] for r in f_) / len(f_) * 1024 * 2,
                               "hbm_write_bytes_per_launch": sum(r[3] for r in w_) / len(w_) * 1024})
        out["other_kernels"] = others
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
