"""Summarises rocprofv3 CSV output (kernel trace + separate --pmc passes) per kernel *instance*
(kernel name x grid size), applying the gfx950 corrections of MI355X_MICROARCH.md §HBM:
FETCH_SIZE is reported in KiB-units of 64-B requests that tally 128-B coalesced requests at 64 B ->
bytes = FETCH_SIZE*1024*2 for wide streaming reads; WRITE_SIZE bytes = WRITE_SIZE*1024.

    python tools/prof_summary.py <prof_dir> [top_n]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load_counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for fn in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def load_trace(d):
    out = defaultdict(list)
    for fn in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(fn)):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            out[(r["Kernel_Name"], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    trace = load_trace(os.path.join(root, "trace"))
    sq = load_counters(os.path.join(root, "pmc_sq"))
    fe = load_counters(os.path.join(root, "pmc_fetch"))
    wr = load_counters(os.path.join(root, "pmc_write"))
    rows = []
    for key, durs in trace.items():
        tot = sum(durs)
        rows.append((tot, key, durs))
    rows.sort(reverse=True)
    allt = sum(r[0] for r in rows)
    print("%-52s %9s %6s %9s %8s %9s %9s %7s" % ("kernel instance (name, grid)", "total_ms", "calls", "avg_us", "share",
                                                 "rdMB/call", "wrMB/call", "mfma%"))
    for tot, key, durs in rows[:top]:
        name = key[0].replace("void ocr::", "").replace("ocr::", "")
        name = name[:name.index("(")] if "(" in name else name
        avg = tot / len(durs)
        f = fe.get(key, {}).get("FETCH_SIZE")
        w = wr.get(key, {}).get("WRITE_SIZE")
        rd = (sum(f) / len(f)) * 1024 * 2 / 1e6 if f else float("nan")
        wb = (sum(w) / len(w)) * 1024 / 1e6 if w else float("nan")
        c = sq.get(key, {})
        mf = float("nan")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            # busy cycles summed over the chip's 1024 SIMDs / (active cycles x 1024/8 XCD-summed)
            mf = 100.0 * sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(c["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0)
        print("%-52s %9.3f %6d %9.1f %7.2f%% %9.1f %9.1f %7.1f" % ("%s g=%d" % (name, key[1]), tot / 1e6, len(durs), avg / 1e3,
                                                                  100.0 * tot / allt, rd, wb, mf))
    print("total kernel time %.3f ms over %d dispatches" % (allt / 1e6, sum(len(r[2]) for r in rows)))


if __name__ == "__main__":
    main()
