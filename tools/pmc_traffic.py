"""HBM traffic of the dominant kernel from the separate rocprofv3 --pmc passes (tools/run_profile.sh),
with the gfx950 corrections of MI355X_MICROARCH.md §HBM: bytes_read = FETCH_SIZE * 1024 * 2 (FETCH_SIZE
tallies the 128-B requests of a wide coalesced stream at 64 B), bytes_written = WRITE_SIZE * 1024.

The dominant kernel of bench.py is rec op 30 (conv1x1 480->480 at H/8 = 6 rows, reading the SE-gated
depthwise output): conv_mfma_kernel<3, ...> on the 1872-line launch (all width-320 lines of the 64 images)
has grid 35100 workgroups x 256 threads (M = 1872*6*80 rows / 128 x 5 column groups).  rec ops 25, 32 and
34 have the same grid; see sel() for which dispatches are op 30.

    python tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/<name>.json
"""
import csv
import glob
import json
import os
import sys


def rows(d, counter):
    out = []
    for fn in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"]),
                            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return sorted(out)


def main():
    root, dst = sys.argv[1], sys.argv[2]
    # round 3: the recognizer's lines run as ONE ragged launch (every tensor width together), so the row count of op 30
    # comes from the bench line of the same code (roofline.algorithmic_flops_per_launch = 2 * M * 480 * 480) instead
    # of being "1872 lines x 6 x 80"
    LINES, M_ROWS, NAME = 1872, 1872 * 6 * 80, None
    if len(sys.argv) > 3:
        rl = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["roofline"]
        M_ROWS = int(round(rl["algorithmic_flops_per_launch"] / (2.0 * 480 * 480)))
        NAME = rl["kernel"]
        LINES = int(NAME.split("@")[1].split("x")[0])
    grid = ((M_ROWS + 127) // 128) * 5 * 256
    # with the SE gate folded into its consumers (net.hip) ops 25 and 30 run the GATE instantiation
    # conv_mfma_kernel<3, 0, true, true> on this grid, in that order: op 30 is every second dispatch of it;
    # with OCR_FUSE_GATE=0 ops 25, 30, 32, 34 share one kernel and op 30 is every fourth starting at the second
    # round 3 (second half): the big 1x1 convs run conv_mfma_mt_kernel<3, 2, GATE> (two pixel tiles per wave): ops 25 and 30
    # are its gated instantiation on a grid of ceil(M / 256) x 5 workgroups, in that order
    grid2 = ((M_ROWS + 255) // 256) * 5 * 256
    def sel(rs):
        mt = [r for r in rs if "conv_mfma_mt_kernel<3, 2, true>" in r[1] and r[2] == grid2]
        if mt:
            return mt[1::2]
        gated = [r for r in rs if "conv_mfma_kernel<3, 0, true, true>" in r[1] and r[2] == grid]
        if gated:
            return gated[1::2]
        return [r for r in rs if "conv_mfma_kernel<3" in r[1] and r[2] == grid][1::4]
    fe = sel(rows(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"))
    wr = sel(rows(os.path.join(root, "pmc_write"), "WRITE_SIZE"))
    mf = sel(rows(os.path.join(root, "pmc_sq"), "SQ_VALU_MFMA_BUSY_CYCLES"))
    gui = sel(rows(os.path.join(root, "pmc_sq"), "GRBM_GUI_ACTIVE"))
    rd = sum(r[3] for r in fe) / len(fe) * 1024 * 2
    wb = sum(r[3] for r in wr) / len(wr) * 1024
    M, K, N = M_ROWS, 480, 480
    alg = 4.0 * (M * K + M * N + K * N)
    out = {
        "kernel": NAME or "rec.30.conv1x1_480_480%s@%dx48x320" % ("_gated" if any("true>" in r[1] for r in fe) else "", LINES),
        "launch": "%s, grid %dx256 (the %d-line launch)" % (fe[0][1].split("(")[0], fe[0][2] // 256, LINES),
        "dispatches_averaged": len(fe),
        "hbm_read_bytes_per_launch": rd,
        "hbm_write_bytes_per_launch": wb,
        "traffic_bytes_per_launch": rd + wb,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (rd + wb) / alg,
        "mfma_busy_fraction": sum(r[3] for r in mf) / (sum(r[3] for r in gui) / 8.0 * 1024.0),
        "avg_duration_us_under_pmc": sum(r[4] for r in fe) / len(fe) / 1e3,
        "corrections": "FETCH_SIZE*1024*2, WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section); separate --pmc passes",
    }
    # the same dispatches in the plain --kernel-trace --stats run (no counters): the duration bench.py's events must agree with
    tr = []
    for fn in glob.glob(os.path.join(root, "trace", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(fn)):
            tr.append((int(r["Start_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size") or r["Grid_Size_X"]), 0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    tr = sel(sorted(tr))
    if tr:
        out["avg_duration_us_kernel_trace"] = sum(r[4] for r in tr) / len(tr) / 1e3
        out["dispatches_in_kernel_trace"] = len(tr)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
