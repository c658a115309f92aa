"""Per-kernel HBM traffic of one bench step from the rocprofv3 passes of tools/run_profile.sh: for every (kernel symbol, grid)
group of the --kernel-trace run, its dispatch count, average duration, share of the kernel time and - from the separate
--pmc FETCH_SIZE / WRITE_SIZE passes, with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE * 1024 * 2,
WRITE_SIZE * 1024) - the bytes it read from and wrote to HBM per launch and the rate they amount to.

    python tools/hbm_table.py gpurun_out/prof_<tag> [rows] > profiles/<name>.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_traffic as P


def main():
    root = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    groups = {}
    for r in P.trace_rows(root):
        g = groups.setdefault((r[1], r[2]), [0, 0])
        g[0] += r[4]
        g[1] += 1
    fe = P.pmc_rows(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    wr = P.pmc_rows(os.path.join(root, "pmc_write"), "WRITE_SIZE")

    def avg(rs, k):
        v = [r[3] for r in rs if (r[1], r[2]) == k]
        return sum(v) / len(v) if v else 0.0

    total = sum(g[0] for g in groups.values())
    print("# kernel (symbol, grid threads) | dispatches | avg us | share of kernel time | HBM read GB | HBM write GB | TB/s   (per launch)")
    for k, (dur, n) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:rows]:
        rd, wb, us = avg(fe, k) * 2048, avg(wr, k) * 1024, dur / n / 1e3
        print("%-92s %9d  x%-3d %8.1f us  %5.2f%%  read %7.3f  write %7.3f  %5.2f TB/s" %
              (k[0].split("(")[0][:92], k[1], n, us, 100.0 * dur / total, rd / 1e9, wb / 1e9, (rd + wb) / us / 1e6 if us else 0.0))


if __name__ == "__main__":
    main()
