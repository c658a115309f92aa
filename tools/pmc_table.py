"""Per-dispatch counter table from tools/pmc_net.sh output: python tools/pmc_table.py gpurun_out/pmc_<tag> [name-filter]
Dispatches of the LAST forward only (net_bench runs warm-up + 1), in launch order."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    disp = defaultdict(dict)
    meta = {}
    for fn in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")) + glob.glob(os.path.join(root, "*", "*", "*", "*_counter_collection.csv")):
        ps = os.path.relpath(fn, root).split(os.sep)[0]
        for r in csv.DictReader(open(fn)):
            key = (ps, int(r["Dispatch_Id"]))
            disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta[key] = (r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    passes = sorted({k[0] for k in disp})
    for ps in passes:
        ids = sorted(k[1] for k in disp if k[0] == ps)
        half = ids[len(ids) // 2:]          # last forward
        names = sorted({c for k in disp if k[0] == ps for c in disp[k]})
        print("== pass", ps)
        print("%4s %-40s %10s %8s " % ("#", "kernel", "grid", "us") + " ".join("%14s" % n[-14:] for n in names))
        for j, i in enumerate(half):
            m = meta[(ps, i)]
            if flt and flt not in m[0]:
                continue
            print("%4d %-40s %10d %8.1f " % (j, m[0], m[1], m[2] / 1e3) + " ".join("%14.4g" % disp[(ps, i)].get(n, 0) for n in names))


if __name__ == "__main__":
    main()
