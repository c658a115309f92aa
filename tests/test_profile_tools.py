"""tools/pmc_traffic.py and tools/hbm_table.py on a synthetic rocprofv3 output tree: the (symbol, grid) grouping, the gfx950
corrections (FETCH_SIZE * 1024 * 2, WRITE_SIZE * 1024), the named-kernel section and the agreement check against a bench line."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def _tree(tmp):
    # two kernels: A (grid 512) twice at 3 ms, B (grid 256) once at 1 ms, A on another grid once at 0.5 ms
    launches = [("A<1>(args)", 512, 0, 3_000_000), ("A<1>(args)", 512, 4_000_000, 7_000_000), ("dw_lds_kernel<5>(args)", 256, 8_000_000, 9_000_000),
                ("A<1>(args)", 128, 9_500_000, 10_000_000)]
    _write(os.path.join(tmp, "trace", "run", "1_kernel_trace.csv"), ["Kernel_Name", "Grid_Size", "Start_Timestamp", "End_Timestamp"],
           [[n, g, s, e] for n, g, s, e in launches])
    def pmc(sub, counter, values):
        _write(os.path.join(tmp, sub, "run", "1_counter_collection.csv"),
               ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"],
               [[i + 1, n, g, counter, v, s, e] for i, ((n, g, s, e), v) in enumerate(zip(launches, values))])
    pmc("pmc_fetch", "FETCH_SIZE", [1000, 1000, 100, 10])   # KB units of 64-B requests: x 1024 x 2
    pmc("pmc_write", "WRITE_SIZE", [500, 500, 50, 5])       # x 1024
    return launches


def test_pmc_traffic_groups_by_symbol_and_grid(tmp_path):
    tmp = str(tmp_path)
    _tree(tmp)
    bench = {"roofline": {"kernel": "net.A [ops 1,2]", "bound": "mfma", "peak": 157.3, "frac": 0.5, "avg_launch_ms": 3.05,
                          "algorithmic_bytes_per_launch": 2.0e6, "algorithmic_flops_per_launch": 2.4e11}}
    bl = os.path.join(tmp, "bench.json")
    open(bl, "w").write(json.dumps(bench) + "\n")
    out = os.path.join(tmp, "out.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), tmp, out, bl, "dw_lds_kernel"], stdout=subprocess.DEVNULL)
    d = json.load(open(out))
    assert d["dispatches_in_kernel_trace"] == 2 and abs(d["avg_duration_us_kernel_trace"] - 3000.0) < 1e-6   # A on grid 512 only
    assert d["hbm_read_bytes_per_launch"] == 1000 * 1024 * 2 and d["hbm_write_bytes_per_launch"] == 500 * 1024
    assert abs(d["traffic_over_algorithmic"] - (1000 * 2048 + 500 * 1024) / 2.0e6) < 1e-9
    assert d["durations_agree_within_10pct"] is True
    assert abs(d["frac_from_kernel_trace"] - 2.4e11 / 3e-3 / 1e12 / 157.3) < 1e-9
    assert [o["launch"] for o in d["other_kernels"]] == ["dw_lds_kernel<5> grid 256"]
    assert d["other_kernels"][0]["hbm_read_bytes_per_launch"] == 100 * 2048 and d["other_kernels"][0]["hbm_write_bytes_per_launch"] == 50 * 1024
    assert [g["launch"] for g in d["next_groups"]] == ["dw_lds_kernel<5> grid 256", "A<1> grid 128"]


def test_hbm_table_lists_every_group(tmp_path):
    tmp = str(tmp_path)
    _tree(tmp)
    txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "hbm_table.py"), tmp, "10"]).decode()
    rows = [l for l in txt.splitlines() if l and not l.startswith("#")]
    assert len(rows) == 3 and rows[0].startswith("A<1>") and " 512 " in rows[0] and "x2" in rows[0]
    assert "read   0.002  write   0.001" in rows[0]          # 2.048 MB read, 0.512 MB written per launch
    assert rows[1].startswith("dw_lds_kernel<5>") and rows[2].startswith("A<1>") and " 128 " in rows[2]
