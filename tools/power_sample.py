"""Development tool: runs a command while sampling the GPU's package power and shader clock from the amdgpu hwmon files
(readable without privileges on the GPU box), prints the command's output and the samples' statistics over the busy part.
    python tools/power_sample.py [--ms 20] -- <command ...>
"""
import glob
import subprocess
import sys
import threading
import time


def find():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        try:
            open(d + "/power1_input").read()
            return d
        except OSError:
            continue
    return None


def main():
    args = sys.argv[1:]
    period = 0.02
    if args and args[0] == "--ms":
        period = float(args[1]) / 1000.0
        args = args[2:]
    if args and args[0] == "--":
        args = args[1:]
    d = find()
    samples = []
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            try:
                p = int(open(d + "/power1_input").read()) / 1e6
                f = int(open(d + "/freq1_input").read()) / 1e6
                samples.append((time.time(), p, f))
            except (OSError, ValueError):
                pass
            time.sleep(period)

    if d:
        threading.Thread(target=loop, daemon=True).start()
    t0 = time.time()
    rc = subprocess.call(args)
    stop.set()
    busy = [s for s in samples if s[1] > 0.5 * max(x[1] for x in samples)] if samples else []
    if busy:
        ps = sorted(s[1] for s in busy)
        fs = sorted(s[2] for s in busy)
        print("power_sample: %d samples (%d busy) over %.1f s: power W median %.0f max %.0f | sclk MHz median %.0f min %.0f max %.0f | cap %s W"
              % (len(samples), len(busy), time.time() - t0, ps[len(ps) // 2], ps[-1], fs[len(fs) // 2], fs[0], fs[-1],
                 open(d + "/power1_cap").read().strip()[:-6]))
    else:
        print("power_sample: no hwmon samples")
    return rc


if __name__ == "__main__":
    sys.exit(main())
