"""Samples the GPU's shader clock, memory clock and socket power (rocm-smi's sysfs sources) every 50 ms while a command runs.
usage: python tools/clock_watch.py <out.txt> -- <command ...>     (the command runs as a child; its output passes through)
The point: the matrix-heavy kernels of the step run under the board's power limit, so the MFMA "peak" that bounds them is the
peak at the clock the limit allows, not at the 2.4 GHz of the data sheet."""
import glob, os, subprocess, sys, threading, time

out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]


def first(pattern):
    g = sorted(glob.glob(pattern))
    return g[0] if g else None


dev = None
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    if os.path.exists(d + "/pp_dpm_sclk"):
        dev = d
        break
hw = first(dev + "/hwmon/hwmon*") if dev else None


def cur(path):
    try:
        for ln in open(path):
            if ln.rstrip().endswith("*"):
                return ln.split(":")[1].strip().rstrip("*").strip()
    except OSError:
        pass
    return "?"


def num(path):
    try:
        return float(open(path).read())
    except (OSError, ValueError):
        return float("nan")


rows, stop = [], threading.Event()


def watch():
    t0 = time.time()
    while not stop.is_set():
        p = num(hw + "/power1_average") if hw else float("nan")
        if p != p and hw:
            p = num(hw + "/power1_input")
        f = num(hw + "/freq1_input") if hw else float("nan")
        rows.append((time.time() - t0, cur(dev + "/pp_dpm_sclk") if dev else "?", cur(dev + "/pp_dpm_mclk") if dev else "?", p / 1e6, f / 1e6))
        time.sleep(0.05)


th = threading.Thread(target=watch, daemon=True)
th.start()
rc = subprocess.call(cmd)
stop.set(); th.join()
with open(out, "w") as o:
    print("device %s hwmon %s; %d samples" % (dev, hw, len(rows)), file=o)
    cap = num(hw + "/power1_cap") / 1e6 if hw else float("nan")
    print("power cap %.0f W" % cap, file=o)
    print("%8s %10s %10s %8s %8s" % ("t s", "sclk", "mclk", "W", "freq1 MHz"), file=o)
    for r in rows[::4]:
        print("%8.2f %10s %10s %8.1f %8.0f" % r, file=o)
sys.exit(rc)
