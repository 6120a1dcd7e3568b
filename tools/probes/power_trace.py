"""Sample the GPU's socket power and shader clock (sysfs hwmon / pp_dpm_sclk, ~100 Hz) while a command runs, and summarise the
samples taken while the GPU was busy (power above the idle level + 40 %).  Question it answers: is the transition running against
the POWER cap - i.e. does a kernel that keeps the matrix pipe busier just lower the clock for everything else?
Usage: python tools/probes/power_trace.py <tag> -- <command ...>     (writes gpurun_out/<tag>_power.json)"""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def find(paths):
    for pat in paths:
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[0]
    return None


def read_num(path):
    try:
        with open(path) as fh:
            return float(fh.read().split()[0])
    except Exception:
        return None


def main():
    tag = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    p_path = find(["/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"])
    f_path = find(["/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"])
    cap_path = find(["/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap"])
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), read_num(p_path) if p_path else None, read_num(f_path) if f_path else None))
            time.sleep(0.01)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    rc = subprocess.call(cmd)
    stop.set()
    th.join()
    pw = [s[1] / 1e6 for s in samples if s[1] is not None]            # microwatts -> W
    fq = [s[2] / 1e6 for s in samples if s[2] is not None]            # Hz -> MHz
    out = {"tag": tag, "rc": rc, "seconds": time.perf_counter() - t0, "samples": len(samples), "power_path": p_path, "freq_path": f_path,
           "power_cap_W": (read_num(cap_path) or 0) / 1e6 if cap_path else None}
    if pw:
        idle = statistics.quantiles(pw, n=20)[0]
        busy = [(p, f) for (p, f) in zip(pw, fq or [0] * len(pw)) if p > idle * 1.4]
        out.update({"power_W_idle_p5": idle, "power_W_max": max(pw), "busy_samples": len(busy)})
        if busy:
            bp, bf = [b[0] for b in busy], [b[1] for b in busy]
            out.update({"busy_power_W_mean": statistics.mean(bp), "busy_power_W_p50": statistics.median(bp), "busy_power_W_p95": statistics.quantiles(bp, n=20)[-1],
                        "busy_sclk_MHz_mean": statistics.mean(bf), "busy_sclk_MHz_p50": statistics.median(bf), "busy_sclk_MHz_p5": statistics.quantiles(bf, n=20)[0],
                        "busy_sclk_MHz_max": max(bf)})
        # the last 30 % of the run is the timed loop of bench.py (weights, recording and warm-up come first)
        tail = samples[int(len(samples) * 0.7):]
        tp = [s[1] / 1e6 for s in tail if s[1] is not None]
        tf = [s[2] / 1e6 for s in tail if s[2] is not None]
        if tp:
            out.update({"tail_power_W_mean": statistics.mean(tp), "tail_power_W_p50": statistics.median(tp)})
        if tf:
            out.update({"tail_sclk_MHz_mean": statistics.mean(tf), "tail_sclk_MHz_p50": statistics.median(tf)})
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/{tag}_power.json", "w"), indent=1)
    print(json.dumps(out))
    return rc


if __name__ == "__main__":
    sys.exit(main())
