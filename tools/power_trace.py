#!/usr/bin/env python3
"""Socket power and clocks while a command runs: is the run power-managed?
    python tools/power_trace.py OUT.json -- python bench.py --gpus 1 --steps 20 --warmup 5
Starts the command as a CHILD process (this process never touches the GPU: rocm-smi is a process of its own) and samples
`rocm-smi -P -c -t --showmaxpower --json` about twice a second until the child ends.  Writes the samples and a summary (percentiles of
power and of the shader clock over the samples above half the power cap) to OUT.json; the child's stdout / stderr pass through."""
import json
import subprocess
import sys
import threading
import time


SEEN = []


def parse(card):
    """One card's record of `rocm-smi -P -c -t --showmaxpower --json` -> {power_w, cap_w, sclk_mhz, mclk_mhz, tj_c} (what is there)."""
    rec = {}
    for k, v in card.items():
        kl = k.lower()
        if "power" in kl and "max" in kl:
            rec["cap_w"] = float(v)
        elif "power" in kl:
            rec["power_w"] = float(v)
        elif "sclk" in kl and "level" not in kl or kl.startswith("sclk clock speed"):
            rec["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
        elif "mclk" in kl and "level" not in kl or kl.startswith("mclk clock speed"):
            rec["mclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
        elif "temperature" in kl and "junction" in kl:
            rec["tj_c"] = float(v)
    return rec


def sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-P", "-c", "-t", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
    except Exception as e:      # a failed sample is a hole in the trace, not an error of the run
        return {"error": str(e)}
    rec = parse(card)
    if not SEEN:
        SEEN.append(1)
        rec["raw"] = card      # the first sample keeps rocm-smi's own record (key names differ between releases)
    return rec


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    samples, stop = [], threading.Event()
    t0 = time.time()

    def loop():
        while not stop.is_set():
            r = sample()
            r["t"] = round(time.time() - t0, 2)
            samples.append(r)
            stop.wait(0.25)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=15)
    good = [s for s in samples if "power_w" in s]
    cap = max([s.get("cap_w", 0) for s in good] + [0])
    busy = [s for s in good if cap and s["power_w"] > 0.5 * cap] or good
    summ = {"samples": len(samples), "busy_samples": len(busy), "cap_w": cap,
            "power_w": {q: pct([s["power_w"] for s in busy], p) for q, p in (("p10", .1), ("p50", .5), ("p90", .9))},
            "sclk_mhz": {q: pct([s["sclk_mhz"] for s in busy if "sclk_mhz" in s], p) for q, p in (("p10", .1), ("p50", .5), ("p90", .9))},
            "tj_c_max": max([s.get("tj_c", 0) for s in good] + [0])}
    json.dump({"cmd": cmd, "rc": rc, "summary": summ, "samples": samples}, open(out_path, "w"))
    print("power_trace:", json.dumps(summ), file=sys.stderr)
    sys.exit(rc)


if __name__ == "__main__":
    main()
