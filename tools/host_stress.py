#!/usr/bin/env python3
"""Node readiness without a node: the HOST work of N ranks side by side -- no GPU call anywhere.

SURVEY 5 predicted that the weak-scaling risk of this path is host-side: every rank of an 8-GPU job runs one scheduler thread,
1 + chains launch threads, the noise skipper + 4 drawers (63 MB of Exp(1) values per BAIR batch: `helpers/pipeline.py:NoiseFeed`) and
torch's intra-op pool, and no round has had more than one GPU to see eight of them share a host.  This tool starts N processes,
each pinned by `ccvs_amd.tools.affinity.pin_rank` exactly like a bench.py rank (LOCAL_RANK / LOCAL_WORLD_SIZE), and in each:
  * a `NoiseFeed` in its host-only mode is asked for BAIR noise streams ([960, 16, 1024] fp32 = 63 MB) at `--rate` batches per
    second (default 1.0 -- above what one GPU consumes: 229 frames/s = 0.95 batches/s), `--batches` of them;
  * `--launch-threads` dummy launch threads spin the way a hipGraph-replaying worker does between blocking launches (a short
    Python loop + a 0.2 ms sleep per "step": GIL pressure, not CPU saturation);
and reports per rank the Exp(1) MB/s the feed SUSTAINED, the worst lateness of a stream against its deadline, and the slow-down of
8 ranks against 1.  Done-criterion of the verdict: >= 63 MB/s per rank at 8 processes with < 10 % slow-down.

    python tools/host_stress.py --ranks 1 8 [--batches 12] [--rate 1.0] [--burst]
Two passes per rank count: burst (every request queued at once: the feed's CAPACITY per rank, the MB/s and slow-down figures) and
paced (one request per batch period: how long a stream takes from request to "there"); `--burst`: the first pass only.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS, STEPS, WIDTH = 16, 960, 1024          # one BAIR batch: 960 picks x 16 clips x 1024 logits
STREAM_MB = 4.0 * ROWS * STEPS * WIDTH / 2 ** 20


def worker(args):
    sys.path.insert(0, ROOT)
    from ccvs_amd.tools.affinity import pin_rank
    cores = pin_rank(numa=False)            # before torch's pools start (bench.py: before the first GPU call)
    import torch
    from ccvs_amd.helpers.pipeline import NoiseFeed
    stop = threading.Event()

    def launcher():                         # a token worker between graph launches: bookkeeping + a blocking call
        acc = 0
        while not stop.is_set():
            for i in range(2000):
                acc += i & 3
            time.sleep(2e-4)

    threads = [threading.Thread(target=launcher, daemon=True) for _ in range(args.launch_threads)]
    for th in threads:
        th.start()
    gen = torch.Generator().manual_seed(1234 + int(os.environ.get("LOCAL_RANK", 0)))
    feed = NoiseFeed(gen, "cpu")
    t0 = time.perf_counter()
    tickets, waiters = [], []

    def stamp(rec):                        # when the stream is THERE (drawn; on a GPU box: its upload enqueued)
        assert rec["ticket"]["done"].wait(1800)
        rec["t_done"] = time.perf_counter()

    for i in range(args.batches):
        if not args.burst:
            due = t0 + i / args.rate
            while time.perf_counter() < due:
                time.sleep(1e-3)
        rec = {"t_req": time.perf_counter(), "ticket": feed.request(ROWS, STEPS, WIDTH), "t_done": None}
        tickets.append(rec)
        waiters.append(threading.Thread(target=stamp, args=(rec,), daemon=True))
        waiters[-1].start()
    for th in waiters:
        th.join(1800)
    for rec in tickets:
        assert rec["ticket"]["error"] is None, rec["ticket"]["error"]
    lat = [rec["t_done"] - rec["t_req"] for rec in tickets]
    dt = max(rec["t_done"] for rec in tickets) - t0
    feed.close()
    stop.set()
    print(json.dumps({"rank": int(os.environ.get("LOCAL_RANK", 0)), "cores": cores, "seconds": dt, "mb_per_s": args.batches * STREAM_MB / dt,
                      "max_latency_s": max(lat), "mean_latency_s": sum(lat) / len(lat), "parallel_feed": feed.parallel, "drawers": feed.n_drawers,
                      "torch_threads": torch.get_num_threads()}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--rate", type=float, default=1.0)
    ap.add_argument("--burst", action="store_true")
    ap.add_argument("--launch-threads", type=int, default=3)
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    print(f"# host: {os.cpu_count()} hardware threads, {len(os.sched_getaffinity(0))} allowed; one stream = {STREAM_MB:.1f} MB; "
          f"{'burst (capacity)' if args.burst else f'paced at {args.rate} batches/s per rank'}; {args.launch_threads} dummy launch threads per rank; "
          f"load average before: {os.getloadavg()}")
    for burst in ((True, False) if not args.burst else (True,)):
        base = None
        print("## burst: every request queued at once -- the feed's CAPACITY per rank" if burst else
              f"## paced: one request every {1 / args.rate:.2f} s per rank -- is every stream there within a batch period?")
        for n in args.ranks:
            procs = []
            for r in range(n):
                env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), WORLD_SIZE=str(n))
                env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // (2 * n)))))     # bench.py's rule for N > 1
                cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--batches", str(args.batches), "--rate", str(args.rate),
                       "--launch-threads", str(args.launch_threads)] + (["--burst"] if burst else [])
                procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True))
            rows = []
            for p_ in procs:
                out, _ = p_.communicate(timeout=3600)
                rows.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
            worst = min(r["mb_per_s"] for r in rows)
            mean = sum(r["mb_per_s"] for r in rows) / len(rows)
            if base is None:
                base = mean
            pin = f"{len(rows[0]['cores'])} cores per rank" if rows[0]["cores"] else "unpinned (one rank)"
            if burst:
                print(f"{n} rank(s): Exp(1) per rank mean {mean:7.1f} MB/s, slowest rank {worst:7.1f} MB/s ({worst / STREAM_MB:.2f} BAIR batches/s), "
                      f"slow-down of the slowest rank against {args.ranks[0]} rank(s) {100 * (1 - worst / base):5.1f} %   [{pin}; drawers {rows[0]['drawers']}; "
                      f"torch threads {rows[0]['torch_threads']}]")
            else:
                print(f"{n} rank(s): request -> stream drawn: mean {sum(r['mean_latency_s'] for r in rows) / len(rows):5.2f} s, worst {max(r['max_latency_s'] for r in rows):5.2f} s "
                      f"(batch period {1 / args.rate:.2f} s)   [{pin}]")
    print(f"# load average after: {os.getloadavg()}")


if __name__ == "__main__":
    main()
