#!/usr/bin/env python3
"""Benchmark of the CCVS synthesis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the synthesis path over one batch per GPU (BAIR 256x256, 1
conditioning frame -> 15 synthesized frames, batch 16 per GPU: BASELINE.json configs[1]):
encode all frames, crop, token synthesis, flow-guided decode, uint8 pack, RCCL all-gather of
the clips -- frames already resident in HBM when the clock starts.  K steps = K batches, timed
from the first encode to the last gathered clip.  Two schedules of exactly that work:
  pipelined (default)  two batches in flight per GPU: the token loop of batch i+1 on a
                       high-priority stream beside the encoder / decoder of batch i
                       (Generator.run_pipelined; bit-identical clips); fill and drain are inside
                       the timed region;
  serial               Generator.generate_vid batch after batch.
Weights are random-init (the reference's initialisers), frames are seeded synthetic tensors.
Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement" for every field.
`python bench.py --gpus N` without a torch.distributed environment starts the N ranks itself.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

NOISE_SEED = 20261004    # host-drawn sampling noise: the process generator is re-seeded in front of every pass, like a fresh reference process


def noise_seed_of_rank(rank):
    """Seed of rank r's process generator (host-drawn sampling noise).  The reference never seeds (tools/engine.py): its N processes
    draw N unrelated streams.  Until round 6 every rank was seeded with NOISE_SEED itself, so rank r's clip i sampled with the noise
    of rank 0's clip i -- harmless to frames/s, wrong as a generation job.  `--sample-noise device` is the
    world-size-invariant sampler (Philox keyed by the GLOBAL clip index); this one is per process, like the reference's."""
    return NOISE_SEED + int(rank)


FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20,
                    help="timed batches (default 20: ~30 s; the fill and drain of the pipelined schedule are inside the timed region and weigh 1/K)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU (weak scaling)")
    ap.add_argument("--config", type=str, default="bair", choices=["bair", "kinetics", "bair-p2p", "drums"],
                    help="bair = BASELINE.json configs[1] (the metric); kinetics / bair-p2p / drums = configs[2] / [3] / [4] at their real geometry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sample-noise", type=str, default=None, choices=["device", "host"],
                    help="host (default for bair / bair-p2p): the Exp(1) stream torch.multinomial draws from the process generator under the reference's "
                         "seed, pre-drawn per batch on a noise thread and read inside the captured decode step -- the sampler every token-for-token "
                         "oracle test pins; device (default for kinetics / drums): in-kernel Philox keyed by the global clip index (bair: reported "
                         "beside the headline as `sampling_device_noise`).  Kinetics draws 704 x 64 x 16384 values = 3 GB of host noise per batch "
                         "(~10 s of one CPU thread, what the reference's CPU path spends too); Drums' ancillary picks are not one plain stream")
    ap.add_argument("--no-other-noise-leg", action="store_true", help="skip the second timed pass (same K batches) with the other noise source")
    ap.add_argument("--conv-precision", type=str, default=None, choices=["bf16x3", "f32"])
    ap.add_argument("--schedule", type=str, default=os.environ.get("CCVS_BENCH_SCHEDULE", "pipelined"), choices=["serial", "stream", "pipelined"],
                    help="serial: batches one after the other, token loop then decode (the single-call drop-in, Generator.generate_vid); stream: batches one "
                         "after the other, the decoder of a batch following ITS OWN token loop frame by frame (generate_vid(schedule='stream')); pipelined: "
                         "several batches in flight (token loops of batches i+1.. beside the decoder of batch i)")
    ap.add_argument("--cu-limit", type=int, default=None, help="pipelined: CUs the decoder stream may occupy while token loops are in flight")
    ap.add_argument("--lanes", type=int, default=None, help="pipelined: batches whose token loops run as ONE loop over their stacked rows (a token group)")
    ap.add_argument("--chains", type=int, default=None, help="pipelined: token groups that run beside each other (one stream each)")
    ap.add_argument("--ramp", type=str, default=None, help="pipelined: sizes of the first token groups, e.g. 1,2 (then --lanes)")
    ap.add_argument("--no-strict-f32", action="store_true", help="skip the exact-fp32-convolution leg reported as strict_f32 (the same K batches on the same schedule)")
    ap.add_argument("--rec-pass", action="store_true", help="also run the reference's teacher-forced reconstruction decode (not counted)")
    ap.add_argument("--encode", choices=["cond", "all"], default="all",
                    help="frames of the input clip the encoder sees: all of them, as the reference's generate_vid does (default; its rec pass reads those "
                         "codes), or only the conditioning frames synthesis reads (the default line carries that figure as `encode_cond_only`)")
    ap.add_argument("--no-encode-cond-leg", action="store_true", help="skip the second timed pass (same K batches) with only the conditioning frames encoded")
    return ap.parse_args()


def _kill_tree(proc, grace=10.0):
    """Stop `proc` and EVERY descendant.  torch.distributed.run starts each rank in a session of its own, so killing the agent's
    process group leaves hung ranks behind -- holding the GPUs and the write end of our stdout pipe.  SIGTERM goes to the
    direct child first (torchrun forwards it and reaps its workers); whatever of the tree is still alive after the grace
    period is SIGKILLed (the descendants are listed BEFORE anything is signalled: orphans cannot be found afterwards)."""
    import signal
    import psutil
    try:
        tree = psutil.Process(proc.pid).children(recursive=True)
    except psutil.Error:
        tree = []
    try:
        proc.send_signal(signal.SIGTERM)
    except OSError:
        pass
    deadline = time.time() + grace
    while time.time() < deadline and (proc.poll() is None or any(p_.is_running() for p_ in tree)):
        time.sleep(0.1)
    for p_ in tree:
        try:
            p_.kill()
        except psutil.Error:
            pass
    if proc.poll() is None:
        proc.kill()


def supervise(args, cmd, env):
    """Run the measuring process(es) `cmd` as a child of this one -- which never touches the GPU -- under a time limit and relay
    the JSON line.  The schedule itself cannot hang silently any more (every wait in Generator.run_pipelined has a time limit
    and raises); this limit is the harness's own backstop against anything else (a box whose host stalls, a dead RCCL rank).
    A child that runs into it is stopped with all its descendants and the SAME measurement is started once more, without the
    host-side cpu_baseline leg (the one part whose duration the GPU does not decide); the line then says so
    (`supervisor.attempts`, `supervisor.note`).  What is measured never changes between attempts."""
    import subprocess
    limit = float(os.environ.get("CCVS_BENCH_TIME_LIMIT", 600 + 12 * (args.steps + args.warmup)))
    notes = []
    for attempt in range(1, 3):
        extra = ["--no-cpu-baseline"] if attempt >= 2 and not args.no_cpu_baseline else []
        proc = subprocess.Popen(cmd + extra, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            _kill_tree(proc, grace=float(os.environ.get("CCVS_BENCH_KILL_GRACE", "10")))
            try:
                proc.communicate(timeout=10)
            except subprocess.TimeoutExpired:   # an orphan still holds the pipe: stop reading
                proc.stdout.close()
            notes.append(f"attempt {attempt} stopped after {limit:.0f} s without a result")
            print(f"bench.py: {notes[-1]}", file=sys.stderr)
            continue
        lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if not lines:
            sys.stderr.write(out[-4000:])
            return proc.returncode or 1
        line = lines[-1]
        if notes:
            rec = json.loads(line)
            rec["supervisor"] = {"attempts": attempt, "note": "; ".join(notes) + ("; this line was measured with " + " ".join(extra) if extra else "")}
            line = json.dumps(rec)
        print(line)
        return proc.returncode
    return 3


def launch_ranks(args):
    """`python bench.py --gpus N` invoked bare (no torch.distributed environment): start the rank process(es) as children of
    this process, which has not touched the GPU (torch.cuda.device_count() does not initialise it), under `supervise`.
    N > 1: N fresh ranks through torch.distributed.run, as the reference's launch line does
    (scripts/bairhd/save_videos_p2p.sh:6); N = 1: this script again as the one rank."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this host driver
    if args.gpus > 1:   # N ranks share the host: a few intra-op threads each (the 1 + chains launch threads per rank are extra)
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // (2 * args.gpus)))))
    env["CCVS_BENCH_CHILD"] = "1"
    if args.gpus == 1 and os.environ.get("CCVS_BENCH_FORCE_LAUNCHER") != "1":
        return supervise(args, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return supervise(args, cmd, env)


def build_generator(args):
    from ccvs_amd.tools.options import Options, BAIR_ARGV, KINETICS_ARGV, BAIR_P2P_ARGV, DRUMS_ARGV
    from ccvs_amd.helpers.generator import Generator
    argv = list({"bair": BAIR_ARGV, "kinetics": KINETICS_ARGV, "bair-p2p": BAIR_P2P_ARGV, "drums": DRUMS_ARGV}[args.config])
    # the teacher-forced "rec" decode of the REAL codes is not part of the metric (synthesized frames only)
    argv += ["--batch_size_vid", str(args.batch), "--x_sample_noise", args.sample_noise, "--rec_pass", "true" if args.rec_pass else "false",
             "--encode_all", "true" if args.encode == "all" else "false"]
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, load_stft_ae=(args.config == "drums"), argv=argv)
    torch.manual_seed(0)  # reference initialisers under seed 0
    with contextlib.redirect_stdout(sys.stderr):   # "Loading untrained ... net": stdout carries the ONE JSON line only
        gen = Generator(opt).build_models()
    return gen, opt


@torch.no_grad()
def calibrate_codebook(gen, data):
    """Documented synthetic codebook randn * std(z_e) (SURVEY section 7 hard part 2) and
    non-zero positional tables, so that argmin / attention see realistic statistics."""
    z_e, _ = gen.vid_model.net_e(data["vid"][:2, :1])
    cb = gen.vid_model.net_q.embedding.weight
    g = torch.Generator().manual_seed(4)
    cb.copy_((torch.randn(cb.shape, generator=g) * float(z_e.std())).to(cb.device))
    t = gen.transformer_model.net_t
    g = torch.Generator().manual_seed(3)
    t.s_emb.copy_((torch.randn(t.s_emb.shape, generator=g) * 0.02).to(cb.device))
    t.t_emb.copy_((torch.randn(t.t_emb.shape, generator=g) * 0.02).to(cb.device))


def cpu_baseline(gen, opt):
    """The oracle (CPU restatement of the reference algorithm, incl. its no-KV-cache token loop) timed on this host on a
    BOUNDED sample of config 1 (BAIR, batch 1) and extrapolated to one clip: encoder on 1 frame, decoder on 1 frame with k = 1
    and k = 3 contexts, GPT forward at five sequence lengths, integrated over the 960-token loop (piecewise-linear between the
    sampled lengths: on these hosts the time is not a clean quadratic in the length) and the 15-frame decode loop.
    The GPU box's host is shared, so the sample defends itself: the thread count is chosen by a short sweep (8 / 16 / 32 / 64
    on one mid-size GPT forward, stopped as soon as more threads get slower -- all 256 hardware threads took 81 s for a
    forward that takes 0.24 s on 16), every quantity is the MEDIAN of three calls, the load average is reported, and the figure
    is marked `unreliable` when the calls of any one quantity spread by more than 10 % around their median."""
    import statistics
    import numpy as np
    from oracle import ccvs_oracle as O
    qopt, xopt = opt["qvid_generator"], opt["transformer"]
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    nets = {"e": cpu(gen.vid_model.net_e), "q": cpu(gen.vid_model.net_q), "g": cpu(gen.vid_model.net_g),
            "t": cpu(gen.transformer_model.net_t)}
    all_cores = os.cpu_count() or torch.get_num_threads()
    g = torch.Generator().manual_seed(1)
    frame = torch.rand(1, 1, 3, qopt.max_dim, qopt.max_dim, generator=g) * 2 - 1
    t_all = time.perf_counter()
    load0 = os.getloadavg()
    budget = float(os.environ.get("CCVS_CPU_BASELINE_BUDGET", "45"))   # seconds; past it the repeats are dropped (median of what was taken)
    spreads = {}

    def timed(name, fn, reps=3):
        fn()    # untimed: the first call of a kind pays one-time costs (allocator growth, thread-pool start) that are not the host's noise
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget:
                break
        med = statistics.median(ts)
        spreads[name] = (max(ts) - min(ts)) / med if len(ts) > 1 else None
        return med, len(ts)

    calls = 0
    with torch.no_grad():
        O.encoder_forward(nets["e"], qopt, frame[:, :, :, :64, :64])  # untimed warm-up (thread pool, allocator)
        idx_probe = torch.randint(0, xopt.z_num, (1, 384), generator=g)
        sweep = {}
        for nthr in [n for n in (8, 16, 32, 64) if n <= all_cores] or [all_cores]:
            torch.set_num_threads(nthr)
            O.gpt_forward(nets["t"], xopt, idx_probe[:, :64])
            t0 = time.perf_counter()
            O.gpt_forward(nets["t"], xopt, idx_probe)
            sweep[nthr] = time.perf_counter() - t0
            if sweep[nthr] > 1.5 * min(sweep.values()):
                break                                   # more threads only get slower from here
        cores = min(sweep, key=sweep.get)
        torch.set_num_threads(cores)
        t_enc, n = timed("encoder", lambda: O.qvid_encode(nets, qopt, frame)); calls += n
        enc = O.qvid_encode(nets, qopt, frame)
        z = enc["z"]
        ctx = [f for f in enc["inter"]]
        t_dec1, n = timed("decoder k=1", lambda: O.decoder_forward(nets["g"], qopt, z, [ctx])); calls += n
        t_dec3, n = timed("decoder k=3", lambda: O.decoder_forward(nets["g"], qopt, z, [ctx, ctx, ctx]), reps=2); calls += n
        t_dec2 = 0.5 * (t_dec1 + t_dec3)   # linear in the number of contexts
        ts = {}
        for T in (64, 1023, 512, 256, 768):      # the ends first: they bound the integral if the budget runs out
            idx = torch.randint(0, xopt.z_num, (1, T), generator=g)
            ts[T], n = timed(f"GPT T={T}", lambda: O.gpt_forward(nets["t"], xopt, idx), reps=3 if T < 1000 else 2); calls += n
        ts = dict(sorted(ts.items()))
    # decode: frame with k contexts costs a + b*k ; 1 cond frame (k=1) + 15 frames with k = 1..15
    b = max(t_dec2 - t_dec1, 0.0)
    a = max(t_dec1 - b, 0.0)
    t_decode = (a + b) + sum(a + b * k for k in range(1, 16)) + 15 * t_enc          # + 15 re-encodes
    # GPT: one full forward per new token, T = 64 .. 1023: piecewise-linear through the sampled lengths
    tt, yy = np.array(list(ts.keys()), dtype=np.float64), np.array(list(ts.values()))
    t_gpt = float(np.interp(np.arange(64, 1024), tt, yy).sum())
    coef = np.polyfit(tt, yy, 2)
    fit_err = float(np.max(np.abs(np.polyval(coef, tt) / yy - 1.0)))   # (how far from a quadratic: information, not the estimate)
    t_encode = 16 * t_enc
    total = t_encode + t_gpt + t_decode
    load1 = os.getloadavg()
    worst = max((v for v in spreads.values() if v is not None), default=0.0)
    return {"value": 15.0 / total, "unit": "frames/s", "cores": cores, "kind": "port", "unreliable": bool(worst > 0.10),
            "worst_call_spread": round(worst, 3), "call_spread": {k: (None if v is None else round(v, 3)) for k, v in spreads.items()},
            "threads_swept": {str(k): round(v, 3) for k, v in sweep.items()}, "host_cores": all_cores,
            "load_average": {"before": [round(v, 1) for v in load0], "after": [round(v, 1) for v in load1]},
            "sample": (f"oracle on BAIR batch 1, extrapolated from {time.perf_counter() - t_all:.1f}s of CPU work on {cores} threads (fastest of the sweep; "
                       f"{all_cores} hardware threads on the host): medians of up to 3 calls -- encoder 1 frame {t_enc:.2f}s, decoder 1 frame k=1 {t_dec1:.2f}s / "
                       f"k=3 {t_dec3:.2f}s, GPT forward T={'/'.join(str(k) for k in ts)} {'/'.join(f'{v:.2f}' for v in ts.values())}s ({calls} timed calls; "
                       f"piecewise-linear over T; a quadratic would miss a sample by {100 * fit_err:.0f}%; worst spread of repeated calls {100 * worst:.0f}%) "
                       f"-> clip = encode {t_encode:.0f}s + no-cache token loop {t_gpt:.0f}s + decode {t_decode:.0f}s")}


HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s


@torch.no_grad()
def decode_kernel_rooflines(gen, xopt, batch):
    """HBM rooflines of the decode step's kernels (the token loop is a chain of weight / KV-cache streams): every GEMM
    shape of a layer, the head and the cached attention at the mean cache length of the loop are captured `reps` times back
    to back in a hipGraph and replayed between two HIP events (outside the timed region; the same weights are re-read by
    every launch, so these are L2 / Infinity-Cache-warm figures for the small matrices -- the in-situ durations, where every
    launch streams its own layer's weights from HBM, are in the rocprofv3 kernel stats under profiles/).  achieved = algorithmic bytes (weights, or keys + values) / average duration."""
    from ccvs_amd import ops
    net = gen.transformer_model.net_t
    cfg = net.config
    C, H, V = cfg.n_embd, cfg.n_head, net.head.weight.shape[0]
    D = C // H
    dev = net.head.weight.device
    blk = net.blocks[len(net.blocks) // 2]
    (qw, qb, qs), (fw, fb, fs) = blk.folded()
    hw, hb, hs = net._head_packed()[1]
    x = torch.randn(batch, C, device=dev)
    h = torch.randn(batch, 4 * C, device=dev)
    tmax = xopt.z_len
    kc, vc = torch.randn(batch, H, tmax, D, device=dev), torch.randn(batch, H, tmax, D, device=dev)
    t_mean = (xopt.cond_len + xopt.z_len) // 2
    q3 = torch.randn(batch, 1, C, device=dev)
    cases = [
        ("gemm16 ln1+qkv (+KV scatter)", 4.0 * qw.numel(), lambda: ops.gemm_ln_qkv(x, qw, qb, qs, kc, vc, batch, 1, tmax - 1)),
        ("gemm16 proj+residual", 4.0 * blk.attn.proj.weight.numel(), lambda: ops.gemm_nt(x, blk.attn.proj.weight, blk.attn.proj.bias, ops.EPI_RESIDUAL, residual=x)),
        ("gemm16 ln2+fc+GELU", 4.0 * fw.numel(), lambda: ops.gemm_ln(x, fw, fb, fs, epilogue=ops.EPI_GELU)),
        ("gemm16 fc2+residual (split-K)", 4.0 * blk.mlp[3].weight.numel(), lambda: ops.gemm_nt(h, blk.mlp[3].weight, blk.mlp[3].bias, ops.EPI_RESIDUAL, residual=x)),
        ("gemm16 ln_f+head", 4.0 * hw.numel(), lambda: ops.gemm_ln(x, hw, hb, hs)),
        (f"attention_decode_kernel<{D}> at cache length {t_mean}", 8.0 * batch * H * t_mean * D, lambda: ops.attention(q3, kc, vc, t_mean - 1)),
    ]
    out, reps = [], 100
    for name, nbytes, fn in cases:
        fn()
        graph = torch.cuda.CUDAGraph()             # `reps` launches in one hipGraph: launch gaps as in the captured decode step
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (3 * reps)
        gbps = nbytes / (us * 1e-6) / 1e9
        out.append({"kernel": name, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": us, "launches_timed": 3 * reps, "traffic": None})
    return out


@torch.no_grad()
def gemm16_alone_us(gen, rows):
    """Average duration of a decode-step GEMM at `rows` stacked rows with the chip to itself: the four GEMMs of EVERY layer (own
    weights, so each launch streams from HBM as in the loop) + the head, captured in one hipGraph, replayed between two HIP events."""
    from ccvs_amd import ops
    net = gen.transformer_model.net_t
    C = net.config.n_embd
    dev = net.head.weight.device
    x = torch.randn(rows, C, device=dev)
    h = torch.randn(rows, 4 * C, device=dev)
    H = net.config.n_head
    kc, vc = torch.zeros(rows, H, 8, C // H, device=dev), torch.zeros(rows, H, 8, C // H, device=dev)
    hw, hb, hs = net._head_packed()[1]
    packed = [blk.folded() for blk in net.blocks]

    def step():
        n = 0
        for blk, ((qw, qb, qs), (fw, fb, fs)) in zip(net.blocks, packed):
            ops.gemm_ln_qkv(x, qw, qb, qs, kc, vc, rows, 1, 3)
            ops.gemm_nt(x, blk.attn.proj.weight, blk.attn.proj.bias, ops.EPI_RESIDUAL, residual=x)
            ops.gemm_ln(x, fw, fb, fs, epilogue=ops.EPI_GELU)
            ops.gemm_nt(h, blk.mlp[3].weight, blk.mlp[3].bias, ops.EPI_RESIDUAL, residual=x)
            n += 4
        ops.gemm_ln(x, hw, hb, hs)
        return n + 1

    step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        n = step()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * n)


def gemm16_profile(args):
    """What only a rocprofv3 run can say about the decode GEMM (in-run duration from a kernel trace of the pipelined bench, the CUs'
    read requests to L2 from a --pmc pass): the committed record, with its sources.  None when nothing is committed for the config."""
    path = os.path.join(ROOT, "profiles", "gemm16_inrun.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(f"{args.config}-b{args.batch}")


def conv_traffic(args, kind, launches):
    """HBM bytes per launch of the convolution kernel, from the rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate runs, KiB units; profiles/r01_conv_traffic.json says how it was
    collected).  None when no matching profile is committed: PMC counters cannot be read from inside bench.py."""
    path = os.path.join(ROOT, "profiles", "conv_traffic.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    key = f"{args.config}-b{args.batch}-{kind}"
    if key not in rec:
        return None
    r = rec[key]
    return r


def main():
    args = parse_args()
    if args.sample_noise is None:
        args.sample_noise = "host" if args.config in ("bair", "bair-p2p") else "device"
    profiled = bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprof" in os.environ.get("LD_PRELOAD", "")   # under rocprofv3: measure in THIS process
    if ("WORLD_SIZE" not in os.environ and os.environ.get("CCVS_BENCH_CHILD") != "1" and os.environ.get("CCVS_BENCH_SUPERVISE", "1") != "0"
            and not (profiled and args.gpus == 1)):
        # invoked bare: the measurement runs in child process(es) under a time limit (CCVS_BENCH_FORCE_LAUNCHER=1 takes the
        # torch.distributed.run path with one rank, which exercises the N-rank launch on a one-GPU box)
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    from ccvs_amd.tools.affinity import pin_rank
    cpu_set = pin_rank()      # N > 1: this rank's own block of cores (NUMA-local to its GPU when rocm-smi says so), before the first GPU call
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    from ccvs_amd import lib, ops
    from ccvs_amd.tools.engine import Engine
    lib.load()
    if args.conv_precision:
        ops.CONV_PRECISION = args.conv_precision

    with Engine() as engine:
        gen, opt = build_generator(args)
        gen.engine = engine
        lo, _ = engine.shard_batch(args.batch * world)
        xopt = opt["transformer"]
        predicted = xopt.vid_len - xopt.cond_len // 64 - (1 if xopt.p2p else 0)   # p2p: the last frame is the given end frame
        dev = torch.device("cuda", torch.cuda.current_device())

        def make_batch(step):
            data = {"vid": gen.synthetic_batch(args.batch, seed=1 + step, first_clip=lo)["vid"].to(dev)}
            if xopt.stft:   # audio-conditioned: one 64 x 16 spectrogram frame per video frame (a_stft_shape 8 2 after three stride-2 levels)
                g_ = torch.Generator().manual_seed(77 + step)
                data["stft"] = (torch.rand(args.batch, xopt.vid_len, 1, 64, 16, generator=g_) * 2 - 1).to(dev)
            return data

        # the same clips on every rank (global clips 0-1 of seed 1): the codebook scale, hence the replica, is identical everywhere
        calibrate_codebook(gen, {"vid": gen.synthetic_batch(2, seed=1, first_clip=0)["vid"].to(dev)})

        kept = {}

        def finish(step, fake):
            """uint8 pack on the decode stream, then the RCCL all-gather on a side stream (it overlaps the next batch)."""
            if step == kept.get("step"):     # the self-check below compares this batch with a serial run of the same inputs
                kept["fake"] = fake
            return engine.all_gather_clips_async(ops.pack_u8(fake["vid"]))

        rank_seed = noise_seed_of_rank(engine.rank)    # this rank's own stream (rank 0: NOISE_SEED, as every round so far)

        def run(first, batches, then=None):
            """K batches from resident inputs to all-gathered uint8 clips; returns the gathered clips of the last one.  `then`: the
            length of the run that follows (the warm-up run captures the decode step of THAT run's token-group sizes too: with
            --warmup 1 --steps 3 the timed run's group of three would otherwise be captured inside the clock)."""
            torch.manual_seed(rank_seed)
            if args.schedule == "pipelined":
                ramp = tuple(int(v) for v in args.ramp.split(",") if v) if args.ramp is not None else None
                res = gen.run_pipelined(iter(batches), first_iter=first, cu_limit=args.cu_limit, finish=finish, lanes=args.lanes, chains=args.chains,
                                        ramp=ramp, n_batches=(len(batches), then) if then else len(batches))
                handles = [r["finished"] for r in res]
                stages = gen.pipeline_stage_ms()       # every rank: the line reports the stages per rank
                if engine.is_main:
                    stages["timeline"] = gen.pipeline_timeline()
                    stages["token_groups"] = gen.pipeline_token_groups()   # of THIS run: later passes (single_call: generate_vid streams through the same machinery) replace the generator's record
            else:
                ops.CONV_CU_LIMIT = args.cu_limit or 0     # (experiments: the cost of capping the convolutions, in isolation)
                handles, stages = [], {"encode": 0.0, "transformer": 0.0, "decode": 0.0}
                for i, data in enumerate(batches):
                    out = gen.generate_vid(data, first + i, schedule="stream" if args.schedule == "stream" else "serial")
                    handles.append(finish(first + i, out["fake"]))
                    for k, v in gen.stage_ms().items():
                        stages[k] += v
            clips = None
            for h in handles:
                clips = h.wait()
            return clips, stages

        if args.warmup > 0:
            run(-args.warmup, [make_batch(1000 + w) for w in range(args.warmup)], then=args.steps)
        batches = [make_batch(i) for i in range(args.steps)]   # inputs resident in HBM before the clock starts
        timer = ops.KernelTimer()
        kept["step"] = 0
        torch.cuda.synchronize()
        engine.barrier()
        ops.KERNEL_TIMER = timer
        t0 = time.perf_counter()
        clips, stage = run(0, batches)
        torch.cuda.synchronize()
        engine.barrier()
        elapsed = engine.all_reduce_max(time.perf_counter() - t0)
        ops.KERNEL_TIMER = None
        kept["step"] = None
        assert clips.shape[0] == args.batch * world

        # per-rank stage times (rank order) for the line: which rank, and which stage of it, bounds a multi-GPU run
        rank_stages = [{k: v / args.steps for k, v in stage.items() if k not in ("timeline", "token_groups")}]
        if engine.distributed:
            import torch.distributed as dist
            gathered = [None] * world
            dist.all_gather_object(gathered, rank_stages[0])
            rank_stages = gathered
        alone, self_check, alone_passes, single_call = None, None, None, None
        if engine.is_main and args.schedule == "pipelined":
            # timed batch 0 once more, ALONE on the serial schedule (generate_vid: same inputs, same iteration index, the process
            # generator re-seeded as in front of the timed pass -- batch 0 is its first consumer in both), outside the timed region:
            # (1) the self-check of the line -- the clip the pipelined schedule produced for batch 0 beside up to 15 other batches
            #     in flight must be the serial clip, bit for bit (tokens and fp32 pixels);
            # (2) the same convolution launches with the chip to themselves: the kernel's roofline (in the pipelined schedule
            #     the launches are timed beside the token loops of other batches)
            # The pass is run three times and the MEDIAN pass is the one reported (all of them listed in the line): right
            # after the timed region single passes of one box came out between 183 and 266 TFLOP/s (the same pass in a process of
            # its own: 266-267 every time, tools/r05/alone_var.py) -- the first pass runs into whatever the timed region left behind
            # (clock state; worker threads winding down on the host side of the launch-bound small convolutions).
            passes, serial_call_s = [], []
            for _ in range(3):       # ALONE_PASSES
                timer_p = ops.KernelTimer()
                ops.KERNEL_TIMER = timer_p
                torch.manual_seed(rank_seed)
                torch.cuda.synchronize()
                t_call = time.perf_counter()
                out_p = gen.generate_vid({k: v.clone() for k, v in batches[0].items()}, 0, schedule="serial")
                torch.cuda.synchronize()
                serial_call_s.append(time.perf_counter() - t_call)
                ops.KERNEL_TIMER = None
                n_p, f_p, ms_p = timer_p.summary("conv2d_" + ops.CONV_PRECISION)
                passes.append((f_p / (ms_p * 1e-3) / 1e12 if ms_p > 0 else 0.0, timer_p, (n_p, f_p, ms_p)))
                if len(passes) == 1:
                    out_s = out_p          # the self-check compares the FIRST pass
                del out_p
            alone_passes = [round(v[0], 1) for v in passes]
            _, timer_alone, (n_a, f_a, ms_a) = sorted(passes, key=lambda v: v[0])[len(passes) // 2]
            alone = f_a / (ms_a * 1e-3) / 1e12 if ms_a > 0 else None
            del passes
            if "fake" in kept:
                same_tok = bool(torch.equal(kept["fake"]["code"], out_s["fake"]["code"]))
                d = (kept["fake"]["vid"] - out_s["fake"]["vid"]).abs().max().item()
                self_check = {"pipelined_equals_serial": bool(same_tok and d == 0.0), "tokens_equal": same_tok, "max_abs": d,
                              "what": f"timed batch 0 ({args.batch} clips x {xopt.vid_len} frames, produced with up to {gen.last_lanes * (gen.last_chains + 2)} "
                                      f"batches in flight: token groups of {gen.last_lanes} x {gen.last_chains} chains, {getattr(gen, 'last_dec_streams', 1)} decode "
                                      "streams) against generate_vid of the same inputs alone, same noise seed: tokens and fp32 pixels"}
                if not self_check["pipelined_equals_serial"]:
                    print(f"bench.py: SELF-CHECK FAILED: {self_check}", file=sys.stderr, flush=True)
            kept.pop("fake", None)
            # the single-call drop-in (Generator.generate_vid, ONE batch, nothing else in flight): the three passes above are that call in the
            # reference's order (token loop, then decode); the same call with the decoder following its own token loop frame by frame
            # (schedule="stream", the default of generate_vid since round 6): one untimed call (its decode step's capture key), two timed
            stream_call_s, same_stream = [], None
            for rep in range(3):
                torch.manual_seed(rank_seed)
                torch.cuda.synchronize()
                t_call = time.perf_counter()
                out_q = gen.generate_vid({k: v.clone() for k, v in batches[0].items()}, 0, schedule="stream")
                torch.cuda.synchronize()
                if rep:
                    stream_call_s.append(time.perf_counter() - t_call)
                else:
                    same_stream = bool(torch.equal(out_q["fake"]["code"], out_s["fake"]["code"]) and torch.equal(out_q["fake"]["vid"], out_s["fake"]["vid"]))
                del out_q
            single_call = {"serial_frames_per_s": predicted * args.batch / min(serial_call_s), "stream_frames_per_s": predicted * args.batch / min(stream_call_s),
                           "serial_ms_per_call": [round(1e3 * v) for v in serial_call_s], "stream_ms_per_call": [round(1e3 * v) for v in stream_call_s],
                           "stream_equals_serial": same_stream,
                           "note": "Generator.generate_vid on ONE batch with nothing else in flight, wall time per call (best of the calls listed): `serial` = encode, "
                                   "the whole token loop, then the decode (the reference's order; these are the three alone passes of the convolution roofline, "
                                   "HIP-event timers on); `stream` = the decoder follows the call's own token loop frame by frame (generate_vid's default); "
                                   "same clips bit for bit"}
            del out_s
        if engine.is_main:
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            frames = predicted * args.batch * world * args.steps
            kind = ops.CONV_PRECISION
            n_conv, conv_flops, conv_ms = timer.summary("conv2d_" + kind)
            achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
            # Pipelined schedule: a launch duration in the timed region includes the time the convolution shares the chip with the
            # token loops of other batches -- a figure of the schedule, not of the kernel.  The kernel's roofline is taken from
            # the same launches of one more batch with nothing beside them (HIP events, right after the timed region; these are
            # the durations rocprofv3's kernel trace shows for the serial schedule, where nothing runs beside a convolution); the timed-region
            # figure stays in the line as `in_timed_region`.
            shared = None
            if alone:
                shared = {"achieved": achieved, "frac": achieved / (BF16_MFMA_PEAK_TFLOPS if kind == "bf16x3" else FP32_MFMA_PEAK_TFLOPS),
                          "launches": n_conv, "avg_launch_us": 1e3 * conv_ms / max(n_conv, 1),
                          "stream_seconds_per_step_second": conv_ms * 1e-3 / elapsed,
                          "whole_job_tflops": conv_flops / elapsed / 1e12,
                          "whole_job_frac": conv_flops / elapsed / 1e12 / (BF16_MFMA_PEAK_TFLOPS if kind == "bf16x3" else FP32_MFMA_PEAK_TFLOPS),
                          "note": f"HIP events around every convolution launch of the {args.steps} timed batches: the launches share the chip with "
                                  f"{gen.last_chains} token loops (each over the stacked rows of up to {gen.last_lanes} other batches) -- all of them: "
                                  "the decoder follows the token loops frame by frame, so the run has no quiet tail in which the last decodes ran "
                                  "alone (rounds 2-3 and the first half of round 4 averaged such a tail into this figure)"
                                  + (f"; and with the decodes of {gen.last_dec_streams} batches side by side on {gen.last_dec_streams} streams: a launch's "
                                     "duration here is that of a kernel sharing the chip with the other decode's kernels as well -- a figure of the "
                                     "schedule (compare `value`), not of the kernel (`achieved`)" if getattr(gen, "last_dec_streams", 1) > 1 else "")}
                n_conv, conv_flops, conv_ms, achieved = n_a, f_a, ms_a, alone
                timer = timer_alone
            # bf16x3: three bf16 MFMA products per algorithmic fp32 product; peak = dense bf16 MFMA
            peak = BF16_MFMA_PEAK_TFLOPS if kind == "bf16x3" else FP32_MFMA_PEAK_TFLOPS
            products = 3 if kind == "bf16x3" else 1
            n_enc = gen._frames_to_encode(xopt.vid_len, int(torch.prod(torch.tensor(opt["qvid_generator"].z_shape))))
            line = {
                "metric": {"bair": "synthesized frames/sec (BAIR 256x256, cond=1, pred=15), whole job",
                           "kinetics": "synthesized frames/sec (Kinetics-600 64x64, cond=5, pred=11), whole job",
                           "bair-p2p": "synthesized frames/sec (BAIR 256x256 point-to-point, start + end frame given, 14 interpolated), whole job",
                           "drums": "synthesized frames/sec (AudioSet-Drums 128x128, audio-conditioned, cond=15, pred=30), whole job"}[args.config],
                "value": frames / elapsed,
                "unit": "frames/s", "per_gpu": frames / elapsed / world, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 (convolutions as split-bf16 x3 on bf16 MFMA, fp32 accumulate)" if kind == "bf16x3" else "f32",
                "data": "synthetic",
                "config": {"workload": {"bair": "BAIR 256x256 1->15 frames, batch 16 per GPU (BASELINE.json configs[1])",
                                        "kinetics": "Kinetics-600 64x64 5->11 frames (BASELINE.json configs[2])",
                                        "bair-p2p": "BAIR 256x256 point-to-point, 14 interpolated frames (BASELINE.json configs[3], scripts/bairhd/save_videos_p2p.sh)",
                                        "drums": "AudioSet-Drums 128x128 audio-conditioned, 15 -> 30 frames, STFT tokens given (BASELINE.json configs[4], "
                                                 "scripts/drums/save_videos_audio_on.sh)"}[args.config], "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                           "predicted_frames_per_clip": predicted,
                           "encode": (f"all {xopt.vid_len} frames of every input clip, as the reference's generate_vid does (helpers/generator.py:69)" if n_enc == xopt.vid_len
                                      else f"the {n_enc} conditioning frame(s) per clip that synthesis reads (`--encode cond`)"),
                           "rec_pass": ("also run, not counted" if args.rec_pass else "off: the reference's extra teacher-forced reconstruction decode is not part of "
                                        "the synthesized-frames metric (SURVEY 8d)"), "sampling": f"top-k {xopt.top_k}, T={xopt.temperature}, noise={args.sample_noise}" + (
                               " (the Exp(1) blocks torch.multinomial draws from the process generator, transformer_model.py:395-409: pre-drawn batch after "
                               "batch by a noise thread, read inside the captured decode step of every token group)" if args.sample_noise == "host" else
                               " (in-kernel Philox4x32-10 keyed by the global clip index)"),
                           "schedule": (f"pipelined: up to {gen.last_lanes * (gen.last_chains + 2)} batches in flight per GPU -- {gen.last_chains} token loops on "
                                        f"their own streams, each ONE loop over the stacked rows of {gen.last_lanes} consecutive batches (weights streamed "
                                        f"once per token for all of them, per-batch KV rows and sampler words), beside the encoder and {getattr(gen, 'last_dec_streams', 1)} decode streams "
                                        f"(the decoder takes a batch's tokens frame by frame; the decode of batch i runs on stream i % {getattr(gen, 'last_dec_streams', 1)})"
                                        + (f", whose kernels are capped to {gen.last_cu_limit} of {n_cu} CUs" if gen.last_cu_limit else "")
                                        + f"; every generate call is one batch of {args.batch} clips; "
                                        "K batches timed from resident inputs to gathered clips, fill and drain included") if args.schedule == "pipelined"
                                       else ("serial: one batch at a time" if args.schedule == "serial" else
                                             "stream: one batch at a time, its decoder following its own token loop frame by frame (generate_vid(schedule='stream'))"),
                           "parallelism": f"dp{world} (batch sharded, one RCCL all-gather of uint8 clips on a side stream)",
                           "conv_intermediates": ("packed split-bf16 (hi + lo, 4 bytes per element like fp32; bit-identical results) between the convolutions of "
                                                  "Matching / Subpixel" if (ops.CONV_P8 and kind == "bf16x3") else "fp32")},
                "self_check": self_check,
                "single_call": single_call,
                "stage_ms_per_step": {k: v / args.steps for k, v in stage.items() if k not in ("timeline", "token_groups")},
                "timeline_ms": stage.get("timeline"),
                "stage_note": "per-batch stage durations from HIP events on the stage's own stream" +
                              ("; in the pipelined schedule encode+decode (stream D) and the transformer stages (one stream per lane) overlap" if args.schedule == "pipelined" else ""),
                "roofline": {"kernel": ("conv2d_bf16x3_pc_kernel<TW,MB,NTY> + conv2d_bf16x3_kernel<TW,MB> (every instantiation: all conv launches)" +
                                        (" + conv2d_bf16x3_pt_kernel<MB,PP,P8IN> (CCVS_CONV_PT=" + os.environ["CCVS_CONV_PT"] + ": everywhere)" if os.environ.get("CCVS_CONV_PT", "") not in ("", "0") else
                                         (" + conv2d_bf16x3_pt_kernel<4,2,false> in the ALONE passes only: a single generate call runs the fp32-input 128-channel 3 x 3 layers "
                                          "as persistent tiles (ccvs_conv_persistent_tiles, mode " + str(ops.conv_persistent_tiles()) + "); while several batches are in flight "
                                          "-- the timed region -- helpers/pipeline.py keeps the per-tile kernels, which are the faster ones there"
                                          if "CCVS_CONV_PT" not in os.environ and ops.conv_persistent_tiles() else ""))
                                        if kind == "bf16x3" else "conv2d_mfma_kernel<TW,MB>"), "bound": "mfma",
                             "power_limited_peak": {"note": "NOT measured by this command (profiles/r06_mfma_rate_probe.txt: tools/micro/mfma_rate_probe.hip, back-to-back "
                                                            "v_mfma_f32_32x32x16_bf16 on all 256 CUs): on RANDOM data the chip runs bf16 matrix instructions at 1.63 GHz -- 1713 "
                                                            "TFLOP/s dense (2459 on all-zero data = the guide's 2.5 PFLOP/s `peak` above); with a convolution tap's LDS operand "
                                                            "fetches 1566, with a staging wave beside 1454.  Three bf16 products per algorithmic product: `achieved` x 3 is the "
                                                            "bf16 matrix work actually executed",
                                                    "dense_bf16_random_data_tflops": 1713.0, "with_lds_operand_fetches": 1566.0, "with_staging_wave": 1454.0, "zero_data": 2459.0,
                                                    "frac_of_power_limited": (3.0 * achieved / 1454.0) if kind == "bf16x3" else None},
                             "clock_note": ("NOT measured by this command (profiles/r06_power_trace.txt, tools/power_trace.py around this command on another box): the pipelined "
                                            "passes run power-managed -- 1319 W of the socket's 1400 W at a shader clock of 2209 MHz -- while one batch alone (the `alone_passes`) "
                                            "runs at 1159 W and 2389 MHz; `peak` is the 2.4 GHz figure for both") if args.schedule == "pipelined" else None,
                             "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                             "frac_in_run": (shared["whole_job_frac"] if shared else achieved / peak),
                             "frac_note": ("`frac` = the kernel with the chip to itself (the alone pass below); `frac_in_run` = the same launches' algorithmic FLOP "
                                           "over the WALL time of the timed region (all decode streams together, token loops beside them) -- the figure of the schedule; "
                                           "in_timed_region.frac is per launch duration, i.e. per decode stream" if shared else None),
                             "measured": ("HIP events around every convolution launch of one more batch run with nothing beside it, right after the timed "
                                          "region: the median of three such passes, `alone_passes` (pipelined schedule: see in_timed_region)" if shared else
                                          "HIP events around every convolution launch of the timed region"),
                             "alone_passes": alone_passes if shared else None,
                             "in_timed_region": shared,
                             "cu_share": (f"launches capped to {gen.last_cu_limit} of {n_cu} CUs while a token loop is in flight: against that share of the "
                                          f"peak the fraction is {achieved / (peak * gen.last_cu_limit / n_cu):.4f}") if args.schedule == "pipelined" and gen.last_cu_limit else None,
                             "traffic": (conv_traffic(args, kind, n_conv) or {}).get("bytes_per_launch"),
                             "traffic_over_algorithmic": ((conv_traffic(args, kind, n_conv) or {}).get("bytes_per_launch", 0.0) /
                                                          (timer.total_bytes("conv2d_" + kind) / max(n_conv, 1)) if conv_traffic(args, kind, n_conv) else None),
                             "traffic_note": "PMC counters cannot be read inside the timed run: FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc run of the "
                                             "same convolutions (tools/pmc_conv_traffic.sh -> profiles/conv_traffic.json), not re-measured by this command",
                             "traffic_detail": {k_: v_ for k_, v_ in (conv_traffic(args, kind, n_conv) or {}).items() if k_ != "per_instantiation"} or None,
                             "mfma_products_per_flop": products, "mfma_issue_frac": products * achieved / peak,
                             "vs_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                             "launches": n_conv, "avg_launch_us": 1e3 * conv_ms / max(n_conv, 1),
                             "algorithmic_gflop_per_launch": conv_flops / max(n_conv, 1) / 1e9,
                             "algorithmic_bytes_per_launch": timer.total_bytes("conv2d_" + kind) / max(n_conv, 1),
                             "clock_note": "GRBM cycles / time = 2.0 GHz under this kernel (power-limited), i.e. a 2083 TF/s bf16 ceiling",
                             "stream_seconds_per_step_second": (shared["stream_seconds_per_step_second"] if shared else conv_ms * 1e-3 / elapsed),
                             "stream_seconds_note": "sum of the convolution launches' durations over the wall time of the timed region; > 1 when the launches of "
                                                    "several decode streams overlap in time (until round 5 this field was called share_of_step_time)"},
            }
            tl = stage.get("timeline")
            if tl:   # the price of the schedule: a batch is in flight from its encode to the end of its decode
                lat = [t["d1"] - t["e0"] for t in tl]
                line["latency_ms_per_batch"] = {"mean": sum(lat) / len(lat), "max": max(lat),
                                                "note": "encode start to decode end of one batch (HIP events); the serial schedule's is its step time"}
                if args.schedule == "pipelined":   # stream D: encodes + the pieces of the decodes, against its span in the timed region
                    busy = stage["encode"] + stage["decode"]
                    span = max(t["d1"] for t in tl) - min(t["e0"] for t in tl)
                    line["decoder_stream"] = {"busy_ms": busy, "span_ms": span, "idle_frac": max(0.0, 1.0 - busy / span),
                                              "first_decode_starts_ms": min(t["d0"] for t in tl), "token_stages_end_ms": max(t["t1"] for t in tl),
                                              "decode_streams": getattr(gen, "last_dec_streams", 1),
                                              "note": "HIP events around every encode and every piece (conditioning frames / one frame) of every decode, on the "
                                                      "stream the piece runs on; with more than one decode stream the pieces of two batches overlap in time and "
                                                      "busy_ms (their sum) exceeds the span: idle_frac is then a lower bound of 0"}
            line["hbm_peak_allocated_gb"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30   # every batch in flight + weights + graphs' pools (torch allocator)
            # the token loop as a whole, in situ: weights ONCE per step of a token group + the keys and values of every batch in it,
            # against the HBM peak
            net_t = gen.transformer_model.net_t
            n_tok = predicted * 64    # frame tokens sampled per clip (drums: the window slides, so every slide also re-prefills ~1200 tokens;
                                      # the byte count below is the plain loop's and only approximate for that config)
            w_bytes = 4.0 * sum(p_.numel() for n_, p_ in net_t.named_parameters() if n_.startswith("blocks.") and p_.dim() == 2) + 4.0 * net_t.head.weight.numel()
            kv_bytes = 8.0 * args.batch * net_t.config.n_embd * len(net_t.blocks) * (xopt.cond_len + xopt.z_len) / 2
            if args.schedule == "pipelined":
                groups = stage["token_groups"]        # (recorded with the timed run's stages)
                n_chains = gen.last_chains
            else:
                groups, n_chains = [(1, stage["transformer"] / args.steps)] * args.steps, 1
            loop_ms = sum(ms for _, ms in groups)
            loop_bytes = sum(n_tok * (w_bytes + g * kv_bytes) for g, _ in groups)
            gbps = loop_bytes / (loop_ms * 1e-3) / 1e9
            mean_group = sum(g for g, _ in groups) / len(groups)
            line["roofline_token_loop"] = {
                "kernel": "token loop (ccvs_gpt_decode_step replayed as a hipGraph: 5 x n_layer + 3 launches per step; one step serves every batch of the token group)",
                "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "ms_per_step": loop_ms / (len(groups) * n_tok), "ms_per_token_per_batch": loop_ms / (sum(g for g, _ in groups) * n_tok),
                "tokens_per_clip": n_tok, "batches_per_token_group": [g for g, _ in groups], "mean_batches_per_group": mean_group,
                "algorithmic_bytes_per_step": w_bytes + mean_group * kv_bytes, "weights_bytes": w_bytes, "mean_kv_bytes_per_batch": kv_bytes,
                "concurrent_token_loops": n_chains, "aggregate_frac": n_chains * gbps / HBM_PEAK_GBPS, "approximate": args.config == "drums",
                "note": "HIP events on each token stream around the whole loop of every token group of the timed region (prefill of the conditioning frame "
                        "included); bytes = weights counted ONCE per step + mean KV bytes of every batch in the group; `frac` is ONE loop's stream while "
                        "`concurrent_token_loops` loops and the decoder share the memory system (aggregate_frac = loops x frac)"}
            if args.schedule == "pipelined" and args.config in ("bair", "bair-p2p", "kinetics"):
                # the same token loop with the chip to itself (per step, alone against in-run): one whole loop over the
                # stacked rows of a full token group, graph-replayed, in-kernel noise, outside the timed region
                g_rows = max(1, int(round(mean_group))) * args.batch
                n_g = g_rows // args.batch
                import copy
                code0 = torch.randint(0, xopt.z_num, (g_rows, xopt.cond_len), device=dev)
                probe = copy.copy(net_t)          # the same parameters, its own engine state (KV cache, captured steps): the run's stay untouched
                probe.drop_engine_state()
                probe.noise_key, probe.row_offset, probe.noise_call = [(11 + g_, 7 + g_) for g_ in range(n_g)], [0] * n_g, 0
                try:
                    n_new = min(n_tok, xopt.z_len - xopt.cond_len)
                    probe.generate(code0, 24, sample=True, top_k=xopt.top_k, temperature=xopt.temperature, noise="device")   # capture
                    torch.cuda.synchronize()
                    loops_ms = []
                    for _ in range(2):     # two loops, the faster one reported: one in four single loops came out at 9.4 instead of 3.2 ms per step
                        t0 = time.perf_counter()
                        probe.generate(code0, n_new, sample=True, top_k=xopt.top_k, temperature=xopt.temperature, noise="device")
                        torch.cuda.synchronize()
                        loops_ms.append(1e3 * (time.perf_counter() - t0) / n_new)
                    alone_ms = min(loops_ms)
                    line["roofline_token_loop"]["alone"] = {
                        "ms_per_step": alone_ms, "loops_ms_per_step": [round(v, 3) for v in loops_ms],
                        "rows": g_rows, "tokens": n_new, "in_run_over_alone": line["roofline_token_loop"]["ms_per_step"] / alone_ms,
                        "frac": (w_bytes + n_g * kv_bytes) / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "note": f"one loop over {g_rows} stacked rows ({n_g} batches), prefill of the conditioning frame included, nothing beside it; "
                                "in the timed region the same step shares the chip with the other token loop and two decodes"}
                finally:
                    probe.drop_engine_state()
                    del probe
            # everything that runs in the timed region shares one memory system: the algorithmic bytes of the two big consumers
            conv_bytes_step = timer.total_bytes("conv2d_" + kind) / (1 if alone else args.steps)   # the convolutions of one batch
            line["pipeline_hbm"] = {"token_loop_GB_per_step": loop_bytes / args.steps / 1e9, "conv_GB_per_step": conv_bytes_step / 1e9,
                                    "GBps_of_these_two": (loop_bytes / args.steps + conv_bytes_step) / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                                    "note": "algorithmic bytes per batch of the token loops (weights once per step of a group + keys and values) and of the "
                                            "decoder's convolutions over the measured step time; the stencil kernels of the decoder (~0.43 TB per BAIR batch, "
                                            "profiles/r04_pmc_decoder_kernels.txt) and the encoder come on top -- DESIGN.md 4.3"}
            line["roofline_decode_kernels"] = decode_kernel_rooflines(gen, xopt, args.batch)
            # The top kernel by time (VERDICT r5, weak 6 / line hygiene 9): the decode GEMM as ONE entry -- its launches of a step stream the
            # step's weights once: bytes per launch = weights / launches; `alone` from the token-loop probe above is per step, so the
            # per-launch figure here is measured live on the stacked rows of a token group with every layer's OWN weights (HBM-cold,
            # unlike roofline_decode_kernels' warm re-launches); in-run duration and L2 requests are rocprofv3's (committed record).
            n_gemm = 4 * len(net_t.blocks) + 1
            g16 = {"kernel": "gemm16_kernel<WNT, RB, CB, U> (every GEMM of a decode step: ln1+qkv, proj, ln2+fc, fc2 per layer, ln_f+head)",
                   "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "launches_per_step": n_gemm,
                   "algorithmic_bytes_per_launch": w_bytes / n_gemm,
                   "note": "weights only (the activations of 16-128 rows are KBs and L2-resident); one step = one pass over the weights for every batch of the token group"}
            try:
                rows_g = max(1, int(round(mean_group))) * args.batch
                us_alone = gemm16_alone_us(gen, rows_g)
                g16["alone"] = {"rows": rows_g, "avg_launch_us": us_alone, "achieved": w_bytes / n_gemm / (us_alone * 1e-6) / 1e9,
                                "frac": w_bytes / n_gemm / (us_alone * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                "measured": "HIP events around a hipGraph of the step's GEMMs over ALL layers (each launch its own layer's weights: 1.2 GB per "
                                            "pass, nothing re-read from cache), nothing beside it; attention / embed / pick left out"}
                g16["achieved"], g16["frac"] = g16["alone"]["achieved"], g16["alone"]["frac"]
            except Exception as exc:   # never lose the line to a side measurement
                g16["alone"] = {"error": repr(exc)}
            prof = gemm16_profile(args)
            if prof:
                by = w_bytes / n_gemm
                g16["in_run"] = {"avg_launch_us": prof["in_run_avg_us"], "achieved": by / (prof["in_run_avg_us"] * 1e-6) / 1e9,
                                 "frac": by / (prof["in_run_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, "rows": prof["rows"], "source": prof["in_run_source"],
                                 "not_measured_by_this_command": True}
                g16["l2"] = {"read_requests_per_launch": prof["l2_read_requests_per_launch"],
                             "bytes_through_the_cu_l2_path_per_launch": prof["l2_read_requests_per_launch"] * prof["l2_request_bytes"],
                             "over_algorithmic": prof["l2_read_requests_per_launch"] * prof["l2_request_bytes"] / by,
                             "hbm_fetch_over_algorithmic": prof["hbm_fetch_bytes_per_launch"] / by, "source": prof["counters_source"]}
            line["roofline_gemm16"] = g16
            line["multi_gpu"] = {"rccl_ranks": (torch.distributed.get_world_size() if engine.distributed else 1), "backend": engine.backend if engine.distributed else None,
                                 "stage_ms_per_step_by_rank": rank_stages,
                                 "host_threads_per_rank": {"torch_intra_op": torch.get_num_threads(), "launch_threads": 1 + (gen.last_chains if args.schedule == "pipelined" else 0),
                                                           "noise_thread": int(args.schedule == "pipelined" and args.sample_noise == "host")},
                                 "cpu_affinity": ({"cores_of_rank_0": cpu_set, "note": "every rank pins itself to its own block of cores before its first GPU call "
                                                   "(ccvs_amd/tools/affinity.py: NUMA-local to its GPU when `rocm-smi --showtoponuma` parses, else contiguous blocks)"}
                                                  if cpu_set is not None else "one rank: not pinned"),
                                 "noise_seed": {"rank_0": rank_seed, "rule": "rank r seeds its process generator with NOISE_SEED + r (bench.py: noise_seed_of_rank): "
                                                "the ranks' host-drawn sampling streams differ, as the reference's unseeded processes' do; `--sample-noise device` "
                                                "is the world-size-invariant sampler (Philox keyed by the global clip index)"},
                                 "scaling_curve": "NO SCALING CURVE EXISTS: no round of this build has had more than one GPU (SCALE_r01-r05: skipped, no 8-GPU node); "
                                                  "this run is one line for N = " + str(world) + "; the N > 1 path is covered by code, by the world-2 gloo tests and by "
                                                  "tools/host_stress.py (profiles/r06_host_stress.txt: 8 ranks' host work side by side, no GPU) -- the driver derives "
                                                  "efficiency from the N = 1, 2, 4, 8 lines when a node exists (tools/scale_sweep.sh runs them)"}
            if args.encode == "all" and args.config == "bair" and world == 1 and not args.no_encode_cond_leg and not args.rec_pass:
                # the same K batches and schedule with only the conditioning frame of every clip encoded: synthesis reads nothing else
                # ("from conditioning frames resident on GPU"); the reference also encodes the 15 frames it is about to
                # replace, for its rec pass, and that is what `value` above includes.  Reported beside it, never as `value`.
                xopt.encode_all = False
                try:
                    run(4000, [make_batch(4000 + w) for w in range(max(1, min(args.warmup, 3)))])   # untimed
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, stage_c = run(5000, batches)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    line["encode_cond_only"] = {"value": frames / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / args.steps,
                                                "stage_ms_per_step": {k: v / args.steps for k, v in stage_c.items() if k not in ("timeline", "token_groups")},
                                                "frames_encoded_per_clip": gen._frames_to_encode(xopt.vid_len, int(torch.prod(torch.tensor(opt["qvid_generator"].z_shape)))),
                                                "note": f"same {args.steps} batches and schedule, `--encode_all false`: the encoder sees the conditioning frame(s) only; "
                                                        "the synthesized clips are the same bits (tests/test_features_gpu.py::test_encode_conditioning_frames_only)"}
                finally:
                    xopt.encode_all = True
            if args.config == "bair" and world == 1 and not args.no_other_noise_leg and args.schedule == "pipelined" and not args.rec_pass:
                # the same K batches and schedule with the OTHER noise source: reference-seed sampling (host) must cost the schedule nothing
                other = "device" if args.sample_noise == "host" else "host"
                trs = [gen.transformer_model] + [tr_ for tr_, _ in gen._chains]
                for tr_ in trs:
                    tr_.sample_noise = other
                try:
                    run(6000, [make_batch(6000 + w) for w in range(max(1, min(args.warmup, 3)))])   # untimed (captures the other sampler's step)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, stage_o = run(0, batches)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    line["sampling_" + other + "_noise"] = {
                        "value": frames / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / args.steps, "vs_headline": (frames / dt) / line["value"],
                        "stage_ms_per_step": {k: v / args.steps for k, v in stage_o.items() if k not in ("timeline", "token_groups")},
                        "note": f"same {args.steps} batches, same schedule, noise={other}" + (
                            ": in-kernel Philox keyed by the global clip index (validated distributionally; world-size invariant)" if other == "device" else
                            ": the reference's seeded torch.multinomial stream, pre-drawn per batch by the noise thread")}
                finally:
                    for tr_ in trs:
                        tr_.sample_noise = args.sample_noise
            if args.config == "bair" and world == 1 and not args.conv_precision and not args.no_strict_f32:
                # the same path with EXACT fp32 convolutions (v_mfma_f32_32x32x2_f32 instead of the split-bf16 products): what the
                # arithmetic choice of `dtype` buys, on the SAME schedule and the same K batches, in the driver's own line; plus the
                # one B = 16 comparison of the two arithmetics: timed batch 0 decoded by both (GPU vs GPU, teacher-forced on the
                # headline's tokens so that a VQ near-tie flipped by the encoder's arithmetic cannot hide the pixel difference)
                torch.manual_seed(rank_seed)
                ref_out = gen.generate_vid({k: v.clone() for k, v in batches[0].items()}, 0, schedule="serial")          # split-bf16, batch 0 (= the self-check's clip)
                ops.CONV_PRECISION = "f32"
                try:
                    t32 = ops.KernelTimer()
                    ops.KERNEL_TIMER = t32
                    ws32 = gen.condition({k: v.clone() for k, v in batches[0].items()})
                    vid32 = gen.decode_codes(ws32, ref_out["fake"]["code"])["vid"]
                    torch.cuda.synchronize()
                    ops.KERNEL_TIMER = None
                    n32, f32flops, ms32 = t32.summary("conv2d_f32")
                    tf = f32flops / (ms32 * 1e-3) / 1e12 if ms32 > 0 else 0.0
                    line["dtype_check"] = {
                        "max_abs_pixel_diff": (vid32 - ref_out["fake"]["vid"]).abs().max().item(), "bound": 1e-3,
                        "vq_codes_equal": bool(torch.equal(ws32["encoded"]["code"], ref_out["enc_code"])),
                        "what": f"timed batch 0, {args.batch} clips x {xopt.vid_len} frames at 256x256: split-bf16 x3 convolutions (the headline) against exact "
                                "fp32 convolutions, both on the GPU, both decoding the headline's sampled tokens through the 15-frame recurrence "
                                "(every synthesized frame re-encoded into the context ring); north_star's bound is 1e-3 against the fp32 CPU path"}
                    del ws32, vid32, ref_out
                    if args.schedule == "pipelined":
                        run(3000, [make_batch(3000 + w) for w in range(max(1, min(args.warmup, 2)))])   # untimed: fp32 weight forms packed
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        _, stage32 = run(0, batches)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        n_b, sched = args.steps, f"pipelined, the same {args.steps} batches and schedule as the headline"
                    else:
                        torch.manual_seed(rank_seed)
                        t0 = time.perf_counter()
                        for i in range(2):
                            gen.generate_vid({k: v.clone() for k, v in batches[i % len(batches)].items()}, i, schedule="serial")
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        n_b, sched, stage32 = 2, "serial, 2 batches", None
                    line["strict_f32"] = {"frames_per_s": predicted * args.batch * n_b / dt, "vs_headline": (predicted * args.batch * n_b / dt) / line["value"],
                                          "schedule": sched, "ms_per_step": 1e3 * dt / n_b,
                                          "stage_ms_per_step": ({k: v / n_b for k, v in stage32.items() if k not in ("timeline", "token_groups")} if stage32 else None),
                                          "conv_tflops": tf, "frac_of_fp32_mfma_peak": tf / FP32_MFMA_PEAK_TFLOPS, "conv_launches": n32,
                                          "conv_measured": "HIP events around every convolution launch of one batch (encode + 15-frame decode) with nothing beside it"}
                finally:
                    ops.CONV_PRECISION = kind
                    ops.KERNEL_TIMER = None
            if not args.no_cpu_baseline and args.config == "bair" and world == 1:   # rank 0 at N = 1 only
                line["cpu_baseline"] = cpu_baseline(gen, opt)
            print(json.dumps(line))


if __name__ == "__main__":
    main()
