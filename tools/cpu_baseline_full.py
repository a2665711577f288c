#!/usr/bin/env python3
"""The CPU baseline run ONCE IN FULL: the oracle -- the CPU restatement of the
reference's algorithm, with its no-KV-cache token loop (transformer_model.py:343-350: one full forward per new token) -- on one
whole BAIR clip (256x256, 1 conditioning frame -> 15 synthesized frames, batch 1: BASELINE.json configs[0]), per-stage seconds,
core count and load average printed; the anchor of the extrapolated `cpu_baseline` in bench.py's line.

Run on the GPU box it doubles as the one FREE-RUNNING full-size parity check: the HIP path generates the same clip first (same
weights, same frames, sampled with host-drawn noise under the same `torch.manual_seed`), and the oracle's 960 sampled tokens and
15 decoded frames are compared with it -- tokens must be identical, pixels within 1e-3.

    python tools/cpu_baseline_full.py [--threads 16] [--out profiles/r05_cpu_baseline_full.txt]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--seed", type=int, default=2026)
    args = ap.parse_args()
    from ccvs_amd.tools.options import Options, BAIR_ARGV
    from ccvs_amd.helpers.generator import Generator
    from oracle import ccvs_oracle as O
    import bench

    opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                          argv=list(BAIR_ARGV) + ["--batch_size_vid", "1", "--x_sample_noise", "host", "--rec_pass", "false"])
    qopt, xopt = opt["qvid_generator"], opt["transformer"]
    torch.manual_seed(0)
    gen = Generator(opt).build_models()
    vid = gen.synthetic_batch(1, seed=1)["vid"]
    bench.calibrate_codebook(gen, {"vid": gen.synthetic_batch(2, seed=1, first_clip=0)["vid"].cuda()})
    lines = []

    def say(msg):
        print(msg, flush=True)
        lines.append(msg)

    # the HIP path first: the clip the oracle has to reproduce
    torch.manual_seed(args.seed)
    t0 = time.perf_counter()
    out = gen.generate_vid({"vid": vid.clone().cuda()}, 0)
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    hip_code, hip_vid, hip_enc = out["fake"]["code"].cpu(), out["fake"]["vid"].cpu(), out["enc_code"].cpu()

    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    nets = {"e": cpu(gen.vid_model.net_e), "q": cpu(gen.vid_model.net_q), "g": cpu(gen.vid_model.net_g), "t": cpu(gen.transformer_model.net_t)}
    del gen
    torch.cuda.empty_cache()
    torch.set_num_threads(args.threads)
    say(f"# CPU baseline in full: oracle (reference algorithm, no KV cache) on ONE BAIR clip 256x256 1->15, batch 1 (BASELINE.json configs[0])")
    say(f"host: {os.cpu_count()} hardware threads, torch threads {args.threads}, load average before {tuple(round(v, 1) for v in os.getloadavg())}")
    size = qopt.z_shape[0] * qopt.z_shape[1]
    with torch.no_grad():
        torch.manual_seed(args.seed)
        t0 = time.perf_counter()
        enc = O.qvid_encode(nets, qopt, vid)                              # all 16 frames, as generate_vid does (generator.py:69)
        t_enc = time.perf_counter() - t0
        code = enc["code"][:, :xopt.cond_len]
        inter = [f[:, :1].contiguous() for f in enc["inter"]]
        t0 = time.perf_counter()
        fake_code = O.generate_fake(nets["t"], xopt, code, xopt.vid_len * size)   # 960 sampled tokens, a full forward each
        t_gpt = time.perf_counter() - t0
        t0 = time.perf_counter()
        fake = O.qvid_decode(nets, qopt, fake_code, inter)
        t_dec = time.perf_counter() - t0
    total = t_enc + t_gpt + t_dec
    say(f"encode 16 frames {t_enc:.1f} s | token loop (960 tokens, no cache) {t_gpt:.1f} s | decode 1 + 15 frames (15 re-encodes) {t_dec:.1f} s | "
        f"clip {total:.1f} s")
    say(f"cpu_baseline (full run): {15.0 / total:.5f} synthesized frames/s on {args.threads} threads; load average after "
        f"{tuple(round(v, 1) for v in os.getloadavg())}")
    say(f"HIP path, the same clip alone (batch 1, serial schedule, first call incl. graph capture): {t_hip:.2f} s")
    same_enc = bool(torch.equal(hip_enc, enc["code"]))
    same_tok = bool(torch.equal(hip_code, fake_code))
    n_diff = int((hip_code != fake_code).sum())
    say(f"free-running parity on this clip (sampled, top-k {xopt.top_k}, host noise under torch.manual_seed({args.seed})): VQ codes of the 16 input "
        f"frames equal: {same_enc}; the 960 sampled tokens equal: {same_tok}" + ("" if same_tok else f" ({n_diff} differ, first at {int((hip_code != fake_code).nonzero()[0, 1])})"))
    if same_tok:
        per_frame = [(hip_vid[:, t] - fake[:, t]).abs().max().item() for t in range(fake.shape[1])]
        say("decoded pixels, max|HIP - oracle| per frame: " + " ".join(f"{d:.1e}" for d in per_frame) + f" (bound 1e-3: {'ok' if max(per_frame) < 1e-3 else 'EXCEEDED'})")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    return 0 if (same_enc and same_tok) else 1


if __name__ == "__main__":
    sys.exit(main())
