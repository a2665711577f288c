"""Checkpoint I/O with the reference's API (models/__init__.py:5-132): same file naming
(`{label}_net_{iter}.pth`, `_latest_`, `_best_`), `block_delta` key shifting and non-strict
loading, so reference checkpoints load into the MI355X modules unchanged."""
import os
from glob import glob

import torch


def save_network(net, label, which_iter, opt, latest=False, best=False):
    if net is None:
        return
    assert not (latest and best), "Either 'latest' or 'best' but both were given"
    kind = "latest_" if latest else ("best_" if best else "")
    save_path = os.path.join(opt.checkpoint_path, f"{label}_{kind}net_{which_iter}.pth")
    old_paths = glob(os.path.join(opt.checkpoint_path, f"{label}_{kind}net_*.pth")) if kind else []
    os.makedirs(opt.checkpoint_path, exist_ok=True)
    torch.save(net.state_dict(), save_path)
    for old in old_paths:
        if os.path.abspath(old) != os.path.abspath(save_path):
            os.unlink(old)


def load_state_dict(net, state_dict, strict=True, block_delta=None):
    if block_delta is not None:
        shifted = {}
        for key, val in state_dict.items():
            if "blocks" in key:
                pre, post = key.split("blocks")
                parts = post.split(".")
                parts[1] = str(int(parts[1]) + block_delta)
                key = pre + "blocks" + ".".join(parts)
            shifted[key] = val
        state_dict = shifted
    if strict:
        net.load_state_dict(state_dict)
        return
    own = net.state_dict()
    keep = {k: v for k, v in state_dict.items() if k in own and own[k].shape == v.shape}
    for k in state_dict:
        if k not in keep:
            print(f"Skipping {k} (missing in model or size mismatch)")
    own.update(keep)
    net.load_state_dict(own)


def _resolve(load_path, label, which_iter, required):
    if which_iter in ("latest", "best"):
        found = glob(os.path.join(load_path, f"{label}_{which_iter}_net_*.pth"))
        assert len(found) > 0, f"Did not find any checkpoint for {label} net at {which_iter} iter and path {load_path}"
        assert len(found) == 1
        return found[0]
    for name in (f"{label}_net_{which_iter}.pth", f"{label}_latest_net_{which_iter}.pth", f"{label}_best_net_{which_iter}.pth"):
        cand = os.path.join(load_path, name)
        if os.path.exists(cand):
            return cand
    if required:
        raise ValueError(f"No checkpoint for {label} net at iter {which_iter} and path {load_path}")
    return None


def load_network(net, label, opt, override_iter=None, override_load_path=None, required=True, head_to_n=0, block_delta=None):
    if net is None:
        return None
    which_iter = override_iter if override_iter is not None else getattr(opt, "which_iter", 0)
    load_path = override_load_path if override_load_path is not None else getattr(opt, "load_path", None)
    if load_path is not None and str(which_iter) != "0":
        path = _resolve(load_path, label, which_iter, required)
        if path is None:
            print(f"Loading untrained {label} net")
            return net
        if head_to_n != 0:
            raise NotImplementedError("head_to_n (continuous multi-proposal head) is outside the hot path")
        state = torch.load(path, map_location="cpu")
        state = {k: v for k, v in state.items() if not k.endswith(".attn.mask")}  # 1024^2 causal-mask buffers: not needed
        load_state_dict(net, state, strict=not getattr(opt, "not_strict", False), block_delta=block_delta)
        print(f"Loading checkpoint for {label} net from {path}")
    else:
        print(f"Loading untrained {label} net")
    return net


def print_network(net):
    if net is not None:
        print(net)
        print("Total number of parameters: %d" % sum(p.numel() for p in net.parameters()))
