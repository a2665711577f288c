"""-m gpu: the decode step as ONE launch (`ccvs_gpt_decode.persistent`, gpt.hip: gpt_step_kernel -- resident workgroups walking the
step's phases with in-launch grid barriers and sc1 hand-offs) against the launch chain it replaces (5 n_layer + 3 launches): the
same tile bodies in the same order, so EVERYTHING the step leaves behind must be the same bits -- tokens, logits, the residual
stream, the KV cache rows it appended, the device-resident counters -- for every sampler, for row groups, through hipGraph
replays and eagerly, at 16 rows (one block per workgroup) and at stacked rows (the 2 x 2 block tile, split-K tickets inside a
phase).  Reference of the arithmetic: mingpt.py:99-117,232-305 + transformer_model.py:395-409 (through the oracle-pinned chain)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(seed=11, n_layer=3, n_embd=256, n_head=4, vocab=200):
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(seed)
    net = mingpt.GPT(vocab_size=vocab, block_size=400, num_blocks=25, n_layer=n_layer, n_head=n_head, n_embd=n_embd, emb_mode="temporal",
                     shape=(4, 4)).cuda()
    for p in net.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    return net


def _run(net, persistent, codes, n_new, groups, use_graph, sample, noise="device", host_streams=None):
    net.drop_engine_state()
    net.persistent_step = persistent
    keys = [(0x1234567 + 977 * g, 0xabcdef01 ^ (g << 7)) for g in range(groups)]
    if groups > 1:
        net.noise_key, net.row_offset, net.noise_call = list(keys), [32] * groups, 0
    else:
        net.noise_key, net.row_offset, net.noise_call = keys[0], 32, 0
    if host_streams is not None:
        net.noise_streams = [s.clone() for s in host_streams]
    out = net.generate(codes, n_new, sample=sample, top_k=20, noise=noise, use_graph=use_graph)
    net.check_steps()
    torch.cuda.synchronize()
    c = net._cache
    state = {"tokens": out.clone(), "logits": c["logits"].clone(), "x": c["x"].clone(), "q": c["q"].clone(), "att": c["att"].clone(), "h": c["h"].clone(),
             "k": [k.clone() for k in c["k"]], "v": [v.clone() for v in c["v"]], "len": c["len_dev"].clone(), "widx": c["widx"].clone(),
             "state": c["state"].clone()}
    net.noise_key, net.row_offset, net.noise_streams = None, 0, None
    return state


def _assert_same(a, b, what):
    for key in ("tokens", "logits", "x", "q", "att", "h", "len", "widx", "state"):
        assert torch.equal(a[key], b[key]), f"{what}: `{key}` differs between the launch chain and the persistent step"
    L = int(a["len"].max())     # (the caches are torch.empty: only the appended rows are defined)
    for l, (ka, kb, va, vb) in enumerate(zip(a["k"], b["k"], a["v"], b["v"])):
        assert torch.equal(ka[:, :, :L], kb[:, :, :L]) and torch.equal(va[:, :, :L], vb[:, :, :L]), f"{what}: KV cache of layer {l} differs"


@pytest.mark.parametrize("batch,groups", [(16, 1), (7, 1), (16, 3), (5, 3), (16, 4), (24, 2), (16, 6)])
@pytest.mark.parametrize("use_graph", [True, False])
def test_persistent_step_equals_launch_chain(batch, groups, use_graph):
    """Sampled (in-kernel Philox), graph-replayed and eager: 16 / 7 rows (one block per workgroup), 48 / 15 / 64 / 48 / 96 stacked rows
    (2 x 2 blocks; 5 rows per group puts two groups into one 16-row tile; 96 rows = three row pairs)."""
    net = _net()
    codes = torch.randint(0, 200, (batch * groups, 16), device="cuda")
    n_new = 60 if use_graph else 12
    chain = _run(net, False, codes, n_new, groups, use_graph, sample=True)
    pers = _run(net, True, codes, n_new, groups, use_graph, sample=True)
    assert not torch.equal(chain["tokens"][:, 16:], chain["tokens"][:, 16:17].expand(-1, n_new)), "degenerate sample"
    _assert_same(chain, pers, f"{batch} x {groups} rows, graph={use_graph}")


def test_persistent_step_greedy_and_host_noise_streams():
    """The other two samplers of the step: greedy, and the reference's host-drawn Exp(1) stream read block by block inside the step
    (`ccvs_gpt_decode.noise_stream`), per row group."""
    net = _net(seed=5)
    batch, groups, n_new = 16, 3, 40
    codes = torch.randint(0, 200, (batch * groups, 16), device="cuda")
    chain = _run(net, False, codes, n_new, groups, True, sample=False)
    pers = _run(net, True, codes, n_new, groups, True, sample=False)
    _assert_same(chain, pers, "greedy")
    g = torch.Generator().manual_seed(9)
    streams = [torch.empty(n_new, batch, 200).exponential_(1, generator=g).cuda() for _ in range(groups)]
    chain = _run(net, False, codes, n_new, groups, True, sample=True, noise="host", host_streams=streams)
    pers = _run(net, True, codes, n_new, groups, True, sample=True, noise="host", host_streams=streams)
    _assert_same(chain, pers, "host noise streams")


def test_persistent_step_many_sequences_back_to_back():
    """The barrier words are never zeroed between launches (arrival counters run on, `base` moves once per launch): hundreds of
    steps and several sequences through ONE workspace, alternating row counts (two instantiations, two phase tables), stay equal to
    the chain -- and the step's status word stays clean."""
    net = _net(seed=2, n_layer=2)
    for rnd in range(3):
        for batch, groups in ((16, 1), (16, 4)):
            codes = torch.randint(0, 200, (batch * groups, 16), device="cuda")
            chain = _run(net, False, codes, 120, groups, True, sample=True)
            pers = _run(net, True, codes, 120, groups, True, sample=True)
            _assert_same(chain, pers, f"round {rnd}, {batch} x {groups}")


def test_persistent_step_full_size_gpt_under_load():
    """BAIR geometry (24 x 1024, 16 heads, head dim 64, V = 1024), 64 stacked rows, 200 tokens from a cache of 64: the persistent
    step against the chain while ANOTHER stream keeps the chip unevenly busy with convolutions (workgroups of the step then become
    resident late and at different times: the hand-offs are tested under uneven load, consumers warm) -- tokens and every buffer
    equal; then the same with nothing beside it."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    from ccvs_amd import ops
    torch.manual_seed(0)
    net = mingpt.GPT(vocab_size=1024, block_size=1024, num_blocks=16, n_layer=24, n_head=16, n_embd=1024, emb_mode="temporal", shape=(8, 8)).cuda()
    codes = torch.randint(0, 1024, (64, 64), device="cuda")
    chain = _run(net, False, codes, 200, 4, True, sample=True)
    pers = _run(net, True, codes, 200, 4, True, sample=True)
    _assert_same(chain, pers, "full size, alone")
    side = torch.cuda.Stream()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(24, 96, 256, 256, generator=g).cuda()
    w = ops.pack_conv_weight(torch.randn(128, 96, 3, 3, generator=g).cuda())
    bias = torch.zeros(128, device="cuda")
    stop = torch.zeros(1, device="cuda")

    def load():
        with torch.cuda.stream(side):
            for i in range(300):
                ops.conv2d(x[: 3 + (i % 5) * 5], w, bias, 128, 3, pad=1, act=True)      # ragged launches: 3 ... 23 images
    import threading
    th = threading.Thread(target=load)
    side.wait_stream(torch.cuda.current_stream())
    th.start()
    pers2 = _run(net, True, codes, 200, 4, True, sample=True)
    th.join()
    torch.cuda.synchronize()
    del stop
    _assert_same(chain, pers2, "full size, beside convolutions")
