"""GPU parity of the decode-step helpers and of the whole decode engine.
Floating-point kernels: compared with a plain PyTorch fp32 reference of the same op (tolerances in each test)."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O
from util import REL_TOL, elem_err_ok, layer_to_torch, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("m,h", [(1, 4096), (3, 5120), (2, 256)])
def test_rmsnorm_vs_torch_fp32(m, h):
    from qeft_amd import _lib
    torch.manual_seed(0)
    x = torch.randn(m, h, device=DEV).half()
    add = torch.randn(m, h, device=DEV).half()
    g = (1 + 0.1 * torch.randn(h, device=DEV)).half()
    y = torch.empty_like(x)
    res = torch.empty_like(x)
    _lib.check(_lib.lib().qeft_rmsnorm(x.data_ptr(), None, g.data_ptr(), None, y.data_ptr(), m, h, 1e-5, _st()))
    ref = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5) * g.float()
    assert torch.allclose(y.float(), ref, rtol=2e-3, atol=2e-3)
    _lib.check(_lib.lib().qeft_rmsnorm(x.data_ptr(), add.data_ptr(), g.data_ptr(), res.data_ptr(), y.data_ptr(), m, h,
                                       1e-5, _st()))
    hsum = (x.float() + add.float()).half().float()
    ref = hsum * torch.rsqrt(hsum.pow(2).mean(-1, keepdim=True) + 1e-5) * g.float()
    assert torch.equal(res.float(), hsum)
    assert torch.allclose(y.float(), ref, rtol=2e-3, atol=2e-3)


def test_silu_mul_vs_torch_fp32():
    from qeft_amd import _lib
    torch.manual_seed(1)
    g = torch.randn(11008, device=DEV).half() * 3
    u = torch.randn(11008, device=DEV).half()
    o = torch.empty_like(g)
    _lib.check(_lib.lib().qeft_silu_mul(g.data_ptr(), u.data_ptr(), o.data_ptr(), g.numel(), _st()))
    ref = torch.nn.functional.silu(g.float()) * u.float()
    assert torch.allclose(o.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("n_heads,n_kv,steps,split", [(4, 4, 70, 1), (8, 2, 300, 1), (32, 32, 200, 4), (8, 2, 300, 2),
                                                       (32, 8, 330, 8), (4, 4, 40, 4)])
def test_rope_attention_decode_vs_torch_fp32(n_heads, n_kv, steps, split):
    from qeft_amd import _lib
    nws = _lib.lib().qeft_attn_workspace_bytes(n_heads, split)
    assert (nws > 0) == (split > 1)
    ws = torch.zeros(max(nws // 4, 4), dtype=torch.float32, device=DEV)
    torch.manual_seed(2)
    hd, max_seq = 128, 512
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, dtype=torch.float64) / 64))
    ang = torch.arange(max_seq, dtype=torch.float64)[:, None] * inv[None]
    cs, sn = ang.cos().float().to(DEV), ang.sin().float().to(DEV)
    kc = torch.zeros(n_kv, max_seq, hd, device=DEV, dtype=torch.float16)
    vc = torch.zeros_like(kc)
    pos = torch.zeros(1, dtype=torch.int32, device=DEV)
    Q = torch.randn(steps, n_heads, hd, device=DEV).half()
    K = torch.randn(steps, n_kv, hd, device=DEV).half()
    V = torch.randn(steps, n_kv, hd, device=DEV).half()
    out = torch.empty(n_heads * hd, device=DEV, dtype=torch.float16)

    def rope(x, t):
        a, b = x[..., :64].float(), x[..., 64:].float()
        return torch.cat([a * cs[t] - b * sn[t], b * cs[t] + a * sn[t]], -1)

    kr = torch.stack([rope(K[t], t) for t in range(steps)]).half().float()   # cache holds fp16
    for t in range(steps):
        pos.fill_(t)
        # odd steps hand over the pre-selected rotary row (tab_rows = 1), even steps the whole table
        row = t % 2 == 1
        _lib.check(_lib.lib().qeft_rope_attn_decode(Q[t].data_ptr(), K[t].data_ptr(), V[t].data_ptr(),
                                                    cs[t].data_ptr() if row else cs.data_ptr(),
                                                    sn[t].data_ptr() if row else sn.data_ptr(), 1 if row else max_seq,
                                                    kc.data_ptr(), vc.data_ptr(), pos.data_ptr(),
                                                    None, out.data_ptr(), ws.data_ptr(), split, n_heads, n_kv, max_seq,
                                                    _st()))
        if t in (0, 1, steps // 2, steps - 1) or (split > 1 and t % 7 == 3):
            q = rope(Q[t], t)                                                  # [H, 128]
            rep = n_heads // n_kv
            kk = kr[:t + 1].repeat_interleave(rep, 1)                          # [L, H, 128]
            vv = V[:t + 1].float().repeat_interleave(rep, 1)
            att = torch.einsum("hd,lhd->hl", q, kk) / math.sqrt(hd)
            ref = torch.einsum("hl,lhd->hd", att.softmax(-1), vv).reshape(-1)
            torch.cuda.synchronize()
            assert torch.allclose(out.float(), ref, rtol=5e-3, atol=5e-3), t
    assert torch.allclose(kc[:, :steps].float().transpose(0, 1), kr, rtol=2e-3, atol=1e-3)   # 1 fp16 ulp (fma vs mul+sub)
    # out_pos: the same output, scattered (used to pre-apply o_proj's column order)
    perm = torch.randperm(n_heads * hd, device=DEV).to(torch.int32)
    out2 = torch.empty_like(out)
    pos.fill_(steps - 1)
    _lib.check(_lib.lib().qeft_rope_attn_decode(Q[-1].data_ptr(), K[-1].data_ptr(), V[-1].data_ptr(), cs.data_ptr(),
                                                sn.data_ptr(), max_seq, kc.data_ptr(), vc.data_ptr(), pos.data_ptr(),
                                                perm.data_ptr(), out2.data_ptr(), ws.data_ptr(), split, n_heads, n_kv,
                                                max_seq, _st()))
    torch.cuda.synchronize()
    assert torch.equal(out2[perm.long()], out)
    # the split merge is order-fixed: repeated launches (back to back, no host sync) give the same bits
    outs = [torch.empty_like(out) for _ in range(20)]
    for o in outs:
        _lib.check(_lib.lib().qeft_rope_attn_decode(Q[-1].data_ptr(), K[-1].data_ptr(), V[-1].data_ptr(), cs.data_ptr(),
                                                    sn.data_ptr(), max_seq, kc.data_ptr(), vc.data_ptr(), pos.data_ptr(),
                                                    None, o.data_ptr(), ws.data_ptr(), split, n_heads, n_kv, max_seq,
                                                    _st()))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, out)


@pytest.mark.parametrize("ns,k,r", [((4096, 4096, 4096), 4096, 128), ((11008, 11008), 4096, 128), ((256, 64, 64), 512, 0),
                                    ((512, 256), 11008, 128)])
def test_grouped_gemv_equals_separate_launches(ns, k, r):
    """q/k/v (gate/up) in one launch == the per-linear GEMV, bit for bit (same per-row arithmetic)."""
    from qeft_amd import _lib, qeft_cuda
    g = 128
    layers = [layer_to_torch(O.make_layer(n, k, r, g, seed=10 + i), DEV) for i, n in enumerate(ns)]
    x = torch.from_numpy(O.make_activation(1, k, r, seed=3)).to(DEV)
    ys = [torch.empty(1, n, device=DEV, dtype=torch.float16) for n in ns]

    def arr(ts):
        a = (ctypes.c_void_p * len(ts))()
        for i, t in enumerate(ts):
            a[i] = t.data_ptr()
        return a
    _lib.check(_lib.lib().qeft_gemv_w4_group(
        x.data_ptr(), None, 0.0, len(ns), arr([l["qweight"] for l in layers]), arr([l["scales"] for l in layers]),
        arr([l["scaled_zeros"] for l in layers]), arr([l["oweight_interleaved"] for l in layers]) if r else None, None, None,
        arr(ys), (ctypes.c_int * len(ns))(*ns), k, g, r, _st()))
    for l, n, y in zip(layers, ns, ys):
        if r:
            ref = qeft_cuda.gemv_4bit_qeft(x, l["qweight"], l["scales"], l["scaled_zeros"], l["oweight_interleaved"],
                                           1, n, k, g)
        else:
            ref = qeft_cuda.gemv_4bit(x, l["qweight"], l["scales"], l["scaled_zeros"], 1, n, k, g)
        torch.cuda.synchronize()
        assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("use_graph", [False, True])
def test_decode_engine_matches_dense_fp32_model(use_graph):
    """Teacher-forced logits / NLL of the HIP decode engine vs a plain PyTorch fp32 model over the dense
    dequantised weights, on a tiny Llama-shaped model (SURVEY.md §8d 'PPL parity on synthetic weights')."""
    from qeft_amd.llama import DecodeEngine, QuantLlama, nll_from_logits, tiny_shape
    shape = tiny_shape(n_layers=3, hidden=256, inter=512, n_heads=2, vocab=384, max_seq=64)
    model = QuantLlama(shape, DEV, seed=1)
    eng = DecodeEngine(model, use_graph=use_graph)
    tokens = torch.randint(0, shape.vocab, (40,), generator=torch.Generator().manual_seed(0)).to(DEV)
    got = eng.teacher_forced_logits(tokens)
    ref = model.forward_dense_reference(tokens)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() / scale < 2e-2
    assert abs(nll_from_logits(got, tokens) - nll_from_logits(ref, tokens)) < 5e-3


def test_decode_engine_greedy_graph_equals_eager():
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=2, hidden=256, inter=512, n_heads=2, vocab=384, max_seq=64)
    model = QuantLlama(shape, DEV, seed=2)
    seqs = []
    for use_graph in (False, True):
        eng = DecodeEngine(model, use_graph=use_graph)
        eng.greedy = True
        eng.reset()
        eng.tok.fill_(5)
        toks = []
        for _ in range(24):
            eng.step()
            toks.append(int(eng.tok.item()))
        seqs.append(toks)
    assert seqs[0] == seqs[1]


def _arr(ts):
    a = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        a[i] = t.data_ptr()
    return a


@pytest.mark.parametrize("ns,k", [((4096, 4096, 4096), 4096), ((11008, 11008), 4096), ((256, 256), 512)])
def test_fused_rmsnorm_group_equals_unfused(ns, k):
    """RMSNorm folded into the grouped GEMV == qeft_rmsnorm followed by the grouped GEMV (within the path's tolerance),
    and both within it of the float64 result on the oracle-dequantised weights."""
    from qeft_amd import _lib
    lib, r, g = _lib.lib(), 128, 128
    bufs = [O.make_layer(n, k, r, g, seed=20 + i) for i, n in enumerate(ns)]
    layers = [layer_to_torch(b, DEV) for b in bufs]
    torch.manual_seed(5)
    x = (torch.randn(1, k, device=DEV) * 2).half()
    gamma = (1 + 0.1 * torch.randn(k, device=DEV)).half()
    xn = torch.empty_like(x)
    y0 = [torch.empty(1, n, device=DEV, dtype=torch.float16) for n in ns]
    y1 = [torch.empty(1, n, device=DEV, dtype=torch.float16) for n in ns]
    packs = (_arr([l["qweight"] for l in layers]), _arr([l["scales"] for l in layers]),
             _arr([l["scaled_zeros"] for l in layers]), _arr([l["oweight_interleaved"] for l in layers]))
    nn_ = (ctypes.c_int * len(ns))(*ns)
    _lib.check(lib.qeft_rmsnorm(x.data_ptr(), None, gamma.data_ptr(), None, xn.data_ptr(), 1, k, 1e-5, _st()))
    _lib.check(lib.qeft_gemv_w4_group(xn.data_ptr(), None, 0.0, len(ns), *packs, None, None, _arr(y0), nn_, k, g, r,
                                      _st()))
    from qeft_amd import qeft_cuda
    szp = [qeft_cuda.pack_scales(l["scales"], l["scaled_zeros"], n, k, g) for l, n in zip(layers, ns)]
    _lib.check(lib.qeft_gemv_w4_group(x.data_ptr(), gamma.data_ptr(), 1e-5, len(ns), *packs, None, _arr(szp), _arr(y1),
                                      nn_, k, g, r, _st()))
    torch.cuda.synchronize()
    # the fused kernel stages x * gamma and applies 1/rms to the finished dot products: one fp16 rounding per
    # activation like the unfused norm, but not the same one -- equal within the path's tolerance, not bit for bit
    x64 = x.cpu().numpy().astype(np.float64)[0]
    xn64 = x64 / np.sqrt((x64 ** 2).mean() + 1e-5) * gamma.cpu().numpy().astype(np.float64)
    for a, b, bf in zip(y0, y1, bufs):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < REL_TOL
        assert elem_err_ok(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-3, atol_scale=2e-3)
        w = O.dequant_dense(bf["qweight"], bf["scales"], bf["scaled_zeros"], bf["oweight"], g).astype(np.float64)
        ref = w @ xn64
        assert rel_err(a.cpu().numpy()[0], ref) < REL_TOL and rel_err(b.cpu().numpy()[0], ref) < REL_TOL


@pytest.mark.parametrize("n,k", [(4096, 11008), (256, 512)])
def test_fused_silu_down_equals_unfused(n, k):
    from qeft_amd import _lib
    lib, r, g = _lib.lib(), 128, 128
    l = layer_to_torch(O.make_layer(n, k, r, g, seed=31), DEV)
    torch.manual_seed(6)
    gate = (torch.randn(k, device=DEV) * 2).half()
    up = torch.randn(k, device=DEV).half()
    res = torch.randn(1, n, device=DEV).half()
    act = torch.empty_like(gate)
    y0 = torch.empty(1, n, device=DEV, dtype=torch.float16)
    y1 = torch.empty_like(y0)
    _lib.check(lib.qeft_silu_mul(gate.data_ptr(), up.data_ptr(), act.data_ptr(), k, _st()))
    _lib.check(lib.qeft_gemv_w4_fused(act.data_ptr(), l["qweight"].data_ptr(), l["scales"].data_ptr(),
                                      l["scaled_zeros"].data_ptr(), l["oweight_interleaved"].data_ptr(), None, None,
                                      res.data_ptr(), None, y0.data_ptr(), 1, n, k, g, r, _st()))
    from qeft_amd import qeft_cuda
    szp = qeft_cuda.pack_scales(l["scales"], l["scaled_zeros"], n, k, g)
    _lib.check(lib.qeft_gemv_w4_silu(gate.data_ptr(), up.data_ptr(), l["qweight"].data_ptr(), l["scales"].data_ptr(),
                                     l["scaled_zeros"].data_ptr(), l["oweight_interleaved"].data_ptr(), None,
                                     res.data_ptr(), szp.data_ptr(), y1.data_ptr(), n, k, g, r, _st()))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("n,k,g", [(4096, 4096, 128), (256, 512, 128), (64, 2048, 2048)])
def test_pack_scales_shadow_layout_and_equivalence(n, k, g):
    """qeft_pack_scales: out[n/16][grp][n%16] = scale | scaled_zero << 16, and the GEMV gives identical results
    with and without the shadow."""
    from qeft_amd import qeft_cuda
    l = layer_to_torch(O.make_layer(n, k, 128, g, seed=41), DEV)
    szp = qeft_cuda.pack_scales(l["scales"], l["scaled_zeros"], n, k, g)
    s16 = l["scales"].view(torch.int16).to(torch.int32) & 0xFFFF
    z16 = l["scaled_zeros"].view(torch.int16).to(torch.int32) & 0xFFFF
    word = (s16 | (z16 << 16)).to(torch.int32)                     # [groups, N]
    ref = word.reshape(k // g, n // 16, 16).permute(1, 0, 2).contiguous()
    torch.cuda.synchronize()
    assert torch.equal(szp, ref)
    x = torch.from_numpy(O.make_activation(2, k, 128, seed=2)).to(DEV)
    y0 = qeft_cuda.gemv_4bit_fused(x, l["qweight"], l["scales"], l["scaled_zeros"], l["oweight_interleaved"], None, None,
                                   None, 2, n, k, g)
    y1 = qeft_cuda.gemv_4bit_fused(x, l["qweight"], l["scales"], l["scaled_zeros"], l["oweight_interleaved"], None, None,
                                   None, 2, n, k, g, szp)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("vocab,hidden", [(32000, 4096), (1003, 256)])
def test_token_begin_end_match_torch(vocab, hidden):
    """Embedding lookup + rotary-row select in front of the layers, greedy argmax + pos += 1 behind lm_head."""
    from qeft_amd import _lib
    lib = _lib.lib()
    torch.manual_seed(5)
    max_seq = 64
    embed = torch.randn(vocab, hidden, device=DEV).half()
    tab = torch.randn(max_seq, 128, device=DEV)
    tok = torch.tensor([vocab - 3], dtype=torch.long, device=DEV)
    pos = torch.tensor([17], dtype=torch.int32, device=DEV)
    h = torch.zeros(hidden, device=DEV, dtype=torch.float16)
    row = torch.zeros(128, device=DEV)
    _lib.check(lib.qeft_token_begin(embed.data_ptr(), tok.data_ptr(), tab.data_ptr(), pos.data_ptr(), h.data_ptr(),
                                    row.data_ptr(), hidden, vocab, max_seq, _st()))
    torch.cuda.synchronize()
    assert torch.equal(h, embed[vocab - 3]) and torch.equal(row, tab[17])
    for trial in range(4):
        logits = torch.randn(vocab, device=DEV).half()
        if trial == 1:       # ties: the lowest index wins, like torch.argmax
            logits[[5, 700, vocab - 1]] = 9.0
        if trial == 2:
            logits[vocab - 1] = 11.0
        _lib.check(lib.qeft_token_end(logits.data_ptr(), tok.data_ptr(), pos.data_ptr(), vocab, 1, _st()))
        torch.cuda.synchronize()
        assert int(tok.item()) == int(torch.argmax(logits.float()).item()), trial
        assert int(pos.item()) == 18 + trial
    before = int(tok.item())
    _lib.check(lib.qeft_token_end(None, None, pos.data_ptr(), vocab, 0, _st()))    # not greedy: only the position moves
    torch.cuda.synchronize()
    assert int(tok.item()) == before and int(pos.item()) == 22


def test_prefill_matches_dense_model_and_hands_over_to_decode():
    """Batched prompt pass through the packed linears (GEMM path) == the dense fp32 model within tolerance, and decoding
    from its KV caches == decoding the whole sequence token by token."""
    from qeft_amd.llama import DecodeEngine, QuantLlama, nll_from_logits, prefill, tiny_shape
    shape = tiny_shape(n_layers=2, hidden=256, inter=512, n_heads=2, vocab=384, max_seq=64)
    model = QuantLlama(shape, DEV, seed=6)
    tokens = torch.randint(0, shape.vocab, (37,), generator=torch.Generator().manual_seed(3)).to(DEV)
    T = 32
    eng = DecodeEngine(model, use_graph=False)
    got = prefill(model, tokens[:T], eng).float()
    ref = model.forward_dense_reference(tokens[:T])
    torch.cuda.synchronize()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-2
    assert abs(nll_from_logits(got, tokens[:T]) - nll_from_logits(ref, tokens[:T])) < 5e-3
    assert int(eng.pos.item()) == T
    # continue decoding 5 tokens from the prefilled caches; compare with a token-by-token run of everything
    outs = []
    for t in tokens[T:].tolist():
        eng.tok.fill_(t)
        eng.step()
        outs.append(eng.logits[0].float().clone())
    cont = torch.stack(outs)
    full = DecodeEngine(model, use_graph=False).teacher_forced_logits(tokens)[T:]
    torch.cuda.synchronize()
    assert (cont - full).abs().max().item() / full.abs().max().item() < 2e-2
    assert (cont.argmax(-1) == full.argmax(-1)).float().mean().item() >= 0.8


def test_attention_split_follows_the_position():
    """Past 256 cached positions the engine switches to 4 attention blocks per head (its own captured graph); logits
    stay within tolerance of the one-block-per-head run and the switch happens where it should."""
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=2, hidden=256, inter=512, n_heads=2, vocab=384, max_seq=320)
    model = QuantLlama(shape, DEV, seed=8)
    tokens = torch.randint(0, shape.vocab, (300,), generator=torch.Generator().manual_seed(5))
    eng = DecodeEngine(model, use_graph=True)
    assert [eng._split_for(p) for p in (0, 255, 256, 1535, 1536)] == [1, 1, 4, 4, 8]
    got = eng.teacher_forced_logits(tokens)
    assert sorted(eng.graphs) == [(1, False), (4, False)] and eng.host_pos == 300
    one = DecodeEngine(model, use_graph=False)
    one.attn_split_forced = 1
    ref = one.teacher_forced_logits(tokens)
    torch.cuda.synchronize()
    assert torch.equal(got[:256], ref[:256])                       # same kernels, same order before the switch
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 5e-3


def test_engine_refuses_to_run_past_the_kv_cache():
    """host-side guard: at position max_seq the device side would skip the attention and keep counting (stale output)."""
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=1, hidden=256, inter=512, n_heads=2, vocab=128, max_seq=32)
    eng = DecodeEngine(QuantLlama(shape, DEV, seed=3), use_graph=False)
    eng.set_position(31)
    eng.step()
    with pytest.raises(RuntimeError, match="KV cache full"):
        eng.step()
    with pytest.raises(ValueError):
        eng.set_position(33)


def test_graphs_are_keyed_by_split_and_greedy():
    """A graph captured with greedy token_end must not be replayed for a teacher-forced run (and vice versa)."""
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=1, hidden=256, inter=512, n_heads=2, vocab=128, max_seq=32)
    eng = DecodeEngine(QuantLlama(shape, DEV, seed=4), use_graph=True)
    eng.greedy = True
    eng.tok.fill_(3)
    eng.step()
    first = int(eng.tok.item())
    eng.reset()
    eng.greedy = False
    eng.tok.fill_(3)
    eng.step()
    torch.cuda.synchronize()
    assert int(eng.tok.item()) == 3                      # the non-greedy graph leaves the fed token alone
    assert sorted(eng.graphs) == [(1, False), (1, True)]
    assert first == int(eng.logits[0].float().argmax().item())


# ---- the reference module's FasterTransformer-derived entries (qeft_cuda.cpp:22-26) through the `qeft_cuda` shim -------
def test_shim_layernorm_forward_cuda():
    import qeft_cuda                                            # the top-level alias a reference caller imports
    torch.manual_seed(0)
    x = (torch.randn(2, 5, 512, device=DEV) * 3).half()
    gamma = (1 + 0.1 * torch.randn(512, device=DEV)).half()
    out = torch.empty_like(x)
    assert qeft_cuda.layernorm_forward_cuda(x, gamma, out, 1e-5) is None       # positional, writes `out` (layernorm.cu:94-110)
    torch.cuda.synchronize()
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma.float()
    assert (out.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    with pytest.raises(RuntimeError):
        qeft_cuda.layernorm_forward_cuda(x.float(), gamma, out, 1e-5)


@pytest.mark.parametrize("B,H,Hkv,rot,neox,alibi", [(1, 4, 4, 128, True, False), (2, 8, 2, 128, True, False), (1, 2, 2, 0, True, False),
                                                     (2, 4, 2, 128, True, True), (1, 4, 4, 0, True, True), (2, 4, 2, 64, True, False),
                                                     (1, 4, 4, 128, False, False), (2, 4, 4, 32, False, True)])
def test_shim_single_query_attention_reference_cache_layout(B, H, Hkv, rot, neox, alibi, D=128):
    """Positional call exactly as ftllama_modeling.py:139-153 makes it, caches in the reference's layouts
    (k_cache [B, Hkv, Dh/8, L, 8], v_cache [B, Hkv, L, Dh]); checked against a plain fp32 PyTorch decode -- the neox rotary
    of the Llama path, no rotary, a partial rotary_embedding_dim, the interleaved GPT-J style
    (decoder_masked_multihead_attention_utils.h: pairs (2i, 2i + 1), angle t / base^(2i / rot)) and ALiBi slopes
    (decoder_masked_multihead_attention_template.hpp:1335-1345: slope * (key - query position) on the scaled score)."""
    import math
    import qeft_cuda
    L, T = 64, 21
    torch.manual_seed(B * 10 + H)
    k_cache = torch.zeros(B, Hkv, D // 8, L, 8, dtype=torch.float16, device=DEV)
    v_cache = torch.zeros(B, Hkv, L, D, dtype=torch.float16, device=DEV)
    half = max(rot // 2, 1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, dtype=torch.float64) * 2.0 / max(rot, 1)))
    slopes = (2.0 ** (-8.0 * torch.arange(1, H + 1, dtype=torch.float32) / H)).to(DEV) if alibi else None

    def rope(x, p):       # [.., 128] fp32: the first `rot` dims rotated, neox pairs (i, i + rot/2) or GPT-J pairs (2i, 2i + 1)
        if rot == 0:
            return x
        ang = (p * inv).float().to(x.device)
        c, s = ang.cos(), ang.sin()
        o = x.clone()
        if neox:
            a, b = x[..., :half], x[..., half:rot]
            o[..., :half], o[..., half:rot] = a * c - b * s, b * c + a * s
        else:
            a, b = x[..., 0:rot:2], x[..., 1:rot:2]
            o[..., 0:rot:2], o[..., 1:rot:2] = a * c - b * s, b * c + a * s
        return o

    ks, vs = [], []
    for t in range(T):
        q = torch.randn(B, H, D, device=DEV).half()
        k = torch.randn(B, Hkv, D, device=DEV).half()
        v = torch.randn(B, Hkv, D, device=DEV).half()
        out = qeft_cuda.single_query_attention(q, k, v, k_cache, v_cache, None, slopes, t, rot, 10000.0, neox)
        torch.cuda.synchronize()
        assert out.shape == q.shape and out.dtype == torch.float16
        ks.append(rope(k.float(), t).half().float())
        vs.append(v.float())
        K, V = torch.stack(ks, 2), torch.stack(vs, 2)                          # [B, Hkv, t+1, D]
        K, V = K.repeat_interleave(H // Hkv, 1), V.repeat_interleave(H // Hkv, 1)
        att = torch.einsum("bhd,bhtd->bht", rope(q.float(), t), K) / math.sqrt(D)
        if alibi:
            att = att + slopes[None, :, None] * (torch.arange(t + 1, device=DEV, dtype=torch.float32) - t)[None, None, :]
        ref = torch.einsum("bht,bhtd->bhd", att.softmax(-1), V)
        assert (out.float() - ref).abs().max().item() < 2e-2 * max(ref.abs().max().item(), 1.0), t
    # the caches hold what the reference's would: rotated keys in the FT layout, values as they came
    Kc = k_cache.permute(0, 1, 3, 2, 4).reshape(B, Hkv, L, D)[:, :, :T].float()
    assert (Kc - torch.stack(ks, 2)).abs().max().item() < 2e-3 * 4
    assert torch.equal(v_cache[:, :, :T].float(), torch.stack(vs, 2))
    assert float(k_cache.permute(0, 1, 3, 2, 4).reshape(B, Hkv, L, D)[:, :, T:].abs().max()) == 0.0
    # unsupported arguments raise (documented in the shim)
    with pytest.raises(RuntimeError, match="alibi_slopes_"):
        qeft_cuda.single_query_attention(q, k, v, k_cache, v_cache, None, torch.zeros(H + 1, device=DEV), T, rot, 10000.0, neox)
    with pytest.raises(RuntimeError, match="rotary_embedding_dim"):
        qeft_cuda.single_query_attention(q, k, v, k_cache, v_cache, None, None, T, 63, 10000.0, neox)
    with pytest.raises(RuntimeError, match="timestep"):
        qeft_cuda.single_query_attention(q, k, v, k_cache, v_cache, None, None, L, rot, 10000.0, neox)
    # per-sample lengths instead of the common timestep (ft_attention.cpp:143-149)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    o1 = qeft_cuda.single_query_attention(q, k, v, k_cache.clone(), v_cache.clone(), lens, slopes, 0, rot, 10000.0, neox)
    o2 = qeft_cuda.single_query_attention(q, k, v, k_cache.clone(), v_cache.clone(), None, slopes, T, rot, 10000.0, neox)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


def test_multi_token_graphs_equal_single_token_steps():
    """DecodeEngine.run(n): graphs of 8 greedy tokens (token and position handed over on the device) give the same tokens,
    logits and cache as n calls of step(); a run that is not a multiple of 8 and one that starts mid-way included."""
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=2, hidden=256, inter=512, n_heads=2, vocab=512, max_seq=64)
    model = QuantLlama(shape, DEV, seed=5)
    a, b = DecodeEngine(model, use_graph=True), DecodeEngine(model, use_graph=True)
    for e in (a, b):
        e.greedy = True
        e.reset()
        e.tok.fill_(7)
    toks = []
    for _ in range(3 + 21):
        a.step()
        toks.append(int(a.tok.item()))
    for _ in range(3):
        b.step()
    b.run(21)                       # 8 + 8 + 5 single steps
    torch.cuda.synchronize()
    assert any(len(k) == 3 for k in b.graphs)                     # a multi-token graph was captured and used
    assert int(b.tok.item()) == toks[-1] and b.host_pos == a.host_pos == 24
    assert torch.equal(a.logits, b.logits) and torch.equal(a.pos, b.pos)
    for li in range(shape.n_layers):
        assert torch.equal(a.kc[li][:, :24], b.kc[li][:, :24]) and torch.equal(a.vc[li][:, :24], b.vc[li][:, :24])


@pytest.mark.parametrize("D,B,H,Hkv,rot,neox,alibi", [(64, 2, 4, 2, 64, True, False), (96, 1, 4, 4, 32, False, True), (32, 1, 8, 8, 0, True, False),
                                                       (256, 2, 2, 1, 256, True, False), (80, 1, 4, 2, 80, True, True), (192, 1, 2, 2, 64, True, False)])
def test_shim_single_query_attention_other_head_sizes(D, B, H, Hkv, rot, neox, alibi):
    """The head sizes the reference instantiates besides 128 (ft_attention.cpp:110-181: 32 .. 256) through the generic kernel
    (round 4): the same checks as the 128 case -- outputs vs a plain fp32 PyTorch decode over 21 steps, the caches' contents in
    the reference's layouts, per-sample lengths, the argument errors."""
    test_shim_single_query_attention_reference_cache_layout(B, H, Hkv, rot, neox, alibi, D=D)
    from qeft_amd import _lib
    assert _lib.last_variant() == "sqa_generic"
