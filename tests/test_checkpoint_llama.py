"""A WHOLE tiny Llama written by the reference's own `save_model` (tests/golden/make_golden_ckpt_llama.py: 2 layers, all seven
linears, norms, embedding, head, HF key names) loads through QuantLlama.from_packed -- the counterpart of load_owqmodel
(qeft/utils/modelutils.py:147-183) in front of the reference's decode benchmark (qeft/main.py:310-371).

CPU: keys / buffers / inferred shape, and the oracle's dequantisation of the loaded buffers reproduces the fixture's logits
(dense float64 forward over the fake-quantised weights, computed by the generator).  GPU: the decode engine, the prefill pass
and the module forwards on the loaded model against the same logits."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import qeft_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLDEN, "ref_ckpt_llama2l.pth")
LOGIT_TOL = 1e-2      # max |dlogit| / max |logit| vs the fake-quantised model: fp16 scale / scaled-zero rounding through 2 layers


def _io():
    return np.load(os.path.join(GOLDEN, "ref_ckpt_llama2l_io.npz"))


def test_from_packed_on_cpu_reads_the_reference_checkpoint():
    from qeft_amd.llama import QuantLlama
    from qeft_amd.qlinear import QuantLinear
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    model = QuantLlama.from_packed(CKPT, device="cpu", max_seq=64)
    s = model.shape
    assert (s.hidden, s.inter, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab, s.n_out, s.group_size, s.bits) == \
        (256, 384, 2, 2, 2, 96, 128, 128, 4)
    assert model.unexpected_keys == []
    sd = model.state_dict()
    for k, v in ck["model_state_dict"].items():          # every tensor of the checkpoint sits in the model under its own key, bit for bit
        assert k in sd, k
        assert sd[k].dtype == v.dtype and torch.equal(sd[k].cpu(), v), k
    extra = sorted(set(sd) - set(ck["model_state_dict"]))
    assert all(k.endswith("reorder_ids") for k in extra), extra      # registered by set_kernel() on the o_proj layers (qlinear.py:227-229)
    assert isinstance(model.model.layers[1].mlp.down_proj, QuantLinear)
    assert model.model.layers[0].self_attn.o_proj.forward.__name__ == "forward_outlier_out_proj"
    io = _io()
    for li in range(2):
        assert np.array_equal(model.model.layers[li].self_attn.o_proj.outlieridx.numpy(),
                              io[f"outids__model__layers__{li}__self_attn__o_proj"])


def _oracle_logits(model, tokens):
    """float64 forward with every packed linear dequantised by the ORACLE from the loaded buffers."""
    s = model.shape
    f = lambda t: t.detach().cpu().double()   # noqa: E731
    n = tokens.numel()
    h = f(model.model.embed_tokens.weight)[tokens]
    half = 64
    inv = 1.0 / (s.rope_theta ** (torch.arange(0, half, dtype=torch.float64) / half))
    ang = torch.arange(n, dtype=torch.float64)[:, None] * inv[None, :]
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]

    def rms(x, g):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + s.rms_eps) * f(g)

    def rope(x):
        a, b = x[..., :64], x[..., 64:]
        return torch.cat([a * cos - b * sin, b * cos + a * sin], -1)

    def w(ql):
        return torch.from_numpy(O.dequant_dense(ql.qweight.cpu().numpy(), ql.scales.cpu().numpy(), ql.scaled_zeros.cpu().numpy(),
                                                ql.oweight.cpu().numpy(), ql.group_size)).double()

    mask = torch.full((n, n), float("-inf"), dtype=torch.float64).triu(1)
    for L in model.model.layers:
        at, mlp = L.self_attn, L.mlp
        x = rms(h, L.input_layernorm.weight)
        q = rope((x @ w(at.q_proj).T).view(n, s.n_heads, 128))
        k = rope((x @ w(at.k_proj).T).view(n, s.n_kv_heads, 128))
        v = (x @ w(at.v_proj).T).view(n, s.n_kv_heads, 128)
        att = torch.einsum("thd,shd->hts", q, k) / math.sqrt(128) + mask
        a = torch.einsum("hts,shd->thd", att.softmax(-1), v).reshape(n, s.hidden)
        a = a[:, torch.from_numpy(O.sparse_to_dense_ids(at.o_proj.outlieridx.cpu().numpy(), s.hidden))]
        h = h + a @ w(at.o_proj).T
        x = rms(h, L.post_attention_layernorm.weight)
        h = h + (torch.nn.functional.silu(x @ w(mlp.gate_proj).T) * (x @ w(mlp.up_proj).T)) @ w(mlp.down_proj).T
    return rms(h, model.model.norm.weight) @ f(model.lm_head.weight).T


def test_loaded_buffers_reproduce_the_fixture_logits_through_the_oracle():
    from qeft_amd.llama import QuantLlama
    io = _io()
    model = QuantLlama.from_packed(CKPT, device="cpu", max_seq=64)
    tokens = torch.from_numpy(io["tokens"])
    got = _oracle_logits(model, tokens).numpy()
    ref = io["logits"]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < LOGIT_TOL, err
    assert (got.argmax(-1) == ref.argmax(-1)).mean() > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [True, False])
def test_engine_on_the_reference_checkpoint(use_graph):
    """The decode engine (the thing bench.py times) on the reference-written checkpoint, teacher-forced over the fixture's
    tokens: logits vs the fixture (fake-quantised float64 model) and vs the build's own dense fp32 model over the same
    buffers; the prefill pass (GEMM path) on the same model; QuantLinear.forward of a loaded layer through the v3 GEMV."""
    from qeft_amd import _lib
    from qeft_amd.llama import DecodeEngine, QuantLlama, prefill
    io = _io()
    model = QuantLlama.from_packed(CKPT, device="cuda:0", max_seq=64)
    tokens = torch.from_numpy(io["tokens"]).to("cuda:0")
    ref = torch.from_numpy(io["logits"]).to("cuda:0")
    scale = ref.abs().max().item()
    eng = DecodeEngine(model, use_graph=use_graph)
    got = eng.teacher_forced_logits(tokens).double()
    torch.cuda.synchronize()
    assert (got - ref).abs().max().item() / scale < LOGIT_TOL
    dense = model.forward_dense_reference(tokens).double()
    assert (got - dense).abs().max().item() / scale < 5e-3          # the same buffers, fp32 PyTorch: only fp16 activations differ
    top2 = ref.topk(2, dim=-1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL * scale
    assert torch.equal(got.argmax(-1)[sure], ref.argmax(-1)[sure])
    pre = prefill(model, tokens).double()
    torch.cuda.synchronize()
    assert (pre - ref[-pre.shape[0]:]).abs().max().item() / scale < LOGIT_TOL
    x = torch.randn(3, 256, device="cuda:0").half()
    y = model.model.layers[0].self_attn.o_proj(x)
    assert _lib.last_variant() == "gemv_v3_mb" and y.shape == (3, 256)


@pytest.mark.gpu
def test_replace_oweight_reaches_prefill_and_stops_a_stale_engine():
    """checkpoint.replace_oweight (fine-tuned delta, modelutils.py:185-198) re-derives what the kernels read: the prefill pass
    drops its cached fused operands (its output changes), and a DecodeEngine built before the call refuses to run."""
    from qeft_amd.checkpoint import replace_oweight
    from qeft_amd.llama import DecodeEngine, QuantLlama, prefill
    io = _io()
    model = QuantLlama.from_packed(CKPT, device="cuda:0", max_seq=64)
    tokens = torch.from_numpy(io["tokens"]).to("cuda:0")
    eng = DecodeEngine(model, use_graph=False)
    before = prefill(model, tokens).float()
    assert model._prefill_ops is not None
    name = "model.layers.1.mlp.up_proj"
    layer = model.model.layers[1].mlp.up_proj
    new_ow = (layer.oweight.float() + 0.05 * torch.randn_like(layer.oweight.float())).half()
    replace_oweight(model, {name: new_ow})
    after = prefill(model, tokens).float()
    torch.cuda.synchronize()
    assert (after - before).abs().max().item() > 1e-3
    # the same through the per-module path (no cache at all): the two agree
    ref = model.forward_dense_reference(tokens)
    assert (after - ref).abs().max().item() / ref.abs().max().item() < 5e-3
    with pytest.raises(RuntimeError, match="build a new DecodeEngine"):
        eng.step()
    eng2 = DecodeEngine(model, use_graph=False)
    got = eng2.teacher_forced_logits(tokens)
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 5e-3


def test_from_packed_head_split_is_explicit_or_refused(tmp_path):
    """hidden % 128 == 0 does not make head_dim 128.  An explicit n_heads (or a config.json beside the file) that gives
    another head size must raise instead of decoding with the wrong head split / rotary."""
    import json
    import shutil
    import warnings
    from qeft_amd.llama import QuantLlama
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    hidden = ck["model_state_dict"]["model.embed_tokens.weight"].shape[1]
    with pytest.raises(ValueError, match="head_dim"):
        QuantLlama.from_packed(CKPT, device="cpu", max_seq=64, n_heads=hidden // 64, n_kv_heads=hidden // 64)
    p = tmp_path / "m.pth"
    shutil.copy(CKPT, p)
    (tmp_path / "config.json").write_text(json.dumps({"num_attention_heads": hidden // 64, "num_key_value_heads": hidden // 64}))
    with pytest.raises(ValueError, match="head_dim"):
        QuantLlama.from_packed(str(p), device="cpu", max_seq=64)
    (tmp_path / "config.json").write_text(json.dumps({"num_attention_heads": hidden // 128, "rms_norm_eps": 1e-6, "rope_theta": 5e5}))
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # config present: nothing is assumed, nothing warned
        m = QuantLlama.from_packed(str(p), device="cpu", max_seq=64)
    assert m.shape.n_heads == hidden // 128 and m.shape.rms_eps == 1e-6 and m.shape.rope_theta == 5e5


def test_checkpoint_files_load_with_the_restricted_unpickler():
    """The format's only non-tensor objects are argparse.Namespace and a dtype: no arbitrary pickle code is needed (or run)."""
    from qeft_amd.checkpoint import load_checkpoint_file
    a = load_checkpoint_file(CKPT)
    b = load_checkpoint_file(CKPT, unsafe_pickle=True)
    assert a.keys() == b.keys() and a["quantinfos"].keys() == b["quantinfos"].keys()
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ("arbitrary code ran",))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "evil.pth")
        torch.save({"model_state_dict": {}, "x": Evil()}, path)
        with pytest.raises(pickle.UnpicklingError):
            load_checkpoint_file(path)
