"""Self-contained Llama-shaped decode harness around QuantLinear.

The reference measures decode speed by driving HF `LlamaForCausalLM` token by token (qeft/main.py:310-371,
qeft/benchmark.py:293-338).  HF model loading needs the hub, so the harness builds its own Llama-2-shaped module
tree (same attribute names as HF: model.layers[i].self_attn.{q,k,v,o}_proj, mlp.{gate,up,down}_proj) with
synthetic packed weights, and a `DecodeEngine` that runs one token as a fixed sequence of
C-ABI launches on static buffers so the whole step can be captured into one hipGraph:

    [q|k|v] grouped GEMV (input RMSNorm fused) -> rotary + KV append + attention -> o_proj GEMV (+gather, +residual)
            -> [gate|up] grouped GEMV (post-attention RMSNorm fused) -> down_proj GEMV (silu*mul fused, +residual)

`QuantLlama.forward_dense_reference` is a plain fp32 PyTorch implementation over the dense dequantised weights,
used by the tests and by `eval_nll` as the parity target ("PPL vs reference" on synthetic weights).
"""
import ctypes
import os
import math
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _lib, fuse, qeft_cuda
from .qlinear import QuantLinear
from .quant import fake_quantize, minmax_params


@dataclass
class LlamaShape:
    hidden: int
    inter: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab: int
    max_seq: int = 512
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    n_out: int = 128
    group_size: int = 128
    name: str = "llama"
    bits: int = 4          # 3: this build's 3-bit extension layout (qlinear.pack_w3)

    @property
    def head_dim(self):
        return self.hidden // self.n_heads


LLAMA2_7B = LlamaShape(4096, 11008, 32, 32, 32, 32000, name="llama-2-7b")
LLAMA2_13B = LlamaShape(5120, 13824, 40, 40, 40, 32000, name="llama-2-13b")
LLAMA2_70B = LlamaShape(8192, 28672, 80, 64, 8, 32000, name="llama-2-70b")      # (grouped-query attention; not a BASELINE config)


def tiny_shape(n_layers=2, hidden=256, inter=512, n_heads=2, vocab=512, max_seq=64, n_out=128):
    return LlamaShape(hidden, inter, n_layers, n_heads, n_heads, vocab, max_seq, n_out=n_out, name="tiny")


# ----------------------------------------------------------------------------------------------------
# synthetic packed layers: W ~ N(0, 0.02^2), per-group asymmetric min-max INT4, last n_out
# columns kept fp16
# ----------------------------------------------------------------------------------------------------
def synthetic_quantlinear(name, in_f, out_f, n_out, group_size, seed, device, outlieridx=None, fast_init=False,
                          bits=4):
    # CPU generator: identical weights on every device / rank for a given seed
    gdev = "cpu" if (torch.device(device).type == "cpu" or not fast_init) else device
    gen = torch.Generator(device=gdev).manual_seed(seed)
    w = (torch.randn(out_f, in_f, generator=gen, dtype=torch.float32, device=gdev) * 0.02).to(device).half()
    scale, zero = minmax_params(w, group_size, bits)
    wq = fake_quantize(w, scale, zero, group_size, bits).half()
    if n_out > 0:
        wq[:, in_f - n_out:] = w[:, in_f - n_out:]
    lin = nn.Linear(in_f, out_f, bias=False, dtype=torch.float16, device=device)
    lin.weight.data = wq
    ql = QuantLinear(bits, in_f, out_f, False, torch.float16, n_out, group_size, True, name).to(device)
    if outlieridx is None:
        outlieridx = torch.arange(in_f - n_out, in_f, dtype=torch.int32, device=device)
    ql.pack(lin, scale, zero, outlieridx.to(device))
    ql.set_kernel()
    return ql


class _Attn(nn.Module):
    pass


class _Mlp(nn.Module):
    pass


class _Layer(nn.Module):
    pass


class _Inner(nn.Module):
    pass


class _Norm(nn.Module):
    """RMSNorm weight holder with HF's key (`...layernorm.weight`, `model.norm.weight`); the arithmetic runs in the kernels."""

    def __init__(self, weight):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)


class QuantLlama(nn.Module):
    """Module tree with HF's attribute names; every decoder linear is a packed QuantLinear."""

    def __init__(self, shape: LlamaShape, device="cuda:0", seed=0, fast_init=False):
        """fast_init: draw the synthetic weights with the device generator (much faster for 7B/13B shapes)."""
        super().__init__()
        assert shape.head_dim == 128, "the decode attention kernel is built for head_dim 128"
        self.shape = shape
        s = shape
        gen = torch.Generator(device="cpu").manual_seed(seed)
        self.model = _Inner()
        self.model.embed_tokens = nn.Embedding(s.vocab, s.hidden, dtype=torch.float16, device=device)
        self.model.embed_tokens.weight.data = (torch.randn(s.vocab, s.hidden, generator=gen) * 0.5).half().to(device)
        self.model.norm = _Norm((1.0 + 0.1 * torch.randn(s.hidden, generator=gen)).half().to(device))
        self.lm_head = nn.Linear(s.hidden, s.vocab, bias=False, dtype=torch.float16, device=device)
        self.lm_head.weight.data = (torch.randn(s.vocab, s.hidden, generator=gen) * 0.02).half().to(device)
        layers = []
        kv = s.n_kv_heads * s.head_dim
        for li in range(s.n_layers):
            L = _Layer()
            L.self_attn = _Attn()
            L.mlp = _Mlp()
            pre = f"model.layers.{li}."
            sd = 1000 * li + seed * 7919
            L.self_attn.q_proj = synthetic_quantlinear(pre + "self_attn.q_proj", s.hidden, s.hidden, s.n_out, s.group_size, sd + 0, device, fast_init=fast_init, bits=s.bits)
            L.self_attn.k_proj = synthetic_quantlinear(pre + "self_attn.k_proj", s.hidden, kv, s.n_out, s.group_size, sd + 1, device, fast_init=fast_init, bits=s.bits)
            L.self_attn.v_proj = synthetic_quantlinear(pre + "self_attn.v_proj", s.hidden, kv, s.n_out, s.group_size, sd + 2, device, fast_init=fast_init, bits=s.bits)
            # o_proj has its OWN outlier columns (per layer, reorder.py:38-46) -> runtime gather of its input
            g2 = torch.Generator(device="cpu").manual_seed(sd + 99)
            oidx = torch.randperm(s.hidden, generator=g2)[:max(s.n_out, 1)].sort().values.to(torch.int32) if s.n_out else None
            L.self_attn.o_proj = synthetic_quantlinear(pre + "self_attn.o_proj", s.hidden, s.hidden, s.n_out, s.group_size, sd + 3, device, oidx, fast_init=fast_init, bits=s.bits)
            L.mlp.gate_proj = synthetic_quantlinear(pre + "mlp.gate_proj", s.hidden, s.inter, s.n_out, s.group_size, sd + 4, device, fast_init=fast_init, bits=s.bits)
            L.mlp.up_proj = synthetic_quantlinear(pre + "mlp.up_proj", s.hidden, s.inter, s.n_out, s.group_size, sd + 5, device, fast_init=fast_init, bits=s.bits)
            L.mlp.down_proj = synthetic_quantlinear(pre + "mlp.down_proj", s.inter, s.hidden, s.n_out, s.group_size, sd + 6, device, fast_init=fast_init, bits=s.bits)
            L.input_layernorm = _Norm((1.0 + 0.1 * torch.randn(s.hidden, generator=gen)).half().to(device))
            L.post_attention_layernorm = _Norm((1.0 + 0.1 * torch.randn(s.hidden, generator=gen)).half().to(device))
            layers.append(L)
        self.model.layers = nn.ModuleList(layers)
        for prm in self.parameters():
            prm.requires_grad_(False)
        half = s.head_dim // 2
        inv = 1.0 / (s.rope_theta ** (torch.arange(0, half, dtype=torch.float64) / half))
        ang = torch.arange(s.max_seq, dtype=torch.float64)[:, None] * inv[None, :]
        self.register_buffer("rope_cos", ang.cos().float().to(device), persistent=False)
        self.register_buffer("rope_sin", ang.sin().float().to(device), persistent=False)

    # ------------------------------------------------------------------ packed checkpoint -> model
    @classmethod
    def from_packed(cls, checkpoint_path, device="cuda:0", max_seq=512, rms_eps=1e-5, rope_theta=10000.0, name=None,
                    n_heads=None, n_kv_heads=None, unsafe_pickle=False):
        """A packed checkpoint in the reference's on-disk format (save_model, qeft/utils/modelutils.py:248-268) as a QuantLlama
        the DecodeEngine / prefill / eval_nll run on -- the counterpart of load_owqmodel (modelutils.py:147-183) feeding the
        reference's decode benchmark (qeft/main.py:310-371, 510-553).  The state dict carries HF's LlamaForCausalLM keys
        (`model.embed_tokens.weight`, `model.layers.i.self_attn.q_proj.qweight` ..., `model.layers.i.input_layernorm.weight`,
        `model.norm.weight`, `lm_head.weight`); the shape is read off the tensors (head_dim 128: Llama-2 7B / 13B; HF hub
        loading is out of scope, so rms_eps / rope_theta / max_seq are arguments; a `config.json` beside the checkpoint, or the
        n_heads / n_kv_heads arguments, say how the projections split into heads -- without either, head_dim 128 is assumed, with
        a warning outside the Llama-2 shapes; a split that is not head_dim 128 raises rather than decode with the wrong rotary).
        A fine-tuned delta
        (`{oweight_state_dict, base_path}`, save_wctmodel :270-284) loads its base first and replaces the outlier slices."""
        from .checkpoint import replace_oweight, load_checkpoint_file
        ckpt = load_checkpoint_file(checkpoint_path, unsafe_pickle)
        if "base_path" in ckpt:
            base = ckpt["base_path"]
            if not os.path.isabs(base) and not os.path.exists(base):
                base = os.path.join(os.path.dirname(os.path.abspath(checkpoint_path)), base)
            model = cls.from_packed(base, device, max_seq, rms_eps, rope_theta, name, n_heads, n_kv_heads, unsafe_pickle)
            replace_oweight(model, ckpt["oweight_state_dict"])
            return model
        assert ckpt.get("packing", False), "not a packed checkpoint"
        sd, infos = ckpt["model_state_dict"], ckpt["quantinfos"]
        vocab, hidden = sd["model.embed_tokens.weight"].shape
        n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("model.layers."))
        info0 = infos["model.layers.0.self_attn.q_proj"]
        kv = sd["model.layers.0.self_attn.k_proj.scales"].shape[1]
        inter = sd["model.layers.0.mlp.gate_proj.scales"].shape[1]
        # head split: explicit arguments > config.json beside the file (HF keys) > the known Llama-2 shapes
        cfg_path = os.path.join(os.path.dirname(os.path.abspath(checkpoint_path)), "config.json")
        if n_heads is None and os.path.exists(cfg_path):
            import json
            with open(cfg_path) as fh:
                cfg = json.load(fh)
            n_heads = int(cfg["num_attention_heads"])
            n_kv_heads = int(cfg.get("num_key_value_heads", n_heads))
            rms_eps = float(cfg.get("rms_norm_eps", rms_eps))
            rope_theta = float(cfg.get("rope_theta", rope_theta))
        if n_heads is None:
            if hidden % 128 or kv % 128:
                raise ValueError(f"cannot infer the head split of hidden={hidden}, kv={kv}: pass n_heads / n_kv_heads or put config.json beside the checkpoint")
            if (hidden, inter) not in ((4096, 11008), (5120, 13824), (8192, 28672)):
                import warnings
                warnings.warn(f"{checkpoint_path}: no config.json and no n_heads given; head_dim 128 assumed for hidden={hidden}")
            n_heads = hidden // 128
        if n_kv_heads is None:
            n_kv_heads = kv // (hidden // n_heads)
        if hidden % n_heads or hidden // n_heads != 128 or kv != n_kv_heads * 128:
            raise ValueError(f"head_dim {hidden // n_heads if hidden % n_heads == 0 else hidden / n_heads} (hidden {hidden}, "
                             f"{n_heads} heads, kv width {kv}): the decode attention kernel is built for head_dim 128 only")
        gs = info0.group_size if info0.group_size and info0.group_size > 0 else hidden
        shape = LlamaShape(hidden, inter, n_layers, n_heads, n_kv_heads, vocab, max_seq, rms_eps, rope_theta,
                           n_out=int(info0.n_out), group_size=int(gs), name=name or os.path.basename(checkpoint_path),
                           bits=int(info0.bits))
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.shape = shape
        self.model = _Inner()
        self.model.embed_tokens = nn.Embedding(vocab, hidden, dtype=torch.float16)
        self.model.norm = _Norm(torch.empty(hidden, dtype=torch.float16))
        self.lm_head = nn.Linear(hidden, vocab, bias=False, dtype=torch.float16)
        layers = []
        for li in range(n_layers):
            L = _Layer()
            L.self_attn, L.mlp = _Attn(), _Mlp()
            for grp, mod, names in (("self_attn", L.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")),
                                    ("mlp", L.mlp, ("gate_proj", "up_proj", "down_proj"))):
                for nm in names:
                    full = f"model.layers.{li}.{grp}.{nm}"
                    qi = infos[full]
                    n_rows = sd[full + ".scales"].shape[1]
                    k_cols = hidden if nm != "down_proj" else inter
                    g = qi.group_size if qi.group_size and qi.group_size > 0 else -1
                    setattr(mod, nm, QuantLinear(int(qi.bits), k_cols, n_rows, (full + ".bias") in sd, torch.float16, int(qi.n_out),
                                                 g, bool(getattr(qi, "reorder", True)), full))
            L.input_layernorm = _Norm(torch.empty(hidden, dtype=torch.float16))
            L.post_attention_layernorm = _Norm(torch.empty(hidden, dtype=torch.float16))
            layers.append(L)
        self.model.layers = nn.ModuleList(layers)
        # reorder_ids is registered by set_kernel(); a checkpoint saved after set_kernel() carries it already
        missing, unexpected = self.load_state_dict({k: v for k, v in sd.items() if not k.endswith("reorder_ids")}, strict=False)
        missing = [k for k in missing if not k.endswith(("rope_cos", "rope_sin"))]
        assert not missing, f"checkpoint lacks {missing[:4]} ..."
        self.unexpected_keys = list(unexpected)      # e.g. rotary inv_freq buffers of an HF export: not used here
        self.to(device)
        for L in self.model.layers:
            for mod in (L.self_attn.q_proj, L.self_attn.k_proj, L.self_attn.v_proj, L.self_attn.o_proj, L.mlp.gate_proj,
                        L.mlp.up_proj, L.mlp.down_proj):
                mod.set_kernel()
        for prm in self.parameters():
            prm.requires_grad_(False)
        half = shape.head_dim // 2
        inv = 1.0 / (rope_theta ** (torch.arange(0, half, dtype=torch.float64) / half))
        ang = torch.arange(max_seq, dtype=torch.float64)[:, None] * inv[None, :]
        self.register_buffer("rope_cos", ang.cos().float().to(device), persistent=False)
        self.register_buffer("rope_sin", ang.sin().float().to(device), persistent=False)
        return self

    # ------------------------------------------------------------------ dense fp32 reference
    @torch.no_grad()
    def dense_weights(self):
        """Dense dequantised fp32 weights per layer (through the HIP dequant kernel, itself bit-exact vs the oracle)."""
        from . import qeft_cuda
        out = []
        for L in self.model.layers:
            d = {}
            for grp, names in (("self_attn", ("q_proj", "k_proj", "v_proj", "o_proj")), ("mlp", ("gate_proj", "up_proj", "down_proj"))):
                for n in names:
                    ql = getattr(getattr(L, grp), n)
                    qw = ql._qweight4().clone() if ql.bits == 3 else ql.qweight
                    d[n] = qeft_cuda.dequantize_weight_4bit_qeft(qw, ql.scales, ql.scaled_zeros,
                                                                 ql.oweight if ql.outlierfeatures else None).float()
            out.append(d)
        return out

    @torch.no_grad()
    def forward_dense_reference(self, tokens, dense=None):
        """Plain PyTorch fp32 causal forward over the dense dequantised weights.  tokens [T] -> logits [T, vocab]."""
        s = self.shape
        dense = dense or self.dense_weights()
        T = tokens.numel()
        h = self.model.embed_tokens.weight[tokens].float()
        cos, sin = self.rope_cos[:T], self.rope_sin[:T]

        def rms(x, g):
            return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + s.rms_eps) * g.float()

        def rope(x):  # [T, H, 128]
            a, b = x[..., :64], x[..., 64:]
            c, sn = cos[:, None, :], sin[:, None, :]
            return torch.cat([a * c - b * sn, b * c + a * sn], dim=-1)

        mask = torch.full((T, T), float("-inf"), device=h.device).triu(1)
        for L, d in zip(self.model.layers, dense):
            x = rms(h, L.input_layernorm.weight)
            q = rope((x @ d["q_proj"].T).view(T, s.n_heads, 128))
            k = rope((x @ d["k_proj"].T).view(T, s.n_kv_heads, 128))
            v = (x @ d["v_proj"].T).view(T, s.n_kv_heads, 128)
            rep = s.n_heads // s.n_kv_heads
            k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
            att = torch.einsum("thd,shd->hts", q, k) / math.sqrt(128) + mask
            a = torch.einsum("hts,shd->thd", att.softmax(-1), v).reshape(T, s.hidden)
            a = a[:, L.self_attn.o_proj.reorder_ids] if hasattr(L.self_attn.o_proj, "reorder_ids") else a
            h = h + a @ d["o_proj"].T
            x = rms(h, L.post_attention_layernorm.weight)
            act = torch.nn.functional.silu(x @ d["gate_proj"].T) * (x @ d["up_proj"].T)
            h = h + act @ d["down_proj"].T
        return rms(h, self.model.norm.weight) @ self.lm_head.weight.float().T


def _rmsnorm(x, gamma, eps):
    """[T, H] fp16 -> fp16 through the C ABI (qeft_rmsnorm), current stream."""
    y = torch.empty_like(x)
    _lib.check(_lib.lib().qeft_rmsnorm(x.data_ptr(), None, gamma.data_ptr(), None, y.data_ptr(), x.shape[0], x.shape[1],
                                       eps, torch.cuda.current_stream(x.device).cuda_stream))
    return y


def _add_rmsnorm(x, add, gamma, eps):
    """h = x + add (fp16, rounded once like the torch add), y = rmsnorm(h) * gamma: one launch (qeft_rmsnorm's fused form).
    Returns (h, y)."""
    h, y = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.lib().qeft_rmsnorm(x.data_ptr(), add.data_ptr(), gamma.data_ptr(), h.data_ptr(), y.data_ptr(), x.shape[0],
                                       x.shape[1], eps, torch.cuda.current_stream(x.device).cuda_stream))
    return h, y


def _prefill_operands(model):
    """Per layer: q|k|v as one GEMM operand and gate|up interleaved in blocks of 64 rows (fuse.py), derived from the 4-bit
    modules on the first prompt and kept on the model (a second copy of those weights, as the decode engine keeps its own;
    `model._prefill_ops = None` drops them, e.g. after fine-tuned outlier weights were loaded)."""
    ops = getattr(model, "_prefill_ops", None)
    if ops is None:
        ops = []
        for L in model.model.layers:
            at, mlp, o = L.self_attn, L.mlp, {}
            qkv, gu = [at.q_proj, at.k_proj, at.v_proj], [mlp.gate_proj, mlp.up_proj]
            plain = lambda ls: all(getattr(l, "bits", 4) == 4 and not l.training and getattr(l, "fused", True) for l in ls)   # noqa: E731
            if plain(qkv) and all(l.outfeatures % 128 == 0 for l in qkv):
                o["qkv"] = fuse.concat_gemm_operand(qkv)
            if plain(gu) and mlp.gate_proj.outfeatures % 64 == 0:
                o["gu"] = fuse.pair64_gemm_operand(*gu)
            ops.append(o)
        model._prefill_ops = ops
    return ops


@torch.no_grad()
def prefill(model: "QuantLlama", tokens, engine=None):
    """Batched forward over T prompt tokens through the packed QuantLinears: T >= 8 rows take the MFMA GEMM path
    (fused outlier slice; BASELINE config 3), attention is torch's causal SDPA in fp16.  Returns fp16 logits
    [T, vocab].  With `engine` (a DecodeEngine of the same model) the rotated keys and the values are written into its
    KV caches and its position is set to T, so decoding continues from the prompt (main.py:340-371 feeds the prompt
    token by token; this is the batched equivalent)."""
    s = model.shape
    T = tokens.numel()
    assert T <= s.max_seq and (engine is None or engine.P == 1), "prefill hands over to a single-GPU engine"
    h = model.model.embed_tokens.weight[tokens]                       # [T, hidden] fp16
    cos, sin = model.rope_cos[:T, None, :], model.rope_sin[:T, None, :]

    cos_t, sin_t = model.rope_cos[:T].contiguous(), model.rope_sin[:T].contiguous()
    lib = _lib.lib()

    def rope(x, heads=None):                                          # [T, H, 128] fp16 -> rotated fp16 (fp32 math), in place
        if x.is_cuda and x.stride(-1) == 1 and x.stride(1) == 128:     # rows may be wider than their heads (fused q|k|v)
            _lib.check(lib.qeft_rope_rows(x.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), T, heads or x.shape[1],
                                          x.stride(0), torch.cuda.current_stream(x.device).cuda_stream))
            return x
        a, b = x[..., :64].float(), x[..., 64:].float()
        return torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1).half()

    fused = _prefill_operands(model) if T >= 8 and h.is_cuda else None
    hq, hkv = s.n_heads * 128, s.n_kv_heads * 128
    delta = None                                                      # the previous layer's down_proj output, added by the next norm
    for li, L in enumerate(model.model.layers):
        at, mlp = L.self_attn, L.mlp
        if delta is None:
            x = _rmsnorm(h, L.input_layernorm.weight, s.rms_eps)
        else:
            h, x = _add_rmsnorm(h, delta, L.input_layernorm.weight, s.rms_eps)
        fo = fused[li] if fused is not None else {}
        if "qkv" in fo:
            # q|k|v as one GEMM (N = 3 x 4096: 768 tiles of 256 x 128 = three whole rounds of the chip); q, k, v are views of
            # its output, rotary over the q and k heads of every row in one launch
            op = fo["qkv"]
            y = qeft_cuda.gemm_4bit_qeft(x, op.qweight, op.scales, op.scaled_zeros, op.oweight, op.bias)
            rope(y.view(T, s.n_heads + 2 * s.n_kv_heads, 128), s.n_heads + s.n_kv_heads)
            q, k, v = (y[:, a:b].view(T, -1, 128) for a, b in ((0, hq), (hq, hq + hkv), (hq + hkv, hq + 2 * hkv)))
        else:
            q = rope(at.q_proj(x).view(T, s.n_heads, 128))
            k = rope(at.k_proj(x).view(T, s.n_kv_heads, 128))
            v = at.v_proj(x).view(T, s.n_kv_heads, 128)
        if engine is not None:
            engine.kc[li][:, :T] = k.transpose(0, 1)
            engine.vc[li][:, :T] = v.transpose(0, 1)
        rep = s.n_heads // s.n_kv_heads
        kk, vv = (k, v) if rep == 1 else (k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1))
        # 4-D operands: torch routes an unbatched [H, T, 128] call to its math path (1.6 ms per layer at T = 2048 on this
        # build) and a batched one to the fused kernel (0.14 ms) -- tools/sdpa_probe.py
        a = torch.nn.functional.scaled_dot_product_attention(q.transpose(0, 1)[None], kk.transpose(0, 1)[None],
                                                             vv.transpose(0, 1)[None], is_causal=True)[0]    # [H, T, 128]
        a = a.transpose(0, 1).reshape(T, s.hidden).contiguous()
        h, x = _add_rmsnorm(h, at.o_proj(a), L.post_attention_layernorm.weight, s.rms_eps)     # o_proj gathers its own column order
        if "gu" in fo and qeft_cuda.gemm_gateup_supported(T, fo["gu"]):
            act = qeft_cuda.gemm_4bit_gateup(x, fo["gu"])             # gate|up as one GEMM, SiLU(gate) * up its epilogue
        else:
            act = mlp.up_proj.forward_silu_mul(x, mlp.gate_proj(x))   # SiLU(gate) * up in the up_proj GEMM's epilogue
        delta = mlp.down_proj(act)
    if engine is not None:
        engine.set_position(T)
    _, hn = _add_rmsnorm(h, delta, model.model.norm.weight, s.rms_eps)
    return torch.matmul(hn, model.lm_head.weight.t())


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class DecodeEngine:
    """One decode token = a fixed list of C-ABI launches on static buffers (hipGraph-capturable).

    With `tp_group` (world size P > 1) the quantized linears are sharded over the group in Megatron pairs: q|k|v and
    gate|up by output rows (a rank owns the rows of its own heads / its slice of the MLP width, so attention, the KV caches
    and SiLU stay local), o_proj and down_proj by input columns (fuse.column_shard), so that a row-sharded producer feeds
    its column-sharded consumer without a collective; the partial outputs (+ the fp32 residual on rank 0) are summed by ONE
    fp32 all-reduce of `hidden` floats after o_proj and one after down_proj: 2 collectives per layer (16 KB each for 7B;
    xGMI latency-bound).  Norms, embedding and lm_head are replicated.  Layers the v3 GEMV does not take (3-bit, odd
    shapes) keep the round-1 scheme: every linear row-sharded, 4 all-gathers per layer.
    """

    def __init__(self, model: QuantLlama, use_graph=True, tp_group=None, collective="rccl"):
        """collective (tensor-parallel only): "rccl" = torch.distributed's all_reduce on the group (RCCL ring over xGMI; gloo in the
        one-GPU rehearsals), "oneshot" = the hand-written single-kernel all-reduce over IPC-mapped mailboxes (qeft_amd/oneshot.py,
        SURVEY.md section 8e: the decode payload is 16 KB, latency-bound); `self.collective` says which one runs -- a request for
        "oneshot" that cannot be served raises; "auto" tries it, checks it against the group's own all_reduce on a probe vector and
        falls back to "rccl" on every rank alike, leaving the reason in `self.collective_note`."""
        import torch.distributed as dist
        from .sharded import shard_quantlinear
        self.m = model
        s = model.shape
        dev = model.lm_head.weight.device
        self.dev = dev
        self.lib = _lib.lib()
        # the engine streams COPIES of the outlier weights (fuse.py operands): checkpoint.replace_oweight bumps this counter
        # and step() / run() then refuse to run on the stale copies
        self._ow_version = getattr(model, "_oweight_version", 0)
        self.tp_group = tp_group
        self.bits = getattr(s, "bits", 4)
        # tp_group: a torch.distributed group (RCCL), or any object with .world, .rank and .all_gather(out, inp) -- the
        # tests drive two ranks of one process in lock step through such an object (tests/test_gpu_tp.py)
        self.sim_group = tp_group if hasattr(tp_group, "all_gather") or hasattr(tp_group, "all_reduce") else None
        self.n_collectives = 0              # collectives issued by the last eager _launch_token (tests: 2 per layer)
        if self.sim_group is not None:
            self.P, self.rank = tp_group.world, tp_group.rank
        else:
            self.P = dist.get_world_size(tp_group) if tp_group is not None else 1
            self.rank = dist.get_rank(tp_group) if tp_group is not None else 0
        self.tp = tp_group is not None      # a 1-rank group still runs the sharded launch sequence + collectives
        P, tp = self.P, self.tp
        f16 = dict(dtype=torch.float16, device=dev)
        self.tok = torch.zeros(1, dtype=torch.long, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.hbuf = [torch.zeros(s.hidden, **f16), torch.zeros(s.hidden, **f16)]
        self.xn = torch.zeros(s.hidden, **f16)
        kvd = s.n_kv_heads * s.head_dim
        self.hs, self.kvs, self.its = s.hidden // P, kvd // P, s.inter // P
        self.q = torch.zeros(s.hidden, **f16)
        self.k = torch.zeros(kvd, **f16)
        self.v = torch.zeros(kvd, **f16)
        self.att = torch.zeros(s.hidden, **f16)
        self.act = torch.zeros(s.inter, **f16)
        # local (per-rank) slices; with P == 1 they alias the full buffers
        # tensor-parallel: a rank owns the rows of q/k/v of ITS heads, so attention runs on local heads and local KV
        # caches; what is gathered is the attention output (and h, and silu(gate)*up)
        assert s.n_heads % P == 0 and s.n_kv_heads % P == 0, "heads must divide over the ranks"
        self.heads_l, self.kv_heads_l = s.n_heads // P, s.n_kv_heads // P
        self.qkv_loc = torch.zeros(self.hs + 2 * self.kvs, **f16) if tp else None
        self.att_loc = torch.zeros(self.hs, **f16) if tp else None
        self.att_g = torch.zeros(s.hidden, **f16) if tp else None       # w3 only: o_proj input gathered by torch
        self.h_loc = torch.zeros(self.hs, **f16) if tp else None
        self.gate_loc = torch.zeros(self.its, **f16)
        self.up_loc = torch.zeros(self.its, **f16)
        self.act_loc = torch.zeros(self.its, **f16) if tp else self.act
        self.hn = torch.zeros(1, s.hidden, **f16)
        self.logits = torch.zeros(1, s.vocab, **f16)
        self.kc = [torch.zeros(self.kv_heads_l, s.max_seq, s.head_dim, **f16) for _ in range(s.n_layers)]
        self.vc = [torch.zeros(self.kv_heads_l, s.max_seq, s.head_dim, **f16) for _ in range(s.n_layers)]
        # A head's context is dealt over attn_split blocks.  One CU pulls ~50 GB/s, so past a few hundred cached positions
        # one block per head is bound by its own fetch; below that the merge hand-off (~2.5 us) costs more than the split
        # saves.  Measured on Llama-2-7B: contexts 64..192 -> 653 / 641 / 638 tokens/s at split 1 / 2 / 4; 100..700 ->
        # 604 / 624 / 631 / 611 at 1 / 2 / 4 / 8; 500..2000 -> 435 vs 570 at 1 vs 4.  The split follows the position
        # (one captured graph per split); QEFT_ATTN_SPLIT pins it.
        self.rope_tab = torch.cat([model.rope_cos, model.rope_sin], 1).contiguous()   # [max_seq][cos 64 | sin 64]
        self.rope_row = torch.zeros(1, 128, dtype=torch.float32, device=dev)
        forced = os.environ.get("QEFT_ATTN_SPLIT")
        self.attn_split_forced = int(forced) if forced else None
        self.attn_split = self.attn_split_forced or 1
        nws = self.lib.qeft_attn_workspace_bytes(self.heads_l, 8)
        self.attn_ws = torch.zeros(nws // 4, dtype=torch.float32, device=dev)
        self.host_pos = 0          # host mirror of self.pos (chooses the split; any split is correct at any position)
        self.graphs = {}
        self.greedy = False
        self.graph = None
        self.use_graph = use_graph
        # v3 path (single GPU, 4 bits, the shapes gemv_v3.h takes): raw-x GEMVs with the norms / SiLU on the producers'
        # epilogues and an fp32 residual stream.  QEFT_ENGINE_V2=1 keeps the round-1 launch sequence (A/B timing).
        g_, k_ok = s.group_size, (s.hidden % 128 == 0 and s.inter % 128 == 0)
        self.v3 = (self.bits in (3, 4) and not tp and k_ok and s.n_out in (0, 128) and g_ == 128 and s.hidden % 16 == 0
                   and s.inter % 16 == 0 and kvd % 16 == 0 and os.environ.get("QEFT_ENGINE_V2") != "1")
        self.tp3 = (tp and self.bits == 4 and k_ok and s.n_out in (0, 128) and g_ == 128 and self.hs % 16 == 0
                    and self.kvs % 16 == 0 and self.its % 16 == 0 and self.its >= 128
                    and os.environ.get("QEFT_ENGINE_V2") != "1")
        self.collective = "rccl" if tp else None
        self.collective_note = None
        self.oneshot = None
        if self.tp3 and collective in ("oneshot", "auto") and self.sim_group is None:
            # "auto": the one-shot kernel if every rank can build it AND it reproduces the group's own all_reduce on a probe vector;
            # anything else falls back to the group's collective, and says so (collective_note)
            from .oneshot import OneShotAllReduce
            # Every rank issues the same collectives of the group in the same order whatever fails locally (a rank that left the
            # sequence early would leave its peers waiting in a collective it never joins).
            ok, why = True, None

            def failed(e):
                nonlocal ok, why
                if ok:
                    ok, why = False, f"{type(e).__name__}: {e}"[:200]

            try:
                if self.P > self.lib.qeft_oneshot_max_world():
                    raise ValueError(f"{self.P} ranks > {self.lib.qeft_oneshot_max_world()}")
                self.oneshot = OneShotAllReduce(s.hidden, dev, tp_group)        # (raises on every rank alike, or on none)
            except Exception as e:      # noqa: BLE001 -- any failure of the optional fast path means: use the group's collective
                failed(e)
            probe = torch.randn(s.hidden, generator=torch.Generator().manual_seed(1234 + self.rank)).to(dev)
            ref = probe.clone()
            on_cpu = dist.get_backend(tp_group) == "gloo"
            for _ in range(3):              # both mailbox parities
                if ok:
                    try:
                        self.oneshot.all_reduce(probe)
                    except Exception as e:  # noqa: BLE001
                        failed(e)
                r = ref.cpu() if on_cpu else ref
                dist.all_reduce(r, group=tp_group)
                ref = r.to(dev)
            if ok:
                try:
                    torch.cuda.synchronize(dev)
                    self.oneshot.check_status()
                    # (rank-order fp32 sum vs the ring's order: equal for two ranks, within rounding beyond)
                    if not torch.allclose(probe, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max())):
                        raise RuntimeError(f"probe mismatch: max |d| {(probe - ref).abs().max().item():.3e}")
                except Exception as e:      # noqa: BLE001
                    failed(e)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            flag = flag if dist.get_backend(tp_group) == "gloo" else flag.to(dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=tp_group)        # every rank takes the same path
            if int(flag.item()) == 1:
                self.collective = "oneshot"
            else:
                if self.oneshot is not None:
                    self.oneshot.close()
                self.oneshot = None
                self.collective_note = why or "another rank could not build the one-shot collective"
                if collective == "oneshot":
                    raise RuntimeError(f"collective='oneshot' was requested and is not available: {self.collective_note}")
        elif tp and collective == "oneshot":
            raise RuntimeError("collective='oneshot' needs the v3 tensor-parallel path and a real process group")
        if self.tp3:
            self.h32 = torch.zeros(s.hidden, dtype=torch.float32, device=dev)
            self.part32 = torch.zeros(s.hidden, dtype=torch.float32, device=dev)    # partial o_proj / down_proj output -> all-reduce
            self.zero32 = torch.zeros(s.hidden, dtype=torch.float32, device=dev)    # the "residual" of ranks > 0
            self.n_ssq_tb = self.lib.qeft_token_begin_norm_blocks(s.hidden)
            self.ssq = torch.zeros((self.n_ssq_tb + 3) // 4 * 4, dtype=torch.float32, device=dev)
            self.tp3ops = []
        if self.v3:
            self.h32 = torch.zeros(s.hidden, dtype=torch.float32, device=dev)
            self.n_ssq_tb = self.lib.qeft_token_begin_norm_blocks(s.hidden)
            self.n_ssq_lin = self.lib.qeft_decode_linear_blocks(s.hidden)
            self.ssq = torch.zeros((max(self.n_ssq_tb, self.n_ssq_lin) + 3) // 4 * 4, dtype=torch.float32, device=dev)
            self.qkv_out = torch.zeros(s.hidden + 2 * kvd, **f16)      # q | k | v of one launch; self.q / k / v are views
            self.q, self.k, self.v = self.qkv_out[:s.hidden], self.qkv_out[s.hidden:s.hidden + kvd], self.qkv_out[s.hidden + kvd:]
            self.v3ops = []
        # per-layer local linears + argument packs (host arrays of device pointers must stay alive)
        self.lin = []
        self.packs = []
        self.att_pos = []   # per layer: where attention output element i goes so that o_proj needs no gather
        for L in model.model.layers:
            a, mlp = L.self_attn, L.mlp
            names = dict(q=a.q_proj, k=a.k_proj, v=a.v_proj, o=a.o_proj, g=mlp.gate_proj, u=mlp.up_proj, d=mlp.down_proj)
            if self.tp3:
                self.tp3ops.append(self._build_tp3_layer(names))
                continue
            if tp:
                names = {kk: shard_quantlinear(vv, self.rank, P).to(dev) for kk, vv in names.items()}
            self.lin.append(names)
            o = names["o"]
            if hasattr(o, "reorder_ids"):
                inv = torch.empty_like(o.reorder_ids)
                inv[o.reorder_ids] = torch.arange(o.reorder_ids.numel(), device=inv.device)
                self.att_pos.append(inv.to(torch.int32).to(dev))
            else:
                self.att_pos.append(None)
            qkv = [names["q"], names["k"], names["v"]]
            gu = [names["g"], names["u"]]
            no = s.n_out
            if self.v3:
                # derived operands (qeft_amd/fuse.py): q|k|v concatenated, gate|up pair-interleaved for the SiLU epilogue
                from . import fuse
                self.v3ops.append(dict(qkv=fuse.concat_linears(qkv), o=fuse.single(names["o"]),
                                       gu=fuse.pair_interleave(*gu), d=fuse.single(names["d"])))
            if tp:
                qkv_y = [self.qkv_loc[:self.hs], self.qkv_loc[self.hs:self.hs + self.kvs], self.qkv_loc[self.hs + self.kvs:]]
            else:
                qkv_y = [self.q, self.k, self.v]
            self.packs.append(dict(
                qkv=(_ptr_array([l.qweight for l in qkv]), _ptr_array([l.scales for l in qkv]),
                     _ptr_array([l.scaled_zeros for l in qkv]),
                     _ptr_array([l.oweight_interleaved for l in qkv]) if no else None,
                     _ptr_array(qkv_y), (ctypes.c_int * 3)(self.hs, self.kvs, self.kvs),
                     _ptr_array([l._szp(l.scales) for l in qkv])),
                gu=(_ptr_array([l.qweight for l in gu]), _ptr_array([l.scales for l in gu]),
                    _ptr_array([l.scaled_zeros for l in gu]),
                    _ptr_array([l.oweight_interleaved for l in gu]) if no else None,
                    _ptr_array([self.gate_loc, self.up_loc]), (ctypes.c_int * 2)(self.its, self.its),
                    _ptr_array([l._szp(l.scales) for l in gu])),
            ))

    @property
    def h(self):
        return self.hbuf[0]

    def _build_tp3_layer(self, names):
        """Operands of one decoder layer for this rank (Megatron pairing, fuse.py): row shards of q|k|v and gate|up, column
        shards of o_proj and down_proj with their zero-initialised x vectors (only owned positions are ever written)."""
        from . import fuse
        from .sharded import shard_quantlinear
        s, P, rk, dev = self.m.shape, self.P, self.rank, self.dev
        row = {kk: shard_quantlinear(names[kk], rk, P).to(dev) for kk in ("q", "k", "v", "g", "u")}
        o, d = names["o"], names["d"]
        if hasattr(o, "reorder_ids"):                   # natural index -> o_proj's column (outliers last)
            inv = torch.empty_like(o.reorder_ids)
            inv[o.reorder_ids] = torch.arange(o.reorder_ids.numel(), device=inv.device)
        else:
            inv = torch.arange(s.hidden, device=dev)
        o_op, att_pos = fuse.column_shard(o, inv[rk * self.hs:(rk + 1) * self.hs])
        d_op, d_pos = fuse.column_shard(d, torch.arange(rk * self.its, (rk + 1) * self.its, device=dev))
        lead = int(d_pos[0])
        assert torch.equal(d_pos.long(), lead + torch.arange(self.its, device=d_pos.device)), \
            "down_proj's owned columns must be one run of its x vector (outlier columns last, owned by the last rank)"
        f16 = dict(dtype=torch.float16, device=dev)
        return dict(qkv=fuse.concat_linears([row["q"], row["k"], row["v"]]), o=o_op, gu=fuse.pair_interleave(row["g"], row["u"]),
                    d=d_op, x_o=torch.zeros(o_op.infeatures, **f16), att_pos=att_pos.to(dev).contiguous(),
                    x_d=torch.zeros(d_op.infeatures, **f16), lead_d=lead)

    def _all_reduce(self, t):
        """In-place sum over the tensor-parallel group (fp32 partial outputs; RCCL ring over xGMI, or the tests' stand-in)."""
        self.n_collectives += 1
        if self.oneshot is not None:
            self.oneshot.all_reduce(t)
        elif self.sim_group is not None:
            self.sim_group.all_reduce(t)
        else:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.tp_group)

    def _all_gather(self, out, inp):
        self.n_collectives += 1
        if self.sim_group is not None:
            self.sim_group.all_gather(out, inp)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(out, inp, group=self.tp_group)

    def _split_for(self, pos):
        if self.attn_split_forced:
            return self.attn_split_forced
        return 1 if pos < 256 else (4 if pos < 1536 else 8)

    def reset(self):
        self.set_position(0)

    def verify_collective(self, when="a run"):
        """Tensor-parallel engines with the one-shot collective: True if no rank's status word holds a give-up code.  Otherwise the
        collective is dropped on EVERY rank alike (one MAX all-reduce of the group says so), the captured graphs are discarded and
        the group's own all-reduce serves from here (`collective` becomes "rccl", the reason goes to `collective_note`): results
        produced since the last check are not to be trusted, the caller re-runs them.  Synchronises; call it at token boundaries."""
        if self.oneshot is None:
            return True
        import torch.distributed as dist
        torch.cuda.synchronize(self.dev)
        word = int(self.oneshot.status[0].item())
        on_cpu = dist.get_backend(self.tp_group) == "gloo"
        flag = torch.tensor([word], dtype=torch.int64, device="cpu" if on_cpu else self.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.tp_group)
        code = int(flag.item())
        if not code:
            return True
        self.oneshot.close()
        self.oneshot, self.collective = None, "rccl"
        self.collective_note = (f"the one-shot collective gave up during {when} (largest status word over the ranks {code:#x}): "
                                "the group's all-reduce from there")
        self.graph, self.graphs = None, {}
        return False

    def set_position(self, t):
        """The next token to be fed sits at position t (the KV caches hold positions < t)."""
        if not 0 <= int(t) <= self.m.shape.max_seq:
            raise ValueError(f"position {t} outside the KV cache (max_seq = {self.m.shape.max_seq})")
        self.pos.fill_(t)
        self.host_pos = int(t)

    def gemv_kernel_name(self):
        """The kernel(s) behind the quantized linears of a token (for bench.py's roofline record)."""
        if self.bits == 3 and not self.v3:
            return "qeft::gemv_w4_mfma_group_kernel / gemv_w4_mfma_kernel <BITS = 3>"
        if self.v3 or self.tp3:
            return "qeft::gemv_v3_kernel"
        return "qeft::gemv_w4_mfma_group_kernel / gemv_w4_mfma_kernel"

    def weight_bytes_per_token(self):
        """Algorithmic HBM bytes of the quantized linears one token streams on THIS rank (SURVEY.md §8d formula)."""
        tot = 0
        if self.tp3:        # the operands as sharded: whole boundary groups and the full outlier slice count (they are streamed)
            for ops in self.tp3ops:
                for op in (ops["qkv"], ops["o"], ops["gu"], ops["d"]):
                    n, k, r, g = op.outfeatures, op.infeatures, op.outlierfeatures, op.group_size
                    tot += n * (k - r) // 2 + 2 * (k // g) * n * 2 + n * r * 2 + 2 * k + 2 * n
            return tot
        for names in self.lin:
            for l in names.values():
                n, k, r, g = l.outfeatures, l.infeatures, l.outlierfeatures, l.group_size
                tot += n * (k - r) * l.bits // 8 + 2 * (k // g) * n * 2 + n * r * 2 + 2 * k + 2 * n
        return tot

    # -- the launch sequence ---------------------------------------------------------------------------
    @torch.no_grad()
    def _launch_token(self, linears_only=False, only=None):
        """only (with linears_only): launch just one GEMV of every layer -- "qkv", "o", "gu" or "d" -- for per-kernel timing."""
        import torch.distributed as dist
        self.n_collectives = 0
        if self.v3:
            return self._launch_token_v3(linears_only, only)
        if self.tp3:
            return self._launch_token_tp3(linears_only, only)
        s, lib, ck, P, tp = self.m.shape, self.lib, _lib.check, self.P, self.tp
        w3 = self.bits == 3
        gemv_group = lib.qeft_gemv_w3_group if w3 else lib.qeft_gemv_w4_group
        gemv_silu = lib.qeft_gemv_w3_silu if w3 else lib.qeft_gemv_w4_silu

        def gemv_fused(x, ql, ow, residual, szp, y, n, k, ids=None):
            """one linear, batch 1, + residual.  ids: o_proj's input gather (tensor-parallel only; on one GPU the
            attention kernel already writes o_proj's column order)"""
            if w3:
                return lib.qeft_gemv_w3(x, ql.qweight.data_ptr(), ql.scales.data_ptr(), ql.scaled_zeros.data_ptr(), ow,
                                        None, residual, szp, y, 1, n, k, g, no, st)
            return lib.qeft_gemv_w4_fused(x, ql.qweight.data_ptr(), ql.scales.data_ptr(), ql.scaled_zeros.data_ptr(), ow,
                                          None, ids, residual, szp, y, 1, n, k, g, no, st)

        def pick(tag, fn):      # per-kernel timing (bench.py): keep one GEMV of the layer, drop the others
            return fn if only in (None, tag) else (lambda *a: 0)
        group_qkv, group_gu = pick("qkv", gemv_group), pick("gu", gemv_group)
        fused_o, fused_d, silu_d = pick("o", gemv_fused), pick("d", gemv_fused), pick("d", gemv_silu)

        st = torch.cuda.current_stream(self.dev).cuda_stream
        h, h2 = self.hbuf
        if not linears_only:
            # h = embed[tok] and this position's rotary row, selected once per token (no attention launch waits for
            # pos before its rotary)
            ck(lib.qeft_token_begin(self.m.model.embed_tokens.weight.data_ptr(), self.tok.data_ptr(),
                                    self.rope_tab.data_ptr(), self.pos.data_ptr(), h.data_ptr(), self.rope_row.data_ptr(),
                                    s.hidden, s.vocab, s.max_seq, st))
        g, no = s.group_size, s.n_out
        r0 = self.rank * self.hs
        for li, L in enumerate(self.m.model.layers):
            lin, pk = self.lin[li], self.packs[li]
            # input_layernorm is fused into the q|k|v launch (x is normalised while it is staged)
            qw, sc, sz, ow, ys, ns, szp = pk["qkv"]
            ck(group_qkv(h.data_ptr(), L.input_layernorm.weight.data_ptr(), s.rms_eps, 3, qw, sc, sz, ow, None, szp, ys, ns,
                         s.hidden, g, no, st))
            if not linears_only:
                if tp:      # local heads only: q/k/v slices straight from the grouped GEMV, natural output order
                    qp = self.qkv_loc.data_ptr()
                    ck(lib.qeft_rope_attn_decode(qp, qp + self.hs * 2, qp + (self.hs + self.kvs) * 2,
                                                 self.rope_row.data_ptr(), self.rope_row.data_ptr() + 64 * 4, 1,
                                                 self.kc[li].data_ptr(), self.vc[li].data_ptr(), self.pos.data_ptr(), None,
                                                 self.att_loc.data_ptr(),
                                                 self.attn_ws.data_ptr() if self.attn_ws is not None else None,
                                                 self.attn_split, self.heads_l, self.kv_heads_l, s.max_seq, st))
                else:
                    ck(lib.qeft_rope_attn_decode(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(),
                                                 self.rope_row.data_ptr(), self.rope_row.data_ptr() + 64 * 4, 1,
                                                 self.kc[li].data_ptr(), self.vc[li].data_ptr(), self.pos.data_ptr(),
                                                 self.att_pos[li].data_ptr() if self.att_pos[li] is not None else None,
                                                 self.att.data_ptr(),
                                                 self.attn_ws.data_ptr() if self.attn_ws is not None else None,
                                                 self.attn_split, s.n_heads, s.n_kv_heads, s.max_seq, st))
            if tp:
                self._all_gather(self.att, self.att_loc)      # natural head order (ranks own consecutive heads)
            o = lin["o"]
            ow_o = o.oweight_interleaved.data_ptr() if no else None
            szp_o = o._szp(o.scales)
            szp_o = szp_o.data_ptr() if szp_o is not None else None
            if tp:
                ids32 = getattr(o, "reorder_ids32", None)
                if ids32 is not None and w3:                   # the 3-bit GEMV has no gather: torch does it
                    torch.index_select(self.att, 0, o.reorder_ids, out=self.att_g)
                    ids32 = None
                    xin = self.att_g
                else:
                    xin = self.att
                ck(fused_o(xin.data_ptr(), o, ow_o, h[r0:r0 + self.hs].data_ptr(), szp_o, self.h_loc.data_ptr(),
                           self.hs, s.hidden, ids32.data_ptr() if ids32 is not None else None))
                self._all_gather(h2, self.h_loc)
                h, h2 = h2, h
            else:
                ck(fused_o(self.att.data_ptr(), o, ow_o, h.data_ptr(), szp_o, h.data_ptr(), s.hidden, s.hidden))
            qw, sc, sz, ow, ys, ns, szp = pk["gu"]
            ck(group_gu(h.data_ptr(), L.post_attention_layernorm.weight.data_ptr(), s.rms_eps, 2, qw, sc, sz, ow, None, szp,
                        ys, ns, s.hidden, g, no, st))
            if tp:
                if not linears_only:
                    ck(lib.qeft_silu_mul(self.gate_loc.data_ptr(), self.up_loc.data_ptr(), self.act_loc.data_ptr(),
                                         self.its, st))
                self._all_gather(self.act, self.act_loc)
            d = lin["d"]
            ow_d = d.oweight_interleaved.data_ptr() if no else None
            szp_d = d._szp(d.scales)
            szp_d = szp_d.data_ptr() if szp_d is not None else None
            if tp:
                ck(fused_d(self.act.data_ptr(), d, ow_d, h[r0:r0 + self.hs].data_ptr(), szp_d, self.h_loc.data_ptr(),
                           self.hs, s.inter))
                self._all_gather(h2, self.h_loc)
                h, h2 = h2, h
            else:
                # silu(gate) * up is formed while down_proj stages its input
                ck(silu_d(self.gate_loc.data_ptr(), self.up_loc.data_ptr(), d.qweight.data_ptr(),
                          d.scales.data_ptr(), d.scaled_zeros.data_ptr(), ow_d, None, h.data_ptr(),
                          szp_d, h.data_ptr(), s.hidden, s.inter, g, no, st))
        if linears_only:
            return
        # an even number of buffer swaps per token: the result is back in hbuf[0]
        ck(lib.qeft_rmsnorm(h.data_ptr(), None, self.m.model.norm.weight.data_ptr(), None, self.hn.data_ptr(), 1,
                            s.hidden, s.rms_eps, st))
        torch.matmul(self.hn, self.m.lm_head.weight.t(), out=self.logits)
        ck(lib.qeft_token_end(self.logits.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(), s.vocab,
                              1 if self.greedy else 0, st))

    def _token_tail(self, h32, st):
        """final RMSNorm + fp16 lm_head (one launch where the head's width is one the fused kernel takes) + token end"""
        s, lib, ck = self.m.shape, self.lib, _lib.check
        w = self.m.lm_head.weight
        if s.hidden in (512, 1024, 2048, 4096, 5120, 8192) and w.dtype == torch.float16 and w.is_contiguous() \
                and os.environ.get("QEFT_LM_HEAD_TORCH") != "1":
            ck(lib.qeft_lm_head_f16(h32, self.m.model.norm.weight.data_ptr(), w.data_ptr(), self.logits.data_ptr(), s.hidden, s.vocab,
                                    s.rms_eps, st))
        else:
            ck(lib.qeft_rmsnorm_f32(h32, self.m.model.norm.weight.data_ptr(), self.hn.data_ptr(), 1, s.hidden, s.rms_eps, st))
            torch.matmul(self.hn, w.t(), out=self.logits)
        ck(lib.qeft_token_end(self.logits.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(), s.vocab,
                              1 if self.greedy else 0, st))

    @torch.no_grad()
    def _launch_token_v3(self, linears_only=False, only=None):
        """One token on the v3 GEMV (gemv_v3.h): every quantized linear reads its fp16 input vector as it is; the RMSNorms
        are split into (h * gamma, partial sums of h^2) on the producer's epilogue and a deferred 1/rms on the consumer's;
        SiLU(gate) * up is formed in the gate|up launch; the residual stream h32 stays fp32."""
        s, lib, ck = self.m.shape, self.lib, _lib.check
        st = torch.cuda.current_stream(self.dev).cuda_stream
        g, no, eps = s.group_size, s.n_out, s.rms_eps
        layers = self.m.model.layers
        xn, ssq, h32 = self.xn.data_ptr(), self.ssq.data_ptr(), self.h32.data_ptr()

        entry = lib.qeft_decode_linear_w3 if self.bits == 3 else lib.qeft_decode_linear

        def lin(op, x, y, mode=0, residual=None, ssq_in=None, n_ssq=0, gamma_out=None):
            return entry(x, op.qweight.data_ptr(), op.sz_packed.data_ptr(),
                                          op.oweight.data_ptr() if no else None, None, y, op.outfeatures, op.infeatures, g, no,
                                          mode, residual, ssq_in, n_ssq, eps, gamma_out, xn if gamma_out else None,
                                          ssq if gamma_out else None, st)

        def pick(tag):
            return lin if only in (None, tag) else (lambda *a, **kw: 0)
        if not linears_only:
            ck(lib.qeft_token_begin_norm(self.m.model.embed_tokens.weight.data_ptr(), self.tok.data_ptr(),
                                         self.rope_tab.data_ptr(), self.pos.data_ptr(), h32, self.rope_row.data_ptr(),
                                         layers[0].input_layernorm.weight.data_ptr(), xn, ssq, s.hidden, s.vocab, s.max_seq, st))
        n_ssq = self.n_ssq_tb
        for li, L in enumerate(layers):
            pk = self.v3ops[li]
            ck(pick("qkv")(pk["qkv"], xn, self.qkv_out.data_ptr(), ssq_in=ssq, n_ssq=n_ssq))
            if not linears_only:
                ck(lib.qeft_rope_attn_decode(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(),
                                             self.rope_row.data_ptr(), self.rope_row.data_ptr() + 64 * 4, 1,
                                             self.kc[li].data_ptr(), self.vc[li].data_ptr(), self.pos.data_ptr(),
                                             self.att_pos[li].data_ptr() if self.att_pos[li] is not None else None,
                                             self.att.data_ptr(),
                                             self.attn_ws.data_ptr() if self.attn_ws is not None else None,
                                             self.attn_split, s.n_heads, s.n_kv_heads, s.max_seq, st))
            ck(pick("o")(pk["o"], self.att.data_ptr(), h32, residual=h32, gamma_out=L.post_attention_layernorm.weight.data_ptr()))
            n_ssq = self.n_ssq_lin
            ck(pick("gu")(pk["gu"], xn, self.act.data_ptr(), mode=1, ssq_in=ssq, n_ssq=n_ssq))
            nxt = layers[li + 1].input_layernorm.weight.data_ptr() if li + 1 < len(layers) else None
            ck(pick("d")(pk["d"], self.act.data_ptr(), h32, residual=h32, gamma_out=nxt))
        if linears_only:
            return
        self._token_tail(h32, st)

    @torch.no_grad()
    def _launch_token_tp3(self, linears_only=False, only=None):
        """One token of one tensor-parallel rank on the v3 GEMV: q|k|v (RMSNorm of the all-reduced h inside the launch) ->
        attention (local heads, output scattered into o_proj's x vector) -> o_proj partial (+ h on rank 0) -> all-reduce ->
        gate|up (norm inside, SiLU epilogue, output straight into down_proj's x vector) -> down_proj partial (+ h on rank 0)
        -> all-reduce: 5 launches and 2 collectives per layer.  The fp32 residual stream alternates between two buffers
        (a partial-output launch reads h from one and writes into the other, which the all-reduce then completes)."""
        s, lib, ck = self.m.shape, self.lib, _lib.check
        st = torch.cuda.current_stream(self.dev).cuda_stream
        g, no, eps = s.group_size, s.n_out, s.rms_eps
        layers = self.m.model.layers
        cur, oth = self.h32, self.part32
        zero = self.zero32.data_ptr()

        def hnorm(op, h, gamma, y, mode=0):
            return lib.qeft_decode_linear_hnorm(h.data_ptr(), gamma.data_ptr(), op.qweight.data_ptr(), op.sz_packed.data_ptr(),
                                                op.oweight.data_ptr() if no else None, None, y, op.outfeatures, op.infeatures, g,
                                                no, mode, eps, st)

        def partial(op, x, h, y):           # y (fp32) = op . x + (rank 0: h)
            return lib.qeft_decode_linear(x, op.qweight.data_ptr(), op.sz_packed.data_ptr(), op.oweight.data_ptr() if no else None,
                                          None, y.data_ptr(), op.outfeatures, op.infeatures, g, no, 0,
                                          h.data_ptr() if self.rank == 0 else zero, None, 0, eps, None, None, None, st)

        def pick(tag, fn):
            return fn if only in (None, tag) else (lambda *a, **kw: 0)
        if not linears_only:
            ck(lib.qeft_token_begin_norm(self.m.model.embed_tokens.weight.data_ptr(), self.tok.data_ptr(),
                                         self.rope_tab.data_ptr(), self.pos.data_ptr(), cur.data_ptr(), self.rope_row.data_ptr(),
                                         layers[0].input_layernorm.weight.data_ptr(), self.xn.data_ptr(), self.ssq.data_ptr(), s.hidden,
                                         s.vocab, s.max_seq, st))
        qp = self.qkv_loc.data_ptr()
        for li, L in enumerate(layers):
            pk = self.tp3ops[li]
            ck(pick("qkv", hnorm)(pk["qkv"], cur, L.input_layernorm.weight, qp))
            if not linears_only:
                ck(lib.qeft_rope_attn_decode(qp, qp + self.hs * 2, qp + (self.hs + self.kvs) * 2,
                                             self.rope_row.data_ptr(), self.rope_row.data_ptr() + 64 * 4, 1,
                                             self.kc[li].data_ptr(), self.vc[li].data_ptr(), self.pos.data_ptr(),
                                             pk["att_pos"].data_ptr(), pk["x_o"].data_ptr(),
                                             self.attn_ws.data_ptr() if self.attn_ws is not None else None,
                                             self.attn_split, self.heads_l, self.kv_heads_l, s.max_seq, st))
            ck(pick("o", partial)(pk["o"], pk["x_o"].data_ptr(), cur, oth))
            if not linears_only:
                self._all_reduce(oth)
            cur, oth = oth, cur
            ck(pick("gu", hnorm)(pk["gu"], cur, L.post_attention_layernorm.weight, pk["x_d"].data_ptr() + pk["lead_d"] * 2, mode=1))
            ck(pick("d", partial)(pk["d"], pk["x_d"].data_ptr(), cur, oth))
            if not linears_only:
                self._all_reduce(oth)
            cur, oth = oth, cur
        if linears_only:
            return
        self._token_tail(cur.data_ptr(), st)         # an even number of swaps: cur is self.h32 again

    def capture(self, linears_only=False, only=None, split=None):
        """Capture one token into a hipGraph (after a warm-up launch on a side stream, as torch requires).  `split`:
        attention blocks per head baked into this graph (default: what the current position asks for)."""
        self.attn_split = split or self._split_for(self.host_pos)
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        pos0, tok0 = self.pos.clone(), self.tok.clone()
        with torch.cuda.stream(side):
            self._launch_token(linears_only, only)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.pos.copy_(pos0)
        self.tok.copy_(tok0)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._launch_token(linears_only, only)
        self.pos.copy_(pos0)
        self.tok.copy_(tok0)
        if linears_only:
            return graph
        # a graph bakes in the attention split AND whether token_end writes the greedy token
        self.graph = self.graphs[(self.attn_split, bool(self.greedy))] = graph
        return graph

    def precapture(self, max_pos):
        """Capture the graphs of every attention split the positions [host_pos, max_pos) will use (so that none is
        captured inside a timed region)."""
        for sp in sorted({self._split_for(p) for p in (self.host_pos, 255, 256, 1535, 1536, max_pos - 1)
                          if self.host_pos <= p < max_pos}):
            if (sp, bool(self.greedy)) not in self.graphs:
                self.capture(split=sp)

    def _check_fresh(self):
        if getattr(self.m, "_oweight_version", 0) != self._ow_version:
            raise RuntimeError("the model's outlier weights were replaced (checkpoint.replace_oweight) after this DecodeEngine "
                               "derived its operands from them: build a new DecodeEngine")

    def step(self):
        """Run one token: consumes self.tok at position self.pos, leaves logits (and, if greedy, the next token)."""
        if self.host_pos >= self.m.shape.max_seq:
            # the device side would skip the attention (stale output) and keep counting: refuse instead
            raise RuntimeError(f"KV cache full: position {self.host_pos} >= max_seq {self.m.shape.max_seq}")
        self._check_fresh()
        sp = self._split_for(self.host_pos)
        if self.use_graph:
            g = self.graphs.get((sp, bool(self.greedy)))
            if g is None:
                g = self.capture(split=sp)
            g.replay()
        else:
            self.attn_split = sp
            self._launch_token()
        self.host_pos += 1

    MULTI = int(os.environ.get("QEFT_MULTI_TOKENS", "8"))       # tokens per multi-token graph (greedy decoding only; 16 / 32 measured in round 4)

    def run(self, n_tokens):
        """Greedy decoding of n_tokens tokens (self.greedy must be set): like n_tokens calls of step(), but with hipGraphs of
        MULTI tokens where the positions allow it -- token_end writes the next token and position on the device, so
        consecutive tokens need nothing from the host, and one replay per 8 tokens saves 7 of 8 inter-replay gaps (about
        1 % of a 7B token).  logits hold the last token's values."""
        assert self.greedy, "run() feeds every token's argmax to the next: greedy decoding only"
        self._check_fresh()
        multi_ok = self.use_graph and os.environ.get("QEFT_MULTI_TOKEN_GRAPH") != "0"
        while n_tokens > 0:
            p, m = self.host_pos, self.MULTI
            sp = self._split_for(p)
            if multi_ok and n_tokens >= m and p + m <= self.m.shape.max_seq and self._split_for(p + m - 1) == sp:
                key = (sp, True, m)
                g = self.graphs.get(key)
                if g is None:
                    g = self.graphs[key] = self._capture_multi(m, sp)
                g.replay()
                self.host_pos += m
                n_tokens -= m
            else:
                self.step()
                n_tokens -= 1

    def _capture_multi(self, m, split):
        self.attn_split = split
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        pos0, tok0 = self.pos.clone(), self.tok.clone()
        with torch.cuda.stream(side):
            self._launch_token()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.pos.copy_(pos0)
        self.tok.copy_(tok0)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(m):
                self._launch_token()
        # the capture itself executes nothing, but the warm-up launch above wrote one token's K/V at pos0 (rewritten by the
        # real run) and advanced pos / tok: restore them
        self.pos.copy_(pos0)
        self.tok.copy_(tok0)
        return graph

    @torch.no_grad()
    def teacher_forced_logits(self, tokens):
        """Feed `tokens` one by one from position 0; returns fp32 logits [T, vocab] (main.py:340-371 protocol)."""
        self.reset()
        self.greedy = False
        outs = []
        for t in tokens.tolist():
            self.tok.fill_(t)
            self.step()
            outs.append(self.logits[0].float().clone())
        return torch.stack(outs)


def nll_from_logits(logits, tokens):
    """Mean next-token NLL of a teacher-forced run (main.py:291-305 / :369-371)."""
    return torch.nn.functional.cross_entropy(logits[:-1].float(), tokens[1:].to(logits.device)).item()
