"""Generate a WHOLE tiny Llama as a packed checkpoint in the REFERENCE's on-disk format: 2 decoder layers with all seven
linears, the RMSNorm weights, the embedding and the head, HF's LlamaForCausalLM key names -- written by the reference's own
``qeft.utils.modelutils.save_model`` (modelutils.py:219-268: lm_pack -> QuantLinear.pack -> state_dict + quantinfos).  This is
the file ``load_owqmodel`` (modelutils.py:147-183) feeds to the reference's decode benchmark (main.py:310-371, 510-553).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden_ckpt_llama.py

Writes ``ref_ckpt_llama2l.pth`` (tensors + Namespace objects) and ``ref_ckpt_llama2l_io.npz``: a token sequence and the
logits of a dense float64 causal forward over the FAKE-QUANTISED weights the checkpoint was packed from (what the reference
evaluates, recon.py:573), with the o_proj input gathered by the reference's sparse_to_dense_ids (qlinear.py:275).
The module tree below only gives the reference's writer HF's names to walk; qeft_amd.llama.QuantLlama.from_packed rebuilds
its own tree from the state dict.
"""
import math
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.dirname(os.path.dirname(HERE)))
from qeft.quant import Quantizer, quantize  # noqa: E402
from qeft.reorder import sparse_to_dense_ids  # noqa: E402
from qeft.utils.modelutils import save_model  # noqa: E402

HID, INTER, LAYERS, HEADS, VOCAB, R, G, T = 256, 384, 2, 2, 96, 128, 128, 24
EPS, THETA = 1e-5, 10000.0


class _Norm(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter((1.0 + 0.1 * torch.randn(HID)).half())


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(self, n, nn.Linear(HID, HID, bias=False, dtype=torch.float16))


class _Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.gate_proj = nn.Linear(HID, INTER, bias=False, dtype=torch.float16)
        self.up_proj = nn.Linear(HID, INTER, bias=False, dtype=torch.float16)
        self.down_proj = nn.Linear(INTER, HID, bias=False, dtype=torch.float16)


class _Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn, self.mlp = _Attn(), _Mlp()
        self.input_layernorm, self.post_attention_layernorm = _Norm(), _Norm()


class _Inner(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed_tokens = nn.Embedding(VOCAB, HID, dtype=torch.float16)
        self.layers = nn.ModuleList([_Layer() for _ in range(LAYERS)])
        self.norm = _Norm()


class TinyLlama(nn.Module):
    dtype = torch.float16

    def __init__(self):
        super().__init__()
        self.model = _Inner()
        self.lm_head = nn.Linear(HID, VOCAB, bias=False, dtype=torch.float16)


def dense_forward(sd, tokens, oids):
    """float64 causal Llama forward over a plain state dict (fake-quantised weights)."""
    f = lambda k: sd[k].double()   # noqa: E731
    n = tokens.numel()
    h = f("model.embed_tokens.weight")[tokens]
    half = 64
    inv = 1.0 / (THETA ** (torch.arange(0, half, dtype=torch.float64) / half))
    ang = torch.arange(n, dtype=torch.float64)[:, None] * inv[None, :]
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]

    def rms(x, g):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + EPS) * g

    def rope(x):
        a, b = x[..., :64], x[..., 64:]
        return torch.cat([a * cos - b * sin, b * cos + a * sin], -1)

    mask = torch.full((n, n), float("-inf"), dtype=torch.float64).triu(1)
    for li in range(LAYERS):
        p = f"model.layers.{li}."
        x = rms(h, f(p + "input_layernorm.weight"))
        q = rope((x @ f(p + "self_attn.q_proj.weight").T).view(n, HEADS, 128))
        k = rope((x @ f(p + "self_attn.k_proj.weight").T).view(n, HEADS, 128))
        v = (x @ f(p + "self_attn.v_proj.weight").T).view(n, HEADS, 128)
        att = torch.einsum("thd,shd->hts", q, k) / math.sqrt(128) + mask
        a = torch.einsum("hts,shd->thd", att.softmax(-1), v).reshape(n, HID)
        a = a[:, sparse_to_dense_ids(oids[p + "self_attn.o_proj"].long(), HID)]          # qlinear.py:275
        h = h + a @ f(p + "self_attn.o_proj.weight").T
        x = rms(h, f(p + "post_attention_layernorm.weight"))
        act = torch.nn.functional.silu(x @ f(p + "mlp.gate_proj.weight").T) * (x @ f(p + "mlp.up_proj.weight").T)
        h = h + act @ f(p + "mlp.down_proj.weight").T
    return rms(h, f("model.norm.weight")) @ f("lm_head.weight").T


def main():
    torch.manual_seed(2024)
    model = TinyLlama()
    model.model.embed_tokens.weight.data = (torch.randn(VOCAB, HID) * 0.5).half()
    model.lm_head.weight.data = (torch.randn(VOCAB, HID) * 0.05).half()
    quantizers, oids = {}, {}
    for name, lin in [(n, m) for n, m in model.named_modules() if isinstance(m, nn.Linear) and n != "lm_head"]:
        n, k = lin.weight.shape
        w = (torch.randn(n, k) * 0.05).half()
        qz = Quantizer(bits=4, perchannel=True, sym=False, mse=False, group_size=G)
        wq = torch.empty(n, k)
        for g0 in range(0, k, G):                      # the layer-wise flow: one find_params per group (recon.py:488-573)
            slab = w[:, g0:g0 + G].float()
            qz.find_params(slab, weight=True)
            wq[:, g0:g0 + G] = quantize(slab, qz.scale, qz.zero, qz.minq, qz.maxq)
            qz.append_params()
        wq = wq.half()
        wq[:, k - R:] = w[:, k - R:]                   # OGR: the retained fp16 columns are the LAST r (reorder.py:148-176)
        lin.weight.data = wq.clone()
        qz.out_ids = torch.randperm(k)[:R].sort().values.to(torch.int32) if "o_proj" in name \
            else torch.arange(k - R, k, dtype=torch.int32)
        qz.n_out, qz.reorder = R, True
        quantizers[name] = qz
        oids[name.rsplit(".", 0)[0]] = qz.out_ids
    fake_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}          # BEFORE packing: the fake-quantised model
    tokens = torch.randint(0, VOCAB, (T,))
    logits = dense_forward(fake_sd, tokens, oids)

    path = os.path.join(HERE, "ref_ckpt_llama2l.pth")
    save_model(model, quantizers, path, packing=True, fake=False)          # the reference's writer, unmodified
    ck = torch.load(path, weights_only=False)
    print(sorted(ck["model_state_dict"])[:12], len(ck["model_state_dict"]), "keys")
    np.savez_compressed(os.path.join(HERE, "ref_ckpt_llama2l_io.npz"), tokens=tokens.numpy(), logits=logits.numpy(),
                        **{"outids__" + n.replace(".", "__"): v.numpy() for n, v in oids.items() if "o_proj" in n})
    print("wrote", path, os.path.getsize(path), "bytes; logits", tuple(logits.shape), float(logits.abs().max()))


if __name__ == "__main__":
    main()
