"""Generate golden vectors by running the REFERENCE's own Python packer.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

Writes small ``.npz`` fixtures next to this file.  Each fixture holds inputs
(fake-quantised fp16 weight, fp32 scale/zero, outlier index list) and the
outputs of the reference functions

    qeft.qlinear.pack_intweight / pack_oweight / QuantLinear.pack   (qlinear.py:70-215)
    qeft.quant.Quantizer.find_params (min-max) / quantize            (quant.py:8-10,142-158)
    qeft.reorder.sparse_to_dense_ids                                 (reorder.py:6-12)

The fixtures are data only; no reference source text is stored.
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference")
from qeft.qlinear import QuantLinear, pack_intweight, pack_oweight  # noqa: E402
from qeft.quant import Quantizer, quantize  # noqa: E402
from qeft.reorder import sparse_to_dense_ids  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (N, K, n_out, group, sym, bias)
CASES = [
    (8, 128, 0, 128, False, False),
    (8, 128, 64, 64, False, False),
    (16, 256, 128, 128, False, True),
    (64, 512, 128, 128, False, False),
    (64, 512, 64, 128, False, False),
    (256, 1024, 128, 128, False, False),
    (32, 256, 32, 128, True, False),
    (24, 384, 0, 128, False, True),
]


def ref_minmax_fakequant(w, group, sym):
    """Per-group parameters exactly as layerwise quantisation drives the reference
    Quantizer (find_params on each [N, g] slab, weight=True, perchannel)."""
    n, k = w.shape
    scales, zeros, wq = [], [], torch.empty_like(w, dtype=torch.float32)
    for g0 in range(0, k, group):
        qz = Quantizer(bits=4, perchannel=True, sym=sym, mse=False, group_size=group)
        slab = w[:, g0:g0 + group].float()
        qz.find_params(slab, weight=True)
        wq[:, g0:g0 + group] = quantize(slab, qz.scale, qz.zero, qz.minq, qz.maxq)
        scales.append(qz.scale.reshape(n, 1))
        zeros.append(qz.zero.reshape(n, 1))
    return torch.cat(scales, 1), torch.cat(zeros, 1), wq, (qz.minq, qz.maxq)


def main():
    for ci, (n, k, r, g, sym, bias) in enumerate(CASES):
        torch.manual_seed(1000 + ci)
        w = (torch.randn(n, k) * 0.02).half()
        scale, zero, wq, (minq, maxq) = ref_minmax_fakequant(w, g, sym)
        wq = wq.half()
        if r > 0:
            wq[:, k - r:] = w[:, k - r:]
        perm = torch.randperm(k)[:max(r, 1)].sort().values.int() if r > 0 else torch.zeros(0, dtype=torch.int)
        lin = torch.nn.Linear(k, n, bias=bias, dtype=torch.float16)
        lin.weight.data = wq.clone()
        if bias:
            lin.bias.data = (torch.randn(n) * 0.1).half()
        ql = QuantLinear(4, k, n, bias, torch.float16, r, g, True, "model.layers.0.self_attn.o_proj")
        ql.pack(lin, scale.clone(), zero.clone(), perm, sym=sym)
        sd = {kk: v.detach().cpu().numpy() for kk, v in ql.state_dict().items()}
        if bias and "bias" not in sd:          # pack() rebinds bias as a plain attribute
            sd["bias"] = ql.bias.detach().cpu().numpy()
        # raw integer matrix through the stand-alone packer as well
        qraw = torch.randint(0, 16, (n, k), dtype=torch.int32)
        out = dict(
            case=np.array([n, k, r, g, int(sym), int(bias)], dtype=np.int64),
            w_orig=w.numpy(), w_fake=wq.numpy(),
            scale=scale.float().numpy(), zero=zero.float().numpy(),
            outlieridx=perm.numpy(),
            qraw=qraw.numpy().astype(np.uint8),
            qraw_packed=pack_intweight(qraw, 4, 64).numpy(),
            **{"sd_" + kk: v for kk, v in sd.items()},
        )
        if r > 0:
            out["reorder_ids"] = sparse_to_dense_ids(perm.long(), k).numpy()
            ow_rand = torch.randn(n, r).half()
            out["ow_rand"] = ow_rand.numpy()
            out["ow_rand_packed"] = pack_oweight(ow_rand, 4).numpy()
        path = os.path.join(HERE, f"pack_case{ci}.npz")
        np.savez_compressed(path, **out)
        print(path, {kk: v.shape for kk, v in out.items() if kk.startswith("sd_")})


if __name__ == "__main__":
    main()
