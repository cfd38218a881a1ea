"""Host-side quantisation helpers that install and feed QuantLinear (reference qeft/quant.py).

Only the pieces on or next to the hot path are restated: the fake-quant formula (:8-10), the asymmetric
min-max parameters of `Quantizer.find_params(mse=False)` (:142-158), `make_quant` (:194-214) and `lm_pack`
(:216-234).  The MSE grid search and the GPTQ/OWQ reconstruction are offline calibration and out of scope.
"""
import torch
import torch.nn as nn

from .qlinear import QuantLinear


def quantize(x, scale, zero, minq, maxq):
    """scale * (clamp(round(x/scale) + zero, minq, maxq) - zero)   (reference quant.py:8-10)"""
    q = torch.clamp(torch.round(x / scale) + zero, minq, maxq)
    return scale * (q - zero)


def minmax_params(w, group_size, bits=4, sym=False):
    """Per-group min-max (scale, zero), fp32 [N, K/g]  (reference quant.py:142-158; sym=True: the symmetric branch,
    grid -2^(b-1)..2^(b-1)-1, zero = 0 -- QuantLinear.pack(sym=True) then shifts the zeros by 2^(b-1))."""
    n, k = w.shape
    g = w.float().reshape(n, k // group_size, group_size)
    xmin = torch.minimum(g.amin(-1), torch.zeros((), device=w.device))
    xmax = torch.maximum(g.amax(-1), torch.zeros((), device=w.device))
    if sym:
        xmax = torch.maximum(xmin.abs(), xmax)
        xmin = torch.where(xmin < 0, -xmax, xmin)
    dead = (xmin == 0) & (xmax == 0)
    xmin = torch.where(dead, -torch.ones_like(xmin), xmin)
    xmax = torch.where(dead, torch.ones_like(xmax), xmax)
    if sym:
        scale = xmax / ((2 ** bits - 1) // 2 + 1)
        return scale, torch.zeros_like(scale)
    maxq = 2 ** bits - 1
    scale = (xmax - xmin) / maxq
    zero = torch.round(-xmin / scale)
    return scale, zero


def fake_quantize(w, scale, zero, group_size, bits=4, sym=False):
    s = torch.repeat_interleave(scale, group_size, dim=1)
    z = torch.repeat_interleave(zero, group_size, dim=1)
    minq, maxq = (-((2 ** bits - 1) // 2 + 1), (2 ** bits - 1) // 2) if sym else (0, 2 ** bits - 1)
    return quantize(w.float(), s, z, minq, maxq)


def find_layers(module, layers=(nn.Linear,), name=""):
    """reference qeft/utils/misc.py:8-16"""
    if isinstance(module, tuple(layers)):
        return {name: module}
    res = {}
    for name1, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + "." + name1 if name != "" else name1))
    return res


def make_quant(module, quantinfos, name=""):
    """Swap every nn.Linear named in `quantinfos` for a QuantLinear (reference quant.py:194-214)."""
    if isinstance(module, QuantLinear):
        return
    for attr in dir(module):
        tmp = getattr(module, attr)
        name1 = name + "." + attr if name != "" else attr
        if name1 in quantinfos:
            info = quantinfos[name1]
            setattr(module, attr, QuantLinear(info.bits, tmp.in_features, tmp.out_features, tmp.bias is not None,
                                              tmp.weight.dtype, getattr(info, "n_out", 0),
                                              getattr(info, "group_size", -1), getattr(info, "reorder", False),
                                              name1).to(tmp.weight.device))
    for name1, child in module.named_children():
        make_quant(child, quantinfos, name + "." + name1 if name != "" else name1)


def lm_pack(model, quantinfos, linears=(nn.Linear,)):
    """Pack every quantised nn.Linear of `model` in place (reference quant.py:216-234).  `quantinfos[name]`
    carries bits / group_size / n_out / reorder / sym and the calibration results scale(_group) / zero(_group) /
    out_ids."""
    layers = find_layers(model, linears)
    layers = {n: layers[n] for n in quantinfos}
    make_quant(model, quantinfos)
    qlayers = find_layers(model, [QuantLinear])
    for name in qlayers:
        info = quantinfos[name]
        qlayers[name].pack(layers[name],
                           scales=getattr(info, "scale_group", getattr(info, "scale", None)),
                           zeros=getattr(info, "zero_group", getattr(info, "zero", None)),
                           outlieridx=getattr(info, "out_ids", None), sym=getattr(info, "sym", False))
    return model
