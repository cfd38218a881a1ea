"""Packed-checkpoint I/O with the reference's on-disk format (qeft/utils/modelutils.py:120-284).

    {'model_state_dict', 'quantinfos': {name: Namespace(bits, sym, group_size, n_out, reorder)},
     'packing': True, 'dtype', 'bits', 'group_size'}                      (save_model :248-268)
    {'oweight_state_dict': {layer_name: oweight}, 'base_path'}            (save_wctmodel :270-284)

The model skeleton (any nn.Module whose quantised layers are nn.Linear with the checkpoint's names) is supplied
by the caller; HF hub loading is out of scope (no network).
"""
import os
from argparse import Namespace
from collections import OrderedDict

import torch

from .qlinear import QuantLinear
from .quant import find_layers, make_quant


def save_packed(model, quantinfos, save_path):
    """model: already packed (its quantised layers are QuantLinear)."""
    infos = {n: Namespace(bits=getattr(q, "bits", 4), sym=getattr(q, "sym", False),
                          group_size=getattr(q, "group_size", -1), n_out=getattr(q, "n_out", 0),
                          reorder=getattr(q, "reorder", False)) for n, q in quantinfos.items()}
    first = next(iter(infos.values()))
    d = os.path.dirname(save_path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save({"model_state_dict": model.state_dict(), "quantinfos": infos, "packing": True,
                "dtype": torch.float16, "bits": first.bits, "group_size": first.group_size}, save_path)


def load_checkpoint_file(checkpoint_path, unsafe_pickle=False):
    """torch.load of a checkpoint in the reference's format (save_model / save_wctmodel, modelutils.py:248-284).  The only
    non-tensor objects the format holds are argparse.Namespace instances (`quantinfos`) and a torch.dtype, so the file is read with
    the restricted unpickler plus that one class; the reference's own `torch.load` runs arbitrary pickle code, which a
    user-supplied .pth (bench.py --ckpt) must not.  unsafe_pickle=True is the reference's behaviour, for files that carry more."""
    if unsafe_pickle:
        return torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    import argparse
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(checkpoint_path, map_location="cpu", weights_only=True)


def load_packed(model, checkpoint_path, device="cuda:0", training=False, unsafe_pickle=False):
    """reference load_owqmodel / hfmodel_to_owqmodel (modelutils.py:120-183): swap modules, load buffers
    (strict=False like the reference), bind kernels."""
    ckpt = load_checkpoint_file(checkpoint_path, unsafe_pickle)
    if "base_path" in ckpt:
        base = ckpt["base_path"]
        if not os.path.isabs(base) and not os.path.exists(base):     # a delta that travelled with its base file
            base = os.path.join(os.path.dirname(os.path.abspath(checkpoint_path)), base)
        model = load_packed(model, base, device=device, training=training, unsafe_pickle=unsafe_pickle)
        replace_oweight(model, ckpt["oweight_state_dict"])
        return model
    assert ckpt.get("packing", False), "not a packed checkpoint"
    make_quant(model, ckpt["quantinfos"])
    sd = ckpt["model_state_dict"]
    # reorder_ids is registered by set_kernel(); a checkpoint saved after set_kernel() carries it already
    model.load_state_dict({k: v for k, v in sd.items() if not k.endswith("reorder_ids")}, strict=False)
    model = model.to(device)
    for layer in find_layers(model, [QuantLinear]).values():
        layer.set_kernel(training)
    return model


def replace_oweight(model, oweight_state_dict):
    """reference replace_oweight (modelutils.py:185-198) + the fix for its quirk: the interleaved copy the GEMV
    reads is re-derived, so fine-tuned outlier weights are what decode uses."""
    qlayers = find_layers(model, [QuantLinear])
    for name, ow in oweight_state_dict.items():
        layer = qlayers[name]
        with torch.no_grad():
            layer.oweight.copy_(ow.to(layer.oweight.dtype).to(layer.oweight.device))
        layer.refresh_interleaved()
    # derived copies of the outlier weights: the prefill operands are rebuilt on the next prompt, a DecodeEngine built before
    # this call refuses to run (its fuse.py operands are copies too)
    if getattr(model, "_prefill_ops", None) is not None:
        model._prefill_ops = None
    model._oweight_version = getattr(model, "_oweight_version", 0) + 1


def save_finetuned(model, base_path, output_dir):
    """reference save_wctmodel (modelutils.py:270-284): only the trained outlier slices + pointer to the base."""
    sd = OrderedDict()
    for name, layer in find_layers(model, [QuantLinear]).items():
        if layer.outlierfeatures > 0:
            sd[name] = layer.oweight.detach().to(torch.float16).cpu()
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "model.pth")
    torch.save({"oweight_state_dict": sd, "base_path": os.path.abspath(base_path)}, path)
    return path
