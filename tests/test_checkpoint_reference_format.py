"""A packed checkpoint WRITTEN BY THE REFERENCE (qeft.utils.modelutils.save_model / save_wctmodel, run by
tests/golden/make_golden_ckpt.py in the build container) loads unchanged through qeft_amd.checkpoint.load_packed:
argparse.Namespace quantinfos, the reference QuantLinear's state_dict keys and layouts, the fine-tuned delta format."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import qeft_oracle as O
from util import rel_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT, WCT = os.path.join(GOLDEN, "ref_ckpt_tiny.pth"), os.path.join(GOLDEN, "ref_ckpt_tiny_wct.pth")
HID, INTER, R, G = 256, 384, 128, 128
NAMES = ["model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.o_proj", "model.layers.0.mlp.up_proj",
         "model.layers.0.mlp.down_proj"]
FAKE_TOL = 5e-3     # packed checkpoint vs the fake-quantised fp16 weights it came from: fp16 rounding of scale / scaled zero


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj = nn.Linear(HID, HID, bias=False, dtype=torch.float16)
        self.o_proj = nn.Linear(HID, HID, bias=False, dtype=torch.float16)


class _Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.up_proj = nn.Linear(HID, INTER, bias=False, dtype=torch.float16)
        self.down_proj = nn.Linear(INTER, HID, bias=True, dtype=torch.float16)


class _Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn, self.mlp = _Attn(), _Mlp()


class _Inner(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([_Layer()])


class Skeleton(nn.Module):
    def __init__(self):
        super().__init__()
        self.model = _Inner()


def _io():
    return np.load(os.path.join(GOLDEN, "ref_ckpt_tiny_io.npz"))


def test_reference_checkpoint_format_is_what_survey_8b_says():
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    assert set(ck) == {"model_state_dict", "quantinfos", "packing", "dtype", "bits", "group_size"}
    assert ck["packing"] is True and ck["bits"] == 4 and ck["group_size"] == G and ck["dtype"] == torch.float16
    assert sorted(ck["quantinfos"]) == sorted(NAMES)
    for info in ck["quantinfos"].values():
        assert isinstance(info, Namespace)
        assert vars(info) == dict(bits=4, sym=False, group_size=G, n_out=R, reorder=True)
    keys = {k.rsplit(".", 1)[1] for k in ck["model_state_dict"] if k.startswith(NAMES[0])}
    assert keys == {"qweight", "scales", "scaled_zeros", "oweight", "oweight_interleaved", "outlieridx"}
    wct = torch.load(WCT, map_location="cpu", weights_only=False)
    assert set(wct) == {"oweight_state_dict", "base_path"} and sorted(wct["oweight_state_dict"]) == sorted(NAMES)


def test_load_packed_on_cpu_restores_every_buffer_and_the_oracle_reproduces_the_outputs():
    from qeft_amd import checkpoint
    from qeft_amd.qlinear import QuantLinear
    from qeft_amd.quant import find_layers
    model = checkpoint.load_packed(Skeleton(), CKPT, device="cpu")
    ql = find_layers(model, [QuantLinear])
    assert sorted(ql) == sorted(NAMES)
    sd = torch.load(CKPT, map_location="cpu", weights_only=False)["model_state_dict"]
    io = _io()
    for name, layer in ql.items():
        for key in ("qweight", "scales", "scaled_zeros", "oweight", "oweight_interleaved", "outlieridx"):
            assert torch.equal(getattr(layer, key), sd[f"{name}.{key}"]), (name, key)
        assert layer.outlierfeatures == R and layer.group_size == G and layer.bits == 4
        key = name.replace(".", "__")
        assert np.array_equal(layer.outlieridx.numpy(), io["outids__" + key])
        x = io["x_inter"] if layer.infeatures == INTER else io["x"]
        ids = None
        if "o_proj" in name:      # set_kernel() registered the gather exactly as the reference does (qlinear.py:227-229)
            ids = layer.reorder_ids.numpy()
            assert np.array_equal(ids, O.sparse_to_dense_ids(io["outids__" + key], layer.infeatures))
        y = O.quant_linear(x, layer.qweight.numpy(), layer.scales.numpy(), layer.scaled_zeros.numpy(), layer.oweight.numpy(),
                           layer.bias.numpy() if layer.bias is not None else None, G, reorder_ids=ids)
        assert rel_err(y, io["y__" + key]) < FAKE_TOL, name
        # the interleaved copy the GEMV reads is the reference's pack_oweight of the plain one
        assert np.array_equal(O.pack_oweight(layer.oweight.numpy()).view(np.uint16),
                              layer.oweight_interleaved.numpy().view(np.uint16))
    assert ql["model.layers.0.mlp.down_proj"].bias is not None


@pytest.mark.gpu
@pytest.mark.parametrize("path,ykey", [(CKPT, "y__"), (WCT, "y_wct__")])
def test_reference_checkpoint_runs_on_the_hip_path(path, ykey):
    """decode (m = 5 -> GEMV) and prefill (m = 40 -> GEMM) forwards of the loaded modules vs the outputs recorded when the
    fixture was written; the fine-tuned delta must reach BOTH the GEMM's oweight and the GEMV's interleaved copy."""
    from qeft_amd import checkpoint
    from qeft_amd.qlinear import QuantLinear
    from qeft_amd.quant import find_layers
    model = checkpoint.load_packed(Skeleton(), path, device="cuda:0")
    io = _io()
    for name, layer in find_layers(model, [QuantLinear]).items():
        key = name.replace(".", "__")
        x = torch.from_numpy(io["x_inter"] if layer.infeatures == INTER else io["x"]).to("cuda:0")
        y = layer(x)
        y8 = layer(x.repeat(8, 1))[:5]
        torch.cuda.synchronize()
        assert rel_err(y.cpu().numpy(), io[ykey + key]) < FAKE_TOL, name
        assert rel_err(y8.cpu().numpy(), io[ykey + key]) < FAKE_TOL, name
