"""Generate a packed checkpoint in the REFERENCE's on-disk format by running the reference's own
``qeft.utils.modelutils.save_model`` (modelutils.py:219-268: lm_pack -> QuantLinear.pack -> state_dict + quantinfos of
argparse.Namespace) on a tiny Llama-shaped skeleton, and the fine-tuned delta of ``save_wctmodel`` (:270-284).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden_ckpt.py

/root/repo is on the path because qeft/monkeypatch/ftllama_modeling.py:18 imports `qeft_cuda` unconditionally: the
reference's modules then bind THIS build's `qeft_cuda` alias -- the drop-in boundary itself; nothing on the GPU runs here.
Writes ``ref_ckpt_tiny.pth``, ``ref_ckpt_tiny_wct.pth`` (data: tensors + Namespace objects) and ``ref_ckpt_tiny_io.npz``
(an input batch and the outputs of a dense float64 forward over the fake-quantised weights the checkpoint was packed
from, before and after the fine-tuned delta).
The skeleton class is re-declared by the test that loads the fixture (tests/test_checkpoint_reference_format.py).
"""
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
from qeft.utils.modelutils import save_model, save_wctmodel  # noqa: E402

HID, INTER, R, G = 256, 384, 128, 128


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
        self.self_attn = _Attn()
        self.mlp = _Mlp()


class _Inner(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([_Layer()])


class TinyLlamaSkeleton(nn.Module):
    dtype = torch.float16

    def __init__(self):
        super().__init__()
        self.model = _Inner()


def main():
    torch.manual_seed(77)
    model = TinyLlamaSkeleton()
    quantizers, dense = {}, {}
    for name, lin in [(n, m) for n, m in model.named_modules() if isinstance(m, nn.Linear)]:
        n, k = lin.weight.shape
        w = (torch.randn(n, k) * 0.02).half()
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
        if lin.bias is not None:
            lin.bias.data = (torch.randn(n) * 0.1).half()
        # out_ids: original (pre-reorder) indices of the retained columns; o_proj has its own set (reorder.py:38-46)
        qz.out_ids = torch.randperm(k)[:R].sort().values.to(torch.int32) if "o_proj" in name \
            else torch.arange(k - R, k, dtype=torch.int32)
        qz.n_out, qz.reorder = R, True
        quantizers[name] = qz
        dense[name] = (wq.float().numpy(), None if lin.bias is None else lin.bias.data.float().numpy())
    path = os.path.join(HERE, "ref_ckpt_tiny.pth")
    save_model(model, quantizers, path, packing=True, fake=False)          # the reference's writer, unmodified
    ck = torch.load(path, weights_only=False)
    print({k: (type(v).__name__) for k, v in ck.items()}, sorted(ck["model_state_dict"])[:8])

    # fine-tuned delta through the reference's writer: set_for_wct() makes oweight an fp32 Parameter, then save_wctmodel
    from qeft.qlinear import QuantLinear
    torch.manual_seed(5)
    new_ow = {}
    for name, mod in model.named_modules():
        if isinstance(mod, QuantLinear):
            mod.set_for_wct()
            with torch.no_grad():
                mod.oweight.add_(torch.randn_like(mod.oweight) * 0.01)
            new_ow[name] = mod.oweight.detach().half().numpy()
    save_wctmodel(model, path, os.path.join(HERE, "_wct_tmp"))
    wct = torch.load(os.path.join(HERE, "_wct_tmp", "model.pth"), weights_only=False)
    wct["base_path"] = "ref_ckpt_tiny.pth"             # relative: the fixture travels (the loader resolves it next to the file)
    torch.save(wct, os.path.join(HERE, "ref_ckpt_tiny_wct.pth"))
    os.remove(os.path.join(HERE, "_wct_tmp", "model.pth"))
    os.rmdir(os.path.join(HERE, "_wct_tmp"))

    # expected outputs: dense float64 forward over the fake-quantised weights the checkpoint was packed from (what the
    # reference evaluates, recon.py:573), o_proj input gathered with the reference's sparse_to_dense_ids (qlinear.py:275)
    from qeft.reorder import sparse_to_dense_ids
    x = (torch.randn(5, HID) * 1.0).half()
    xi = (torch.randn(5, INTER) * 1.0).half()
    out = {"x": x.numpy(), "x_inter": xi.numpy()}
    for name, (w, b) in dense.items():
        key = name.replace(".", "__")
        k = w.shape[1]
        inp = (xi if k == INTER else x).double()
        if "o_proj" in name:
            inp = inp[:, sparse_to_dense_ids(quantizers[name].out_ids.long(), k)]
        y = inp @ torch.from_numpy(w).double().T
        w2 = torch.from_numpy(w).double().clone()
        w2[:, k - R:] = torch.from_numpy(new_ow[name]).double()
        y2 = inp @ w2.T
        if b is not None:
            y, y2 = y + torch.from_numpy(b).double(), y2 + torch.from_numpy(b).double()
        out["y__" + key] = y.numpy()
        out["y_wct__" + key] = y2.numpy()
        out["outids__" + key] = quantizers[name].out_ids.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_ckpt_tiny_io.npz"), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
