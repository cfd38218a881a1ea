"""Shared helpers for the parity tests (oracle = checker, qeft_amd = thing under test)."""
import numpy as np
import torch

from oracle import qeft_oracle as O

REL_TOL = 1e-3  # north_star: outputs within 1e-3 relative (fp16) of the reference definition


def layer_to_torch(bufs, device):
    out = {}
    for k, v in bufs.items():
        if k == "fake_weight":
            continue
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
    return out


def rel_err(y, yref):
    """max |y - yref| / max |yref|  (per-tensor relative error)."""
    y = np.asarray(y, dtype=np.float64)
    yref = np.asarray(yref, dtype=np.float64)
    return float(np.abs(y - yref).max() / max(np.abs(yref).max(), 1e-12))


def elem_err_ok(y, yref, rtol=REL_TOL, atol_scale=1e-3):
    """|y - yref| <= rtol*|yref| + atol where atol = atol_scale * rms(yref): element-wise fp16-style check."""
    y = np.asarray(y, dtype=np.float64)
    yref = np.asarray(yref, dtype=np.float64)
    atol = atol_scale * float(np.sqrt(np.mean(yref ** 2)))
    return bool(np.all(np.abs(y - yref) <= rtol * np.abs(yref) + atol))


def oracle_forward(bufs, x, n_out, group=128, reorder_ids=None):
    return O.quant_linear(x, bufs["qweight"], bufs["scales"], bufs["scaled_zeros"],
                          bufs.get("oweight") if n_out else None, bufs.get("bias"), group,
                          reorder_ids=reorder_ids)
