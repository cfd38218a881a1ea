"""Row-sharded QuantLinear over the GPUs of one node (no reference counterpart).

Output rows of W[N, K] are independent, so rank r keeps rows [r*N/P, (r+1)*N/P) of every packed buffer
(multiples of 8 rows keep the 4-row nibble interleave and the 8-row outlier interleave intact), computes its slice
of y with the same kernels, and ONE collective per linear (all-gather of y[m, N/P] over RCCL/xGMI; the payload is
m*N*2 bytes, latency-bound at decode) rebuilds the full activation.  x is replicated.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from .qlinear import QuantLinear


def shard_bounds(n, rank, world, align=8):
    """Rows of rank `rank`; `align` = 8 keeps both 4-bit interleaves, 16 the row sets of the 3-bit layout."""
    assert n % (align * world) == 0, f"out_features {n} must be a multiple of {align}*world_size ({align * world})"
    per = n // world
    return rank * per, (rank + 1) * per


def shard_quantlinear(ql: QuantLinear, rank: int, world: int) -> QuantLinear:
    """Rows [n0, n1) of a packed QuantLinear as a stand-alone QuantLinear (shares no storage with the original)."""
    n0, n1 = shard_bounds(ql.outfeatures, rank, world, 16 if ql.bits == 3 else 8)
    out = QuantLinear(ql.bits, ql.infeatures, n1 - n0, ql.bias is not None, ql.dtype, ql.outlierfeatures,
                      ql.group_size, getattr(ql, "reorder", False), ql.name)
    rows = 16 if ql.bits == 3 else 4      # weight rows per qweight row
    out.qweight = ql.qweight[n0 // rows:n1 // rows].clone()
    out.scales = ql.scales[:, n0:n1].contiguous()
    out.scaled_zeros = ql.scaled_zeros[:, n0:n1].contiguous()
    if ql.bias is not None:
        out.bias = ql.bias[n0:n1].clone()
    if ql.outlierfeatures > 0:
        out.oweight = ql.oweight.detach()[n0:n1].clone()
        out.oweight_interleaved = ql.oweight_interleaved[n0 // 2:n1 // 2].clone()
        out.outlieridx = ql.outlieridx.clone()
    out.fused = ql.fused
    out.set_kernel(getattr(ql, "training", False))
    return out


class ShardedQuantLinear(nn.Module):
    """Drop-in for a QuantLinear inside a tensor-parallel group: local rows + all-gather of the outputs."""

    def __init__(self, full: QuantLinear, group=None):
        super().__init__()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.outfeatures = full.outfeatures
        self.infeatures = full.infeatures
        self.local = shard_quantlinear(full, self.rank, self.world)
        self._gather = {}       # rows -> [world, rows, N / world] gather buffer

    def forward(self, x):
        """One collective, no per-call allocation beyond the output: the local rows land in a [world, m, N/P] buffer kept per batch
        size (all_gather_into_tensor: hipGraph-capturable on RCCL), and the rank-major buffer is viewed back as [m, N]."""
        y_loc = self.local(x)
        lead = y_loc.shape[:-1]
        y2 = y_loc.reshape(-1, y_loc.shape[-1]).contiguous()
        m, nl = y2.shape
        buf = self._gather.get(m)
        if buf is None or buf.device != y2.device:
            buf = torch.empty(self.world * m, nl, dtype=y2.dtype, device=y2.device)      # (rank-major concatenation: the form gloo takes too)
            self._gather[m] = buf
        dist.all_gather_into_tensor(buf, y2, group=self.group)
        return buf.view(self.world, m, nl).permute(1, 0, 2).reshape(*lead, self.world * nl)
