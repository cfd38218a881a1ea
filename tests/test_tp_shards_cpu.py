"""Megatron-paired tensor parallelism on CPU (world_size-2 gloo; the HIP kernels cannot run here): the shard arithmetic of
the decode engine's o_proj / down_proj -- fuse.column_shard's operands, the zero-padded x vectors and ONE fp32 all-reduce of
the partial outputs (+ the residual on rank 0) -- reproduces the full layer.  The local compute is the oracle's dense
dequantisation of the rank's own operand."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import qeft_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layer(n, k, r, g, seed):
    bufs = O.make_layer(n, k, r, g, seed=seed)
    t = {key: torch.from_numpy(np.ascontiguousarray(bufs[key])) for key in ("qweight", "scales", "scaled_zeros", "oweight")}
    return SimpleNamespace(outfeatures=n, infeatures=k, group_size=g, outlierfeatures=r, bias=None, **t), bufs


def _partial(op, x_p):
    w = O.dequant_dense(op.qweight.numpy(), op.scales.numpy(), op.scaled_zeros.numpy(),
                        op.oweight.numpy() if op.outlierfeatures else None, op.group_size)
    return w.astype(np.float64) @ x_p.astype(np.float64)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("scattered", [False, True])
def test_column_shards_sum_to_the_full_layer(world, scattered):
    """Every rank's partial output on its zero-padded x sums to W x: contiguous ownership (down_proj) and ownership through a
    permutation with the outlier columns spread over the ranks (o_proj behind its reorder)."""
    from qeft_amd import fuse
    n, k, r, g = 32, 1024 + 128, 128, 128          # 8 INT4 groups + the outlier slice; 1152 / 8 = 144 columns per rank
    layer, bufs = _layer(n, k, r, g, seed=world)
    x = O.make_activation(1, k, r, seed=2)[0]
    rng = np.random.default_rng(0)
    if scattered:       # natural index -> kernel column: the r outlier columns sit anywhere in the natural order
        outl = np.sort(rng.choice(k, r, replace=False))
        rest = np.setdiff1d(np.arange(k), outl)
        inv = np.empty(k, dtype=np.int64)
        inv[rest] = np.arange(k - r)
        inv[outl] = k - r + np.arange(r)
    else:
        inv = np.arange(k)
    per = k // world
    total = np.zeros(n)
    seen = np.zeros(k, dtype=int)
    for rank in range(world):
        owned = inv[rank * per:(rank + 1) * per]
        op, pos = fuse.column_shard(layer, torch.from_numpy(owned))
        assert op.infeatures % 128 == 0 and op.qweight.shape == (n // 4, op.infeatures)
        assert op.infeatures <= 128 * (per // 128 + 2) + r          # at most two boundary groups + the outlier slice
        x_p = np.zeros(op.infeatures, dtype=np.float16)
        x_p[pos.numpy()] = x[owned]
        total += _partial(op, x_p)
        seen[owned] += 1
    assert (seen == 1).all()
    ref = O.quant_linear(x[None], bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], None, g,
                         round_fp16=False)[0].astype(np.float64)
    w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], g).astype(np.float64)
    assert np.abs(total - w @ x.astype(np.float64)).max() <= 1e-9 * np.abs(ref).max() + 1e-12


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qeft_amd import fuse
        n, k, r, g = 64, 768 + 128, 128, 128           # 6 INT4 groups over 2 ranks of 448 columns: a shared boundary group
        layer, bufs = _layer(n, k, r, g, seed=11)
        x = O.make_activation(1, k, r, seed=3)[0]
        h = np.random.default_rng(4).standard_normal(n).astype(np.float32)       # the fp32 residual stream
        per = k // world
        owned = torch.arange(rank * per, (rank + 1) * per)
        op, pos = fuse.column_shard(layer, owned)
        x_p = np.zeros(op.infeatures, dtype=np.float16)
        x_p[pos.numpy()] = x[owned.numpy()]
        part = torch.from_numpy((_partial(op, x_p) + (h if rank == 0 else 0.0)).astype(np.float32))    # residual on rank 0 only
        dist.all_reduce(part)                                                    # THE collective of the pair
        w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], g).astype(np.float64)
        want = h.astype(np.float64) + w @ x.astype(np.float64)
        err = float(np.abs(part.numpy().astype(np.float64) - want).max() / np.abs(want).max())
        q.put((rank, int(op.infeatures), err, part.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def test_partial_outputs_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][3] == res[1][3]                  # both ranks hold the same bytes after the all-reduce
    for rank, kp, err, _ in res:
        # rank 0 owns columns 0..447 (3.5 groups -> 4 whole groups), rank 1 columns 448..767 (2.5 -> 3) and the outlier
        # columns; both carry the 128-column outlier slice
        assert kp == (512 + 128 if rank == 0 else 384 + 128)
        assert err < 1e-6, (rank, err)
