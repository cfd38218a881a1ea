"""N > 1 path on CPU: world_size-2 gloo.  The HIP kernels cannot run here, so the tests cover what is host logic:
row-sharding of every packed buffer (shards re-assemble to the full layer bit for bit, each shard is a valid
stand-alone packed layer) and the collective plumbing of ShardedQuantLinear (all-gather of the per-rank output slices
in rank order), with the local compute replaced by a dense reference on the rank's own shard."""
import os
import socket

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


def _make_full(n, k, r, g, seed, name="model.layers.0.mlp.up_proj"):
    from qeft_amd.qlinear import QuantLinear
    bufs = O.make_layer(n, k, r, g, seed=seed, bias=True)
    ql = QuantLinear(4, k, n, True, torch.float16, r, g, True, name)
    for key in ("qweight", "scales", "scaled_zeros", "oweight", "oweight_interleaved", "bias"):
        setattr(ql, key, torch.from_numpy(np.ascontiguousarray(bufs[key])))
    ql.outlieridx = torch.arange(k - r, k, dtype=torch.int32)
    return ql, bufs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shards_reassemble_bit_exact(world):
    from qeft_amd.qlinear import unpack_intweight, unpack_oweight
    from qeft_amd.sharded import shard_bounds, shard_quantlinear
    n, k, r, g = 256, 512, 128, 128
    full, bufs = _make_full(n, k, r, g, seed=1)
    full.set_kernel = lambda training=False: None    # no GPU here: skip kernel binding of the full layer
    parts = []
    for rank in range(world):
        import qeft_amd.qlinear as qm
        orig = qm.QuantLinear.set_kernel
        qm.QuantLinear.set_kernel = lambda self, training=False: None
        try:
            parts.append(shard_quantlinear(full, rank, world))
        finally:
            qm.QuantLinear.set_kernel = orig
        n0, n1 = shard_bounds(n, rank, world)
        p = parts[-1]
        assert p.outfeatures == n1 - n0 and p.qweight.shape == ((n1 - n0) // 4, k)
        # each shard is itself a valid packed layer: unpacking it gives the rows of the full layer
        assert torch.equal(unpack_intweight(p.qweight), unpack_intweight(full.qweight)[n0:n1])
        assert torch.equal(unpack_oweight(p.oweight_interleaved), full.oweight[n0:n1])
    assert torch.equal(torch.cat([p.qweight for p in parts], 0), full.qweight)
    assert torch.equal(torch.cat([p.scales for p in parts], 1), full.scales)
    assert torch.equal(torch.cat([p.scaled_zeros for p in parts], 1), full.scaled_zeros)
    assert torch.equal(torch.cat([p.oweight for p in parts], 0), full.oweight)
    assert torch.equal(torch.cat([p.oweight_interleaved for p in parts], 0), full.oweight_interleaved)
    assert torch.equal(torch.cat([p.bias for p in parts], 0), full.bias)


def test_shard_requires_multiple_of_8_rows_per_rank():
    from qeft_amd.sharded import shard_bounds
    with pytest.raises(AssertionError):
        shard_bounds(40, 0, 2)      # 20 rows per rank would split an 8-row outlier interleave block
    assert shard_bounds(11008, 7, 8) == (9632, 11008)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import qeft_amd.qlinear as qm
        from qeft_amd.qlinear import unpack_intweight
        from qeft_amd.sharded import ShardedQuantLinear
        qm.QuantLinear.set_kernel = lambda self, training=False: None   # no HIP on this host
        n, k, r, g = 128, 256, 128, 128
        full, bufs = _make_full(n, k, r, g, seed=3)
        sh = ShardedQuantLinear(full, dist.group.WORLD)

        def local_dense(x):   # stand-in for the HIP kernels: dense math on THIS rank's shard only
            l = sh.local
            qv = unpack_intweight(l.qweight).float()
            s = l.scales.float().t().repeat_interleave(g, 1)
            z = l.scaled_zeros.float().t().repeat_interleave(g, 1)
            w = qv * s + z
            w[:, k - r:] = l.oweight.float()
            return (x.float() @ w.t() + l.bias.float()).half()
        sh.local.forward = local_dense
        sh.local.__class__.__call__ = lambda self, x: self.forward(x)
        x = torch.from_numpy(O.make_activation(3, k, r, seed=5))
        y = sh(x)
        yref = O.quant_linear(x.numpy(), bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"],
                              bufs["bias"], g, round_fp16=False)
        err = np.abs(y.numpy().astype(np.float64) - yref.astype(np.float64)).max() / np.abs(yref).max()
        q.put((rank, tuple(y.shape), float(err)))
    finally:
        dist.destroy_process_group()


def test_sharded_forward_allgather_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shape, err in res:
        assert shape == (3, 128)
        assert err < 2e-3, (rank, err)
