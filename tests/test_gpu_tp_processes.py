"""The tensor-parallel engine as REAL processes: two ranks (one process each, as `bench.py --gpus 2` launches them).  On a
one-GPU box they share the GPU and reduce over gloo -- torch.distributed's all_reduce on device tensors in place of RCCL,
the rest of the path identical (shard construction per rank, 2 collectives per layer, replicated norms / lm_head); where two
GPUs are visible the same test also runs on RCCL with one GPU per rank.  Every rank's
teacher-forced logits must equal the other's bit for bit and the single-GPU engine's within 1e-3 of the largest logit."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, q, backend):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo: both ranks on the one GPU; nccl (= RCCL): one GPU per rank, as bench.py --gpus 2 runs on a multi-GPU node
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
        shape = tiny_shape(n_layers=2, hidden=512, inter=1024, n_heads=4, vocab=512, max_seq=64)
        model = QuantLlama(shape, dev, seed=9)
        tokens = torch.randint(0, shape.vocab, (10,), generator=torch.Generator().manual_seed(2))
        eng = DecodeEngine(model, use_graph=False, tp_group=dist.group.WORLD)
        assert eng.tp3 and eng.P == world and eng.rank == rank
        got = eng.teacher_forced_logits(tokens)
        n_coll = eng.n_collectives
        ref = DecodeEngine(model, use_graph=False).teacher_forced_logits(tokens) if rank == 0 else None
        torch.cuda.synchronize()
        q.put((rank, got.cpu(), ref.cpu() if ref is not None else None, n_coll))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", pytest.param("nccl", marks=pytest.mark.skipif(
    torch.cuda.device_count() < 2, reason="the RCCL world-2 run needs two GPUs (the one-GPU box runs the gloo form)"))])
def test_two_rank_processes_match_the_single_gpu_engine(backend):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, got0, ref, n0), (_, got1, _, n1) = res
    assert torch.isfinite(got0).all() and torch.equal(got0, got1)
    assert n0 == n1 == 2 * 2                                   # 2 collectives per layer, 2 layers
    assert (got0 - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
