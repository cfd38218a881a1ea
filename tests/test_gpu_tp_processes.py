"""The tensor-parallel engine as REAL processes: two ranks (one process each, as `bench.py --gpus 2` launches them).  On a
one-GPU box they share the GPU and reduce over gloo -- torch.distributed's all_reduce on device tensors in place of RCCL,
the rest of the path identical (shard construction per rank, 2 collectives per layer, replicated norms / lm_head); where two
GPUs are visible the same test also runs on RCCL with one GPU per rank.  Every rank's
teacher-forced logits must equal the other's bit for bit and the single-GPU engine's within 1e-3 of the largest logit.

Third backend, "ipc": the hand-written one-shot all-reduce (csrc/oneshot.hip, qeft_amd/oneshot.py) -- each
process maps the other's mailbox through hipIpcMemHandle (two processes on ONE GPU can do that as well as two GPUs), gloo only
carries the 64-byte handles at construction.  Its logits must equal the gloo run's BIT FOR BIT (a two-rank fp32 sum is
order-free), eagerly and from a captured hipGraph."""
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
        eng = DecodeEngine(model, use_graph=False, tp_group=dist.group.WORLD, collective="oneshot" if backend == "ipc" else "rccl")
        assert eng.tp3 and eng.P == world and eng.rank == rank
        assert eng.collective == ("oneshot" if backend == "ipc" else "rccl")
        got = eng.teacher_forced_logits(tokens)
        n_coll = eng.n_collectives
        extra = None
        if backend == "ipc":
            eng.oneshot.check_status()
            # the same tokens through gloo in the same processes: the one-shot sum must be bit-identical
            eng_g = DecodeEngine(model, use_graph=False, tp_group=dist.group.WORLD)
            via_gloo = eng_g.teacher_forced_logits(tokens)
            # ... and from a captured graph (the collective is one kernel node; its sequence word lives in device memory)
            eng.reset()
            eng.capture()
            graph_logits = []
            for t in tokens.tolist():
                eng.tok.fill_(t)
                eng.graph.replay()
                eng.host_pos += 1
                graph_logits.append(eng.logits[0].float().clone())
            torch.cuda.synchronize()
            eng.oneshot.check_status()
            extra = (via_gloo.cpu().numpy(), torch.stack(graph_logits).cpu().numpy())
        ref = DecodeEngine(model, use_graph=False).teacher_forced_logits(tokens) if rank == 0 else None
        torch.cuda.synchronize()
        q.put((rank, got.cpu().numpy(), ref.cpu().numpy() if ref is not None else None, n_coll, extra))
    finally:
        dist.destroy_process_group()


def _collective_rank(rank, world, port, q):
    """The collective by itself: random vectors, 40 calls eagerly (the mailboxes' two parities many times over), 40 more from a
    captured graph of 4 calls; every result against the fp64 sum of what the ranks contributed and equal on both ranks."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qeft_amd.oneshot import OneShotAllReduce
        n = 4096 + 6                                # not a multiple of the block's 512 elements: ragged tail
        ar = OneShotAllReduce(n, dev, dist.group.WORLD)
        outs = []
        gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
        def contribution(r):
            return torch.randn(n, generator=gens[r], dtype=torch.float32)
        expect = []
        for _ in range(40):
            parts = [contribution(r) for r in range(world)]
            t = parts[rank].to(dev)
            ar.all_reduce(t)
            outs.append(t.cpu())
            expect.append(sum(p.double() for p in parts))
        torch.cuda.synchronize()
        ar.check_status()
        # graph: 4 dependent calls on one static buffer (t -> 2 t -> 4 t ... with two ranks that contribute the same vector)
        buf = torch.zeros(n, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            ar.all_reduce(buf)                       # warm-up outside the capture (both ranks make it)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4):
                ar.all_reduce(buf)
        graph_ok = True
        for it in range(10):
            v = torch.randn(n, generator=torch.Generator().manual_seed(7 + it), dtype=torch.float32)
            buf.copy_(v.to(dev))
            g.replay()
            torch.cuda.synchronize()
            graph_ok = graph_ok and torch.equal(buf.cpu(), v * float(world ** 4))
        ar.check_status()
        import hashlib
        exact = all(torch.equal(o, e.float()) for o, e in zip(outs, expect))       # two fp32 terms: the rounded exact sum
        digest = hashlib.sha256(b"".join(o.numpy().tobytes() for o in outs)).hexdigest()
        # a peer that never arrives: the call gives up after its 2 s and records a code; every call after that returns at once
        # (the group is out of step for good -- the caller reads the status word and falls back).  Last act of this object.
        dist.barrier()
        quick = True
        if rank == 0:
            import time
            lonely = torch.zeros(n, dtype=torch.float32, device=dev)
            t0 = time.time()
            ar.all_reduce(lonely)
            torch.cuda.synchronize()
            first = time.time() - t0
            t0 = time.time()
            for _ in range(3):
                ar.all_reduce(lonely)
            torch.cuda.synchronize()
            later = time.time() - t0
            gave_up = False
            try:
                ar.check_status()
            except RuntimeError:
                gave_up = True
            quick = gave_up and 1.0 < first < 6.0 and later < 0.5
        dist.barrier()
        q.put((rank, exact, digest, graph_ok and quick))        # (plain values: tensors through the queue outlive their process badly)
    finally:
        dist.destroy_process_group()


def test_oneshot_allreduce_two_processes_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_collective_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, exact0, dig0, g0), (_, exact1, dig1, g1) = res
    assert g0 and g1                  # 10 replays of a 4-call graph
    assert exact0 and exact1          # every eager result == the exactly rounded sum
    assert dig0 == dig1               # bit-identical on both ranks


@pytest.mark.parametrize("backend", ["gloo", "ipc", pytest.param("nccl", marks=pytest.mark.skipif(
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
    import numpy as np
    (_, got0, ref, n0, ex0), (_, got1, _, n1, ex1) = res
    got0, got1, ref = torch.from_numpy(got0), torch.from_numpy(got1), torch.from_numpy(ref)
    assert torch.isfinite(got0).all() and torch.equal(got0, got1)
    if backend == "ipc":
        for (via_gloo, via_graph), got in ((ex0, got0), (ex1, got1)):
            assert np.array_equal(via_gloo, got.numpy())                  # the one-shot sum == gloo's, bit for bit
            assert np.array_equal(via_graph, got.numpy())                 # and the same from a captured graph
    assert n0 == n1 == 2 * 2                                   # 2 collectives per layer, 2 layers
    assert (got0 - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
