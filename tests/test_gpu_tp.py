"""Row-sharded decode engine on the real collective library (RCCL): a group of ONE rank runs exactly the launch
sequence of the multi-GPU bench (sharded linears, one all-gather per linear, hipGraph capture of the collectives) and
must reproduce the single-GPU engine bit for bit.  Ranks > 1 are covered on CPU (gloo) in test_sharded_cpu.py."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def world_of_one():
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    yield dist.group.WORLD
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_tp_engine_group_of_one_equals_single_gpu_engine(world_of_one, use_graph):
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=2, hidden=512, inter=1024, n_heads=4, vocab=512, max_seq=64)
    model = QuantLlama(shape, DEV, seed=3)
    tokens = torch.randint(0, shape.vocab, (12,), generator=torch.Generator().manual_seed(1))
    ref = DecodeEngine(model, use_graph=use_graph).teacher_forced_logits(tokens)
    eng = DecodeEngine(model, use_graph=use_graph, tp_group=world_of_one)
    assert eng.tp and eng.P == 1
    got = eng.teacher_forced_logits(tokens)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)


def test_sharded_quantlinear_on_rccl(world_of_one):
    from qeft_amd.llama import synthetic_quantlinear
    from qeft_amd.sharded import ShardedQuantLinear
    ql = synthetic_quantlinear("model.layers.0.mlp.up_proj", 512, 1024, 128, 128, 5, DEV)
    sh = ShardedQuantLinear(ql, world_of_one)
    for m in (1, 5, 40):
        x = torch.randn(m, 512, device=DEV).half()
        assert torch.equal(sh(x), ql(x))


class _LockstepGroup:
    """Two (or more) ranks of ONE process, one thread each, sharing the GPU's default stream: all_gather = everybody
    publishes its slice, meets at a barrier, copies all slices (stream order makes them ready), meets again (nobody
    overwrites a slice somebody still has to copy).  Stands in for the process group so that the P > 1 launch sequence
    of the decode engine -- shard offsets, slice reassembly, residual slices -- runs on a one-GPU box."""

    def __init__(self, world):
        import threading
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)

    def view(self, rank):
        parent = self

        class _Rank:
            world, slots, barrier = parent.world, parent.slots, parent.barrier

            def __init__(self):
                self.rank = rank

            def all_gather(self, out, inp):
                self.slots[self.rank] = inp
                self.barrier.wait()
                out.view(self.world, -1).copy_(torch.stack([t.reshape(-1) for t in self.slots]))
                self.barrier.wait()
        return _Rank()


@pytest.mark.parametrize("world", [2, 4])
def test_tp_engine_two_ranks_in_lockstep_equal_single_gpu_engine(world):
    import threading
    from qeft_amd.llama import DecodeEngine, QuantLlama, tiny_shape
    shape = tiny_shape(n_layers=2, hidden=512, inter=1024, n_heads=4, vocab=512, max_seq=64)
    model = QuantLlama(shape, DEV, seed=9)
    tokens = torch.randint(0, shape.vocab, (10,), generator=torch.Generator().manual_seed(2))
    ref = DecodeEngine(model, use_graph=False).teacher_forced_logits(tokens)
    grp = _LockstepGroup(world)
    engines = [DecodeEngine(model, use_graph=False, tp_group=grp.view(r)) for r in range(world)]
    assert engines[1].P == world and engines[1].rank == 1 and engines[1].hs == shape.hidden // world
    outs, errs = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(DEV)
            outs[r] = engines[r].teacher_forced_logits(tokens)
        except Exception as e:       # a dead rank must not leave the others at the barrier
            errs.append(e)
            grp.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errs, errs
    torch.cuda.synchronize()
    for r in range(world):
        assert outs[r] is not None and torch.isfinite(outs[r]).all()
        assert torch.equal(outs[r], ref), f"rank {r} of {world}"
