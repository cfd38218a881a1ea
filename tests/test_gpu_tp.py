"""Tensor-parallel decode engine (Megatron pairing: q|k|v and gate|up by rows, o_proj and down_proj by columns, one fp32
all-reduce after each of the two, the consumers' RMSNorm inside their own launches, llama.py).  On the real collective library (RCCL) a group of ONE rank runs exactly the
launch sequence of the multi-GPU bench, hipGraph capture of the collectives included; ranks > 1 run as threads of one
process in lock step on the one GPU.  The sharded sum adds the K shards' fp32 partial outputs in rank order, so results
equal the single-GPU engine to rounding, not bit for bit: the tolerance below is 1e-3 of the largest logit.  The sharding
arithmetic alone is covered on CPU with gloo (tests/test_tp_shards_cpu.py)."""
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
    assert eng.tp and eng.P == 1 and eng.tp3
    got = eng.teacher_forced_logits(tokens)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.9
    if not use_graph:
        assert eng.n_collectives == 2 * shape.n_layers          # one all-reduce after o_proj, one after down_proj


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

            def all_reduce(self, t):
                self.slots[self.rank] = t
                self.barrier.wait()
                total = torch.stack(list(self.slots)).sum(0)     # the same reduction on every rank
                self.barrier.wait()                              # everybody has read every slot
                t.copy_(total)
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
    assert all(e.tp3 for e in engines)
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
        assert torch.equal(outs[r], outs[0]), f"rank {r} of {world} disagrees with rank 0"     # every rank holds the same sums
        assert (outs[r] - ref).abs().max().item() <= 1e-3 * ref.abs().max().item(), f"rank {r} of {world}"
        assert engines[r].n_collectives == 2 * shape.n_layers
    assert (outs[0].argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.9


@pytest.mark.parametrize("world,dims", [(2, (4096, 11008, 32)), (8, (4096, 11008, 32)), (4, (5120, 13824, 40)), (8, (5120, 13824, 40))])
def test_tp_engine_7b_13b_layer_shapes_in_lockstep(world, dims):
    """The shard geometry of Llama-2-7B's and -13B's layers (BASELINE config 4): 31 / 39 INT4 groups of o_proj and 85 / 107 of
    down_proj do not divide over the ranks -- boundary groups and the outlier slice are shared as fuse.column_shard says."""
    import threading
    from qeft_amd.llama import DecodeEngine, LlamaShape, QuantLlama
    hidden, inter, heads = dims
    shape = LlamaShape(hidden, inter, 1, heads, heads, 1024, 64, n_out=128, name="1layer")
    model = QuantLlama(shape, DEV, seed=4, fast_init=True)
    tokens = torch.randint(0, shape.vocab, (6,), generator=torch.Generator().manual_seed(5))
    ref = DecodeEngine(model, use_graph=False).teacher_forced_logits(tokens)
    grp = _LockstepGroup(world)
    engines = [DecodeEngine(model, use_graph=False, tp_group=grp.view(r)) for r in range(world)]
    assert all(e.tp3 for e in engines)
    # a rank streams ~1/world of the weights (+ the shared boundary groups and outlier slices)
    full = DecodeEngine(model, use_graph=False).weight_bytes_per_token()
    per_rank = [e.weight_bytes_per_token() for e in engines]
    assert max(per_rank) < full / world * (1.12 if world == 2 else 1.40)      # measured: 1.03x at 2 ranks, 1.13x at 8
    outs, errs = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(DEV)
            outs[r] = engines[r].teacher_forced_logits(tokens)
        except Exception as e:
            errs.append(e)
            grp.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(outs[r], outs[0])
        assert (outs[r] - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
