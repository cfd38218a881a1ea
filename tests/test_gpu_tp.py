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
