"""Launch geometry beyond the BASELINE models: two decoder layers of the Llama-2-70B shape (hidden 8192, MLP 28672, 64 query heads
over 8 KV heads) on the decode engine, teacher-forced, against the plain fp32 PyTorch model over the dense dequantised weights.
Pins grouped-query attention inside the engine (q|k|v concatenated with a narrow k|v), K = 8192 / 28672 GEMVs (512 .. 3584 row
sets) and the 8192-wide fused head.  Not a BASELINE config: the reference's scripts stop at 13B, its kernels do not."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_layers_of_the_70b_shape(use_graph):
    from qeft_amd.llama import DecodeEngine, LlamaShape, QuantLlama
    shape = LlamaShape(8192, 28672, 2, 64, 8, 32000, max_seq=256, name="llama-2-70b-2layers")
    model = QuantLlama(shape, DEV, seed=3, fast_init=True)
    dense = model.dense_weights()
    eng = DecodeEngine(model, use_graph=use_graph)
    assert eng.v3
    tokens = torch.randint(0, shape.vocab, (40,), generator=torch.Generator().manual_seed(5)).to(DEV)
    got = eng.teacher_forced_logits(tokens)
    ref = model.forward_dense_reference(tokens, dense)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    print(f"[70b-shape parity] max|dlogit|/max|logit| = {err:.3e}")
    assert torch.isfinite(got).all() and err < 6e-3, err
    top2 = ref.topk(2, dim=-1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2 * 6e-3 * scale
    assert torch.equal(got.argmax(-1)[sure], ref.argmax(-1)[sure])
    del dense, model, eng
    torch.cuda.empty_cache()
