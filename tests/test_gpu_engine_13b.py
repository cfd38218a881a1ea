"""Whole-model parity at BASELINE config 4's model: the Llama-2-13B shape (5120 / 13824, 40 layers) on the decode engine,
teacher-forced, against the plain fp32 PyTorch model over the dense dequantised weights (52 GB: fits a 288 GB part).
The single-GPU engine is what each rank of the row-sharded run executes on its shard (tests/test_gpu_tp.py covers the
sharding arithmetic); this test pins the 13B launch geometry itself: 320 / 864 / 960 row sets, K = 5120 / 13824."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NLL_TOL = 1e-3
LOGIT_TOL = 1.5e-2    # max |dlogit| / max |logit|: 40 layers of fp16 activations (observed 1.0e-2; the 7B shape's 32 layers give 6e-3)


@pytest.fixture(scope="module")
def model13b():
    import dataclasses
    from qeft_amd.llama import LLAMA2_13B, QuantLlama
    model = QuantLlama(dataclasses.replace(LLAMA2_13B, max_seq=512), DEV, seed=1, fast_init=True)
    dense = model.dense_weights()
    yield model, dense
    del dense, model
    torch.cuda.empty_cache()


def _compare(got, ref, tokens, nll):
    import torch.nn.functional as F
    from qeft_amd.llama import nll_from_logits
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    dn = abs(nll_from_logits(got, tokens) - nll_from_logits(ref, tokens))
    d = F.cross_entropy(got[:-1].float(), tokens[1:], reduction="none") - F.cross_entropy(ref[:-1].float(), tokens[1:], reduction="none")
    sem = d.std().item() / (d.numel() ** 0.5)
    print(f"[13b parity] T={tokens.numel()} max|dlogit|/max|logit|={err:.3e} |dNLL|={dn:.3e} (s.e.m. {sem:.2e})")
    assert err < LOGIT_TOL, err
    assert abs(d.mean().item()) <= 3.5 * sem + 1e-4, (d.mean().item(), sem)
    assert not nll or dn <= NLL_TOL, dn
    top2 = ref.topk(2, dim=-1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL * scale
    assert torch.equal(got.argmax(-1)[sure], ref.argmax(-1)[sure])


def test_engine_13b_first_tokens_eager(model13b):
    from qeft_amd.llama import DecodeEngine
    model, dense = model13b
    eng = DecodeEngine(model, use_graph=False)
    tokens = torch.randint(0, model.shape.vocab, (96,), generator=torch.Generator().manual_seed(11)).to(DEV)
    got = eng.teacher_forced_logits(tokens)
    ref = model.forward_dense_reference(tokens, dense)
    torch.cuda.synchronize()
    _compare(got, ref, tokens, nll=False)


def test_engine_13b_512_tokens_graph(model13b):
    from qeft_amd.llama import DecodeEngine
    model, dense = model13b
    eng = DecodeEngine(model, use_graph=True)
    tokens = torch.randint(0, model.shape.vocab, (512,), generator=torch.Generator().manual_seed(12)).to(DEV)
    got = eng.teacher_forced_logits(tokens)
    assert {key[0] for key in eng.graphs} >= {1, 4}
    ref = model.forward_dense_reference(tokens, dense)
    torch.cuda.synchronize()
    _compare(got, ref, tokens, nll=True)
