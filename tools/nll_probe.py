"""Lab: which tokens carry the |dNLL| of the 7B first-tokens parity test (tests/test_gpu_engine_7b.py)?"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd.llama import LLAMA2_7B, DecodeEngine, QuantLlama  # noqa: E402

DEV = "cuda:0"
model = QuantLlama(dataclasses.replace(LLAMA2_7B, max_seq=512), DEV, seed=0, fast_init=True)
dense = model.dense_weights()
for seed, T in ((1, 96), (2, 96), (3, 96), (1, 192)):
    tokens = torch.randint(0, model.shape.vocab, (T,), generator=torch.Generator().manual_seed(seed)).to(DEV)
    got = DecodeEngine(model, use_graph=False).teacher_forced_logits(tokens)
    ref = model.forward_dense_reference(tokens, dense)
    ce = lambda lg: torch.nn.functional.cross_entropy(lg[:-1].float(), tokens[1:], reduction="none")
    d = (ce(got) - ce(ref))
    top = d.abs().topk(5)
    print(f"seed {seed} T {T}: mean dNLL {d.mean().item():+.3e}  std {d.std().item():.3e}  max|dlogit|/max {((got - ref).abs().max() / ref.abs().max()).item():.3e}")
    print("   largest per-token:", [(int(i), f"{d[i].item():+.2e}") for i in top.indices])
    print("   mean over halves:", f"{d[:T // 2].mean().item():+.2e}", f"{d[T // 2:].mean().item():+.2e}")
