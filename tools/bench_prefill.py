"""Prefill (BASELINE config 3): one batched pass of a prompt through the packed model (MFMA GEMM path), tokens/s."""
import argparse
import dataclasses
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd.llama import LLAMA2_7B, LLAMA2_13B, QuantLlama, prefill  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--model", default="7b", choices=["7b", "13b"])
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shape = dataclasses.replace({"7b": LLAMA2_7B, "13b": LLAMA2_13B}[a.model], max_seq=max(512, a.seq))
    model = QuantLlama(shape, dev, seed=0, fast_init=True)
    tokens = torch.randint(0, shape.vocab, (a.seq,), generator=torch.Generator().manual_seed(0)).to(dev)
    prefill(model, tokens)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        prefill(model, tokens)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    lin_flops = 2.0 * a.seq * shape.n_layers * (4 * shape.hidden * shape.hidden + 3 * shape.hidden * shape.inter)
    print(f"{shape.name} w4 g128 r128 prefill seq={a.seq}: {t * 1e3:.1f} ms  {a.seq / t:.0f} tokens/s  "
          f"(quantized linears {lin_flops / 1e12:.1f} TFLOP -> {lin_flops / t / 1e12:.0f} TFLOP/s incl. attention, norms, lm_head)")


if __name__ == "__main__":
    main()
