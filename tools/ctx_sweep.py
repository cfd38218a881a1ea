"""Decode tokens/s of the 7B engine against the KV context length (the bench's context is 128..256): 64 graph-replayed tokens
from each starting position; the caches hold random rows (timing only)."""
import os, sys, time, dataclasses, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd.llama import LLAMA2_7B, DecodeEngine, QuantLlama
dev = torch.device("cuda:0")
shape = dataclasses.replace(LLAMA2_7B, max_seq=4096)
model = QuantLlama(shape, dev, seed=0, fast_init=True)
eng = DecodeEngine(model, use_graph=True)
eng.greedy = True
for kc, vc in zip(eng.kc, eng.vc):
    kc.normal_(); vc.normal_()
kv_bytes = lambda p: 2 * p * shape.n_kv_heads * 128 * 2 * shape.n_layers
for ctx in (128, 512, 1024, 2048, 3072, 3968):
    eng.reset(); eng.set_position(ctx); eng.tok.fill_(1)
    eng.precapture(ctx + 80)
    for _ in range(8):
        eng.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 64
    for _ in range(n):
        eng.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"context {ctx:5d}: {1 / dt:7.1f} tokens/s  {dt * 1e3:.3f} ms/token  (KV read per token {kv_bytes(ctx) / 1e6:7.1f} MB, split {eng._split_for(ctx)})", flush=True)
