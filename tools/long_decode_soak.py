"""Long-context decode soak (GPU box): the 7B-shape engine decodes N tokens greedily three ways -- multi-token graphs (run()),
single-token graphs (step()), eager launches -- from the same start; the token sequences must be identical.  Crosses both
attention-split thresholds (256, 1536)."""
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qeft_amd.llama import LLAMA2_7B, DecodeEngine, QuantLlama

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
dev = torch.device("cuda:0")
model = QuantLlama(dataclasses.replace(LLAMA2_7B, max_seq=2048), dev, seed=0, fast_init=True)


def decode(mode):
    eng = DecodeEngine(model, use_graph=mode != "eager")
    eng.greedy = True
    eng.reset()
    eng.tok.fill_(1)
    toks = []
    t0 = time.time()
    if mode == "multi":
        done = 0
        while done < N:
            n = min(64, N - done)
            eng.run(n)
            done += n
            toks.append(int(eng.tok.item()))        # every 64th token
    else:
        for i in range(N):
            eng.step()
            if i % 64 == 63 or i == N - 1:
                toks.append(int(eng.tok.item()))
    torch.cuda.synchronize()
    dt = time.time() - t0
    splits = sorted({k[0] for k in eng.graphs}) if eng.graphs else []
    print(f"{mode:6s}: {N} tokens in {dt:.2f} s ({N / dt:.0f} tokens/s incl. captures), attention splits captured {splits}, "
          f"checkpoints {toks[:4]} ... {toks[-2:]}", flush=True)
    return toks


a, b, c = decode("multi"), decode("single"), decode("eager")
print("multi == single:", a == b, " single == eager:", b == c, flush=True)
sys.exit(0 if a == b == c else 1)
