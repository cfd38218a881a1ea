"""Lab: the fused final-norm + lm_head launch against torch's fp16 matmul (hipBLASLt) on the Llama-2 head, cold weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd import _lib  # noqa: E402

dev = "cuda:0"
lib = _lib.lib()
hidden, vocab = 4096, 32000
ws = [(torch.randn(vocab, hidden, device=dev) * 0.02).half() for _ in range(4)]
h = torch.randn(hidden, device=dev)
g = torch.ones(hidden, device=dev).half()
hn = torch.randn(1, hidden, device=dev).half()
out = torch.empty(vocab, dtype=torch.float16, device=dev)
out2 = torch.empty(1, vocab, dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream


def t(fn, n=40):
    for _ in range(20):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


us = t(lambda i: _lib.check(lib.qeft_lm_head_f16(h.data_ptr(), g.data_ptr(), ws[i % 4].data_ptr(), out.data_ptr(), hidden, vocab, 1e-5, st)))
print(f"qeft_lm_head_f16: {us:.1f} us  {vocab * hidden * 2 / us / 1e3:.0f} GB/s")
us = t(lambda i: torch.matmul(hn, ws[i % 4].t(), out=out2))
print(f"torch.matmul    : {us:.1f} us  {vocab * hidden * 2 / us / 1e3:.0f} GB/s")
