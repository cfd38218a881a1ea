"""One GEMM shape, a few launches: the target of rocprofv3 kernel-trace / PMC passes.  python tools/gemm_one.py N K M [bwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd import qeft_cuda  # noqa: E402

n, k, m = (int(v) for v in sys.argv[1:4])
bwd = len(sys.argv) > 4
dev, r, g = "cuda:0", 128, 128
ws = []
for i in range(4):
    qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
    sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
    sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
    ow = (torch.randn(n, r, device=dev) * 0.02).half()
    ws.append((qw, sc, sz, ow))
x = torch.randn(m, k, device=dev).half()
dy = torch.randn(m, n, device=dev).half()
for _ in range(int(os.environ.get("GEMM_ROUNDS", "5"))):      # x 4 weight sets; ~100 for steady-state clocks
    for qw, sc, sz, ow in ws:
        if bwd:
            qeft_cuda.gemm_4bit_dx(dy, qw, sc, sz, ow)
        else:
            qeft_cuda.gemm_4bit_qeft(x, qw, sc, sz, ow)
torch.cuda.synchronize()
print("done")
