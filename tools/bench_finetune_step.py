"""One fine-tune step of a QuantLinear (forward + backward with the fp16 outlier slice trainable), per kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd import qeft_cuda  # noqa: E402


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


dev = "cuda:0"
for n, k in ((4096, 4096), (11008, 4096), (4096, 11008)):
    for m in (2048, 512):
        r, g = 128, 128
        qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
        sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
        sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
        ow = (torch.randn(n, r, device=dev) * 0.02).half()
        x = torch.randn(m, k, device=dev).half()
        dy = torch.randn(m, n, device=dev).half()
        fw = t(lambda: qeft_cuda.gemm_4bit_qeft(x, qw, sc, sz, ow))
        dx = t(lambda: qeft_cuda.gemm_4bit_dx(dy, qw, sc, sz, ow))
        dw = t(lambda: qeft_cuda.grad_oweight(dy, x, r))
        print(f"N={n} K={k} M={m}: forward {fw:7.1f} us | dX {dx:7.1f} us | d(oweight) {dw:7.1f} us | step {fw + dx + dw:7.1f} us", flush=True)
