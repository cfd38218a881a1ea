import os, sys, torch
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.getcwd())
from bench import _event_time_us
from qeft_amd import qeft_cuda
dev = torch.device("cuda:0")
for n, k in ((4096, 4096), (11008, 4096), (4096, 11008)):
    for m in (2048, 512):
        x = torch.randn(m, k, device=dev).half(); dy = torch.randn(m, n, device=dev).half()
        t = _event_time_us(lambda: qeft_cuda.grad_oweight(dy, x, 128), 50, dev)
        print(f"N={n} K={k} M={m}: d(oweight) {t:.1f} us", flush=True)
