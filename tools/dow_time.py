"""d(oweight) kernel time per launch on the fine-tune shapes: 10 launches captured in one hipGraph (no host time in
the figure).  QEFT_DOW_BN / QEFT_DOW_NW force a variant."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import _event_time_us
from qeft_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.lib()
for n, k in ((4096, 4096), (11008, 4096), (4096, 11008)):
    for m in (2048, 512):
        x = torch.randn(m, k, device=dev).half(); dy = torch.randn(m, n, device=dev).half()
        out = torch.empty(n, 128, dtype=torch.float32, device=dev)
        st = torch.cuda.Stream(dev)
        def launch():
            _lib.check(lib.qeft_grad_oweight(dy.data_ptr(), x.data_ptr(), out.data_ptr(), m, n, k, 128,
                                             torch.cuda.current_stream(dev).cuda_stream))
        with torch.cuda.stream(st):
            launch()
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(10):
                    launch()
            t = _event_time_us(g.replay, 20, dev) / 10
        print(f"N={n} K={k} M={m}: d(oweight) {t:.1f} us ({lib.qeft_last_variant().decode()})", flush=True)
