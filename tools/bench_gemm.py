"""Micro-benchmark of the W4 MFMA GEMM (prefill / fine-tune shapes).  Run on the GPU box."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd import qeft_cuda  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="2048,256,64,16")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--reps", type=int, default=25)
    ap.add_argument("--warm-ms", type=float, default=20.0)
    ap.add_argument("--bwd", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    for shp in a.shapes.split(","):
        n, k = [int(v) for v in shp.split("x")]
        r, g = 128, 128
        ws = []
        for i in range(a.layers):
            qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
            sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
            sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
            ow = (torch.randn(n, r, device=dev) * 0.02).half()
            ws.append((qw, sc, sz, ow))
        for m in [int(v) for v in a.ms.split(",")]:
            x = torch.randn(m, k, device=dev).half()
            dy = torch.randn(m, n, device=dev).half()

            def run():
                for qw, sc, sz, ow in ws:
                    if a.bwd:
                        qeft_cuda.gemm_4bit_dx(dy, qw, sc, sz, ow)
                    else:
                        qeft_cuda.gemm_4bit_qeft(x, qw, sc, sz, ow)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()               # steady state: a GPU coming out of idle runs its first ms at low clocks
            while time.perf_counter() - t0 < a.warm_ms * 1e-3:
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (a.reps * a.layers)
            print(f"{'dX ' if a.bwd else ''}N={n} K={k} M={m}: {us:9.1f} us  {2.0 * m * n * k / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
