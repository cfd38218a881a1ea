"""Micro-benchmark of the decode GEMV on Llama shapes (run on the GPU box).
Cycles over `--layers` distinct weight sets so the 256 MiB Infinity Cache cannot serve the stream."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd import _lib, qeft_cuda  # noqa: E402


def algo_bytes(n, k, m, r=128, g=128):
    return n * (k - r) // 2 + 2 * (k // g) * n * 2 + n * r * 2 + 2 * m * k + 2 * m * n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008")
    a = ap.parse_args()
    dev = "cuda:0"
    lib = _lib.lib()
    for shp in a.shapes.split(","):
        n, k = [int(v) for v in shp.split("x")]
        r, g, m = 128, 128, a.m
        ws = []
        for i in range(a.layers):
            qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
            sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
            sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
            ow = (torch.randn(n // 2, 2 * r, device=dev) * 0.02).half()
            ws.append((qw, sc, sz, ow))
        x = torch.randn(m, k, device=dev).half()
        y = torch.empty(m, n, device=dev, dtype=torch.float16)
        def run_all():
            st = torch.cuda.current_stream().cuda_stream
            for qw, sc, sz, ow in ws:
                lib.qeft_gemv_w4_qeft(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), sz.data_ptr(), ow.data_ptr(),
                                      y.data_ptr(), m, n, k, g, r, st)
        for _ in range(3):
            run_all()
        torch.cuda.synchronize()
        # eager, back-to-back
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run_all()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (a.reps * a.layers)
        # graph
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            run_all()
        gph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        usg = e0.elapsed_time(e1) * 1e3 / (a.reps * a.layers)
        b = algo_bytes(n, k, m)
        print(f"N={n} K={k} m={m}: eager {us:.2f} us/launch ({b / us / 1e3:.0f} GB/s)  "
              f"graph {usg:.2f} us/launch ({b / usg / 1e3:.0f} GB/s)  bytes={b}", flush=True)


if __name__ == "__main__":
    main()
