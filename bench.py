#!/usr/bin/env python3
"""Decode benchmark of the MI355X packed-weight quantized linear path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched through torch.distributed.run)

Workload (BASELINE.json configs[1]): Llama-2-7B shapes, w4 g128 r128, batch 1.  A "step" is ONE decode token of
the whole model: token begin (embedding row + rotary row), 32 layers x 5 launches (q|k|v grouped GEMV with the RMSNorm
folded in, rotary+KV-append+attention writing in o_proj's column order, o_proj GEMV + residual, gate|up grouped GEMV
with the RMSNorm folded in, down_proj GEMV with SiLU*up folded in + residual), final norm, fp16 lm_head, token end
(greedy argmax, pos += 1) — captured once into a hipGraph and replayed.  Weights are synthetic (seeded), inputs
are resident in HBM; the timed region is K graph replays bracketed by barrier + synchronize.

N > 1: every quantized linear is row-sharded over the N ranks and one RCCL all-gather per linear rebuilds the
activations (strong scaling: the model is fixed).

The same JSON line carries
  roofline      for the dominant kernel (the W4 GEMV): algorithmic bytes of all GEMV launches of one token /
                their HIP-event-timed duration (the token's GEMV launches replayed back to back from a graph on the
                launch stream; the event time includes the ~1.3 us inter-kernel gaps, as rocprofv3's kernel-trace
                durations on this stack do -- the two agree), against 8 TB/s.
                roofline.traffic = HBM bytes per GEMV launch from the PMC counters: a short child run of this script
                under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (started before this process touches the GPU),
                FETCH_SIZE x 1024 x 2 (gfx950 correction); null if rocprofv3 is not available.
                roofline.per_launch_kind = the same timing, one GEMV of the layer at a time.
  cpu_baseline  the reference's CPU path (dense nn.Linear on the dequantised weights, oracle/) timed on the host cores
                for the 7 linears of one layer, scaled to a token.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def cpu_baseline(shape, reps=8):
    """Reference CPU path on a bounded sample: one decoder layer's 7 linears, m = 1, fp32 nn.Linear on the
    dequantised weights.  Returns tokens/s-equivalent = 1 / (n_layers * sum_t)."""
    import numpy as np
    from oracle import qeft_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    cores = torch.get_num_threads()
    shapes = [(shape.hidden, shape.hidden)] * 4 + [(shape.inter, shape.hidden)] * 2 + [(shape.hidden, shape.inter)]
    uniq = {}
    total = 0.0
    for (n, k) in shapes:
        if (n, k) not in uniq:
            bufs = O.make_layer(n, k, shape.n_out, shape.group_size, seed=n + k)
            w = O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"], shape.group_size)
            lin = torch.nn.Linear(k, n, bias=False)
            lin.weight.data = torch.from_numpy(w)
            x = torch.from_numpy(O.make_activation(1, k, shape.n_out, seed=1).astype(np.float32))
            with torch.no_grad():
                for _ in range(2):
                    lin(x)
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    lin(x)
                    ts.append(time.perf_counter() - t0)
            uniq[(n, k)] = sorted(ts)[len(ts) // 2]
        total += uniq[(n, k)]
    return {"value": round(1.0 / (shape.n_layers * total), 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"7 linears of one {shape.name} layer, m=1, fp32 torch.nn.Linear on oracle-dequantised weights, "
                      f"median of {reps}; linears only, x{shape.n_layers} layers"}


def hbm_traffic_per_gemv_launch(model_flag, bits):
    """HBM bytes per GEMV launch from the PMC counters, collected exactly as MI355X_MICROARCH.md prescribes: a run of
    its own under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (nothing else traced), a short CHILD run of this script
    started before this process touches the GPU; FETCH_SIZE is in KiB and, on gfx950, reports half of the bytes of a
    wide streaming read -> x 1024 x 2.  Returns None if the profiler is not there or anything goes wrong."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    # already running under a profiler (e.g. `rocprofv3 --stats -- python3 bench.py`): do not nest another one
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="qeft_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", tmp, "-o", "run", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "4", "--model", model_flag,
               "--bits", str(bits), "--no-cpu-baseline", "--no-traffic", "--no-per-kind"]
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
        vals = []
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == "FETCH_SIZE" and "gemv_w4_mfma" in r.get("Kernel_Name", ""):
                    vals.append(float(r["Counter_Value"]))
        if not vals:
            return None
        return {"bytes_per_launch": int(sum(vals) / len(vals) * 1024 * 2), "launches_sampled": len(vals)}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--bits", type=int, default=4, choices=[3, 4],
                    help="4: the reference's w4 checkpoint layout (BASELINE configs 1-4, the default); "
                         "3: this build's 3-bit extension layout (config 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the PMC child run that fills roofline.traffic")
    ap.add_argument("--no-per-kind", action="store_true", help="skip the per-launch-kind timing graphs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # the PMC pass is a child process of its own and has to start BEFORE this process initialises the GPU
    traffic = None
    if world == 1 and "RANK" not in os.environ and not args.no_traffic:
        traffic = hbm_traffic_per_gemv_launch(args.model, args.bits)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    # QEFT_BENCH_FORCE_TP=1: rehearse the multi-GPU launch sequence (sharded linears + RCCL all-gathers inside the
    # graph) with a group of one rank on a one-GPU box
    force_tp = os.environ.get("QEFT_BENCH_FORCE_TP") == "1" and "RANK" in os.environ
    if world > 1 or force_tp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        group = dist.group.WORLD

    from qeft_amd.llama import LLAMA2_7B, LLAMA2_13B, DecodeEngine, QuantLlama, tiny_shape
    import dataclasses
    base = {"7b": LLAMA2_7B, "13b": LLAMA2_13B, "tiny": tiny_shape(n_layers=4, hidden=512, inter=1024, n_heads=4, vocab=1024)}[args.model]
    shape = dataclasses.replace(base, max_seq=max(512, (args.warmup + args.steps + 8 + 15) // 16 * 16), bits=args.bits)

    t_build = time.time()
    model = QuantLlama(shape, dev, seed=0, fast_init=True)
    eng = DecodeEngine(model, use_graph=not args.no_graph, tp_group=group)
    eng.greedy = True
    torch.cuda.synchronize(dev)
    t_build = time.time() - t_build

    graph_ok = not args.no_graph
    if graph_ok:
        try:
            eng.capture()
            eng.precapture(args.warmup + args.steps + 1)     # one graph per attention split the run will reach
        except Exception as e:  # e.g. a collective that cannot be captured: fall back to eager launches
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            eng.use_graph, eng.graph, eng.graphs, graph_ok = False, None, {}, False
            torch.cuda.synchronize(dev)

    # ---- warm-up: W tokens (they also build the KV-cache context, cf. benchmark.py ctx 64)
    eng.reset()
    eng.tok.fill_(1)
    for _ in range(args.warmup):
        eng.step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_tok = int(eng.tok.item())

    # ---- roofline of the dominant kernel: the token's GEMV launches alone, back to back, HIP-event timed
    roof = None
    try:
        g2 = eng.capture(linears_only=True) if graph_ok else None
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            g2.replay() if g2 is not None else eng._launch_token(True)
        torch.cuda.synchronize(dev)
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(reps):
            g2.replay() if g2 is not None else eng._launch_token(True)
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        launches = 4 * shape.n_layers
        us_per_launch = e0.elapsed_time(e1) * 1e3 / (reps * launches)
        bytes_per_launch = eng.weight_bytes_per_token() / launches
        achieved = bytes_per_launch / us_per_launch / 1e3  # GB/s
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": "qeft::gemv_w4_mfma_group_kernel / gemv_w4_mfma_kernel (4 launches per layer)",
                "bytes_per_launch": int(bytes_per_launch), "us_per_launch": round(us_per_launch, 3)}
        # the same, one GEMV of the layer at a time (32 launches per replay): per-kernel rates for DESIGN.md / rocprof
        if traffic is not None:
            roof["traffic"] = traffic["bytes_per_launch"]
            roof["traffic_note"] = (f"FETCH_SIZE x 1024 x 2 (gfx950 correction), mean over {traffic['launches_sampled']} "
                                    "GEMV launches of a separate `rocprofv3 --pmc FETCH_SIZE --kernel-trace` child run")
        if g2 is not None and world == 1 and not args.no_per_kind:
            per = {}
            lin0 = eng.lin[0]
            parts = {"qkv": ("q", "k", "v"), "o": ("o",), "gu": ("g", "u"), "d": ("d",)}
            for tag, names in parts.items():
                gk = eng.capture(linears_only=True, only=tag)
                for _ in range(3):
                    gk.replay()
                torch.cuda.synchronize(dev)
                e0.record(torch.cuda.current_stream(dev))
                for _ in range(reps):
                    gk.replay()
                e1.record(torch.cuda.current_stream(dev))
                torch.cuda.synchronize(dev)
                us = e0.elapsed_time(e1) * 1e3 / (reps * shape.n_layers)
                nbytes = 0
                for nm in names:
                    l = lin0[nm]
                    n_, k_, r_, g_ = l.outfeatures, l.infeatures, l.outlierfeatures, l.group_size
                    nbytes += n_ * (k_ - r_) * l.bits // 8 + 2 * (k_ // g_) * n_ * 2 + n_ * r_ * 2 + 2 * k_ + 2 * n_
                per[tag] = {"us": round(us, 2), "bytes": int(nbytes), "GB/s": round(nbytes / us / 1e3, 1)}
            roof["per_launch_kind"] = per
    except Exception as e:  # never lose the headline number because of the side measurement
        if rank == 0:
            print(f"[bench] roofline pass failed: {type(e).__name__}: {e}", file=sys.stderr)

    if rank == 0:
        ms = dt * 1e3 / args.steps
        out = {
            "metric": "decode tokens/sec, Llama-2-7B w4 g128 r128" if (args.model == "7b" and args.bits == 4)
            else f"decode tokens/sec, {shape.name} w{args.bits} g128 r128",
            "value": round(args.steps / dt, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{shape.name} w{args.bits} g{shape.group_size} r{shape.n_out} full decode step, batch 1, "
                                   f"greedy, KV context {args.warmup}..{args.warmup + args.steps} tokens",
                       "layers": shape.n_layers, "hipgraph": graph_ok,
                       "parallelism": f"tp{world} row-sharded QuantLinear + all-gather" if group is not None else "single GPU",
                       "build_s": round(t_build, 1), "last_token": last_tok},
        }
        if roof:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(shape)
            except Exception as e:
                print(f"[bench] cpu baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
        print(json.dumps(out), flush=True)
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
