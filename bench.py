#!/usr/bin/env python3
"""Decode benchmark of the MI355X packed-weight quantized linear path.

    python bench.py --gpus N --steps K --warmup W

N > 1 may be started either way: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the
driver's form: RANK / WORLD_SIZE in the environment) or plainly as `python bench.py --gpus N`, in which case this process
-- before it makes any GPU call -- starts the N ranks itself as a child `torch.distributed.run` and relays rank 0's line.

Workload (BASELINE.json configs[1]): Llama-2-7B shapes, w4 g128 r128, batch 1.  A "step" is ONE decode token of the whole
model: token begin (embedding row + rotary row), 32 layers x 5 launches (q|k|v GEMV, rotary + KV append + attention,
o_proj GEMV + residual, gate|up GEMV + SiLU*up, down_proj GEMV + residual; the RMSNorms ride on the producers' epilogues),
final norm, fp16 lm_head, token end (greedy argmax, pos += 1) -- captured once into a hipGraph and replayed.  Weights are
synthetic (seeded), inputs are resident in HBM.  Protocol of the reference's generation benchmark (benchmark.py:118-119,
293-338): a 64-token context is built first (untimed, whatever --warmup is), then W untimed warm-up tokens, then the timed
region: K graph replays bracketed by barrier + synchronize, max over ranks.

The same JSON line carries
  roofline          the dominant kernel (the W4 GEMV): algorithmic bytes of all GEMV launches of one token / their
                    HIP-event-timed duration (the token's GEMV launches replayed back to back from a graph on the launch
                    stream; the event time includes the inter-kernel gaps, as rocprofv3's kernel-trace durations on this
                    stack do), against 8 TB/s.  roofline.traffic: HBM bytes per GEMV launch from a child run under
                    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (x 1024 x 2, gfx950 correction).  per_launch_kind: the same
                    timing, one GEMV of the layer at a time.
  latency_protocol  the reference's per-token protocol (main.py:340-371, benchmark.py:293-338): synchronize after every
                    token, median / min seconds, tokens/s = 1 / median -- for graph replays and for eager launches.
  prefill_2048      BASELINE config 3: the M = 2048 MFMA GEMM per 7B shape (us, TFLOP/s, fraction of the 2.5 PFLOP/s dense
                    fp16 peak, kernel variant) and a whole 2048-token prompt pass.
  finetune_step     BASELINE config 5 (w4 operands): forward + dX + d(oweight) of one layer at M = 2048 per shape.
  cpu_baseline      the reference's CPU path (dense nn.Linear on the dequantised weights, oracle/) timed on the host cores:
                    fp32 and fp16 (the reference's dtype, recon.py:573), m = 1 on the three 7B shapes and M = 2048 on
                    4096 x 4096; `value` = fp32 tokens/s-equivalent of the linears.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline is 2:1 sparse)
CONTEXT = 64               # benchmark.py:118: tokens in the KV cache before the measured generation starts
SHAPES_7B = ((4096, 4096), (11008, 4096), (4096, 11008))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "70b", "tiny"],
                    help="7b: the BASELINE headline (default); 13b: BASELINE config 4's model; 70b: the Llama-2-70B shapes on one GPU "
                         "(36 GB of packed weights; not a BASELINE config)")
    ap.add_argument("--bits", type=int, default=4, choices=[3, 4],
                    help="4: the reference's w4 checkpoint layout (BASELINE configs 1-4, the default); "
                         "3: this build's 3-bit extension layout (config 5)")
    ap.add_argument("--ckpt", default=None,
                    help="decode a packed checkpoint in the reference's format (save_model, modelutils.py:248-268; HF key names) "
                         "instead of the synthetic weights: QuantLlama.from_packed; the shape is read off the tensors")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the PMC child run that fills roofline.traffic")
    ap.add_argument("--collective", choices=("auto", "rccl", "oneshot"), default="auto",
                    help="N > 1: the layer's all-reduce -- RCCL, the one-shot IPC kernel (qeft_amd/oneshot.py), or auto = one-shot if it "
                         "passes its self-check on every rank, else RCCL")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the PMC child: two tokens' GEMV launches, no timing
    ap.add_argument("--no-per-kind", action="store_true", help="skip the per-launch-kind timing graphs")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip latency_protocol / prefill_2048 / finetune_step (the decode line and its roofline only)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------- N > 1: self-launch
def launch_ranks(args):
    """Parent of a plain `python bench.py --gpus N`: start one rank per GPU as a fresh `torch.distributed.run` child and
    hand its exit code back.  Nothing here touches the GPU (torch.cuda.device_count() does not initialise it), and the
    ranks are child processes, never an exec of this one."""
    import torch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        # rehearsal on a box with fewer GPUs than ranks: the ranks share the devices, collectives go through gloo (RCCL
        # refuses two ranks on one device) and run eagerly.  Correctness of the sharded launch sequence, not a measurement.
        env["QEFT_BENCH_SHARED_GPU"] = str(max(ndev, 1))
        print(f"[bench] {ndev} GPU(s) visible for --gpus {args.gpus}: rehearsal with shared devices over gloo", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(shape, budget_s=24.0):
    """Reference CPU path on a bounded sample: dense torch.nn.Linear over the oracle-dequantised
    weights of the 7B shapes, m = 1 (decode) for all three and M = 2048 (prefill) for 4096 x 4096, in fp32 and in fp16
    (the reference's dtype).  The time budget bounds the repetitions, never the problem."""
    import numpy as np
    import torch
    from oracle import qeft_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    cores = torch.get_num_threads()
    shapes = sorted({(shape.hidden, shape.hidden), (shape.inter, shape.hidden), (shape.hidden, shape.inter)})
    counts = {(shape.hidden, shape.hidden): 4, (shape.inter, shape.hidden): 2, (shape.hidden, shape.inter): 1}
    t_end = time.perf_counter() + budget_s
    per_shape, total = [], {"fp32": 0.0, "fp16": 0.0}

    def timed(lin, x, max_reps, slice_s):
        with torch.no_grad():
            lin(x)
            ts, t_stop = [], time.perf_counter() + slice_s
            while len(ts) < max_reps and (len(ts) < 2 or time.perf_counter() < t_stop):
                t0 = time.perf_counter()
                lin(x)
                ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], len(ts)

    for (n, k) in shapes:
        bufs = O.make_layer(n, k, shape.n_out, shape.group_size, seed=n + k)
        w = torch.from_numpy(O.dequant_dense(bufs["qweight"], bufs["scales"], bufs["scaled_zeros"], bufs["oweight"],
                                             shape.group_size))
        rec = {"shape": f"{n}x{k}"}
        for m in ((1, 2048) if (n, k) == (shape.hidden, shape.hidden) else (1,)):
            x32 = torch.from_numpy(O.make_activation(m, k, shape.n_out, seed=1).astype(np.float32))
            for dt, tag in ((torch.float32, "fp32"), (torch.float16, "fp16")):
                lin = torch.nn.Linear(k, n, bias=False, dtype=dt)
                lin.weight.data = w.to(dt)
                left = max(t_end - time.perf_counter(), 0.5)
                t, reps = timed(lin, x32.to(dt), 20 if m == 1 else 3, min(left / 4, 2.0 if m == 1 else 4.0))
                rec[f"m{m}_{tag}_us"] = round(t * 1e6, 1)
                rec[f"m{m}_{tag}_GBps"] = round(n * k * (4 if dt == torch.float32 else 2) / t / 1e9, 2)
                rec[f"m{m}_{tag}_reps"] = reps
                if m == 1:
                    total[tag] += counts[(n, k)] * t
        per_shape.append(rec)
    return {"value": round(1.0 / (shape.n_layers * total["fp32"]), 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "fp16_value": round(1.0 / (shape.n_layers * total["fp16"]), 3),
            "sample": f"dense torch.nn.Linear on oracle-dequantised weights of the 3 {shape.name} shapes, m=1 (median of <=20 "
                      f"reps) and M=2048 on {shape.hidden}x{shape.hidden} (<=3 reps), fp32 and fp16; value = 1/(layers x sum "
                      "of the 7 linears, fp32), fp16_value the same in the reference's dtype",
            "per_shape": per_shape}


# ---------------------------------------------------------------------------------------------------- PMC child run
def hbm_traffic_per_gemv_launch(model_flag, bits, ckpt=None):
    """HBM bytes per GEMV launch from the PMC counters, collected as MI355X_MICROARCH.md prescribes: a run of its own
    under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (nothing else traced), a short CHILD run of this script started
    before this process touches the GPU; FETCH_SIZE is in KiB and, on gfx950, reports half of the bytes of a wide
    streaming read -> x 1024 x 2.  None if the profiler is not there or anything goes wrong."""
    import csv
    import glob
    import shutil
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
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--model", model_flag, "--bits", str(bits)]
        if ckpt:        # the counters must describe the launches of the model the line is about
            cmd += ["--ckpt", os.path.abspath(ckpt)]
        env = dict(os.environ, TMPDIR="/tmp")
        res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
        if res.returncode != 0:
            print(f"[bench] PMC child run failed (rc {res.returncode}): {res.stderr.decode(errors='replace')[-600:]}", file=sys.stderr)
            return None
        vals = []
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == "FETCH_SIZE" and ("gemv_v3" in r.get("Kernel_Name", "") or "gemv_w" in r.get("Kernel_Name", "")):
                    vals.append(float(r["Counter_Value"]))
        if not vals:
            print("[bench] PMC child run: no FETCH_SIZE rows for the GEMV kernel", file=sys.stderr)
            return None
        return {"bytes_per_launch": int(sum(vals) / len(vals) * 1024 * 2), "launches_sampled": len(vals)}
    except Exception as e:
        print(f"[bench] PMC child run failed: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------- GEMM sub-records
def _event_time_us(fn, reps, dev, warm_ms=20.0):
    """Steady-state time of fn() in us: the launches are timed after `warm_ms` of the same work.  A GPU coming out of idle
    runs its first milliseconds at low clocks -- a 40-launch measurement straight after the operand set-up read 71 us for a
    GEMM that runs 47-53 us per launch from the third millisecond on (profiles/r02_gemm_clock.txt)."""
    import time
    import torch
    fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(reps):
        fn()
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e3 / reps


def gemm_records(dev, m=2048, layers=4, reps=25):
    """prefill_2048.per_shape and finetune_step: forward / dX / d(oweight) of the three 7B shapes at M = 2048, `layers`
    distinct weight sets cycled so that no launch finds its weights in L2, random (gaussian) activations."""
    import torch
    from qeft_amd import _lib, qeft_cuda
    fwd_recs, ft_recs = [], []
    r, g = 128, 128
    for (n, k) in SHAPES_7B:
        ws = []
        for _ in range(layers):
            qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
            sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
            sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
            ow = (torch.randn(n, r, device=dev) * 0.02).half()
            ws.append((qw, sc, sz, ow))
        x = torch.randn(m, k, device=dev).half()
        dy = torch.randn(m, n, device=dev).half()
        flops = 2.0 * m * n * k
        variants = {}

        def fwd():
            for qw, sc, sz, ow in ws:
                qeft_cuda.gemm_4bit_qeft(x, qw, sc, sz, ow)
            variants["fwd"] = _lib.last_variant()

        def dx():
            for qw, sc, sz, ow in ws:
                qeft_cuda.gemm_4bit_dx(dy, qw, sc, sz, ow)
            variants["dx"] = _lib.last_variant()

        def dow():
            for _ in ws:
                qeft_cuda.grad_oweight(dy, x, r)
            variants["dow"] = _lib.last_variant()

        t_f = _event_time_us(fwd, reps, dev) / layers
        t_x = _event_time_us(dx, reps, dev) / layers
        t_w = _event_time_us(dow, reps, dev) / layers
        # context, not a baseline: the library's dense fp16 GEMM of the same shape (hipBLASLt through torch; 4x the weight bytes)
        wd = [torch.randn(n, k, device=dev).half() for _ in range(layers)]

        def dense():
            for w_ in wd:
                torch.matmul(x, w_.t())
        t_d = _event_time_us(dense, reps, dev) / layers
        del wd
        fwd_recs.append({"shape": f"{n}x{k}", "us": round(t_f, 1), "TFLOPs": round(flops / t_f / 1e6, 1),
                         "frac_of_peak": round(flops / t_f / 1e6 / MFMA_PEAK_TFLOPS, 4), "variant": variants["fwd"],
                         "hipblaslt_dense_fp16_us": round(t_d, 1), "hipblaslt_dense_fp16_TFLOPs": round(flops / t_d / 1e6, 1)})
        # d(oweight) does not depend on dX: on a second stream beside it (what an autograd engine with two streams, or a fused
        # backward, gets) -- reported next to the serial sum
        side = torch.cuda.Stream(dev)

        def dx_dow_overlapped():
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in ws:
                    qeft_cuda.grad_oweight(dy, x, r)
            for qw, sc, sz, ow in ws:
                qeft_cuda.gemm_4bit_dx(dy, qw, sc, sz, ow)
            cur.wait_stream(side)
        t_xw = _event_time_us(dx_dow_overlapped, reps, dev) / layers
        # BASELINE config 5 names the 3-bit pack: the same step on the 3-bit stream itself (round 3: the loader-wave tiers unpack
        # the 12-byte lane records in registers; no expansion pass)
        w3 = [(torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16, (k - r) // 128 * 192), dtype=torch.int32, device=dev), sc, sz, ow)
              for (_, sc, sz, ow) in ws]

        def fwd3():
            for q3, sc, sz, ow in w3:
                qeft_cuda.gemm_3bit_qeft(x, q3, sc, sz, ow)
            variants["fwd_w3"] = _lib.last_variant()

        def dx3():
            for q3, sc, sz, ow in w3:
                qeft_cuda.gemm_3bit_dx(dy, q3, sc, sz, ow, k)
            variants["dx_w3"] = _lib.last_variant()
        t_f3 = _event_time_us(fwd3, reps, dev) / layers
        t_x3 = _event_time_us(dx3, reps, dev) / layers
        del w3
        ft_recs.append({"shape": f"{n}x{k}", "forward_us": round(t_f, 1), "dx_us": round(t_x, 1), "dow_us": round(t_w, 1),
                        "step_us": round(t_f + t_x + t_w, 1), "dx_dow_overlapped_us": round(t_xw, 1),
                        "step_overlapped_us": round(t_f + t_xw, 1),
                        "w3_forward_us": round(t_f3, 1), "w3_dx_us": round(t_x3, 1), "w3_step_us": round(t_f3 + t_x3 + t_w, 1),
                        "w3_over_w4": round((t_f3 + t_x3 + t_w) / (t_f + t_x + t_w), 4),
                        "dx_TFLOPs": round(flops / t_x / 1e6, 1), "dx_frac_of_peak": round(flops / t_x / 1e6 / MFMA_PEAK_TFLOPS, 4),
                        "step_TFLOPs": round((2 * flops + 2.0 * m * n * r) / (t_f + t_x + t_w) / 1e6, 1),
                        "variants": dict(variants)})
        del ws, x, dy
        torch.cuda.empty_cache()
    return fwd_recs, ft_recs


def boundary_gemv_records(dev, sets=8, reps=20):
    """boundary_gemv: the reference's own entry points for < 8 rows, timed as a user of the reference reaches them --
    `qeft_cuda.gemv_4bit_qeft(x, qweight, scales, scaled_zeros, oweight_interleaved, m, N, K, G)` (qlinear.py:253-263) and
    `QuantLinear.forward` (forward_outlier; forward_outlier_out_proj with its reorder_ids gather for the o_proj record) at
    m = 1, 2, 4 and 7 on the three 7B shapes.  `sets` distinct weight sets are cycled (nothing is served from L2 / MALL), the
    calls -- output allocation included -- are captured into a graph and replayed; HIP-event time per call, algorithmic
    bytes per call, and the kernel variant the call reached."""
    import torch
    from qeft_amd import _lib, qeft_cuda
    from qeft_amd.qlinear import QuantLinear, pack_oweight
    r, g = 128, 128
    recs = []
    for (n, k) in SHAPES_7B:
        mods = []
        for i in range(sets):
            ql = QuantLinear(4, k, n, False, torch.float16, r, g, True, "mlp.proj")
            ql.qweight = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
            ql.scales = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
            ql.scaled_zeros = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * ql.scales.float()).half()
            ql.oweight = (torch.randn(n, r, device=dev) * 0.02).half()
            ql.oweight_interleaved = pack_oweight(ql.oweight)
            ql.outlieridx = torch.sort(torch.randperm(k, device=dev)[:r]).values.to(torch.int32)
            ql.set_kernel()
            mods.append(ql)
        oproj = None
        if (n, k) == (4096, 4096):          # the o_proj form of the same layers: forward gathers x[:, reorder_ids] first (qlinear.py:275)
            oproj = []
            for ql in mods:
                qo = QuantLinear(4, k, n, False, torch.float16, r, g, True, "self_attn.o_proj")
                for nm in ("qweight", "scales", "scaled_zeros", "oweight", "oweight_interleaved", "outlieridx"):
                    setattr(qo, nm, getattr(ql, nm))
                qo.set_kernel()
                oproj.append(qo)
        rec = {"shape": f"{n}x{k}"}
        for m in (1, 2, 4, 7):
            x = torch.randn(m, k, device=dev).half()
            nbytes = n * (k - r) // 2 + 2 * (k // g) * n * 2 + n * r * 2 + 2 * m * k + 2 * m * n
            calls = {"gemv_4bit_qeft": lambda ql: qeft_cuda.gemv_4bit_qeft(x, ql.qweight, ql.scales, ql.scaled_zeros,
                                                                          ql.oweight_interleaved, m, n, k, g),
                     "QuantLinear.forward": lambda ql: ql.forward(x)}
            for tag, call in calls.items():
                for which, ms in (("", mods), ("_o_proj", oproj)):
                    if ms is None or (which and tag != "QuantLinear.forward"):
                        continue
                    for ql in ms:                       # warm (lazy buffers, attribute set-up) outside the capture
                        call(ql)
                    variant = _lib.last_variant()
                    torch.cuda.synchronize(dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        outs = [call(ql) for ql in ms]
                    us = _event_time_us(graph.replay, reps, dev) / len(ms)
                    del graph, outs
                    rec[f"m{m}_{tag}{which}"] = {"us": round(us, 2), "GB/s": round(nbytes / us / 1e3, 1), "variant": variant}
        recs.append(rec)
        del mods, oproj
        torch.cuda.empty_cache()
    return {"per_shape": recs, "weight_sets_cycled": sets,
            "note": "the reference's entry points as qlinear.py calls them, graph-replayed incl. the output allocation; "
                    "bytes = algorithmic bytes of the call (SURVEY 8d) with m rows of x and y"}


def mid_m_records(dev, ms=(16, 32, 64, 512, 1024), layers=4, reps=25):
    """prefill_mid_m: the forward GEMM below the M = 2048 tier -- M = 16 (a batch of 16 decoding sequences: the rows ride on the
    decode GEMV), 32 and 64 (benchmark.py's 64-token prompt: the weight-stationary tier gemm_ws.hip), 512 and 1024 (short prompts, fine-tune batches) on the three 7B shapes;
    us, TFLOP/s, fraction of the dense fp16 MFMA peak, variant."""
    import torch
    from qeft_amd import _lib, qeft_cuda
    recs, r, g = [], 128, 128
    for (n, k) in SHAPES_7B:
        ws = []
        for _ in range(layers):
            qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
            sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
            sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
            ow = (torch.randn(n, r, device=dev) * 0.02).half()
            ws.append((qw, sc, sz, ow))
        rec = {"shape": f"{n}x{k}"}
        for m in ms:
            x = torch.randn(m, k, device=dev).half()
            var = {}

            def fwd():
                for qw, sc, sz, ow in ws:
                    qeft_cuda.gemm_4bit_qeft(x, qw, sc, sz, ow)
                var["v"] = _lib.last_variant()
            fwd()                               # (allocates the split-K workspace outside the capture)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()      # replayed: two short launches per call are host-bound when issued eagerly
            with torch.cuda.graph(graph):
                fwd()
            t = _event_time_us(graph.replay, reps, dev) / layers
            del graph
            fl = 2.0 * m * n * k
            rec[f"M{m}"] = {"us": round(t, 1), "TFLOPs": round(fl / t / 1e6, 1), "frac_of_peak": round(fl / t / 1e6 / MFMA_PEAK_TFLOPS, 4),
                            "variant": var["v"]}
        recs.append(rec)
        del ws
        torch.cuda.empty_cache()
    return recs


def model_13b_record(dev, steps=128, warmup=64, base=None, bits=4):
    """model_13b: BASELINE config 4's model (Llama-2-13B shapes, w4 g128 r128) on ONE GPU with the headline protocol
    (64-token context, `warmup` untimed tokens, `steps` timed graph-replayed tokens) and the GEMV launches' rate -- the
    single-GPU point of the row-sharded curve; compact on purpose.  `base`: another model shape (the N > 1 line's replica figure)."""
    import dataclasses
    import torch
    from qeft_amd.llama import LLAMA2_13B, DecodeEngine, QuantLlama
    ctx0 = CONTEXT + warmup
    shape = dataclasses.replace(base if base is not None else LLAMA2_13B, max_seq=512, bits=bits)
    model = QuantLlama(shape, dev, seed=0, fast_init=True)
    eng = DecodeEngine(model, use_graph=True)
    eng.greedy = True
    eng.capture()
    eng.precapture(ctx0 + steps + 1)
    eng.reset()
    eng.tok.fill_(1)
    for _ in range(ctx0):
        eng.step()
    eng.run(eng.MULTI)
    eng.set_position(ctx0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.run(steps)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    g2 = eng.capture(linears_only=True)
    launches = 4 * shape.n_layers
    us = _event_time_us(g2.replay, 20, dev) / launches
    nbytes = eng.weight_bytes_per_token() / launches
    rec = {"tokens_per_s": round(steps / dt, 2), "ms_per_step": round(dt * 1e3 / steps, 4), "steps": steps,
           "gemv_us_per_launch": round(us, 3), "gemv_bytes_per_launch": int(nbytes), "gemv_GB/s": round(nbytes / us / 1e3, 1),
           "gemv_frac_of_8TB/s": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4), "workload": f"{shape.name} w{shape.bits} g{shape.group_size} r{shape.n_out}, batch 1, one GPU"}
    del eng, g2, model
    torch.cuda.empty_cache()
    return rec


# ---------------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the PMC pass is a child process of its own and has to start BEFORE this process initialises the GPU
    traffic = None
    if world == 1 and "RANK" not in os.environ and not args.no_traffic and not args.pmc_child:
        traffic = hbm_traffic_per_gemv_launch(args.model, args.bits, args.ckpt)

    import torch
    import torch.distributed as dist
    shared = int(os.environ.get("QEFT_BENCH_SHARED_GPU", "0"))     # rehearsal: ranks share `shared` devices, gloo, eager
    dev_index = local_rank % shared if shared else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    # QEFT_BENCH_FORCE_TP=1: rehearse the multi-GPU launch sequence (sharded linears + RCCL collectives inside the graph)
    # with a group of one rank on a one-GPU box
    force_tp = os.environ.get("QEFT_BENCH_FORCE_TP") == "1" and "RANK" in os.environ
    if world > 1 or force_tp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        group = dist.group.WORLD

    import dataclasses
    from qeft_amd.llama import LLAMA2_7B, LLAMA2_13B, LLAMA2_70B, DecodeEngine, QuantLlama, tiny_shape
    base = {"7b": LLAMA2_7B, "13b": LLAMA2_13B, "70b": LLAMA2_70B,
            "tiny": tiny_shape(n_layers=4, hidden=512, inter=1024, n_heads=4, vocab=1024)}[args.model]
    ctx0 = CONTEXT + args.warmup                                   # position of the first timed token
    shape = dataclasses.replace(base, max_seq=max(512, (ctx0 + args.steps + 128 + 8 + 15) // 16 * 16), bits=args.bits)

    t_build = time.time()
    if args.ckpt:
        model = QuantLlama.from_packed(args.ckpt, dev, max_seq=shape.max_seq)
        shape = model.shape
    else:
        model = QuantLlama(shape, dev, seed=0, fast_init=True)
    eng = DecodeEngine(model, use_graph=not (args.no_graph or shared), tp_group=group,
                       collective=args.collective if group is not None else "rccl")
    if shared and not args.no_graph and eng.collective == "oneshot":
        eng.use_graph = True        # the rehearsal's gloo collectives cannot be captured; the one-shot kernel can
    eng.greedy = True
    torch.cuda.synchronize(dev)
    t_build = time.time() - t_build
    if args.pmc_child:          # under `rocprofv3 --pmc` every launch costs milliseconds: just the 128 GEMV launches, twice
        for _ in range(2):
            eng._launch_token(True)
        torch.cuda.synchronize(dev)
        return

    def capture_graphs():
        """(graph_ok, error): capture the token graphs of this run; every rank ends up on the same side."""
        ok, err = eng.use_graph, None
        if ok:
            try:
                eng.capture()
                eng.precapture(ctx0 + args.steps + 1)     # one graph per attention split the run will reach
            except Exception as e:  # e.g. a collective that cannot be captured: fall back to eager launches
                err = f"{type(e).__name__}: {e}"[:300]
                if rank == 0:
                    print(f"[bench] graph capture failed ({err}); running eager", file=sys.stderr)
                eng.use_graph, eng.graph, eng.graphs, ok = False, None, {}, False
                torch.cuda.synchronize(dev)
        if world > 1:      # one rank falling back to eager while the others replay graphs would desynchronise the collectives
            flag = torch.tensor([0 if ok else 1], device=dev if not shared else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and ok:
                eng.use_graph, eng.graph, eng.graphs, ok = False, None, {}, False
                err = "another rank failed to capture its graph: eager on every rank"
        return ok, err

    graph_ok, graph_capture_error = capture_graphs()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- context (64 tokens, benchmark.py:118) + W warm-up tokens, untimed; then EXACTLY K timed steps
    def build_context():
        eng.reset()
        eng.tok.fill_(1)
        for _ in range(ctx0):
            eng.step()
        if graph_ok:
            eng.run(eng.MULTI)                 # captures the multi-token graph outside the timed region ...
            eng.set_position(ctx0)             # ... and rewinds: the timed tokens start at the protocol's context

    def oneshot_gave_up(when):
        """Every rank alike: True (after dropping the one-shot collective, re-capturing with the group's own all-reduce and
        rebuilding the context) if any rank's status word holds a give-up code."""
        nonlocal graph_ok, graph_capture_error
        if world == 1 or eng.oneshot is None:
            return False
        if eng.verify_collective(when):
            return False
        if rank == 0:
            print(f"[bench] {eng.collective_note}", file=sys.stderr)
        eng.use_graph = not (args.no_graph or shared)
        graph_ok, graph_capture_error = capture_graphs()
        build_context()
        return True

    def timed_tokens():
        barrier()
        t0 = time.perf_counter()
        if graph_ok:
            eng.run(args.steps)                # the same K tokens; graphs of 8 where the attention split does not change
        else:
            for _ in range(args.steps):
                eng.step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev if not shared else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # The one-shot collective passed its probe at construction; the untimed tokens are its first run under the real launch
    # sequence, the timed ones the second.  A wait that gave up (peer stores that never arrived) leaves a code in its status word
    # and makes later calls return at once: the run never reports a number measured on a collective that has failed.
    build_context()
    if os.environ.get("QEFT_BENCH_FAKE_ONESHOT_GIVEUP") == "1" and eng.oneshot is not None and rank == world - 1:
        eng.oneshot.status[0] = 0x7300         # rehearsal hook: what a wait that gave up on the last rank leaves behind
    oneshot_gave_up("the untimed tokens")
    dt = timed_tokens()
    if oneshot_gave_up("the timed tokens"):
        dt = timed_tokens()
    last_tok = int(eng.tok.item())
    # steady_128: 128 further timed tokens on the same graphs (rewound to the protocol's context, one untimed pass first so that
    # nothing is captured inside the timed one) -- shows whether a short --steps run is representative
    steady = None
    if world == 1 and graph_ok:
        extra_n = 128
        for timed in (False, True):
            eng.set_position(ctx0)
            eng.tok.fill_(1)
            barrier()
            t1 = time.perf_counter()
            eng.run(extra_n)
            barrier()
            if timed:
                steady = {"tokens": extra_n, "tokens_per_s": round(extra_n / (time.perf_counter() - t1), 2),
                          "ms_per_step": round((time.perf_counter() - t1) * 1e3 / extra_n, 4)}

    # ---- roofline of the dominant kernel: the token's GEMV launches alone, back to back, HIP-event timed
    roof = None
    try:
        g2 = eng.capture(linears_only=True) if graph_ok else None
        run_lin = (lambda: g2.replay()) if g2 is not None else (lambda: eng._launch_token(True))
        reps = 20
        launches = 4 * shape.n_layers
        us_per_launch = _event_time_us(run_lin, reps, dev) / launches
        bytes_per_launch = eng.weight_bytes_per_token() / launches
        achieved = bytes_per_launch / us_per_launch / 1e3  # GB/s
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": f"{eng.gemv_kernel_name()} (4 launches per layer)",
                "bytes_per_launch": int(bytes_per_launch), "us_per_launch": round(us_per_launch, 3),
                "frac_of_copy_ceiling_6300": round(achieved / 6300.0, 4)}
        if world == 1:
            # a stream ceiling measured on THIS box in the same run -- a 1 GiB device-to-device copy (read +
            # write counted) and a 1 GiB read (sum); the GEMV fraction against the faster of the two beside the nominal one
            src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
            dst = torch.empty_like(src)
            t_copy = _event_time_us(lambda: dst.copy_(src), 10, dev)
            t_read = _event_time_us(lambda: src.sum(), 10, dev)
            copy_gbps, read_gbps = 2 * src.numel() * 4 / t_copy / 1e3, src.numel() * 4 / t_read / 1e3
            roof["stream_ceiling_measured"] = {"copy_GB/s": round(copy_gbps, 1), "read_GB/s": round(read_gbps, 1),
                                               "frac": round(achieved / max(copy_gbps, read_gbps), 4),
                                               "note": "1 GiB torch copy_ (read + write bytes) and 1 GiB torch sum on this box"}
            del src, dst
        if traffic is not None:
            roof["traffic"] = traffic["bytes_per_launch"]
            roof["traffic_note"] = (f"FETCH_SIZE x 1024 x 2 (gfx950 correction), mean over {traffic['launches_sampled']} "
                                    "GEMV launches of a separate `rocprofv3 --pmc FETCH_SIZE --kernel-trace` child run")
        # the same, one GEMV of the layer at a time (32 launches per replay): per-kernel rates for DESIGN.md / rocprof
        if g2 is not None and world == 1 and not eng.tp and not args.no_per_kind:
            per = {}
            lin0 = eng.lin[0]
            parts = {"qkv": ("q", "k", "v"), "o": ("o",), "gu": ("g", "u"), "d": ("d",)}
            for tag, names in parts.items():
                gk = eng.capture(linears_only=True, only=tag)
                us = _event_time_us(gk.replay, reps, dev) / shape.n_layers
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

    multi = None
    if group is not None:
        # ---- diagnostics of the FIRST multi-GPU runs: who took part, what one collective costs, every rank's rate
        try:
            multi = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                     "graph_capture_error": graph_capture_error}
            names = [None] * world
            dist.all_gather_object(names, f"rank {rank}: cuda:{dev_index} {torch.cuda.get_device_name(dev)}")
            multi["devices"] = names
            # the layer's collective alone: an fp32 all-reduce of `hidden` floats (16 KB for 7B), 64 of them per replay
            buf = torch.zeros(shape.hidden, dtype=torch.float32, device=dev)
            red = buf if not shared else buf.cpu()
            n_coll = 64

            def coll():
                for _ in range(n_coll):
                    dist.all_reduce(red)
            coll()
            torch.cuda.synchronize(dev)
            how = "eager"
            run_coll = coll
            if not shared:
                try:
                    gcoll = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gcoll):
                        coll()
                    run_coll, how = gcoll.replay, "hipgraph"
                except Exception as e:
                    multi["collective_capture_error"] = f"{type(e).__name__}: {e}"[:200]
                    torch.cuda.synchronize(dev)
            barrier()
            t_c = _event_time_us(run_coll, 10, dev, warm_ms=5.0) / n_coll if not shared else None
            if shared:
                t1 = time.perf_counter()
                coll()
                t_c = (time.perf_counter() - t1) * 1e6 / n_coll
            tt = torch.tensor([t_c], dtype=torch.float64, device=dev if not shared else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            multi["collective_us"] = {"all_reduce_fp32_bytes": shape.hidden * 4, "us": round(float(tt.item()), 2), "how": how,
                                      "per_token": 2 * shape.n_layers if getattr(eng, "tp3", False) else 4 * shape.n_layers}
            multi["collective"] = eng.collective                    # what the timed tokens used
            if eng.collective_note:
                multi["collective_note"] = eng.collective_note      # why "auto" fell back to the group's collective
            if eng.oneshot is not None:
                # the one-shot kernel alone, the same way (64 dependent calls per replay)
                b2 = torch.zeros(shape.hidden, dtype=torch.float32, device=dev)

                def coll1():
                    for _ in range(n_coll):
                        eng.oneshot.all_reduce(b2)
                coll1()
                torch.cuda.synchronize(dev)
                run1, how1 = coll1, "eager"
                try:
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1):
                        coll1()
                    run1, how1 = g1.replay, "hipgraph"
                except Exception as e:
                    multi["oneshot_capture_error"] = f"{type(e).__name__}: {e}"[:200]
                    torch.cuda.synchronize(dev)
                barrier()
                t1 = _event_time_us(run1, 10, dev, warm_ms=5.0) / n_coll
                tt1 = torch.tensor([t1], dtype=torch.float64, device=dev if not shared else "cpu")
                dist.all_reduce(tt1, op=dist.ReduceOp.MAX)
                eng.oneshot.check_status()
                multi["collective_us"]["oneshot_us"] = round(float(tt1.item()), 2)
                multi["collective_us"]["oneshot_how"] = how1
            per_rank = [None] * world
            dist.all_gather_object(per_rank, {"rank": rank, "gemv_GB/s": roof["achieved"] if roof else None,
                                              "gemv_frac": roof["frac"] if roof else None,
                                              "gemv_us_per_launch": roof["us_per_launch"] if roof else None,
                                              "weight_bytes_per_token": int(eng.weight_bytes_per_token())})
            multi["per_rank_roofline"] = per_rank
        except Exception as e:
            multi = dict(multi or {}, error=f"{type(e).__name__}: {e}"[:300])
            if rank == 0:
                print(f"[bench] multi-GPU diagnostics failed: {type(e).__name__}: {e}", file=sys.stderr)
        # ---- the other way to use N GPUs for a model that fits one: N independent replicas of the single-GPU engine, one sequence
        # each, no collective (what a server does at batch 1 per GPU).  Every rank measures its own replica (no collective inside
        # the measurement: a rank that fails reports None and nobody waits for it); the line carries the sum beside `value`.
        mine = None
        if world > 1 and not args.no_extras:
            try:
                r = model_13b_record(dev, steps=max(32, min(args.steps, 128)), warmup=16, base=base, bits=args.bits)
                mine = {"rank": rank, "tokens_per_s": r["tokens_per_s"], "gemv_frac_of_8TB/s": r["gemv_frac_of_8TB/s"]}
            except Exception as e:
                mine = {"rank": rank, "error": f"{type(e).__name__}: {e}"[:200]}
            reps = [None] * world
            dist.all_gather_object(reps, mine)
            ok = [x["tokens_per_s"] for x in reps if x and "tokens_per_s" in x]
            multi = dict(multi or {}, replicas={"tokens_per_s_sum": round(sum(ok), 1), "ranks_measured": len(ok), "per_rank": reps,
                                               "shared_device": bool(shared),
                                               "note": "N independent single-GPU engines (one sequence each, no collective), each rank timed "
                                                       "on its own; `value` above is ONE sequence decoded by all N ranks (tensor parallel)"})
    tp3 = bool(getattr(eng, "tp3", False))      # (the extras below free the engine)
    extras = {}
    if world == 1 and not args.no_extras:
        # ---- the reference's per-token protocol: synchronize after every token, median / min, tokens/s = 1 / median
        try:
            lat = {}
            for tag, use_graph in (("hipgraph", True), ("eager", False)):
                if use_graph and not graph_ok:
                    continue
                eng.use_graph = use_graph
                eng.set_position(ctx0)
                eng.tok.fill_(1)
                times = []
                n_tok = min(args.steps, 64) if not use_graph else args.steps
                for _ in range(n_tok):
                    torch.cuda.synchronize(dev)
                    tick = time.perf_counter()
                    eng.step()
                    torch.cuda.synchronize(dev)
                    times.append(time.perf_counter() - tick)
                times.sort()
                med = times[len(times) // 2]
                lat[tag] = {"median_s": round(med, 7), "min_s": round(times[0], 7), "tokens_per_s": round(1.0 / med, 1),
                            "tokens": n_tok}
            eng.use_graph = graph_ok
            lat["protocol"] = "per-token torch.cuda.synchronize, 1 / median(seconds) (main.py:357-371, benchmark.py:293-338)"
            extras["latency_protocol"] = lat
        except Exception as e:
            print(f"[bench] latency protocol failed: {type(e).__name__}: {e}", file=sys.stderr)
        # ---- the reference's entry points for < 8 rows: gemv_4bit_qeft / QuantLinear.forward, m = 1 and 4
        if args.model == "7b" and args.bits == 4 and not args.ckpt:
            try:
                extras["boundary_gemv"] = boundary_gemv_records(dev)
            except Exception as e:
                print(f"[bench] boundary_gemv failed: {type(e).__name__}: {e}", file=sys.stderr)
        # ---- BASELINE configs 3 and 5 at M = 2048 (7B shapes, w4 operands)
        if args.model == "7b" and not args.ckpt:
            try:
                del eng
                fwd_recs, ft_recs = gemm_records(dev)
                pre = {"M": 2048, "per_shape": fwd_recs, "peak_TFLOPs": MFMA_PEAK_TFLOPS,
                       "note": "fused-outlier W4 MFMA GEMM, 4 weight sets cycled, gaussian activations"}
                from qeft_amd.llama import prefill
                toks = torch.randint(0, shape.vocab, (2048,), device=dev)
                if shape.max_seq < 2048:
                    model.shape = dataclasses.replace(shape, max_seq=2048)
                    half = 64
                    inv = 1.0 / (shape.rope_theta ** (torch.arange(0, half, dtype=torch.float64) / half))
                    ang = torch.arange(2048, dtype=torch.float64)[:, None] * inv[None, :]
                    model.rope_cos, model.rope_sin = ang.cos().float().to(dev), ang.sin().float().to(dev)
                t_pre = _event_time_us(lambda: prefill(model, toks), 3, dev)
                lin_flops = 2.0 * 2048 * sum(n * k for n, k in ((4096, 4096),) * 4 + ((11008, 4096),) * 2 + ((4096, 11008),)) * shape.n_layers
                pre["whole_model"] = {"ms": round(t_pre / 1e3, 2), "tokens_per_s": round(2048 / t_pre * 1e6, 0),
                                      "linears_TFLOP": round(lin_flops / 1e12, 2),
                                      "note": "2048-token prompt through every packed linear (GEMM path), torch fused SDPA attention"}
                extras["prefill_2048"] = pre
                extras["prefill_mid_m"] = {"per_shape": mid_m_records(dev), "peak_TFLOPs": MFMA_PEAK_TFLOPS,
                                           "note": "forward GEMM (fused outlier slice) below the M = 2048 tier, 4 weight sets cycled"}
                # the same step at M = 1024 (a shorter batch: the 128-row loader-wave tiles of round 3, forward and dX)
                ft1024 = gemm_records(dev, m=1024, reps=15)[1]
                extras["finetune_step_m1024"] = {"M": 1024, "per_shape": [
                    {k: r[k] for k in ("shape", "forward_us", "dx_us", "dow_us", "step_us", "w3_step_us", "step_TFLOPs", "variants")}
                    for r in ft1024]}
                extras["finetune_step"] = {"M": 2048, "per_shape": ft_recs,
                                           "note": "forward + dX + d(oweight) of one QuantLinear, oweight trainable (qlinear.py:13-44); dx_dow_overlapped_us: dX with d(oweight) on a second stream; w3_*: the same step of a 3-bit layer on the 3-bit stream (no expansion pass)"}
            except Exception as e:
                print(f"[bench] GEMM sub-records failed: {type(e).__name__}: {e}", file=sys.stderr)
            # ---- BASELINE config 4's model on one GPU
            try:
                model = None
                torch.cuda.empty_cache()
                extras["model_13b"] = model_13b_record(dev)
            except Exception as e:
                print(f"[bench] model_13b failed: {type(e).__name__}: {e}", file=sys.stderr)
            # ---- where the same launches land when the fixed part is amortised: the Llama-2-70B shapes (36 GB packed, grouped-query
            #      attention) on the one GPU -- not a BASELINE config, context for roofline.frac
            try:
                from qeft_amd.llama import LLAMA2_70B
                torch.cuda.empty_cache()
                extras["model_70b"] = model_13b_record(dev, steps=48, warmup=16, base=LLAMA2_70B)
            except Exception as e:
                print(f"[bench] model_70b failed: {type(e).__name__}: {e}", file=sys.stderr)

    if rank == 0:
        ms = dt * 1e3 / args.steps
        out = {
            "metric": "decode tokens/sec, Llama-2-7B w4 g128 r128" if (args.model == "7b" and args.bits == 4 and not args.ckpt)
            else f"decode tokens/sec, {shape.name} w{args.bits} g128 r128",
            "value": round(args.steps / dt, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f16",
            "data": f"packed checkpoint {os.path.basename(args.ckpt)}, random token ids" if args.ckpt else "synthetic",
            "config": {"workload": f"{shape.name} w{args.bits} g{shape.group_size} r{shape.n_out} full decode step, batch 1, "
                                   f"greedy, KV context {ctx0}..{ctx0 + args.steps} tokens ({CONTEXT}-token context + "
                                   f"{args.warmup} warm-up tokens before the timed region)",
                       "layers": shape.n_layers, "hipgraph": graph_ok,
                       "parallelism": ((f"tp{world}: q/k/v/gate/up row-sharded, o/down column-sharded + one all-reduce each "
                                        f"(2 collectives per layer{', shared-GPU rehearsal' if shared else ''})")
                                       if tp3 else
                                       (f"tp{world}: every linear row-sharded, one all-gather each "
                                        f"(4 collectives per layer{', shared-GPU rehearsal' if shared else ''})"))
                       if group is not None else "single GPU",
                       "build_s": round(t_build, 1), "last_token": last_tok},
        }
        if graph_capture_error:
            out["graph_capture_error"] = graph_capture_error
        if multi:
            out["multi_gpu"] = multi
        if steady:
            out["steady_128"] = steady
        if roof:
            out["roofline"] = roof
        out.update(extras)
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
