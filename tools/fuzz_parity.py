"""Randomised parity sweep of the C-ABI compute entries against the oracle (numpy) on one GPU: seeded shapes drawn from
the whole accepted domain (N % 8, K % 64, group 32..256 dividing K, outlier count a multiple of 32, every batch tier),
forward (GEMV / small-batch / GEMM tiers), dX and d(oweight).  One process, bounded sizes; prints one line per case and
a summary.  A run is committed under profiles/ -- the fixed-shape tests in tests/ stay the gate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import qeft_oracle as O
from util import layer_to_torch, rel_err
from qeft_amd import qeft_cuda, _lib
DEV = "cuda:0"
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 150
LARGE = len(sys.argv) > 2 and sys.argv[2] == "large"       # shapes that reach the MFMA GEMV and the 128- / 256-row GEMM tiers
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "20261004")))
fails, t0, seen = [], time.time(), {}
if len(sys.argv) > 2 and sys.argv[2] == "v3":
    # the engine's decode GEMV (qeft_decode_linear) on random shapes of its own domain: N % 16, K % 128, group 128, r in {0, 128}
    import types
    for case in range(CASES):
        n = 16 * int(rng.integers(1, 901))
        k = 128 * int(rng.integers(1, 91))
        r = int(rng.choice([0, 128])) if k > 128 else 0
        b = O.make_layer(n, k, r, 128, seed=case, bias=bool(case & 1))
        t = layer_to_torch(b, DEV)
        l = types.SimpleNamespace(qweight=t["qweight"], scales=t["scales"], scaled_zeros=t["scaled_zeros"], oweight=t.get("oweight"),
                                  bias=t.get("bias"), outfeatures=n, infeatures=k, group_size=128, outlierfeatures=r)
        l.sz_packed = qeft_cuda.pack_scales(l.scales, l.scaled_zeros, n, k, 128)
        x = O.make_activation(1, k, r, seed=case)
        y = qeft_cuda.decode_linear(torch.from_numpy(x[0]).to(DEV), l)
        v = _lib.last_variant()
        torch.cuda.synchronize()
        ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, b.get("bias"), 128)
        e = rel_err(y.cpu().numpy()[None], ref.astype(np.float64))
        seen[v] = seen.get(v, 0) + 1
        ok = e < 1e-3
        print(f"case {case:3d} n={n:5d} k={k:5d} r={r:3d} bias={case & 1}  {v:10s} y={e:.1e}" + ("" if ok else "   FAIL"), flush=True)
        if not ok:
            fails.append((case, n, k, r, e))
    print(f"{CASES} cases, {len(fails)} failures, {time.time() - t0:.0f} s; variants reached: {seen}")
    sys.exit(1 if fails else 0)
if len(sys.argv) > 2 and sys.argv[2] == "ws":
    # round 4: the weight-stationary tier (gemm_ws.hip) at the GEMM entries: 17..64 rows, N % 16, K % 128, group 128, r in {0, 128},
    # bias on / off, through gemm_4bit_qeft (fused outlier) and gemm_4bit (no slice); every output element against the oracle
    for case in range(CASES):
        n = 16 * int(rng.integers(1, 901))
        k = 128 * int(rng.integers(2, 91))
        r = int(rng.choice([0, 128, 128]))
        m = int(rng.integers(17, 65))
        bias = bool(rng.integers(0, 2))
        b = O.make_layer(n, k, r, 128, seed=case, bias=bias)
        t = layer_to_torch(b, DEV)
        x = O.make_activation(m, k, max(r, 1), seed=case)
        xt = torch.from_numpy(x).to(DEV)
        if r:
            y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t["oweight"], t.get("bias"))
        elif bias:
            y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], None, t.get("bias"))
        else:
            y = qeft_cuda.gemm_4bit(xt, t["qweight"], t["scales"], t["scaled_zeros"])
        v = _lib.last_variant()
        torch.cuda.synchronize()
        ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, b.get("bias"), 128).astype(np.float64)
        got = y.cpu().numpy()
        e = rel_err(got, ref)
        # element-wise: tests/util.elem_err_ok's allowance (1e-3 |ref| + 1e-3 rms).  With up to 10^6 outputs per case the tail of the
        # accumulation noise of the 1024 + q trick (fp32 partial sums of magnitude 1e4, K up to 11520) crosses it by a few per cent on
        # about one element in 10^6: reported (`over`), and a case fails if more than 1e-5 of its elements cross or any by 2 x
        d = np.abs(got - ref)
        allow = 1e-3 * np.abs(ref) + 1e-3 * float(np.sqrt(np.mean(ref ** 2)))
        over = int((d > allow).sum())
        worst = float((d / allow).max())
        ok = e < 1e-3 and over <= 1e-5 * d.size and worst < 2.0 and v == "gemm_ws"
        seen[v] = seen.get(v, 0) + 1
        print(f"case {case:3d} m={m:2d} n={n:5d} k={k:5d} r={r:3d} bias={int(bias)}  {v:8s} y={e:.1e} over={over} worst={worst:.2f}" + ("" if ok else "   FAIL"), flush=True)
        if not ok:
            fails.append((case, m, n, k, r, e))
    print(f"{CASES} cases, {len(fails)} failures, {time.time() - t0:.0f} s; variants reached: {seen}")
    sys.exit(1 if fails else 0)
if len(sys.argv) > 2 and sys.argv[2] == "ref":
    # round 3: the REFERENCE's gemv entries (gemv_4bit / gemv_4bit_qeft / the fused form) on the v3 kernel's domain -- operands as
    # the checkpoint holds them: m = 1..7, group 128 or per-channel, r in {0, 128}, with / without gather, bias, sz_packed shadow
    for case in range(CASES):
        n = 16 * int(rng.integers(1, 701))
        k = 128 * int(rng.integers(2, 91))
        r = int(rng.choice([0, 128]))
        g = int(rng.choice([128, 128, k]))
        m = int(rng.integers(1, 8))
        gather, bias, shadow = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        b = O.make_layer(n, k, r, g, seed=case, bias=bias)
        t = layer_to_torch(b, DEV)
        x = O.make_activation(m, k, r, seed=case)
        ids = O.sparse_to_dense_ids(np.sort(rng.choice(k, size=max(r, 1), replace=False)), k) if gather else None
        szp = qeft_cuda.pack_scales(t["scales"], t["scaled_zeros"], n, k, g) if shadow else None
        y = qeft_cuda.gemv_4bit_fused(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"],
                                      t.get("oweight_interleaved") if r else None, t.get("bias"),
                                      torch.from_numpy(ids.astype(np.int32)).to(DEV) if gather else None, None, m, n, k, g, szp)
        v = _lib.last_variant()
        torch.cuda.synchronize()
        ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, b.get("bias"), g, reorder_ids=ids)
        e = rel_err(y.cpu().numpy(), ref.astype(np.float64))
        seen[v] = seen.get(v, 0) + 1
        ok = e < 1e-3 and v.startswith("gemv_v3")
        print(f"case {case:3d} n={n:5d} k={k:5d} g={g:5d} r={r:3d} m={m} gather={int(gather)} bias={int(bias)} shadow={int(shadow)}  {v:10s} y={e:.1e}" + ("" if ok else "   FAIL"), flush=True)
        if not ok:
            fails.append((case, n, k, g, r, m, gather, bias, shadow, v, e))
    print(f"{CASES} cases, {len(fails)} failures, {time.time() - t0:.0f} s; variants reached: {seen}")
    sys.exit(1 if fails else 0)
if len(sys.argv) > 2 and sys.argv[2] == "rows16":
    # round 3: the GEMM entries (what QuantLinear.forward calls from 8 rows on) with 8..16 rows on the decode GEMV's domain: the
    # rows ride as A rows of its MFMAs (gemv_v3_mb), in one or two launches, or -- K long, rows many -- on the split-K GEMM tier
    worst = 0.0
    for case in range(CASES):
        n = 16 * int(rng.integers(1, 701))
        k = 128 * int(rng.integers(2, 91))
        r = int(rng.choice([0, 128]))
        g = int(rng.choice([128, 128, k]))
        m = int(rng.integers(8, 17))
        bias = bool(rng.integers(0, 2))
        b = O.make_layer(n, k, r, g, seed=case, bias=bias)
        t = layer_to_torch(b, DEV)
        x = O.make_activation(m, k, r, seed=case)
        xt = torch.from_numpy(x).to(DEV)
        if r or bias:
            y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight") if r else None, t.get("bias"))
        else:
            y = qeft_cuda.gemm_4bit(xt, t["qweight"], t["scales"], t["scaled_zeros"])
        v = _lib.last_variant()
        torch.cuda.synchronize()
        ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, b.get("bias"), g).astype(np.float64)
        got = y.cpu().numpy()
        e = rel_err(got, ref)
        seen[v] = seen.get(v, 0) + 1
        # element-wise: |err| / (1e-3 |ref| + 1e-3 rms(ref)), the parity tests' bound.  The oracle rounds every dequantised weight
        # to fp16 as the reference's kernel does (gemv_cuda_qeft.cu:158); the product folds scale and zero in fp32 on the exact
        # q s + sz.  That difference alone is ~2.8e-4 rms(ref) (1 sigma) per output, so among the ~10^7 outputs of a sweep a few
        # reach 1.0-1.15 of the bound (3.6-4 sigma; the m <= 7 entries show the same elements).  A wrong row, scale or column
        # is off by orders of magnitude: the sweep fails from 2.0 (7 sigma) and prints the worst ratio.
        ratio = float(np.max(np.abs(got - ref) / (1e-3 * np.abs(ref) + 1e-3 * np.sqrt(np.mean(ref ** 2)))))
        worst = max(worst, ratio) if case else ratio
        ok = e < 1e-3 and ratio < 2.0 and got.shape == (m, n)
        print(f"case {case:3d} n={n:5d} k={k:5d} g={g:5d} r={r:3d} m={m:2d} bias={int(bias)}  {v:24s} y={e:.1e} elem={ratio:.2f}" + ("" if ok else "   FAIL"), flush=True)
        if not ok:
            fails.append((case, n, k, g, r, m, bias, v, e))
    print(f"{CASES} cases, {len(fails)} failures, worst element-wise ratio {worst:.2f}, {time.time() - t0:.0f} s; variants reached: {seen}")
    sys.exit(1 if fails else 0)
if len(sys.argv) > 2 and sys.argv[2] == "w3gemm":
    # round 3: GEMM forward / dX of 3-bit layers: native tiers where they apply, the expansion route otherwise
    for case in range(CASES):
        n = 64 * int(rng.integers(4, 100))
        k = 128 * int(rng.integers(3, 40))
        r = int(rng.choice([0, 128]))
        g = int(rng.choice([128, 128, 256 if k % 256 == 0 else 128]))
        m = int(rng.choice([130, 300, 520, 1024, 1100, 2048]))
        b = O.make_layer(n, k, r, g, seed=case, bits=3)
        t = layer_to_torch(b, DEV)
        x = O.make_activation(m, k, r, seed=case)
        dy = (np.random.default_rng(case).standard_normal((m, n)) * 0.1).astype(np.float16)
        ow = t.get("oweight") if r else None
        y = qeft_cuda.gemm_3bit_qeft(torch.from_numpy(x).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], ow)
        v_f = _lib.last_variant()
        dx = qeft_cuda.gemm_3bit_dx(torch.from_numpy(dy).to(DEV), t["qweight"], t["scales"], t["scaled_zeros"], ow, k)
        v_dx = _lib.last_variant()
        torch.cuda.synchronize()
        w = O.dequant_dense(b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, g).astype(np.float64)
        e_y = rel_err(y.cpu().numpy(), x.astype(np.float64) @ w.T)
        e_dx = rel_err(dx.cpu().numpy(), dy.astype(np.float64) @ w)
        for v in (v_f, v_dx):
            seen[v] = seen.get(v, 0) + 1
        ok = e_y < 1e-3 and e_dx < 2e-3
        print(f"case {case:3d} n={n:5d} k={k:5d} g={g:3d} r={r:3d} m={m:4d}  {v_f:22s} {v_dx:12s} y={e_y:.1e} dx={e_dx:.1e}" + ("" if ok else "   FAIL"), flush=True)
        if not ok:
            fails.append((case, n, k, g, r, m, v_f, v_dx, e_y, e_dx))
    print(f"{CASES} cases, {len(fails)} failures, {time.time() - t0:.0f} s; variants reached: {seen}")
    sys.exit(1 if fails else 0)
for case in range(CASES):
    k = 64 * int(rng.integers(16, 73) if LARGE else rng.integers(1, 41))      # 64 .. 2560 (large: 1024 .. 4608)
    gs = [g for g in (32, 64, 128, 256) if k % g == 0]
    g = int(rng.choice(gs))
    n = 8 * int(rng.integers(64, 641) if LARGE else rng.integers(1, 161))     # 8 .. 1280 (large: 512 .. 5120)
    r_choices = [r for r in (0, 32, 64, 96, 128, 160, 256) if r < k]
    r = int(rng.choice(r_choices))
    m = int(rng.choice([1, 4, 7, 64, 520, 1024, 1300, 2048, 2100] if LARGE else [1, 2, 3, 5, 7, 8, 9, 31, 64, 100, 129, 255, 300, 513, 1100]))
    b = O.make_layer(n, k, r, g, seed=case)
    t = layer_to_torch(b, DEV)
    x = O.make_activation(m, k, r, seed=case)
    dy = (np.random.default_rng(case).standard_normal((m, n)) * 0.1).astype(np.float16)
    xt, dyt = torch.from_numpy(x).to(DEV), torch.from_numpy(dy).to(DEV)
    ow = t.get("oweight") if r else None
    errs = {}
    try:
        if m <= 7:
            y = qeft_cuda.gemv_4bit_fused(xt, t["qweight"], t["scales"], t["scaled_zeros"], t.get("oweight_interleaved") if r else None,
                                          None, None, None, m, n, k, g)
        else:
            y = qeft_cuda.gemm_4bit_qeft(xt, t["qweight"], t["scales"], t["scaled_zeros"], ow)
        v_f = _lib.last_variant()
        dx = qeft_cuda.gemm_4bit_dx(dyt, t["qweight"], t["scales"], t["scaled_zeros"], ow)
        v_dx = _lib.last_variant()
        dow = qeft_cuda.grad_oweight(dyt, xt, r) if r else None
        v_dow = _lib.last_variant() if r else "-"
        torch.cuda.synchronize()
        ref = O.quant_linear(x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, None, g)
        dx_ref, dow_ref = O.quant_linear_backward(dy, x, b["qweight"], b["scales"], b["scaled_zeros"], b.get("oweight") if r else None, g)
        errs["y"] = rel_err(y.cpu().numpy(), ref.astype(np.float64))
        errs["dx"] = rel_err(dx.cpu().numpy(), dx_ref.astype(np.float64))
        if r:
            errs["dow"] = rel_err(dow.cpu().numpy(), dow_ref)
        bad = {k_: v for k_, v in errs.items() if not (v < (2e-3 if k_ == "dx" else 1e-3))}
    except Exception as e:      # a refused shape is a finding too
        bad, v_f, v_dx, v_dow = {"exception": repr(e)}, "?", "?", "?"
    for v in (v_f, v_dx, v_dow):
        seen[v] = seen.get(v, 0) + 1
    line = f"case {case:3d} n={n:5d} k={k:5d} g={g:3d} r={r:3d} m={m:4d}  {v_f:24s} {v_dx:12s} {v_dow:22s} " + " ".join(f"{a}={v:.1e}" for a, v in errs.items())
    print(line + ("   FAIL " + str(bad) if bad else ""), flush=True)
    if bad:
        fails.append((case, n, k, g, r, m, bad))
print(f"{CASES} cases, {len(fails)} failures, {time.time() - t0:.0f} s; variants reached: {seen}")
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
