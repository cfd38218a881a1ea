"""Lab (needs a lab build of the library: QEFT_BUILD_LAB=1 python -m qeft_amd.build --force):
shader clock during the v3 GEMM's k loop (QEFT_GEMM_ABL=6 selects the kernel variant with time stamps; the 'bias' buffer
receives, per block, [shader cycles, 100 MHz ticks, start tick, end tick] of compute wave 0's k loop)."""
import os
import sys

os.environ["QEFT_GEMM_ABL"] = "6"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd import qeft_cuda  # noqa: E402

dev = "cuda:0"
for n, k in [(4096, 4096), (4096, 11008)]:
    m, r, g = 2048, 128, 128
    qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
    sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
    sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
    ow = (torch.randn(n, r, device=dev) * 0.02).half()
    x = torch.randn(m, k, device=dev).half()
    dbg = torch.zeros(max(n, 8192), dtype=torch.float16, device=dev)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        qeft_cuda.gemm_4bit_qeft(x, qw, sc, sz, ow, dbg)
        e1.record()
        torch.cuda.synchronize()
        d = dbg.view(torch.int64)[:2048].view(256, 8).cpu()
        cyc, ticks = d[:, 0].double(), d[:, 1].double()
        span = (d[:, 3].max() - d[:, 2].min()).item() / 100.0
        tiles = k // 64
        t0 = d[:, 4].min().item()
        rel = lambda col: f"{(d[:, col].min().item() - t0) / 100:.1f}..{(d[:, col].max().item() - t0) / 100:.1f}"
        print(f"   timeline (us after the first block's entry; min..max over blocks): entry {rel(4)}, k loop starts {rel(2)}, "
              f"k loop ends {rel(3)}, stores drained {rel(5)}")
        print(f"N={n} K={k}: kernel {e0.elapsed_time(e1) * 1e3:7.1f} us; k loop per block: {ticks.mean().item() / 100:6.1f} us "
              f"(min {ticks.min().item() / 100:.1f}, max {ticks.max().item() / 100:.1f}), first start -> last end {span:.1f} us; "
              f"shader clock {cyc.sum().item() / ticks.sum().item() * 100:.0f} MHz; cycles per k-tile {cyc.mean().item() / tiles:.0f}", flush=True)

# ---- back-to-back launches: where does the time between one launch's last store and the next launch's first wave go?
n, k, m, r, g = 4096, 4096, 2048, 128, 128
qw = torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev)
sc = (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half()
sz = (-(torch.rand(k // g, n, device=dev) * 8 + 4) * sc.float()).half()
ow = (torch.randn(n, r, device=dev) * 0.02).half()
x = torch.randn(m, k, device=dev).half()
y = torch.empty(m, n, dtype=torch.float16, device=dev)
dbgs = [torch.zeros(8192, dtype=torch.float16, device=dev) for _ in range(12)]
from qeft_amd import _lib  # noqa: E402
lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for d in dbgs:
    _lib.check(lib.qeft_gemm_w4(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), sz.data_ptr(), ow.data_ptr(), d.data_ptr(),
                                y.data_ptr(), m, n, k, g, r, st))
e1.record()
torch.cuda.synchronize()
print(f"12 launches back to back: {e0.elapsed_time(e1) * 1e3 / 12:.1f} us per launch (events)")
prev_end = None
for i, d in enumerate(dbgs):
    v = d.view(torch.int64)[:2048].view(256, 8).cpu()
    first, last = v[:, 4].min().item(), v[:, 5].max().item()
    gap = "" if prev_end is None else f", gap since the previous launch's last store {(first - prev_end) / 100:.1f} us"
    print(f"  launch {i}: first entry -> last store {(last - first) / 100:.1f} us{gap}")
    prev_end = last

# ---- sustained: does the per-launch time drift (power), depend on rotating operands (cache), or on the Python wrapper (host)?
def chunks(label, call, n_chunks=10, per=40):
    out = []
    for c in range(n_chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(per):
            call(c * per + i)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / per)
    print(f"{label}: us per launch in chunks of {per}: " + " ".join(f"{v:.1f}" for v in out), flush=True)

sets = []
for i in range(4):
    sets.append((torch.randint(-32768, 32767, (n // 4, k), dtype=torch.int16, device=dev),
                 (torch.rand(k // g, n, device=dev) * 0.004 + 0.001).half(),
                 (-(torch.rand(k // g, n, device=dev) * 8 + 4) * 0.003).half(), (torch.randn(n, r, device=dev) * 0.02).half()))
dbg = dbgs[0]

def direct(i, nsets=1):
    q_, s_, z_, o_ = sets[i % nsets]
    _lib.check(lib.qeft_gemm_w4(x.data_ptr(), q_.data_ptr(), s_.data_ptr(), z_.data_ptr(), o_.data_ptr(), dbg.data_ptr(),
                                y.data_ptr(), m, n, k, g, r, st))

chunks("C-ABI, one weight set ", lambda i: direct(i, 1))
chunks("C-ABI, 4 weight sets  ", lambda i: direct(i, 4))
def direct_nobias(i):
    q_, s_, z_, o_ = sets[i % 4]
    _lib.check(lib.qeft_gemm_w4(x.data_ptr(), q_.data_ptr(), s_.data_ptr(), z_.data_ptr(), o_.data_ptr(), None,
                                y.data_ptr(), m, n, k, g, r, st))

chunks("C-ABI, 4 sets, no bias", direct_nobias)
chunks("wrapper, 4 weight sets", lambda i: qeft_cuda.gemm_4bit_qeft(x, *sets[i % 4]))
xs = [torch.randn(m, k, device=dev).half() for _ in range(4)]
chunks("wrapper, 4 sets, 4 x  ", lambda i: qeft_cuda.gemm_4bit_qeft(xs[i % 4], *sets[i % 4]))
