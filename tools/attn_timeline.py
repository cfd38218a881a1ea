"""Lab: phase timeline of the decode attention kernel inside the running engine (debug stamps)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qeft_amd import _lib
from qeft_amd.llama import LLAMA2_7B, DecodeEngine, QuantLlama
import dataclasses

dev = torch.device("cuda:0")
shape = dataclasses.replace(LLAMA2_7B, n_layers=4)
model = QuantLlama(shape, dev, seed=0, fast_init=True)
eng = DecodeEngine(model, use_graph=False)
eng.greedy = True
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng.attn_split_forced = S
lib = _lib.lib()
lib.qeft_debug_attn_stamps.argtypes = [ctypes.c_void_p]
lib.qeft_debug_attn_stamps.restype = None
dbg = torch.zeros(128 * S * 4 * 12, dtype=torch.int64, device=dev)   # up to 4 blocks per (head, split)
eng.reset(); eng.tok.fill_(1)
for _ in range(150):
    eng.step()
lib.qeft_debug_attn_stamps(dbg.data_ptr())
for _ in range(3):
    eng.step()
torch.cuda.synchronize()
lib.qeft_debug_attn_stamps(None)
d = dbg.view(128 * S, 4, 12).cpu().double()
d = d[d[:, 0, 1] > 0]                      # blocks that ran
rt = (d[..., 1] - d[..., 0]) * 10.0      # ns (100 MHz)
cyc = (d[..., 9] if S == 1 else d[..., 10]) - d[..., 2]
print("ns per wave (realtime) mean %.0f  max %.0f ; cycles mean %.0f -> %.2f GHz" % (rt.mean(), rt.max(), cyc.mean(), cyc.mean() / rt.mean()))
names = ["issue loads", "pos arrives", "rope+append+sync (cos/sin, q/k/v arrive)", "scores (K arrives)", "PV (V arrives)", "sync"] + (["merge+store"] if S == 1 else ["block merge + publish + vmcnt(0) + barrier", "ticket + barrier"])
ghz = cyc.mean() / rt.mean()
for i, n in enumerate(names):
    seg = (d[..., 3 + i] - d[..., 2 + i]) / ghz
    print("%-45s mean %7.0f ns   max %7.0f ns" % (n, seg.mean(), seg.max()))
span = (d[..., 1].max() - d[..., 0].min()) * 10
print("first entry -> last exit over the grid: %.0f ns" % span)
print("entry skew over blocks: %.0f ns" % ((d[..., 0].max() - d[..., 0].min()) * 10))

if S > 1:
    last = d[..., 11] > 0
    seg = (d[..., 11] - d[..., 10])[last] / ghz
    print("last arriver: merge of the records + store      mean %7.0f ns   max %7.0f ns  (%d waves)" % (seg.mean(), seg.max(), last.sum()))
