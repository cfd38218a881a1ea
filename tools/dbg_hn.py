import sys, types
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import qeft_oracle as O
from test_gpu_gemv_v3 import make, ref, DEV
from qeft_amd import qeft_cuda, _lib
print("lib", _lib.LIB_PATH)
for n in (4096, 1024, 5120, 256):
    k = n; r = g = 128
    (l1, b1) = make(n, k, r, g, seed=1)
    rng = np.random.default_rng(0)
    x = O.make_activation(1, k, r, seed=6)
    h0 = rng.standard_normal(n).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(n)).astype(np.float16)
    gt = torch.from_numpy(gamma).to(DEV); xt = torch.from_numpy(x[0]).to(DEV)
    for rep in range(3):
        h32 = torch.from_numpy(h0).to(DEV)
        y32, hn, ssq = qeft_cuda.decode_linear(xt, l1, residual=h32, out=h32, gamma_out=gt)
        torch.cuda.synchronize()
        want = (y32 * gt.float()).half()
        bad = (hn != want).nonzero().flatten()
        print(n, rep, "mismatches", bad.numel(), bad[:8].tolist())
        for i in bad[:4].tolist():
            print("   row", i, "hn", hn[i].item(), "want", want[i].item(), "y32", y32[i].item(), "gamma", gt[i].item(), "prod", (y32[i] * gt[i].float()).item())
