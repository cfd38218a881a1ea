"""Golden vectors for the 3-bit extension: the REFERENCE's own quantiser run at bits = 3.

    PYTHONPATH=/root/reference python tests/golden/make_golden_w3.py       (build container only)

The reference cannot pack 3 bits (QuantLinear asserts bits == 4, qlinear.py:127) but its Quantizer can produce
3-bit parameters and fake-quantised weights (quant.py:8-10, 142-158 with maxq = 7).  Each fixture holds the input
weight and the reference's scale / zero / fake-quantised weight per 128-column group: data only.
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference")
from qeft.quant import Quantizer, quantize  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(16, 256, 128), (64, 512, 128), (32, 384, 384)]   # (N, K, group)


def main():
    for ci, (n, k, g) in enumerate(CASES):
        torch.manual_seed(3000 + ci)
        w = (torch.randn(n, k) * 0.02).half()
        scales, zeros, wq = [], [], torch.empty(n, k, dtype=torch.float32)
        for g0 in range(0, k, g):
            qz = Quantizer(bits=3, perchannel=True, sym=False, mse=False, group_size=g)
            slab = w[:, g0:g0 + g].float()
            qz.find_params(slab, weight=True)
            wq[:, g0:g0 + g] = quantize(slab, qz.scale, qz.zero, qz.minq, qz.maxq)
            scales.append(qz.scale.reshape(n, 1))
            zeros.append(qz.zero.reshape(n, 1))
        assert int(qz.maxq) == 7
        np.savez_compressed(os.path.join(HERE, f"quant3_case{ci}.npz"), case=np.array([n, k, g], dtype=np.int64),
                            w_orig=w.numpy(), scale=torch.cat(scales, 1).float().numpy(),
                            zero=torch.cat(zeros, 1).float().numpy(), w_fake=wq.numpy())
        print("wrote quant3_case%d.npz" % ci)


if __name__ == "__main__":
    main()
