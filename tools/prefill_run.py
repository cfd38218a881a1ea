"""A 2048-token prompt through the whole Llama-2-7B (w4 g128 r128) model, 3 times: the run `rocprofv3 --kernel-trace --stats`
is pointed at to see what prefill spends outside the GEMMs."""
import os, sys, time, dataclasses, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qeft_amd.llama import LLAMA2_7B, QuantLlama, prefill
if len(sys.argv) > 1 and sys.argv[1] == "unfused":     # A/B: SiLU(gate) * up as its own launch behind the up_proj GEMM
    from qeft_amd import _lib
    from qeft_amd.qlinear import QuantLinear

    def two_launches(self, x, gate):
        up = self.forward(x)
        out = torch.empty_like(up)
        _lib.check(_lib.lib().qeft_silu_mul(gate.data_ptr(), up.data_ptr(), out.data_ptr(), up.numel(),
                                            torch.cuda.current_stream(up.device).cuda_stream))
        return out
    QuantLinear.forward_silu_mul = two_launches
dev = torch.device("cuda:0")
shape = dataclasses.replace(LLAMA2_7B, max_seq=2048)
model = QuantLlama(shape, dev, seed=0, fast_init=True)
tokens = torch.randint(0, shape.vocab, (2048,), generator=torch.Generator().manual_seed(0)).to(dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prefill(model, tokens)
    torch.cuda.synchronize(); print(f"prefill {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
