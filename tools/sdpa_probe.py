import torch, time
from torch.nn.attention import sdpa_kernel, SDPBackend
q = torch.randn(1, 32, 2048, 128, device="cuda", dtype=torch.float16)
k = torch.randn_like(q); v = torch.randn_like(q)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel([be]):
            ms = t(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True))
        print(name, f"{ms:.3f} ms")
    except Exception as e:
        print(name, "unavailable:", str(e)[:200])
ms = t(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True))
print("default 4-D", f"{ms:.3f} ms")
q3, k3, v3 = q[0], k[0], v[0]
ms = t(lambda: torch.nn.functional.scaled_dot_product_attention(q3, k3, v3, is_causal=True))
print("default 3-D", f"{ms:.3f} ms")
