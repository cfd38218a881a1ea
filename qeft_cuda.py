"""Top-level alias so the reference's `import qeft_cuda` (qeft/qlinear.py:8-11,
qeft/monkeypatch/ftllama_modeling.py:18) resolves to the MI355X implementation."""
from qeft_amd.qeft_cuda import *  # noqa: F401,F403
from qeft_amd.qeft_cuda import (gemm_4bit, gemv_4bit, gemv_4bit_qeft, layernorm_forward_cuda,  # noqa: F401
                                single_query_attention)
