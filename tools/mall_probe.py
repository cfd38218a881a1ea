"""How much faster is a decode GEMV whose weights are already in the Infinity Cache?  Each kind of launch of the 7B layer,
32 launches per graph replay: (cold) the 32 layers' operands in turn, (hot) layer 0's operand 32 times."""
import os, sys, dataclasses, torch
sys.path.insert(0, os.getcwd())
from bench import _event_time_us
from qeft_amd import _lib
from qeft_amd.llama import LLAMA2_7B, DecodeEngine, QuantLlama
dev = torch.device("cuda:0")
lib = _lib.lib()
shape = dataclasses.replace(LLAMA2_7B, max_seq=512)
model = QuantLlama(shape, dev, seed=0, fast_init=True)
eng = DecodeEngine(model, use_graph=False)
s = shape
g, no, eps = s.group_size, s.n_out, s.rms_eps
xn, ssq, h32 = eng.xn.data_ptr(), eng.ssq.data_ptr(), eng.h32.data_ptr()
gam = model.model.layers[0].post_attention_layernorm.weight.data_ptr()

def launch(kind, op, st):
    def lin(x, y, mode=0, residual=None, ssq_in=None, n_ssq=0, gamma_out=None):
        _lib.check(lib.qeft_decode_linear(x, op.qweight.data_ptr(), op.sz_packed.data_ptr(), op.oweight.data_ptr() if no else None,
                                          None, y, op.outfeatures, op.infeatures, g, no, mode, residual, ssq_in, n_ssq, eps,
                                          gamma_out, xn if gamma_out else None, ssq if gamma_out else None, st))
    if kind == "qkv":
        lin(xn, eng.qkv_out.data_ptr(), ssq_in=ssq, n_ssq=eng.n_ssq_lin)
    elif kind == "o":
        lin(eng.att.data_ptr(), h32, residual=h32, gamma_out=gam)
    elif kind == "gu":
        lin(xn, eng.act.data_ptr(), mode=1, ssq_in=ssq, n_ssq=eng.n_ssq_lin)
    else:
        lin(eng.act.data_ptr(), h32, residual=h32, gamma_out=gam)

side = torch.cuda.Stream(dev)
for kind in ("qkv", "o", "gu", "d"):
    res = {}
    for mode in ("cold", "hot"):
        ops = [eng.v3ops[li if mode == "cold" else 0][kind] for li in range(32)]
        with torch.cuda.stream(side):
            for op in ops:
                launch(kind, op, side.cuda_stream)
            torch.cuda.synchronize(dev)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for op in ops:
                    launch(kind, op, side.cuda_stream)
            res[mode] = _event_time_us(gr.replay, 20, dev) / 32
    op = eng.v3ops[0][kind]
    nbytes = op.qweight.numel() + op.sz_packed.numel() * op.sz_packed.element_size() + (op.oweight.numel() * 2 if no else 0)
    print(f"{kind}: {nbytes / 1e6:.1f} MB  cold {res['cold']:.2f} us ({nbytes / res['cold'] / 1e6:.2f} TB/s)   hot {res['hot']:.2f} us ({nbytes / res['hot'] / 1e6:.2f} TB/s)", flush=True)
