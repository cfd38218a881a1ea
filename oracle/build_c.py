"""Build and load oracle/_build/libqeft_oracle.so (the C restatement; test infrastructure only)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libqeft_oracle.so")


def build():
    src = os.path.join(HERE, "qeft_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


def load():
    lib = ctypes.CDLL(build())
    p, i = ctypes.c_void_p, ctypes.c_int
    lib.qeft_oracle_dequant.argtypes = [p, p, p, p, p, i, i, i, i, i]
    lib.qeft_oracle_linear.argtypes = [p, p, p, p, i, i, i]
    lib.qeft_oracle_pack_oweight.argtypes = [p, p, i, i]
    lib.qeft_oracle_pack_intweight.argtypes = [p, p, i, i]
    return lib
