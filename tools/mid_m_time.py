"""Graph-replayed forward GEMM times below the M = 2048 tier (bench.mid_m_records) -- run on the GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ms = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,64,128,256,512,1024").split(","))
for r in bench.mid_m_records("cuda:0", ms=ms):
    print(r["shape"], {k: (v["us"], v["frac_of_peak"], v["variant"].replace("gemm_", "")) for k, v in r.items() if k != "shape"})
