"""The bench's boundary_gemv record alone (the reference's gemv entries, graph-replayed) -- run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for r in bench.boundary_gemv_records("cuda:0")["per_shape"]:
    print(r["shape"], {k: (v["us"], v["variant"]) for k, v in r.items() if k != "shape"})
