#!/bin/bash
# rocprofv3 passes over one GEMM shape: kernel trace + three PMC passes (separate runs, counters only).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_gemm
rm -rf $OUT; mkdir -p $OUT
ARGS="${GEMM_ARGS:-4096 4096 2048}"
cd /tmp
GEMM_ROUNDS=150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run --output-format csv -- python3 $OLDPWD/tools/gemm_one.py $ARGS > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/p1 -o run --output-format csv -- python3 $OLDPWD/tools/gemm_one.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p2 -o run --output-format csv -- python3 $OLDPWD/tools/gemm_one.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $OUT/p3 -o run --output-format csv -- python3 $OLDPWD/tools/gemm_one.py $ARGS > $OUT/p3.log 2>&1
cd $OLDPWD
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/prof_gemm")
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/prof_gemm/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_w4" in k or "gemm_" in k and "qeft" in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(p, k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
for f in glob.glob("gpurun_out/prof_gemm/kt/**/*kernel_stats.csv", recursive=True):
    for i, line in enumerate(open(f)):
        if i < 6: print(line.rstrip()[:200])
# steady state: durations of the last 400 launches of the GEMM kernel in the trace (the first ms run at idle clocks)
for f in glob.glob("gpurun_out/prof_gemm/kt/**/*kernel_trace.csv", recursive=True):
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f))
         if "gemm_w4" in r["Kernel_Name"]]
    d.sort()
    dur = [v for _, v in d]
    if dur:
        last = sorted(dur[-400:])
        print(f"kernel_trace: {len(dur)} launches; first 20 avg {sum(dur[:20]) / 20 / 1e3:.1f} us; last {len(last)}: avg {sum(last) / len(last) / 1e3:.1f} us, "
              f"median {last[len(last) // 2] / 1e3:.1f} us, min {last[0] / 1e3:.1f} us")
PY
