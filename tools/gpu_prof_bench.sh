#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (what roofline.us_per_launch is checked against).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o run --output-format csv -- python3 $OLDPWD/bench.py --no-traffic > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
cd $OLDPWD
python3 tools/summarize_rocprof.py $OUT/kt gpurun_out/r04_bench_kernel_stats.txt
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r04_bench_kernel_stats.csv 2>/dev/null
cp $OUT/bench.json gpurun_out/r04_bench_n1_under_rocprof.json.log
du -sh $OUT | tail -1
rm -rf $OUT/kt        # the raw kernel trace is tens of MB: gpurun copies back at most 64 MiB
head -40 gpurun_out/r04_bench_kernel_stats.txt
tail -c 600 $OUT/bench.json
