#!/bin/bash
# final evidence of the round: full GPU suite, smoke, default bench, rocprofv3 summaries
mkdir -p gpurun_out/r03/final
O=gpurun_out/r03/final
R=$GRAFT_REPO_ROOT
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log | cut -c1-400)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 1000 python bench.py > $O/r03_bench.json 2> $O/bench.err; echo "bench rc $?"
python tools/bench_summary.py $O/r03_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ph -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $R/$O/headline_profiled.json 2> $R/$O/ph.err
cp $R/$O/ph/p_kernel_stats.csv $R/$O/r03_headline_kernel_stats.csv
timeout -k 10 1100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/pb -o p -- python3 $R/bench.py > $R/$O/bench_profiled.json 2> $R/$O/pb.err
cp $R/$O/pb/p_kernel_stats.csv $R/$O/r03_bench_kernel_stats.csv
rm -f $R/$O/pb/p_kernel_trace.csv $R/$O/ph/p_kernel_trace.csv
cd $R && python tools/kstats.py $O/r03_headline_kernel_stats.csv 12
