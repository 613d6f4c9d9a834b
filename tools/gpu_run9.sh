#!/bin/bash
# headline step alone under rocprofv3 --kernel-trace --stats  -> profiles/r03_headline_kernel_stats.csv
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/ph -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $R/gpurun_out/r03/bh.json 2> $R/gpurun_out/r03/bh.err || exit 1
cd $R && python tools/kstats.py gpurun_out/r03/ph/p_kernel_stats.csv 14
python tools/bench_summary.py gpurun_out/r03/bh.json
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03/ph/p_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'coarse_scan_kernel<24, 1' in r['Kernel_Name']]
i0=idx[120]; i1=idx[121]
prev=None
for r in rows[i0:i1+1]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'].replace('(anonymous namespace)::','')[:46]:48s} dur {(en-st)/1e3:7.1f} gap {((st-prev)/1e3 if prev else 0):6.1f}")
    prev=en
PY
