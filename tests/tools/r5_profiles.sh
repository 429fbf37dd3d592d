#!/bin/bash
# Round-5 evidence kept under profiles/: rocprofv3 kernel stats of the bench's workload on one device lane (the kernel whose HIP-event time the bench line's roofline quotes)
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks5 -o ks -- python $R/bench.py --lanes 1 --no-pmc --no-lgs --no-phase --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --resident-passes 2 --parity-big-seconds 0 > $O/r5_c5_bench_one_lane.json 2> $O/bench_c5_one_lane.err
python $R/tools/rocprof_summary.py stats $O/ks5/ks_results.db > $O/r5_c5_kernel_stats.txt
rm -rf $O/ks5
head -12 $O/r5_c5_kernel_stats.txt | cut -c1-120
