#!/bin/bash
# Collects the round-3 evidence kept under profiles/: the default bench line (the metric's 3 Gb configuration), rocprofv3 kernel
# stats of the same workload (one device lane, so that kernels of different batches do not overlap and the per-kernel durations are
# the ones the HIP events of bench.py see; no child legs), the two PMC passes over the first four batches, the long-read window's
# kernel stats.  Run on the GPU box from the repo root.
set -x
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r3
mkdir -p $O
python bench.py > $O/r3_c5_bench.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python $R/bench.py --lanes 1 --no-pmc --no-lgs --no-phase --no-cpu-baseline --no-e2e --steps 2 --resident-passes 2 > $O/r3_c5_bench_one_lane.json 2>> $O/bench_c5.err
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- python $R/bench.py --pmc-child --steps 1 --warmup 0 --pmc-batches 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- python $R/bench.py --pmc-child --steps 1 --warmup 0 --pmc-batches 4 > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py stats $O/ks/ks_results.db > $O/r3_c5_kernel_stats.txt
python tools/rocprof_summary.py pmc $O/pf/pf_results.db > $O/r3_c5_pmc_fetch.txt
python tools/rocprof_summary.py pmc $O/pw/pw_results.db > $O/r3_c5_pmc_write.txt
tests/tools/np2_prof.sh gpurun_out/r3/np2 5 > /dev/null 2>&1 || true
cp gpurun_out/r3/np2/np2_kernel_stats.txt $O/r3_np2_kernel_stats_5mb.txt; cp gpurun_out/r3/np2/np2_pmc_fetch.txt $O/r3_np2_pmc_fetch.txt; cp gpurun_out/r3/np2/np2_pmc_write.txt $O/r3_np2_pmc_write.txt
rm -rf gpurun_out/r3/np2
rm -rf $O/ks $O/pf $O/pw
ls -la $O
