#!/bin/bash
# Collects the round-2 evidence kept under profiles/: the default bench line, rocprofv3 kernel stats of the same command
# (one device lane, so that kernels of different batches do not overlap and the per-kernel durations are the ones the HIP
# events of bench.py see), the two PMC passes, and the from-files kernel stats.  Run on the GPU box from the repo root.
set -x
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out/r2
python bench.py > gpurun_out/r2/r2_c3_bench.json 2> gpurun_out/r2/bench_c3.err
python bench.py --lanes 1 --no-pmc --no-lgs --no-cpu-baseline --no-e2e > gpurun_out/r2/r2_c3_bench_one_lane.json 2>> gpurun_out/r2/bench_c3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2/ks -o ks -- python $R/bench.py --lanes 1 --no-pmc --no-lgs --no-cpu-baseline --no-e2e > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r2/pf -o pf -- python $R/bench.py --pmc-child --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r2/pw -o pw -- python $R/bench.py --pmc-child --steps 1 --warmup 0 > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py stats gpurun_out/r2/ks/ks_results.db > gpurun_out/r2/r2_c3_kernel_stats.txt
python tools/rocprof_summary.py pmc gpurun_out/r2/pf/pf_results.db > gpurun_out/r2/r2_c3_pmc_fetch.txt
python tools/rocprof_summary.py pmc gpurun_out/r2/pw/pw_results.db > gpurun_out/r2/r2_c3_pmc_write.txt
tests/tools/np1_e2e_prof.sh 100 30 0 > gpurun_out/r2/r2_e2e_100mb_kernel_stats.txt 2>&1
python tests/tools/np1_e2e_timing.py 100 30 0 2>&1 | grep -v "^\[np1" > gpurun_out/r2/r2_e2e_100mb_timing.txt
python tests/tools/np1_e2e_timing.py 100 30 1 2>&1 | grep -v "^\[np1" > gpurun_out/r2/r2_e2e_100mb_random_qualities_timing.txt
for q in 0 1; do python tests/tools/np1_inflate_prof.py $q 64; done > gpurun_out/r2/r2_inflate_phase_clocks.txt 2>&1
python tests/tools/np1_inflate_prof.py 0 64 tests/golden/real/sgs.sort.bam >> gpurun_out/r2/r2_inflate_phase_clocks.txt 2>&1
rm -rf gpurun_out/r2/ks gpurun_out/r2/pf gpurun_out/r2/pw gpurun_out/e2e_prof/*.db
ls -la gpurun_out/r2
