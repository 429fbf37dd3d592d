set -x
cd /root/repo
mkdir -p gpurun_out/r2
python bench.py > gpurun_out/r2/bench_c3.json 2> gpurun_out/r2/bench_c3.err
cd /tmp && export TMPDIR=/tmp
# kernel stats of the same command (resident passes + streamed + e2e legs inside one process), no pmc/lgs/cpu legs
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2/ks -o ks -- python /root/repo/bench.py --no-pmc --no-lgs --no-cpu-baseline --no-e2e > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/r2/pf -o pf -- python /root/repo/bench.py --pmc-child --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/r2/pw -o pw -- python /root/repo/bench.py --pmc-child --steps 1 --warmup 0 > /dev/null 2>&1
cd /root/repo
python tools/rocprof_summary.py stats gpurun_out/r2/ks/ks_results.db > gpurun_out/r2/r2_c3_kernel_stats.txt
python tools/rocprof_summary.py pmc gpurun_out/r2/pf/pf_results.db > gpurun_out/r2/r2_c3_pmc_fetch.txt
python tools/rocprof_summary.py pmc gpurun_out/r2/pw/pw_results.db > gpurun_out/r2/r2_c3_pmc_write.txt
tests/tools/np1_e2e_prof.sh 100 30 0 > gpurun_out/r2/r2_e2e_100mb_kernel_stats.txt 2>&1
rm -rf gpurun_out/r2/ks gpurun_out/r2/pf gpurun_out/r2/pw gpurun_out/e2e_prof/*.db
ls -la gpurun_out/r2
