#!/bin/bash
# Round-4 evidence kept under profiles/ (run on the GPU box from the repo root; PARTS selects, default all):
#   bench    the default bench line (the metric's 3 Gb configuration)                               -> r4_c5_bench.json
#   stats    rocprofv3 kernel stats of the same workload, one device lane                           -> r4_c5_kernel_stats.txt
#   pmc      the two PMC passes over the first four batches                                          -> r4_c5_pmc_fetch.txt / _write.txt
#   phase    rocprofv3 kernel stats of the snp_phase pass on the bench's 20 Mb diploid draft        -> r4_snp_phase_kernel_stats_20mb.txt
#   e2e      rocprofv3 kernel stats of `nextpolish1 scorechain` FROM FILES, 300 Mb / 30x, binned q. -> r4_e2e_300mb_kernel_stats.txt
#   np2      the long-read window's kernel stats + PMC                                               -> r4_np2_*.txt
#   tile9    k_tile3 against k_tile9 on config 3 with k_tile9's phase clocks                         -> r4_tile9_ab.txt
set -x
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r4
mkdir -p $O
PARTS=${PARTS:-bench stats pmc phase e2e np2 tile9}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python bench.py > $O/r4_c5_bench.json 2> $O/bench_c5.err
  tail -c 400 $O/bench_c5.err
fi
cd /tmp && export TMPDIR=/tmp
if has stats; then
  rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python $R/bench.py --lanes 1 --no-pmc --no-lgs --no-phase --no-cpu-baseline --no-e2e --steps 2 --resident-passes 2 > $O/r4_c5_bench_one_lane.json 2>> $O/bench_c5.err
  python $R/tools/rocprof_summary.py stats $O/ks/ks_results.db > $O/r4_c5_kernel_stats.txt
  rm -rf $O/ks
fi
if has pmc; then
  rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- python $R/bench.py --pmc-child --steps 1 --warmup 0 --pmc-batches 4 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- python $R/bench.py --pmc-child --steps 1 --warmup 0 --pmc-batches 4 > /dev/null 2>&1
  python $R/tools/rocprof_summary.py pmc $O/pf/pf_results.db > $O/r4_c5_pmc_fetch.txt
  python $R/tools/rocprof_summary.py pmc $O/pw/pw_results.db > $O/r4_c5_pmc_write.txt
  rm -rf $O/pf $O/pw
fi
if has phase; then
  rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/tests/tools/np1_phase_bench.py synth:20 3 --no-oracle > $O/r4_snp_phase_bench_20mb.txt 2>&1
  python $R/tools/rocprof_summary.py stats $O/ph/ph_results.db > $O/r4_snp_phase_kernel_stats_20mb.txt
  rm -rf $O/ph
fi
cd $R
if has e2e; then
  python - <<'PY'
import sys, os, ctypes as C
sys.path.insert(0, ".")
from nextpolish_amd import _native as nat
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(4) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([25000000, 12500000], depth=30, seed=300 + k, with_qual=1, prefix="e%dc" % k), range(8)))
L = nat.lib()
L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files_q(arr, len(sts), b"/tmp/r4_g.fa", b"/tmp/r4_r.bam", 1, 1) == 0
print("300 Mb draft, BAM MB", os.path.getsize("/tmp/r4_r.bam") / 1e6, "records", sum(s.n_reads for s in sts))
PY
  cd /tmp
  NP1_BATCH_BP=260000000 $R/nextpolish_amd/bin/nextpolish1 scorechain /tmp/r4_g.fa /tmp/r4_r.bam > /dev/null 2> $O/e2e_warm.err
  NP1_BATCH_BP=260000000 NP1_TIMING=1 rocprofv3 --kernel-trace --stats -d $O/e2 -o e2 -- $R/nextpolish_amd/bin/nextpolish1 scorechain /tmp/r4_g.fa /tmp/r4_r.bam > /dev/null 2> $O/r4_e2e_300mb_cli.err
  python $R/tools/rocprof_summary.py stats $O/e2/e2_results.db > $O/r4_e2e_300mb_kernel_stats.txt
  rm -rf $O/e2 /tmp/r4_g.fa* /tmp/r4_r.bam*
  cd $R
fi
if has np2; then
  tests/tools/np2_prof.sh gpurun_out/r4/np2 5 > /dev/null 2>&1 || true
  cp gpurun_out/r4/np2/np2_kernel_stats.txt $O/r4_np2_kernel_stats_5mb.txt; cp gpurun_out/r4/np2/np2_pmc_fetch.txt $O/r4_np2_pmc_fetch.txt; cp gpurun_out/r4/np2/np2_pmc_write.txt $O/r4_np2_pmc_write.txt
  rm -rf gpurun_out/r4/np2
fi
if has tile9; then
  { echo "# bench.py --workload c3_100mb_30x --steps 3, tile stage per 13 Mb batch: NP1_TILE=3 (k_tile3, default) vs NP1_TILE=9 (k_tile9)";
    PROBE=0 TESTS=0 KINDS="3 9" bash tests/tools/r4_tile9_ab.sh 2>&1 | grep -E "bench tile|stage_ms|value";
    echo "# k_tile9 phase clocks (NP1_T9_PHASES=1; shader-clock cycles per wave)";
    NP1_T9_PHASES=1 PROBE=0 TESTS=0 KINDS=9 bash tests/tools/r4_tile9_ab.sh > /dev/null 2>&1; grep "k_tile9 cycles" gpurun_out/r4_tile9/bench_tile9.err | tail -2; } > $O/r4_tile9_ab.txt
fi
ls -la $O
