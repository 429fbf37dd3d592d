#!/bin/bash
# Round-6 evidence kept under profiles/:
#  (1) rocprofv3 kernel stats of the bench's workload on one device lane (the kernel whose HIP-event time the bench line's roofline quotes);
#  (2) rocprofv3 kernel stats of the from-files leg on a 300 Mb draft (the BGZF block decoder k_inflate_lds, the CRC pass, the record split);
#  (3) the decoder's HBM traffic: two --pmc passes (FETCH_SIZE, WRITE_SIZE) over the A/B tool's 34 k-block launch.
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks6 -o ks -- python3 $R/bench.py --lanes 1 --no-pmc --no-lgs --no-phase --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --streamed-passes 1 --parity-big-seconds 0 > $O/r6_c5_bench_one_lane.json 2> $O/bench_c5_one_lane.err
python3 $R/tools/rocprof_summary.py stats $O/ks6/ks_results.db > $O/r6_c5_kernel_stats.txt
rm -rf $O/ks6
head -12 $O/r6_c5_kernel_stats.txt | cut -c1-120
python3 - <<PY
import os, sys, ctypes as C
sys.path.insert(0, "$R")
from nextpolish_amd import _native as nat
from concurrent.futures import ThreadPoolExecutor
d = "/tmp/np1_r6prof"; os.makedirs(d, exist_ok=True)
with ThreadPoolExecutor(8) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([2500000] * 12, depth=30.0, seed=300 + k, with_qual=0, prefix="b%dc" % k), range(10)))
L = nat.lib()
L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files_q(arr, len(sts), (d + "/g.fa").encode(), (d + "/r.bam").encode(), 1, 1) == 0
PY
rocprofv3 --kernel-trace --stats -d $O/ks6f -o ks -- $R/nextpolish_amd/bin/nextpolish1 scorechain /tmp/np1_r6prof/g.fa /tmp/np1_r6prof/r.bam > /dev/null 2> $O/e2e_300mb.err
python3 $R/tools/rocprof_summary.py stats $O/ks6f/ks_results.db > $O/r6_e2e_300mb_kernel_stats.txt
rm -rf $O/ks6f
head -12 $O/r6_e2e_300mb_kernel_stats.txt | cut -c1-120
for ctr in FETCH_SIZE WRITE_SIZE; do
  NP1_INFLATE=lds75 rocprofv3 --pmc $ctr -d $O/pmc6 -o pmc -- python3 $R/tests/tools/r6_inflate_ab.py --child /tmp/np1_r6prof/r.bam > $O/pmc_$ctr.out 2>&1
  python3 $R/tools/rocprof_summary.py pmc $O/pmc6/pmc_results.db | grep -i "inflate\|kernel " > $O/r6_inflate_pmc_$ctr.txt
  rm -rf $O/pmc6
  cat $O/r6_inflate_pmc_$ctr.txt | cut -c1-160
done
rm -rf /tmp/np1_r6prof
