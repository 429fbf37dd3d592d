#!/bin/bash
# rocprofv3 kernel stats of the from-files short-read CLI on a generated workload: np1_e2e_prof.sh [Mb] [depth] [with_qual]
set -e
cd "$(dirname "$0")/../.."
MB=${1:-50}; DEPTH=${2:-30}; WQ=${3:-0}
python - <<PY
import sys, os, ctypes as C
sys.path.insert(0, ".")
from nextpolish_amd import _native as nat
from concurrent.futures import ThreadPoolExecutor
nb = max(1, int($MB / 12.5))
with ThreadPoolExecutor(8) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([2500000] * int($MB / nb / 2.5), depth=$DEPTH, seed=100 + k, with_qual=$WQ, prefix="b%dc" % k), range(nb)))
L = nat.lib()
L.np1_streams_write_files.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files(arr, len(sts), b"/tmp/prof_g.fa", b"/tmp/prof_r.bam", 1) == 0
print("BAM MB", os.path.getsize("/tmp/prof_r.bam") / 1e6)
PY
EXE=$PWD/nextpolish_amd/bin/nextpolish1
OUT=$PWD/gpurun_out/e2e_prof
mkdir -p $OUT
OLDPWD_=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o e2e -- $EXE scorechain /tmp/prof_g.fa /tmp/prof_r.bam > /dev/null 2> $OUT/cli.err || true
python $OLDPWD_/tools/rocprof_summary.py stats $OUT/e2e_results.db
