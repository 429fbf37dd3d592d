#!/bin/bash
# Round 4: k_tile9 against k_tile3 -- parity tests, then the same bench workload with either kernel (NP1_TILE=3 | 9).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4_tile9
mkdir -p "$OUT"
W=${1:-c3_100mb_30x}
if [[ "${PROBE:-1}" == 1 ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -I nextpolish_amd/csrc -o /tmp/np_probe tests/tools/efence_probe.hip > "$OUT/probe.log" 2>&1 && timeout 120 /tmp/np_probe fault >> "$OUT/probe.log" 2>&1
  echo "probe rc=$?"; tail -25 "$OUT/probe.log"
fi
if [[ "${TESTS:-1}" == 1 ]]; then
  timeout 900 python -m pytest tests/test_gpu_score_chain.py tests/test_real_data.py -m gpu -x -q -p no:cacheprovider > "$OUT/tests.log" 2>&1
  echo "tests rc=$?"; tail -15 "$OUT/tests.log"
fi
for k in ${KINDS:-3 9}; do
  NP1_TILE=$k timeout 900 python bench.py --workload "$W" --steps 3 --warmup 1 --no-pmc --no-e2e --no-lgs --no-phase --no-cpu-baseline > "$OUT/bench_tile$k.json" 2> "$OUT/bench_tile$k.err"
  echo "bench tile$k rc=$?"
  python - "$OUT/bench_tile$k.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print({k: j.get(k) for k in ("value", "ms_per_step")}, "resident", j.get("resident"), "parity", j.get("parity"))
    print("stage_ms", r.get("stage_ms"), "frac", r.get("frac"))
except Exception as e:
    print("no json:", e)
PY
  tail -3 "$OUT/bench_tile$k.err"
done
