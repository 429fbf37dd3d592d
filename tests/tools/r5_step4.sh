#!/bin/bash
# Round 5, fourth GPU step: what has not run on the GPU yet (tiles over two ranks, the drop-in calls of smoke, the exact task-1 list), the long-read tests on
# page-locked record arrays, and the long-read leg alone
cd "$(dirname "$0")/../.."
O=gpurun_out/r5
mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_tiling.py "tests/test_gpu_score_chain.py::test_dropin_abi_and_cli" -x -q -m gpu -p no:cacheprovider > $O/tiling_tests.log 2>&1; echo "tiling + dropin rc=$? $(tail -1 $O/tiling_tests.log)"; grep -a -E "Error|assert|FAILED" $O/tiling_tests.log | head -8
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $O/smoke.log | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_np2.py -x -q -m gpu -p no:cacheprovider > $O/np2_tests.log 2>&1; echo "np2 tests rc=$? $(tail -1 $O/np2_tests.log)"
timeout 200 python tests/tools/r4_lgs_quick.py 24 12 2> $O/lgs_quick.err | cut -c1-400
NP2_PAGEABLE_RECORDS=1 timeout 200 python tests/tools/r4_lgs_quick.py 24 12 2>> $O/lgs_quick.err | cut -c1-400
