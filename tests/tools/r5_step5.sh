#!/bin/bash
# Round 5, fifth GPU step: the structural layer with its read coordinates from the device (goldens, a slice of the structural fuzz against oracle/_ref)
cd "$(dirname "$0")/../.."
O=gpurun_out/r5
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_np2.py -x -q -m gpu -p no:cacheprovider -k "structural or two_windows or real" > $O/np2_sv_tests.log 2>&1; echo "np2 sv tests rc=$? $(tail -1 $O/np2_sv_tests.log)"
timeout 300 python -m pytest tests/test_real_data.py -x -q -m gpu -p no:cacheprovider -k "long_read" > $O/np2_real_tests.log 2>&1; echo "real long-read tests rc=$? $(tail -1 $O/np2_real_tests.log)"
timeout 400 python tests/tools/np2_fuzz_gpu.py 500 512 sv > $O/np2_fuzz_sv.log 2>&1; echo "sv fuzz rc=$? $(tail -3 $O/np2_fuzz_sv.log | tr '\n' ' ')"
