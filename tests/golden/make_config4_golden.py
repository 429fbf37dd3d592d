#!/usr/bin/env python
"""tests/golden/config4_golden.json: md5 + length of ctg_cns_core's output for every contig of the config-4-size workload
(tests/np2_cases.py:config4_groups), produced by the COMPILED REFERENCE (oracle/_ref/nextpolish2.so).  Build container only
(about 170 core-seconds of the reference)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.exit(subprocess.call([sys.executable, os.path.join(HERE, "..", "tools", "check_config4.py"), "--make-golden", "--library",
                          os.path.join(HERE, "..", "..", "oracle", "_ref", "nextpolish2.so")] + sys.argv[1:]))
