#!/bin/bash
# whole GPU suite on the current tree, then the darts numbers (weight step, architect step, encoder B=16)
set -o pipefail
timeout -k 10 1150 python -m pytest tests -q -m gpu -x --timeout 900 2>&1 | tail -15 > gpurun_out/r3_full_suite.txt
rc=$?
cat gpurun_out/r3_full_suite.txt
if [ $rc -ne 0 ]; then exit $rc; fi
