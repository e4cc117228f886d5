#!/bin/bash
# GPU test driver: the parity suite under every kernel variant, each test under a watchdog.
set -u
python -m pytest tests -m gpu -q --timeout=90 -x 2>&1 | tail -8
for kd in 0 4; do
  echo "== DR_FORCE_KIND=$kd"
  DR_FORCE_KIND=$kd timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=60 -x 2>&1 | tail -4
done
