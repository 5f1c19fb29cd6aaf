#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_r03q.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_r03q.log | cut -c1-300
