#!/bin/bash
cd "$(dirname "$0")/.."
for d in 0.7 1.0 1.4 2.0 2.8; do echo -n "density $d: "; CLWRAP_GRID_DENSITY=$d python3 tools/run_config.py c4 --frames 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'])"; done
bash tools/ab_cfg2.sh "c4" "" _gw4 _gw6 _gw8
