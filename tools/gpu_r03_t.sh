#!/bin/bash
python3 tools/run_config.py c3 --frames 6 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 default', d['kernel_ms'], d['lane_util'])"
python3 tools/run_config.py c3 --frames 6 --variant 64 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 low-occupancy flavour + tail', d['kernel_ms'], d['lane_util'], d['counters'].get('tpt_tiles'))"
CLWRAP_TPT_MAX=24 python3 tools/run_config.py c3 --frames 6 --variant 64 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 low-occ + tail<=24', d['kernel_ms'], d['lane_util'])"
