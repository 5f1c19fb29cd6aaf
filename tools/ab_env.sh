#!/bin/bash
# per-frame kernel times under environment settings; usage: tools/ab_env.sh cfg "VAR=1" "VAR=0" ...
cd "$(dirname "$0")/.."
cfg=$1; shift
for round in 1 2 3; do
for e in "$@"; do
  env $e python3 tools/frame_times.py $cfg --frames 24 2>/dev/null | python3 -c "
import sys,json,statistics; d=json.loads(sys.stdin.read()); m=d['ms']; print('%-24s' % '$e', 'first', m[:3], 'median of rest', round(statistics.median(m[4:]),4), 'max', max(m[4:]))"
done
done
