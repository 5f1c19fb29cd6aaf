#!/bin/bash
# C3: depth-sized scratch stack (variant 0) against the full-depth one (variant 2048): time and L2 <-> fabric traffic
mkdir -p gpurun_out
for v in 4096 0 4096 0; do timeout -k 10 200 python3 tools/run_config.py c3 --frames 20 --variant $v | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'],'variant',d['variant'],d['kernel_ms'],'pushes',d['counters']['pushes'])"; done
for v in 2048 0; do timeout -k 10 200 python3 tools/run_config.py ref800 --frames 30 --variant $v | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'],'variant',d['variant'],d['kernel_ms'])"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/pmc_r03g_v${v}_$ctr; rm -rf $out
    rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 tools/run_config.py c3 --frames 6 --variant $v > $out.log 2>&1
    python3 - $out $ctr $v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wt_trace" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
            agg[r["Kernel_Name"] + " scratch " + r["Scratch_Size"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("variant", sys.argv[3], k[:70], sys.argv[2], "KiB/launch mean", round(sum(v) / len(v)), "launches", len(v))
PY
  done
done
