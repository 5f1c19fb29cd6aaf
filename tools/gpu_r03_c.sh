#!/bin/bash
# stamps + instruction counters: VIS kernel (variant 0) against the batch kernel (variant 256) at C2
mkdir -p gpurun_out; rm -f gpurun_out/stamps_r03c.log
for v in 0 256; do timeout -k 10 200 python tools/stamp_phases.py c2 --variant $v >> gpurun_out/stamps_r03c.log 2>&1; timeout -k 10 200 python tools/stamp_phases.py c2 --strict --variant $v >> gpurun_out/stamps_r03c.log 2>&1; done
cat gpurun_out/stamps_r03c.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 256; do
  out=gpurun_out/pmc_r03c_v$v; rm -rf $out
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out -- python3 tools/run_config.py c2 --frames 10 --variant $v > $out.log 2>&1
  python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wt_trace" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in agg:
    print(k[:60], "launches", n[k], {c: round(v / max(n[k], 1)) for c, v in agg[k].items()})
PY
done
