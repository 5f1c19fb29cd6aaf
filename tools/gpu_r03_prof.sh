#!/bin/bash
# round-3 evidence: the driver-style bench line, then rocprofv3 kernel stats + PMC passes for C2 (bench command), C3, C4 and the reference's own 800x600 depth 15
mkdir -p gpurun_out/r03
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_n1_steps20.json 2> gpurun_out/r03/bench.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r03/bench_n1_steps20.json
bash tools/profile_round.sh r03 > gpurun_out/r03/profile_round.log 2>&1; tail -2 gpurun_out/r03/profile_round.log | cut -c1-200
python3 tools/summarize_prof.py gpurun_out/prof_r03 gpurun_out/r03 r03_c2 > /dev/null 2>&1; echo "c2 summary rc=$?"
for cfg in c3 c4 ref800; do
  bash tools/profile_config.sh r03 $cfg 8 > gpurun_out/r03/profile_$cfg.log 2>&1; tail -2 gpurun_out/r03/profile_$cfg.log | cut -c1-250
  python3 tools/summarize_prof.py gpurun_out/prof_r03_$cfg gpurun_out/r03 r03_$cfg > /dev/null 2>&1; echo "$cfg summary rc=$?"
done
ls -la gpurun_out/r03
