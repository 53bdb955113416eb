#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel trace and PMC passes of P2 config 3 (tools/time_p2.py):
# kernel durations and HBM traffic of k_p2_rows (vertex rows / edge rows) -> gpurun_out/prof_p2/
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_p2
rm -rf $OUT && mkdir -p $OUT
CMD="python3 tools/time_p2.py 200000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o run -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc2 -o run -- $CMD > $OUT/pmc2.log 2>&1
python3 tools/summarize_pmc.py --pmc $OUT/pmc1 $OUT/pmc2 --trace $OUT/stats --kernel k_p2_rows --out $OUT/pmc_summary.json \
  --command "rocprofv3 --kernel-trace [--stats | --pmc <counters>] --output-format csv -- $CMD"
cat $OUT/stats.log | grep -v amdgpu.ids
