#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel trace and PMC passes of the bench command.
#   bash tools/profile_bench.sh            (writes under gpurun_out/prof_r03/)
# Counters go in their own passes, with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
CMD="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o run -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc2 -o run -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc3 -o run -- $CMD > $OUT/pmc3.log 2>&1
python3 tools/summarize_pmc.py --pmc $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 --trace $OUT/stats --kernel k_p1_rings --out $OUT/pmc_summary.json \
  --command "rocprofv3 --kernel-trace [--stats | --pmc <counters>] --output-format csv -- $CMD (one trace pass, three counter passes)"
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv
