#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel statistics of the configurations beside the bench step:
# P2 (S(707) = config 3, Delaunay) and P1 on a Delaunay mesh (Morton / native numbering).
#   bash tools/prof_other_configs.sh        (writes under gpurun_out/prof_other/)
export TMPDIR=/tmp
OUT=gpurun_out/prof_other
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p2 -o run -- python3 tools/time_p2.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/delaunay -o run -- python3 tools/time_delaunay.py 1000000 > $OUT/delaunay.log 2>&1
for d in p2 delaunay; do
  f=$(ls $OUT/$d/*kernel_stats.csv $OUT/$d/*/*kernel_stats.csv 2>/dev/null | head -1)
  grep -E "Name|tfem::" $f > $OUT/${d}_kernel_stats.csv
done
tail -3 $OUT/p2.log; tail -5 $OUT/delaunay.log
