export TMPDIR=/tmp
rm -rf gpurun_out/prof_d && mkdir -p gpurun_out/prof_d
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -o run -- python3 tools/time_delaunay.py 1000000 > gpurun_out/prof_d/log.txt 2>&1
head -12 gpurun_out/prof_d/*kernel_stats.csv 2>/dev/null || head -12 gpurun_out/prof_d/*/*kernel_stats.csv
cat gpurun_out/prof_d/log.txt | tail -5
