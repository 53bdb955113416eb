#!/bin/bash
# rocprofv3 vector-instruction counts of the variants launched by tools/count_valu.py
export TMPDIR=/tmp
OUT=gpurun_out/valu_$1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o run -- python3 tools/count_valu.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_p1_rings" in r["Kernel_Name"]:
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in rows.items():
    n = len(next(iter(cs.values())))
    print(n, name.split("(")[0])
    print("    " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
