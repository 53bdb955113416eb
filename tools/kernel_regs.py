"""Register / scratch / LDS use of every kernel in a hipcc --save-temps assembly file.

    hipcc --offload-arch=gfx950 ... -c --save-temps -o x.o file.hip
    python tools/kernel_regs.py file-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]
"""
import re
import subprocess
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    names, rows = [], []
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", block).group(1))  # noqa: E731
        names.append(name)
        rows.append((get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"),
                     get("group_segment_fixed_size")))
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for d, r in zip(dem, rows):
        d = d.replace("tfem::", "").split("(")[0]
        if not want or all(w in d for w in want):
            print(f"vgpr {r[0]:4d} sgpr {r[1]:4d} scratch {r[2]:5d} lds {r[3]:6d}  {d}")


if __name__ == "__main__":
    main()
