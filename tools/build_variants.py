"""Developer tool: alternative builds of libtfem_hip.so that differ in ONE translation unit
(extra -D flags), for A/B timing on the GPU box through TFEM_HIP_LIB.

    python tools/build_variants.py tfem_rings_src.hip name1="-DTFEM_SRC_ABL=1 -DTFEM_SRC_BENCH_ONLY" name2=...

writes tools/variants/libtfem_<name>.so (git-ignored; travels with gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as g  # noqa: E402


def main():
    g.build_hip()
    units = sys.argv[1].split(",")  # one or several translation units, comma-separated
    variants = dict(a.split("=", 1) for a in sys.argv[2:])
    objdir = os.path.join(g.CSRC, "build")
    outdir = os.path.join(REPO, "tools", "variants")
    os.makedirs(outdir, exist_ok=True)
    others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir))
              if f.endswith(".o") and f[:-2] not in units]
    base = [f for f in g.HIPCC_FLAGS if f != "-shared"]

    def one(item):
        name, extra = item
        objs = []
        for unit in units:
            obj = os.path.join(outdir, f"{unit}.{name}.o")
            cmd = ["/opt/rocm/bin/hipcc", *base, *g.PER_FILE_FLAGS.get(unit, []), *extra.split(),
                   "-I" + os.path.join(REPO, "include"), "-c", "-o", obj, os.path.join(g.CSRC, unit)]
            subprocess.run(cmd, check=True)
            objs.append(obj)
        lib = os.path.join(outdir, f"libtfem_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *others, *objs], check=True)
        for obj in objs:
            os.remove(obj)
        print("built", lib, flush=True)

    with ThreadPoolExecutor(max_workers=min(len(variants), 6)) as pool:
        list(pool.map(one, variants.items()))


if __name__ == "__main__":
    main()
