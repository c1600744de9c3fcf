# Kernel durations of the bench workload for several builds of the library (run on the GPU box).
# usage: bash tools/variant_stats.sh <tag> <lib1.so> <lib2.so> ...   (paths relative to the repository root)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  export RTX_LIB=$R/$LIB
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vs_${TAG}_$NAME -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS} > $R/gpurun_out/vs_${TAG}_$NAME.log 2>&1 || { echo "$NAME failed"; tail -3 $R/gpurun_out/vs_${TAG}_$NAME.log; exit 1; }
  python3 - $R/gpurun_out/vs_${TAG}_$NAME $NAME <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("k_shade_a", "k_shade_b", "k_trace_queue", "k_shadow", "k_scan_chunks"):
        if k in n: out.append("%s %.1f" % (k[2:], float(r["AverageNs"]) / 1e3))
print("%-22s" % sys.argv[2], "  ".join(out), "(us per launch)")
PY
done
