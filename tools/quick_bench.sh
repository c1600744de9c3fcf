# Quick A/B on the GPU box: bench.py without the CPU baseline and extras, one line per variant.
# usage: bash tools/quick_bench.sh <tag> "<bench args variant 1>" "<bench args variant 2>" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
i=0
for ARGS in "$@"; do
  timeout -k 10 180 python -u $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $ARGS > $R/gpurun_out/qb_${TAG}_$i.json 2> $R/gpurun_out/qb_${TAG}_$i.err || { echo "variant $i failed"; tail -3 $R/gpurun_out/qb_${TAG}_$i.err; exit 1; }
  python - "$R/gpurun_out/qb_${TAG}_$i.json" "$ARGS" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-40s %8.0f Mrays/s  %.4f ms/frame  closest %.4f  shadow %.4f  build %.2f" % (sys.argv[2], d["value"], d["ms_per_frame"], k["closest_ms"], k["shadow_ms"], d["build_ms"]))
PY
  i=$((i+1))
done
