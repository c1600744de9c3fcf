# usage: ab_tunes.sh "<tune strings>" "<configs>"   (bench.py --tune, one run per pair; prints Mrays/s and kernel times)
for c in $2; do for t in $1; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-extras --tune $t > gpurun_out/abt_c${c}_$t.json 2>/dev/null || exit 1
done; done
python - "$1" "$2" <<PY
import json,sys
for c in sys.argv[2].split():
    for l in sys.argv[1].split():
        d=json.loads(open("gpurun_out/abt_c%s_%s.json"%(c,l)).read().strip().splitlines()[-1])
        print("config",c,l, round(d["value"]), "closest %.4f shadow %.4f"%(d["kernels"]["closest_ms"], d["kernels"]["shadow_ms"]))
PY
