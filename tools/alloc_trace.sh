# usage (on the GPU box): bash tools/alloc_trace.sh   -> gpurun_out/alloc_trace.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 5 25; do
  timeout -k 10 200 rocprofv3 --hip-trace --output-format csv -d $R/gpurun_out/alloc_$n -- python3 $R/tools/alloc_trace.py $n > $R/gpurun_out/alloc_$n.log 2>&1 || exit 1
done
python3 - <<PY > $R/gpurun_out/alloc_trace.txt
import csv, glob, collections
for n in (5, 25):
    f = glob.glob("$R/gpurun_out/alloc_%d/**/*hip_api_trace.csv" % n, recursive=True)[0]
    c = collections.Counter()
    for row in csv.DictReader(open(f)):
        name = row.get("Function") or row.get("Name") or ""
        if any(k in name for k in ("Malloc", "Free", "hipHostAlloc", "hipHostRegister", "MemPool", "hipArrayCreate")):
            c[name] += 1
    print("steady-state rounds:", n, dict(sorted(c.items())))
PY
cat $R/gpurun_out/alloc_trace.txt
