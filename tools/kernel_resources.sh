# Registers, scratch and occupancy of the traversal kernels as the compiler reports them (cross-compiles, no GPU needed).
# usage: bash tools/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
for f in rtx_shade.hip rtx_trace.hip; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /dev/null 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        if cur: print(cur)
        cur = {"name": m.group(1)}
    for key in ("VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]", "SGPRs"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m and cur: cur[key.split(" ")[0]] = int(m.group(1))
if cur: print(cur)
' | grep -E "k_shadow|k_trace|k_intersect" | sed -E "s/_ZN12_GLOBAL__N_1[0-9]+//; s/Ev8DevScene[^']*'/'/"
done
