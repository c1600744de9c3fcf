#!/usr/bin/env python3
"""Summarises gpurun_out/pmc_<tag>_*/ counter CSVs per traversal kernel (mean per launch)."""
import collections, csv, glob, json, sys
tag = sys.argv[1]
out = collections.defaultdict(dict)
for f in glob.glob('gpurun_out/pmc_%s_*/vm/*_counter_collection.csv' % tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        short = 'shadow' if 'k_shadow' in k else 'closest' if 'k_trace' in k else None
        if short:
            agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
    for short, d in agg.items():
        for c, v in d.items():
            out[short][c] = sum(v) / len(v)
for short, d in out.items():
    print(short, json.dumps({k: round(v) for k, v in sorted(d.items())}))
    g = d.get
    if g('SQ_THREAD_CYCLES_VALU') and g('SQ_ACTIVE_INST_VALU'):
        print('   lane utilisation (thread cycles / (active valu * 64)): %.2f' % (g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 64)))
    if g('TA_TA_BUSY_sum') and g('GRBM_GUI_ACTIVE'):
        print('   TA busy fraction: %.2f' % (g('TA_TA_BUSY_sum') / 256 / (g('GRBM_GUI_ACTIVE') / 8)))
    if g('TCC_HIT_sum') is not None and g('TCC_REQ_sum'):
        print('   L2 hit rate: %.2f' % (g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))))
    if g('FETCH_SIZE') is not None:
        print('   FETCH_SIZE KB %.0f  WRITE_SIZE KB %.0f' % (g('FETCH_SIZE'), g('WRITE_SIZE') or 0))
json.dump(out, open('gpurun_out/pmc_%s_summary.json' % tag, 'w'), indent=1)
