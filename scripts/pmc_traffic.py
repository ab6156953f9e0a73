"""Mean HBM traffic per launch of the conv kernels over a whole bench.py run, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; KiB; gfx950: FETCH_SIZE counts wide streaming reads at half -> traffic = 2*FETCH + WRITE).
usage: pmc_traffic.py <counter_collection csvs...>"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*$', '', r.get('Kernel_Name', '?').replace('void ', ''))[:48]
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
print('%-44s %8s %14s %14s %16s' % ('kernel', 'launches', 'FETCH KiB/l', 'WRITE KiB/l', 'traffic MB/launch'))
for name, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('FETCH_SIZE', [0]))):
    if not name.startswith('k_conv') or 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs:
        continue
    fe, wr = cs['FETCH_SIZE'], cs['WRITE_SIZE']
    f_, w_ = sum(fe) / len(fe), sum(wr) / len(wr)
    print('%-44s %8d %14.1f %14.1f %16.3f' % (name, len(fe), f_, w_, (2 * f_ + w_) * 1024 / 1e6))
