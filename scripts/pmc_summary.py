"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter per dispatch."""
import csv, sys, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*$', '', r.get('Kernel_Name', r.get('Name', '?')).replace('void ', ''))[:40]
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in agg.items():
    if not name.startswith('k_conv'):
        continue
    print(name, ' '.join('%s=%.4g(n%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
