"""Timeline of ONE replayed training step from a rocprofv3 kernel trace (steps end with k_adam_steps): wall time,
per-queue busy time, union busy / idle, gap histogram, time by kernel, and the largest gaps with their neighbours."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:70]


ends = [i for i, r in enumerate(rows) if 'k_adam_steps' in r['Kernel_Name']]
if len(ends) < 3:
    print('fewer than 3 steps in the trace')
    sys.exit(0)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
lo, hi = ends[which - 1] + 1, ends[which] + 1
step = rows[lo:hi]
t0, t1 = step[0]['s'], max(r['e'] for r in step)
print('step: %d dispatches, %.3f ms wall (first start -> last end); step-to-step period %.3f ms' % (
    len(step), (t1 - t0) / 1e6, (rows[ends[which]]['e'] - rows[ends[which - 1]]['e']) / 1e6))
byq = collections.defaultdict(list)
for r in step:
    byq[(r.get('Queue_Id'), r.get('Stream_Id'))].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print('queue %s: %d kernels, busy %.3f ms' % (q, len(rs), sum(r['e'] - r['s'] for r in rs) / 1e6))
ev = sorted((r['s'], r['e']) for r in step)
cur_s, cur_e = ev[0]
union, gaps = 0, []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print('GPU busy (any queue) %.3f ms, idle %.3f ms in %d gaps; sum of kernel durations %.3f ms' % (
    union / 1e6, (t1 - t0 - union) / 1e6, len(gaps), sum(r['e'] - r['s'] for r in step) / 1e6))
names = ['<1us', '1-2us', '2-4us', '4-8us', '8-20us', '>20us']
cnt, tot = collections.Counter(), collections.Counter()
for g, _ in gaps:
    b = 0 if g < 1000 else 1 if g < 2000 else 2 if g < 4000 else 3 if g < 8000 else 4 if g < 20000 else 5
    cnt[b] += 1
    tot[b] += g
for b in range(6):
    print('  gaps %-7s n=%4d total %.3f ms' % (names[b], cnt[b], tot[b] / 1e6))
fam = collections.defaultdict(lambda: [0, 0])
for r in step:
    f = fam[short(r['Kernel_Name'])]
    f[0] += 1
    f[1] += r['e'] - r['s']
print('%-72s %6s %9s %8s' % ('kernel', 'calls', 'us/step', 'avg_us'))
for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%-72s %6d %9.1f %8.1f' % (k, n, t / 1e3, t / 1e3 / n))
print('largest gaps (us): before-kernel -> after-kernel')
for g, at in sorted(gaps, reverse=True)[:15]:
    before = max((r for r in step if r['e'] <= at), key=lambda r: r['e'])
    after = min((r for r in step if r['s'] >= at + g), key=lambda r: r['s'])
    print('  %7.1f  %-45s -> %s' % (g / 1e3, short(before['Kernel_Name'])[:45], short(after['Kernel_Name'])[:45]))

# per-launch table of the analysed step (kernel, grid, duration) for offline inspection
if len(sys.argv) > 3:
    with open(sys.argv[3], 'w') as f:
        f.write('start_us,dur_us,queue,grid,wg,kernel\n')
        for r in step:
            f.write('%.1f,%.2f,%s,%s,%s,%s\n' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r.get('Queue_Id'),
                                               r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')),
                                               short(r['Kernel_Name']).replace(',', ';')))
