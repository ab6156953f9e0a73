"""Timeline analysis of a rocprofv3 kernel trace: last step only — per-queue busy time, gaps on the busiest queue,
biggest gaps with the kernels around them."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n)
    return n[:60]
# find step boundaries by the fused adam kernel (one group per step)
adam = [i for i, r in enumerate(rows) if 'FusedOptimizerTensorListMetadata' in r['Kernel_Name']]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0]['s'], step[-1]['e']
print('last step: %d dispatches, %.3f ms wall' % (len(step), (t1 - t0) / 1e6))
byq = collections.defaultdict(list)
for r in step:
    byq[(r['Queue_Id'], r['Stream_Id'])].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r['e'] - r['s'] for r in rs)
    print('queue %s: %d kernels, busy %.3f ms' % (q, len(rs), busy / 1e6))
# union busy over all queues
ev = sorted((r['s'], r['e']) for r in step)
cur_s, cur_e = ev[0]; union = 0
gaps = []
prev_name = None
for r in step:
    pass
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print('GPU busy (any queue) %.3f ms, idle %.3f ms in %d gaps' % (union / 1e6, (t1 - t0 - union) / 1e6, len(gaps)))
hist = collections.Counter()
for g, _ in gaps:
    b = 0 if g < 1000 else 1 if g < 2000 else 2 if g < 4000 else 3 if g < 8000 else 4 if g < 20000 else 5
    hist[b] += g
names = ['<1us', '1-2us', '2-4us', '4-8us', '8-20us', '>20us']
cnt = collections.Counter()
for g, _ in gaps:
    b = 0 if g < 1000 else 1 if g < 2000 else 2 if g < 4000 else 3 if g < 8000 else 4 if g < 20000 else 5
    cnt[b] += 1
for b in range(6):
    print('  gaps %-7s n=%4d total %.3f ms' % (names[b], cnt[b], hist[b] / 1e6))
print('largest gaps (us): before-kernel -> after-kernel')
byend = sorted(step, key=lambda r: r['e'])
for g, at in sorted(gaps, reverse=True)[:25]:
    before = max((r for r in step if r['e'] <= at), key=lambda r: r['e'])
    after = min((r for r in step if r['s'] >= at + g), key=lambda r: r['s'])
    print('  %7.1f  %-45s -> %s' % (g / 1e3, short(before['Kernel_Name'])[:45], short(after['Kernel_Name'])[:45]))

# --- gaps between consecutive kernels of the main queue, by (prev family -> next family)
main_q = max(byq.items(), key=lambda kv: len(kv[1]))[0]
mq = sorted(byq[main_q], key=lambda r: r['s'])
fam = lambda n: re.sub(r'<.*', '', short(n))[:28]
pair = collections.defaultdict(list)
for a, b in zip(mq[:-1], mq[1:]):
    pair[(fam(a['Kernel_Name']), fam(b['Kernel_Name']))].append((b['s'] - a['e']) / 1e3)
print('main-queue start-after-end gaps (us): pair n median min')
import statistics
for k, v in sorted(pair.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print('  %-28s -> %-28s n=%3d med %6.1f min %6.1f sum %7.1f' % (k[0], k[1], len(v), statistics.median(v), min(v), sum(v)))
allg = [ (b['s'] - a['e']) / 1e3 for a, b in zip(mq[:-1], mq[1:])]
print('main queue: %d kernels, busy %.3f ms, gaps total %.3f ms, median gap %.2f us' % (len(mq), sum(r['e'] - r['s'] for r in mq) / 1e6, sum(allg) / 1e3, statistics.median(allg)))
