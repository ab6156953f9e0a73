"""Which training-queue kernels wait for the side lanes?  Reads a per-step launch table (scripts/trace_graph.py ->
profiles/rNN_step_launches.csv: start_us, dur_us, queue, grid, wg, kernel of ONE replayed step under rocprofv3 --kernel-trace)
and lists every training-queue kernel whose start follows an idle gap of the training queue, with what the other queue was
doing in that gap: a lane kernel that ENDS inside the gap's last microseconds means the training queue sat waiting for the lane
(a join); no lane activity means a launch/dependency bubble of the queue itself.

    python scripts/lane_waits.py profiles/r06z_step_launches.csv [min_gap_us]
"""
import csv
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = [(float(r['start_us']), float(r['dur_us']), r['queue'], r['kernel']) for r in csv.DictReader(open(path))]
queues = sorted({r[2] for r in rows}, key=lambda q: -sum(1 for r in rows if r[2] == q))
main = queues[0]
tq = [r for r in rows if r[2] == main]
lane = [r for r in rows if r[2] != main]
print('# %s: training queue %s (%d kernels), lanes %s (%d kernels); gaps >= %.1f us' % (path, main, len(tq), queues[1:], len(lane), min_gap))
print('# kind: join = a lane kernel ended in the last 6 us of the gap (the queue waited for it); '
      'lane-busy = the lane ran through the gap (contention or an unrelated wait); idle = nothing ran')
print('%9s %7s  %-9s %-34s %-34s %s' % ('start_us', 'gap_us', 'kind', 'kernel that waited', 'after', 'lane kernel ending in the gap (end - start_us)'))
tot = {'join': 0.0, 'lane-busy': 0.0, 'idle': 0.0}
cnt = {'join': 0, 'lane-busy': 0, 'idle': 0}
small = 0.0
for prev, cur in zip(tq, tq[1:]):
    gap0, gap1 = prev[0] + prev[1], cur[0]
    gap = gap1 - gap0
    if gap < min_gap:
        small += max(gap, 0.0)
        continue
    ending = [l for l in lane if gap0 < l[0] + l[1] <= gap1 + 0.5]
    running = [l for l in lane if l[0] < gap1 and l[0] + l[1] > gap0]
    last = max(ending, key=lambda l: l[0] + l[1]) if ending else None
    if last is not None and gap1 - (last[0] + last[1]) <= 6.0:
        kind = 'join'
    elif running:
        kind = 'lane-busy'
    else:
        kind = 'idle'
    tot[kind] += gap
    cnt[kind] += 1
    print('%9.1f %7.1f  %-9s %-34s %-34s %s' % (cur[0], gap, kind, cur[3][:34], prev[3][:34],
                                               '%s (%.1f)' % (last[3][:40], last[0] + last[1] - gap1) if last else '-'))
print('# totals: ' + ', '.join('%s %d gaps %.3f ms' % (k, cnt[k], tot[k] / 1e3) for k in tot)
      + '; gaps below the threshold %.3f ms' % (small / 1e3))
