"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel, calls, total ms, avg us, %)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('# total kernel time %.2f ms over %g steps = %.2f ms/step' % (tot / 1e6, steps, tot / 1e6 / steps))
print('%-64s %7s %10s %10s %7s' % ('kernel', 'calls', 'ms/step', 'avg_us', '%'))
def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n) if n.startswith('k_') else n
    return n[:64]
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%-64s %7s %10.3f %10.1f %7.2f' % (short(r['Name']), r['Calls'], int(r['TotalDurationNs']) / 1e6 / steps,
                                            float(r['AverageNs']) / 1e3, float(r['Percentage'])))
