"""Group a rocprofv3 kernel_stats.csv by kernel family: ms/step, launches/step."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
cats = [('conv_fwd', r'k_conv_fwd|k_conv_small'), ('conv_dw', r'k_conv_dw'), ('dw_reduce', r'k_dw_reduce'), ('bn', r'k_bn_'),
        ('rulebook_subm', r'k_rulebook|k_vol_mark|k_tile_index'), ('hash', r'k_hash'), ('down2+scan', r'k_down2|k_scan|k_chain|k_compact|k_flag'),
        ('rows', r'k_concat|k_gather|k_add|k_sum_groups|k_scatter|k_repeat|k_sparse_to|k_dense_to|k_expand|k_coords|k_dense_coords'),
        ('linear', r'k_linear'), ('loss', r'k_loss'), ('copy/fill', r'rocclr'), ('adam', r'multi_tensor'),
        ('gemm', r'Cijk'), ('torch', r'at::native|at_cuda|elementwise|reduce_kernel')]
agg = collections.OrderedDict((c, [0, 0.0]) for c, _ in cats + [('other', '')])
for r in rows:
    for c, pat in cats:
        if re.search(pat, r['Name']):
            break
    else:
        c = 'other'
    agg[c][0] += int(r['Calls'])
    agg[c][1] += int(r['TotalDurationNs'])
tot_c = sum(v[0] for v in agg.values()); tot_t = sum(v[1] for v in agg.values())
print('%-14s %10s %10s %8s' % ('family', 'launches', 'ms/step', 'avg_us'))
for c, (n, t) in agg.items():
    if n:
        print('%-14s %10.1f %10.3f %8.1f' % (c, n / steps, t / 1e6 / steps, t / 1e3 / n))
print('%-14s %10.1f %10.3f' % ('total', tot_c / steps, tot_t / 1e6 / steps))
