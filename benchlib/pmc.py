"""rocprofv3 --pmc passes of bench.py.  Counters are collected in their own runs with the kernel trace only (gpurun refuses
--pmc together with other trace domains), one counter group per pass (MI355X_MICROARCH.md, PMC slots).
  * measure_traffic: FETCH_SIZE / WRITE_SIZE of the dominant convolution kernel on an isolated probe -> roofline.traffic
  * instep_counters: MFMA busy, texture-addresser busy, GPU active cycles and FETCH / WRITE per kernel INSIDE the replayed
    training step (scripts/pmc_step.py), for the kernels that carry the step."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def traffic_probe(args):
    """Internal (bench.py --traffic-probe, run under rocprofv3 --pmc): the dominant conv shape on this batch's input level."""
    from sgnn_amd import synth
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    dev = torch.device('cuda', 0)
    cin, cout = (int(v) for v in os.environ.get('SGNN_PROBE_SHAPE', '16,16').split(','))
    data = synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, occupancy=args.occupancy)
    g = Grid(coords_from_locs(data['input'][0], dev))
    tab = g.subm_table()
    x = torch.randn(g.n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.1
    for _ in range(6):
        F_.conv_fwd_raw(x, cin, w, 27, tab, g.ld, g.n, cout, 0, 0)
    torch.cuda.synchronize()
    print(json.dumps({'rows': g.n, 'rules': int((tab.view(27, g.ld)[:, :g.n] >= 0).sum().item())}))


def _run_pmc(counters, cmd, timeout):
    """One rocprofv3 pass; returns ({kernel name: {counter: [values per dispatch]}}, stdout) or (None, '')."""
    d = tempfile.mkdtemp(prefix='sgnn_pmc_', dir='/tmp')
    env = dict(os.environ)
    env.update({'TMPDIR': '/tmp', 'SGNN_NO_BIND': '1'})
    try:
        full = ['rocprofv3', '--pmc'] + list(counters) + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--'] + cmd
        out = subprocess.run(full, capture_output=True, text=True, timeout=timeout, env=env, cwd='/tmp')
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r'\(.*$', '', r.get('Kernel_Name', '?').replace('void ', ''))
                agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
        return (agg if agg else None), out.stdout
    except (subprocess.SubprocessError, OSError, ValueError, KeyError):
        return None, ''
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_traffic(args, dom_key, bench_path):
    """HBM bytes per launch of the dominant conv class from PMC counters: two passes (FETCH_SIZE / WRITE_SIZE, KiB) over the
    probe; gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide reads (MI355X_MICROARCH.md, HBM), so traffic =
    2 * FETCH + WRITE.  Returns None when rocprofv3 is not usable."""
    if shutil.which('rocprofv3') is None or dom_key is None:
        return None
    cin, cout = dom_key[1], dom_key[2]
    os.environ['SGNN_PROBE_SHAPE'] = '%d,%d' % (cin, cout)
    cmd = [sys.executable, bench_path, '--traffic-probe', '--batch', str(args.batch), '--dim', str(args.dim), '--occupancy',
           str(args.occupancy)]
    vals, rows, rules = {}, None, None
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        agg, stdout = _run_pmc([counter], cmd, 240)
        if agg is None:
            return None
        for line in stdout.splitlines():
            if line.startswith('{'):
                info = json.loads(line)
                rows, rules = info['rows'], info['rules']
        got = [v for name, cs in agg.items() if 'k_conv_fwd' in name for v in cs.get(counter, [])]
        if not got:
            return None
        vals[counter] = sum(got) / len(got)
    if rows is None:
        return None
    byts = (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0
    alg = 4.0 * rows * (cin + cout) + 4.0 * 27 * rows + 4.0 * 27 * cin * cout
    return {'bytes_per_launch': round(byts), 'algorithmic_bytes': round(alg), 'ratio': round(byts / alg, 3),
            'kernel': 'conv_fwd<%d,%d>K27' % (cin, cout), 'rows': rows, 'rules': rules,
            'FETCH_SIZE_KiB': round(vals['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vals['WRITE_SIZE'], 1),
            'method': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, kernel trace only), mean per launch of the '
                      'kernel on the batch\'s input level; traffic = 2*FETCH + WRITE (gfx950 FETCH_SIZE correction)'}


PASSES = (('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'GRBM_TA_BUSY'), ('FETCH_SIZE',), ('WRITE_SIZE',))
N_SIMD = 1024      # 256 CUs x 4 SIMDs
N_XCD = 8


def instep_counters(args, kernels=('k_conv_fwd_w<16, 16', 'k_conv_fwd<16, 16, 4', 'k_conv_small<16, 16', 'k_conv_dw<16, 16, false, 0', 'k_bn_apply',
                                   'k_bn_bwd_apply'), settle=60, replays=8):
    """Three passes over scripts/pmc_step.py.  Per kernel (name prefix): launches seen, mean per dispatch of each counter,
    and the derived figures
        mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)    (GUI_ACTIVE sums the 8 XCDs)
        ta_busy    = GRBM_TA_BUSY / GRBM_GUI_ACTIVE
        hbm_MB     = (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch (gfx950 FETCH_SIZE correction)
    over ALL dispatches of that kernel in the run (every level size the step launches it on)."""
    if shutil.which('rocprofv3') is None:
        return None
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'pmc_step.py'), '--batch', str(args.batch), '--dim', str(args.dim),
           '--occupancy', str(args.occupancy), '--settle', str(settle), '--replays', str(replays)]
    merged = collections.defaultdict(dict)
    info = None
    for counters in PASSES:
        agg, stdout = _run_pmc(counters, cmd, 420)
        if agg is None:
            return None
        for line in stdout.splitlines():
            if line.startswith('{'):
                info = json.loads(line)
        for name, cs in agg.items():
            for c, v in cs.items():
                merged[name][c] = (sum(v) / len(v), len(v))
    out = []
    for prefix in kernels:
        names = [n for n in merged if n.startswith(prefix)]
        if not names:
            continue
        name = max(names, key=lambda n: merged[n].get('GRBM_GUI_ACTIVE', (0, 0))[1])
        cs = merged[name]
        mean = lambda c: cs[c][0] if c in cs else None
        gui, mf, ta, fe, wr = (mean(c) for c in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_TA_BUSY', 'FETCH_SIZE',
                                                 'WRITE_SIZE'))
        rec = {'kernel': name, 'dispatches': cs.get('GRBM_GUI_ACTIVE', (0, 0))[1],
               'counters_mean_per_dispatch': dict((c, round(v[0], 1)) for c, v in sorted(cs.items()))}
        if gui and mf is not None:
            rec['mfma_util'] = round(mf / (N_SIMD * gui / N_XCD), 4)
        if gui and ta is not None:
            rec['ta_busy'] = round(ta / gui, 4)
        if fe is not None and wr is not None:
            rec['hbm_MB_per_dispatch'] = round((2.0 * fe + wr) * 1024.0 / 1e6, 3)
        out.append(rec)
    return {'kernels': out, 'workload': info,
            'method': 'rocprofv3 --pmc, three passes (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE GRBM_TA_BUSY | '
                      'FETCH_SIZE | WRITE_SIZE), kernel trace only, over scripts/pmc_step.py: %d untimed + 12 re-capture + %d '
                      'replayed steps of the headline workload; means over ALL dispatches of a kernel name in the run'
                      % (settle, replays)}
