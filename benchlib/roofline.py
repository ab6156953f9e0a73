"""Roofline accounting of bench.py: HIP-event timings of every convolution launch (sgnn_prof_*), the rule counts of the run's
own rulebooks, SURVEY.md section 8d's algorithmic bytes / flops per operator."""
import ctypes

import numpy as np
import torch

from . import HBM_PEAK_GBS, FP32_MFMA_PEAK_TF


def conv_alg_bytes(kind, n_out, cin, cout, K):
    """Algorithmic HBM bytes of one conv launch (HISTORY.md section 4): feature slab read once, output written once, the
    K x n_out int32 rule table, the weights.  The weight gradient reads x and dy and writes K*cin*cout: the same count."""
    b = 4 * n_out * (cin + cout) + 4 * K * n_out + 4 * K * cin * cout
    if kind == 2:      # fused backward (csrc/conv_bwd_fused.hip): dy read, dx written, + the layer's input rows read, dW written
        b += 4 * n_out * cout + 4 * K * cin * cout
    return b


def collect_prof(lib, valid_ratio=None, row_map=None):
    """Aggregate the recorded launches by (kind, cin, cout, K).
    valid_ratio: {(launch rows, K): rules / (K x LIVE rows)} measured on the run's own rulebooks (measure_valid_ratios);
    `flops` counts the rules only (F = 2 R Cin Cout), `flops_exec` every table entry of the live rows.
    row_map (capacity mode): launch rows (= capacities) -> live rows; bytes and flops count the live rows."""
    valid_ratio = valid_ratio or {}
    row_map = row_map or {}
    n = lib.sgnn_prof_count()
    kind, cin, cout, K, flags = (ctypes.c_int() for _ in range(5))
    n_out = ctypes.c_int64()
    ms = ctypes.c_float()
    agg = {}
    for i in range(n):
        rc = lib.sgnn_prof_get(i, ctypes.byref(kind), ctypes.byref(n_out), ctypes.byref(cin), ctypes.byref(cout),
                               ctypes.byref(K), ctypes.byref(flags), ctypes.byref(ms))
        if rc != 0:
            continue
        key = (kind.value, cin.value, cout.value, K.value)
        a = agg.setdefault(key, {'launches': 0, 'ms': 0.0, 'bytes': 0.0, 'flops': 0.0, 'flops_exec': 0.0, 'by_size': {}})
        live = row_map.get(n_out.value, n_out.value)
        # rules of the launch = ratio x K x live rows.  A stride-2 table (K = 8) holds exactly one rule per fine row and its
        # ratio is not measured: left at 1, as before (an upper bound of the coarse-row count it is quoted on).
        ratio = valid_ratio.get((n_out.value, K.value), 1.0)
        assert 0.0 <= ratio <= 1.0 + 1e-9, 'more rules than table entries: %r' % ((n_out.value, K.value, ratio),)
        fl_exec = 2.0 * live * K.value * cin.value * cout.value * (2 if kind.value == 2 else 1)   # kind 2: both products
        fl = fl_exec * ratio
        a['launches'] += 1
        a['ms'] += ms.value
        a['bytes'] += conv_alg_bytes(kind.value, live, cin.value, cout.value, K.value)
        a['flops'] += fl
        a['flops_exec'] += fl_exec
        # the same kernel serves levels of very different size: keep the launches apart by output rows (powers of 4)
        bucket = 0 if live <= 0 else int(np.floor(np.log(max(live, 1)) / np.log(4.0)))
        b = a['by_size'].setdefault(bucket, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'rows': 0, 'rules': 0.0,
                                             'n_fwd': 0, 'ms_fwd': 0.0, 'n_dx': 0, 'ms_dx': 0.0})
        b['launches'] += 1
        b['ms'] += ms.value
        # forward launches and data-gradient launches (SGNN_CONV_TRANSPOSE_W) of the same kernel, kept apart: the fused
        # backward epilogue once made the latter 25-45 % slower — this keeps the ratio visible
        d = 'dx' if (kind.value in (0, 2) and (flags.value & 1)) else 'fwd'
        b['n_' + d] += 1
        b['ms_' + d] += ms.value
        b['flops'] += fl
        b['rows'] += live
        b['rules'] += ratio * K.value * live
    return agg


def capacity_row_map(cap, live):
    """launch rows (capacities) -> live rows of every level a capacity-mode step touches."""
    m = {cap['input']: live['input']}
    for c, n in zip(cap['enc'], live['enc']):
        m.setdefault(c, n)
    for (k, pyr), (nk, npyr) in zip(cap['gen'], live['gen']):
        m.setdefault(k, nk)
        m.setdefault(8 * k, 8 * nk)
        for c, n in zip(pyr, npyr):
            m.setdefault(c, n)
    return m


def measure_valid_ratios(step, i):
    """One extra (untimed) step with a hook on the rulebook builder: rules / (27 x LIVE rows) per table, keyed by the rows
    the launches are recorded with (the capacity in capacity mode) and K.  Only the live prefix of a table is counted: a
    capacity-sized table is written up to roundup256(live rows) and holds uninitialised words beyond (counting
    those gave 28.8 "rules" per row of a 27-offset rulebook)."""
    from sgnn_amd.scn import metadata as MD
    ratios, real = {}, MD.Grid.subm_table

    def hooked(self):
        fresh = self._nbr is None
        tab = real(self)
        if fresh and self.n:
            live = int(self.cnt.item()) if self.cnt is not None else self.n
            if live > 0:
                rules = float((tab.view(27, self.ld)[:, :live] >= 0).sum().item())
                assert rules <= 27.0 * live
                ratios[(self.n, 27)] = rules / (27.0 * live)
        return tab
    MD.Grid.subm_table = hooked
    try:
        step(i)
        torch.cuda.synchronize()
    finally:
        MD.Grid.subm_table = real
    return ratios


def algorithmic_step(model, agg, n_prof_steps, row_map):
    """sum of the ALGORITHMIC bytes and flops of one training step over all sparse operators with the
    run's own row counts N_l and rule counts R_l.  Convolutions (forward, data gradient, weight gradient) come from the
    profiled launch records; BatchNormReLU 12 N C forward + 20 N C backward; UnPooling / AddTable / JoinTable / skip-join /
    linear heads as row movement, forward + backward.  The dense 8^3 bottleneck is excluded."""
    from sgnn_amd.scn import program as P_
    conv_b = sum(a['bytes'] for a in agg.values()) / max(n_prof_steps, 1)
    conv_f = sum(a['flops'] for a in agg.values()) / max(n_prof_steps, 1)
    other_b, n_ops = 0.0, 0
    for prog in P_.programs_of(model):
        lev = getattr(prog, 'last_lev_n', None)
        if lev is None:
            continue
        rows = lambda b: float(row_map.get(int(lev[prog.bufs[b][0]]), int(lev[prog.bufs[b][0]])))
        ch = lambda b: prog.bufs[b][1] if b >= 0 else 0
        for o in prog.ops:
            t, in0, in1, out = o[0], o[1], o[2], o[3]
            n_ops += 1
            if t == P_.OP_BN:
                other_b += 32.0 * rows(out) * ch(out)
            elif t in (P_.OP_UNPOOL, P_.OP_ADD, P_.OP_JOIN):
                srcs = [b for b in (in0, in1) if b >= 0]
                other_b += 2 * (4.0 * sum(rows(b) * ch(b) for b in srcs) + 4.0 * rows(out) * ch(out) + 8.0 * rows(out))
            elif t == P_.OP_CONCAT_IN:
                other_b += 2 * (8.0 * rows(out) * ch(out) + 8.0 * rows(out))
            elif t == P_.OP_LINEAR:
                other_b += 2 * 4.0 * rows(out) * (ch(in0) + ch(out))
    return {'bytes': conv_b + other_b, 'flops': conv_f, 'conv_bytes': conv_b, 'other_bytes': other_b, 'sparse_ops': n_ops}


def kernel_name(key):
    return '%s<%d,%d>K%d' % ({0: 'conv_fwd', 1: 'conv_dw', 2: 'conv_bwd_fused'}[key[0]], key[1], key[2], key[3])


def class_record(key, a):
    """One JSON entry per convolution class: time, rates, and the fraction of the roof that binds it."""
    ms = a['ms']
    gbs = a['bytes'] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    tfs = a['flops'] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    t_hbm, t_mfma = a['bytes'] / (HBM_PEAK_GBS * 1e9), a['flops'] / (FP32_MFMA_PEAK_TF * 1e12)
    bound = 'mfma' if t_mfma >= t_hbm else 'hbm'
    frac = tfs / FP32_MFMA_PEAK_TF if bound == 'mfma' else gbs / HBM_PEAK_GBS
    return {'kernel': kernel_name(key), 'launches': a['launches'], 'ms_total': round(ms, 3),
            'avg_launch_us': round(1e3 * ms / max(a['launches'], 1), 2), 'GBps': round(gbs, 1), 'TFLOPs': round(tfs, 2),
            'bound': bound, 'frac': round(frac, 4)}


def roofline_record(agg, n_prof_steps, timing_note):
    """The `roofline` object of the JSON line for the convolution class with the largest total time: `achieved` =
    algorithmic flops (rules only) or bytes of its launches / their summed HIP-event durations."""
    if not agg:
        return None, None
    ranked = sorted(agg.items(), key=lambda kv: -kv[1]['ms'])
    dom_key, dom = ranked[0]
    if dom['ms'] <= 0:
        return None, None
    top = class_record(dom_key, dom)
    unit = 'TFLOP/s' if top['bound'] == 'mfma' else 'GB/s'
    roof = {'bound': top['bound'], 'achieved': top['TFLOPs'] if top['bound'] == 'mfma' else top['GBps'],
            'peak': FP32_MFMA_PEAK_TF if top['bound'] == 'mfma' else HBM_PEAK_GBS, 'unit': unit, 'frac': top['frac'],
            'traffic': None, 'kernel': top['kernel'], 'avg_launch_us': top['avg_launch_us'], 'launches': dom['launches'],
            'alg_GBps': top['GBps'], 'alg_frac_of_hbm_peak': round(top['GBps'] / HBM_PEAK_GBS, 4),
            'TFLOPs': top['TFLOPs'], 'frac_of_fp32_mfma_peak': round(top['TFLOPs'] / FP32_MFMA_PEAK_TF, 4),
            'flops_counted': 'rules only (2 R Cin Cout, R from the run\'s own rulebooks over the live rows); executed incl. '
                             'empty table entries: %.2f TFLOP/s' % (dom['flops_exec'] / (dom['ms'] * 1e-3) / 1e12),
            'conv_ms_per_step': round(sum(a['ms'] for a in agg.values()) / max(n_prof_steps, 1), 3),
            'profiled_steps': n_prof_steps, 'timing': timing_note,
            # `frac` is over ALL launches of the class; split by level size it is throughput-bound only on the big levels
            'by_level_size': [
                {'mean_rows': int(b['rows'] / b['launches']), 'launches': b['launches'],
                 'rules_per_row': round(b['rules'] / max(b['rows'], 1), 2),
                 'avg_us': round(1e3 * b['ms'] / b['launches'], 1), 'share_of_kernel_time': round(b['ms'] / dom['ms'], 3),
                 'TFLOPs': round(b['flops'] / (b['ms'] * 1e-3) / 1e12, 2),
                 'frac_of_fp32_mfma_peak': round(b['flops'] / (b['ms'] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4),
                 'dx_vs_fwd_us': [round(1e3 * b['ms_dx'] / b['n_dx'], 1) if b['n_dx'] else None,
                                  round(1e3 * b['ms_fwd'] / b['n_fwd'], 1) if b['n_fwd'] else None]}
                for _, b in sorted(dom['by_size'].items(), reverse=True) if b['ms'] > 0],
            # the five classes with the largest total time, each against the roof that binds it
            'top_kernels': [class_record(k, a) for k, a in ranked[:5] if a['ms'] > 0]}
    for b in roof['by_level_size']:
        assert b['rules_per_row'] <= dom_key[3] + 1e-6, 'more rules per row than offsets: %r' % (b,)
    return dom_key, roof
