"""Comparison legs of bench.py: the same workload through the other execution modes, in the same process and from the
headline leg's weights (same masks).  Every leg uses bench.py's `timed` (barrier + synchronize brackets, max over ranks)."""
import torch


def sites(outs):
    return [int(o[0].shape[0]) if len(o[0]) else 0 for o in outs[1]] + [int(outs[0][0].shape[0]) if len(outs[0][0]) else 0]


def live_sites(args, live):
    return [args.batch * (args.dim // 8) ** 3] + [8 * k for k, _ in live['gen'][:-1]] + [live['gen'][-1][0]]


def comparison_legs(ctx):
    """ctx: dict with args, world, dev, lw, batches, make_batches, fresh_model, state, timed, flat_sync, dist_on, teacher."""
    from sgnn_amd.model import GenModel
    from sgnn_amd.train import train_step, FlatGradAllReduce, make_optimizer, GeometryPrefetcher, GraphStep
    args, world, dev, lw, batches = ctx['args'], ctx['world'], ctx['dev'], ctx['lw'], ctx['batches']
    timed, flat_sync, dist_on, teacher = ctx['timed'], ctx['flat_sync'], ctx['dist_on'], ctx['teacher']
    k2 = max(10, args.steps // 3)
    legs = {}

    def make_model():       # every comparison leg starts from the headline leg's weights: the same masks
        m = GenModel(8, (args.dim,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)
        m.load_state_dict(ctx['state'])
        return m

    def rec(el, **kw):
        r = {'steps': k2, 'value': round(args.batch * world * k2 / el, 2), 'ms_per_step': round(1e3 * el / k2, 3)}
        r.update(kw)
        return r

    sync = lambda m: FlatGradAllReduce(m.parameters()) if dist_on else None
    gsync = flat_sync if dist_on else None
    # (a) the classic eager path with the reference's masks: five host read-backs per step, ~680 launches issued from
    #     Python (what BENCH_r01 / BENCH_r02's `other_mask_mode` measured)
    m2 = make_model()
    o2, s2, box = make_optimizer(m2.parameters(), lr=1e-3), sync(m2), {}

    def step2(i):
        box['o'] = train_step(m2, o2, batches[i % 2], lw, grad_sync=s2, teacher_forced=False)[2]
    el = timed(step2, 8, k2)
    legs['classic_eager_free_running'] = rec(el, generated_sites_per_level=sites(box['o']))
    del m2, o2
    # (b) BENCH_r02's headline: teacher-forced masks + geometry built one batch ahead on a second stream
    m3 = make_model()
    o3, s3, p3 = make_optimizer(m3.parameters(), lr=1e-3), sync(m3), GeometryPrefetcher(m3)
    el = timed(lambda i: train_step(m3, o3, batches[i % 2], lw, grad_sync=s3, teacher_forced=True, prefetch=p3,
                                    next_batch=batches[(i + 1) % 2]), 8, k2)
    legs['classic_eager_teacher_forced_prefetch'] = rec(el)
    del m3, o3, p3
    # (c) graph replay with the other mask mode (teacher-forced row counts do not depend on the weights)
    m4 = make_model()
    g4 = GraphStep(m4, lr=1e-3, teacher_forced=not teacher, headroom=args.headroom, settle=teacher, grad_sync=gsync,
                   world_size=world)
    el = timed(lambda i: g4(batches[i % 2], lw), 8, k2)
    legs['graph_teacher_forced' if not teacher else 'graph_free_running'] = rec(
        el, generated_sites_per_level=live_sites(args, g4.capacity.read()), stats=dict(g4.stats))
    del m4, g4
    # (e) a point on the transient, for comparison with earlier rounds' free-running numbers: fresh weights, 40 + 12
    #     untimed steps, then k2 timed ones (final level ~260-320 k sites; BENCH_r02: 212 k)
    if not teacher:
        m6 = ctx['fresh_model']()
        g6 = GraphStep(m6, lr=1e-3, headroom=max(args.headroom, 1.6), grad_sync=gsync, world_size=world)
        for i in range(40):
            g6(batches[i % 2], lw)
        g6.replan()
        el = timed(lambda i: g6(batches[i % 2], lw), 12, k2)
        legs['graph_free_running_early_in_training'] = rec(
            el, generated_sites_per_level=live_sites(args, g6.capacity.read()), stats=dict(g6.stats))
        del m6, g6
    # (d) the fixed cost of a step: the same graph-replayed step on ONE block per GPU
    if args.batch > 1:
        b1 = ctx['make_batches'](1)
        m5 = make_model()
        g5 = GraphStep(m5, lr=1e-3, teacher_forced=teacher, headroom=max(args.headroom, 1.6), settle=False, grad_sync=gsync,
                       world_size=world)
        el = timed(lambda i: g5(b1[i % 2], lw), 12, k2)
        legs['batch1'] = {'steps': k2, 'ms_per_step': round(1e3 * el / k2, 3), 'stats': dict(g5.stats)}
        del m5, g5, b1
    torch.cuda.synchronize()
    return legs
