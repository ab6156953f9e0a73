"""Host-side logic that needs no GPU: module layout / state-dict contract, loss restatement, curriculum,
synthetic data layout."""
import os

import pytest

import numpy as np
import torch

import model_oracle as mo
from sgnn_amd import synth, loss as L
from sgnn_amd.train import get_loss_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_genmodel_layout_matches_oracle_and_reference_param_count():
    from sgnn_amd.model import GenModel
    hm = GenModel(8, (64, 64, 64), 1, 16, 16, 4, True, True, 1, 1)
    om = mo.GenModel(8, (64, 64, 64), 1, 16, 16, 4, True, True, 1, 1)
    assert list(hm.state_dict().keys()) == list(om.state_dict().keys())
    for (k, a), (_, b) in zip(hm.state_dict().items(), om.state_dict().items()):
        assert a.shape == b.shape, k
    assert sum(p.numel() for p in hm.parameters()) == 643735
    hm.load_state_dict(om.state_dict(), strict=True)
    # later upstream checkpoints store conv weights as (K, groups=1, nIn, nOut)
    sd = {k: (v.unsqueeze(1) if v.dim() == 3 else v) for k, v in om.state_dict().items()}
    hm.load_state_dict(sd, strict=True)


def test_update_sizes_sets_upper_bounds_per_level():
    from sgnn_amd.model import GenModel
    hm = GenModel(8, (32, 32, 32), 1, 16, 16, 4, True, True, 1, 1)
    hm.update_sizes(np.array([32, 64, 96]), np.array([32, 64, 96]) // 8)      # test_scene.py:78
    assert hm.encoder.process_sparse[0].p0.spatial_size.tolist() == [32, 64, 96]
    assert hm.refinement[0].p0.spatial_size.tolist() == [4, 8, 12]
    assert hm.refinement[2].n0.spatial_size.tolist() == [32, 64, 96]
    assert hm.surfacepred.p0.spatial_size.tolist() == [32, 64, 96]


def test_product_loss_equals_oracle_loss_on_cpu():
    torch.manual_seed(0)
    data = synth.make_batch(2, (16, 16, 16), cfg=5, occupancy=0.1)
    dims = data['sdf'].shape[2:]
    outs = []
    for h, f in enumerate((8, 4, 2, 1)):
        d = [v // f for v in dims]
        n = 200
        locs = torch.stack([torch.randint(0, d[0], (n,)), torch.randint(0, d[1], (n,)), torch.randint(0, d[2], (n,)),
                            torch.randint(0, 2, (n,))], 1)
        outs.append([locs, torch.randn(n, 2, requires_grad=True)])
    sdf_locs = outs[3][0]
    sdf_vals = torch.randn(sdf_locs.shape[0], 1, requires_grad=True)
    lw = np.array([1, 1, 0.5, 1, 2], dtype=np.float32)
    res = []
    for mod in (mo, L):
        t = mod.compute_targets(data['sdf'].clone(), [h.clone() for h in data['hierarchy']], 4, 3, True, data['known'])
        for masking, wgeo in ((True, 5.0), (False, 1.0)):
            loss, losses = mod.compute_loss([sdf_locs, sdf_vals], outs, t[0], t[1], t[2], lw, 3, True, wgeo,
                                            data['input'][0], masking, data['known'])
            g = torch.autograd.grad(loss, [o[1] for o in outs] + [sdf_vals])
            res.append((loss.item(), [float(x) for x in losses], [x.clone() for x in g]))
    for (la, lsa, ga), (lb, lsb, gb) in zip(res[:2], res[2:]):
        assert abs(la - lb) < 1e-5 * max(1.0, abs(la))
        assert np.allclose(lsa, lsb, rtol=1e-5, atol=1e-6)
        for x, y in zip(ga, gb):
            assert (x - y).abs().max().item() < 1e-6


def test_loss_weight_curriculum():
    # train.py:203-231 with num_iters_per_level=2000: level k switches on at iteration 2000*k
    w0 = get_loss_weights(0, 4, 2000, 1.0)
    assert w0.tolist() == [1, 0, 0, 0, 0]
    w = get_loss_weights(1999, 4, 2000, 1.0)
    assert w[0] == 1 and 0 < w[1] <= 1
    assert get_loss_weights(2000, 4, 2000, 1.0).tolist()[:2] == [1, 1]
    assert get_loss_weights(10001, 4, 2000, 1.0).tolist() == [1, 1, 1, 1, 1]
    assert get_loss_weights(8000, 4, 2000, 1.0)[-1] == 1.0


def test_synthetic_batch_layout_and_determinism():
    a = synth.make_batch(3, (32, 32, 32), cfg=7)
    b = synth.make_batch(3, (32, 32, 32), cfg=7)
    locs, feats = a['input']
    assert locs.dtype == torch.int64 and locs.shape[1] == 4 and feats.shape == (locs.shape[0], 1)
    assert torch.equal(locs, b['input'][0]) and torch.equal(feats, b['input'][1])
    assert a['sdf'].shape == (3, 1, 32, 32, 32) and a['known'].dtype == torch.uint8
    assert [h.shape[-1] for h in a['hierarchy']] == [4, 8, 16]
    assert feats.abs().max().item() < 3.0                                   # scene_dataloader.py:102-105
    bcol = locs[:, 3]
    assert torch.all(bcol[1:] >= bcol[:-1])                                 # batch-major like collate()
    key = ((locs[:, 3] * 32 + locs[:, 0]) * 32 + locs[:, 1]) * 32 + locs[:, 2]
    assert torch.unique(key).numel() == key.numel()
    occ = locs.shape[0] / (3 * 32 ** 3)
    assert 0.01 < occ < 0.12


def test_fast_adam_matches_torch_adam():
    """FastAdam = torch.optim.Adam(fused=True) without the per-step Python loop: identical parameters and state,
    also when some parameters receive no gradient on some steps (an empty generative level)."""
    import copy
    from sgnn_amd.train import FastAdam
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3), torch.nn.Linear(3, 3))
    b = copy.deepcopy(a)
    oa = FastAdam(a.parameters(), lr=1e-2, weight_decay=0.01)
    ob = torch.optim.Adam(b.parameters(), lr=1e-2, weight_decay=0.01, fused=True)
    for it in range(8):
        x = torch.randn(4, 5)
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad(set_to_none=True)
            h = net[1](net[0](x))
            (h if it in (2, 3, 6) else net[2](h)).pow(2).sum().backward()
            opt.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
        assert float(oa.state[pa]['step']) == float(ob.state[pb]['step'])
        assert torch.equal(oa.state[pa]['exp_avg_sq'], ob.state[pb]['exp_avg_sq'])
    assert oa.state_dict()['param_groups'][0]['lr'] == 1e-2


def test_numa_cpulist_parsing_and_binding_is_harmless_without_a_gpu():
    from sgnn_amd.train import _parse_cpulist, bind_to_device_numa
    assert _parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    assert _parse_cpulist('') == set()
    import os
    before = os.sched_getaffinity(0)
    assert bind_to_device_numa() is None or torch.cuda.is_available()
    if not torch.cuda.is_available():
        assert os.sched_getaffinity(0) == before


def test_point_cloud_ply_writer(tmp_path):
    """write_points_ply: the vertex-only PLY data_util.visualize_points writes through plyfile (header + float32 xyz)."""
    import numpy as np
    from sgnn_amd.marching_cubes import write_points_ply
    pts = np.array([[0.5, 1.5, 2.5], [3.0, 4.0, 5.0]], dtype=np.float64)
    f = tmp_path / 'p.ply'
    write_points_ply(pts, str(f))
    raw = f.read_bytes()
    head, body = raw.split(b'end_header\n')
    assert head.decode().splitlines() == ['ply', 'format binary_little_endian 1.0', 'element vertex 2', 'property float x',
                                          'property float y', 'property float z']
    assert np.array_equal(np.frombuffer(body, dtype='<f4').reshape(-1, 3), pts.astype(np.float32))


def test_scratch_lanes_are_per_thread_and_levels_remember_their_size():
    """scn.metadata.lane selects the runtime scratch of the geometry-prefetch stream for the CURRENT thread only (a
    worker thread building a plan must not switch the training thread's lane), and a Grid registered in a Metadata
    learns the level's spatial size (what routes its rulebook through the dense index volume)."""
    import threading
    from sgnn_amd.scn import metadata as MD
    assert getattr(MD._lanes, 'n', 0) == 0
    seen = {}

    def worker():
        with MD.lane(1):
            seen['inside'] = getattr(MD._lanes, 'n', 0)
            ready.set()
            done.wait(5)
        seen['after'] = getattr(MD._lanes, 'n', 0)

    ready, done = threading.Event(), threading.Event()
    t = threading.Thread(target=worker)
    t.start()
    assert ready.wait(5)
    assert getattr(MD._lanes, 'n', 0) == 0          # the other thread's lane is its own
    with MD.lane(1):
        assert MD._lanes.n == 1
        with MD.lane(0):
            assert MD._lanes.n == 0
        assert MD._lanes.n == 1
    assert MD._lanes.n == 0
    done.set()
    t.join()
    assert seen == {'inside': 1, 'after': 0}

    coords = torch.zeros(5, 4, dtype=torch.int32)
    g = MD.Grid(coords)
    assert g.dims is None and g.ld == 256
    md = MD.Metadata(3)
    md.set_input((64, 32, 16), g)
    assert g.dims == (64, 32, 16) and md.grid((64, 32, 16)) is g
    md2 = MD.Metadata(3)
    md2.set_input((128, 128, 128), g)                 # the first registration wins: the size is a property of the level
    assert g.dims == (64, 32, 16)


def test_program_planner_turns_join_inputs_into_column_views():
    """Native executor planning (prog.hip: make_plan / make_layout, host code — runs without a GPU): in a Refinement
    program the inputs of both JoinTables live inside the joined buffer (no storage of their own, no concat kernel);
    a join input the caller wants to read back (`keep`) gets its own rows again; materialised buffers never overlap."""
    from sgnn_amd import _lib
    from sgnn_amd.model import GenModel
    from sgnn_amd.scn import program as P
    m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1)
    ref = m.refinement[0]
    prog = P.compile_or_none([ref.p1, ref.p2, ref.p3], ref.nf_in)
    assert prog is not None
    ops, bufs = prog.ops_np, prog.bufs_np
    nops, nbuf, ncls = ops.shape[0], bufs.shape[0], prog.n_classes
    lev_n = np.array([1000, 300, 90], dtype=np.int64)[:ncls]

    def plan(keep_bufs, mode=1):
        keep = np.zeros(nbuf, dtype=np.int32)
        keep[list(keep_bufs)] = 1
        a = (ops.ctypes.data, nops, bufs.ctypes.data, nbuf, prog.n_ext, lev_n.ctypes.data, ncls, keep.ctypes.data)
        total = _lib.query('sgnn_prog_arena_floats', *(a + (mode,)))
        return total, [_lib.query('sgnn_prog_buffer_offset', *(a + (int(mode == 2), b))) for b in range(nbuf)]

    OP_JOIN = 5
    joins = [o for o in ops if o[0] == OP_JOIN]
    assert len(joins) == 2
    total, offs = plan([prog.out])
    spans = []
    for b in range(prog.n_ext, nbuf):
        if offs[b] >= 0:
            spans.append((offs[b], offs[b] + int(lev_n[bufs[b, 0]]) * int(bufs[b, 1]), b))
    spans.sort()
    assert spans[-1][1] <= total
    for (s0, e0, b0), (s1, e1, b1) in zip(spans, spans[1:]):
        assert e0 <= s1, 'buffers %d and %d overlap' % (b0, b1)
    for o in joins:
        a, b, out = int(o[1]), int(o[2]), int(o[3])
        assert offs[a] == -1 and offs[b] == -1 and offs[out] >= 0          # views of the joined buffer
        assert int(bufs[out, 1]) == int(bufs[a, 1]) + int(bufs[b, 1])
    # reading a join input back forces it out of the view
    a0 = int(joins[0][1])
    total2, offs2 = plan([prog.out, a0])
    assert offs2[a0] >= 0 and total2 >= total + int(lev_n[bufs[a0, 0]]) * int(bufs[a0, 1])
    # twice the rows: twice the arena (256-byte alignment aside)
    lev_n = lev_n * 2
    total3, _ = plan([prog.out])
    assert abs(total3 - 2 * total) <= 64 * nbuf


def test_inference_layout_packs_buffers_by_liveness():
    """sgnn_prog_arena_floats(mode 2) / sgnn_prog_forward(training = 2): buffers whose live ranges (first writer .. last
    reader, a JoinTable view's members counted with the joined buffer) do not intersect may share storage, buffers alive
    at the same op never overlap, the outputs stay to the end; the arena is several times smaller than the training
    layout of the same program.  Host arithmetic only."""
    from sgnn_amd import _lib
    from sgnn_amd.model import GenModel
    from sgnn_amd.scn import program as P
    m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1)
    ref = m.refinement[1]
    for mods, cin in (([ref.p1, ref.p2, ref.p3], ref.nf_in), (list(m.encoder.process_sparse[0].children())[1:], None)):
        prog = P.compile_or_none(mods, cin) if cin is not None else m.encoder.process_sparse[0].__dict__.get('_prog')
        if prog is None:
            continue
        ops, bufs = prog.ops_np, prog.bufs_np
        nops, nbuf, ncls, n_ext = ops.shape[0], bufs.shape[0], prog.n_classes, prog.n_ext
        lev_n = np.array([100000, 27000, 6100, 1500, 800000, 5000][:ncls], dtype=np.int64)
        keep = np.zeros(nbuf, dtype=np.int32)
        keep[prog.out] = 1
        a = (ops.ctypes.data, nops, bufs.ctypes.data, nbuf, n_ext, lev_n.ctypes.data, ncls, keep.ctypes.data)
        train_total = _lib.query('sgnn_prog_arena_floats', *(a + (1,)))
        total = _lib.query('sgnn_prog_arena_floats', *(a + (2,)))
        offs = [_lib.query('sgnn_prog_buffer_offset', *(a + (1, b))) for b in range(nbuf)]
        offs_train = [_lib.query('sgnn_prog_buffer_offset', *(a + (0, b))) for b in range(nbuf)]
        assert [o >= 0 for o in offs] == [o >= 0 for o in offs_train]       # same views in both layouts
        # storage owner of every buffer: views (offset -1) belong to the JoinTable output that swallowed them
        owner = list(range(nbuf))
        for o in ops:
            if o[0] == 5:
                for q in (int(o[1]), int(o[2])):
                    if q >= n_ext and offs[q] < 0:
                        owner[q] = int(o[3])
        def root(b):
            while owner[b] != b:
                b = owner[b]
            return b
        first, last, fused_away = {}, {}, set()
        def touch(b, i):
            if b < n_ext:
                return
            r = root(b)
            first[r] = min(first.get(r, i), i)
            last[r] = max(last.get(r, i), i)
        readers = {}
        for o in ops:
            for q in ([int(o[1])] + ([int(o[2])] if o[0] in (4, 5, 6) else [])):
                readers[q] = readers.get(q, 0) + 1
        for i, o in enumerate(ops):
            touch(int(o[1]), i) if o[1] >= 0 else None
            if o[0] in (4, 5, 6) and o[2] >= 0:
                touch(int(o[2]), i)
            nxt = ops[i + 1] if i + 1 < nops else None
            if (o[0] in (0, 1) and nxt is not None and nxt[0] == 4 and int(o[3]) in (int(nxt[1]), int(nxt[2])) and
                    readers.get(int(o[3])) == 1 and nxt[1] != nxt[2]):
                touch(int(nxt[3]), i)       # fused AddTable: the convolution stores the sum, its own buffer is never used
                fused_away.add(int(o[3]))
            elif o[3] >= 0:
                touch(int(o[3]), i)
        for b in fused_away:
            if first.get(b) == last.get(b):
                first.pop(b), last.pop(b)
        last[root(prog.out)] = nops + 1
        spans = [(offs[b], offs[b] + int(lev_n[bufs[b, 0]]) * int(bufs[b, 1]), first[b], last[b], b)
                 for b in first if offs[b] >= 0]
        assert max(e for _, e, _, _, _ in spans) <= total
        shared = 0
        for x in range(len(spans)):
            for y in range(x + 1, len(spans)):
                s0, e0, f0, l0, b0 = spans[x]
                s1, e1, f1, l1, b1 = spans[y]
                if s0 < e1 and s1 < e0:             # same storage: the live ranges must be disjoint
                    assert l0 < f1 or l1 < f0, 'buffers %d and %d are alive together and overlap' % (b0, b1)
                    shared += 1
        assert shared > 0
        assert total * 3 < train_total * 2, (total, train_total)


def test_dense_conv_keeps_native_layout_and_speaks_the_reference_layout():
    """model.DenseConv stores (K, Cin, Cout) — what the rulebook kernels read — while state_dict / load_state_dict /
    named_gradients use nn.Conv3d's (Cout, Cin, k, k, k) and nn.ConvTranspose3d's (Cin, Cout, k, k, k)."""
    from sgnn_amd.model import DenseConv, GenModel, named_gradients, K4S2_SLOT
    torch.manual_seed(3)
    for transposed, (cin, cout, k) in ((False, (16, 24, 4)), (True, (64, 32, 4)), (False, (32, 32, 1))):
        ref = (torch.nn.ConvTranspose3d if transposed else torch.nn.Conv3d)(cin, cout, k, stride=2 if k == 4 else 1,
                                                                             padding=1 if k == 4 else 0, bias=False)
        d = DenseConv(cin, cout, k, 2 if k == 4 else 1, 1 if k == 4 else 0, transposed)
        assert tuple(d.weight.shape) == (k ** 3, cin, cout)
        assert tuple(d.state_dict()['weight'].shape) == tuple(ref.weight.shape)
        d.load_state_dict(ref.state_dict())
        assert torch.equal(d.state_dict()['weight'], ref.weight.detach())
        # tap (a, b, c) of the torch weight is one (Cin, Cout) slice of the native one: slice a*k*k + b*k + c, or — k4/s2
        # layers keep their taps in parity-group order (model.K4S2_TAPS) — the slot that holds that tap
        a, b, c = (1, 2, 3) if k == 4 else (0, 0, 0)
        want = ref.weight[:, :, a, b, c] if transposed else ref.weight[:, :, a, b, c].t()
        slot = K4S2_SLOT[(a * k + b) * k + c] if k == 4 else (a * k + b) * k + c
        assert d.parity_order == (k == 4)
        assert torch.equal(d.weight[slot], want)
        assert torch.equal(d.to_native(d.to_torch(d.weight)), d.weight)
    torch.manual_seed(5)
    hm = GenModel(8, (32,) * 3, 1, 16, 16, 4, True, True, 1, 1)
    torch.manual_seed(5)
    om = mo.GenModel(8, (32,) * 3, 1, 16, 16, 4, True, True, 1, 1)
    for (k1, v1), (k2, v2) in zip(hm.state_dict().items(), om.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1          # same initialisation stream, same layout on the outside
    for p in hm.parameters():
        p.grad = torch.randn_like(p)
    g = named_gradients(hm)
    sd = om.state_dict()
    assert all(tuple(g[n].shape) == tuple(sd[n].shape) for n, _ in om.named_parameters())
    w = hm.encoder.encode_dense0[0]
    assert torch.equal(g['encoder.encode_dense0.0.weight'], w.to_torch(w.weight.grad))


def test_host_side_size_queries_of_the_c_abi():
    """Workspace / blob size queries are plain host arithmetic (callable without a GPU): the caller sizes its buffers
    with them, so their formulas are part of the contract (include/sgnn_hip.h)."""
    from sgnn_amd import _lib
    al = lambda v: (v + 255) // 256 * 256
    # a plain (gather-kernel) launch: one statistics partial per 256-row workgroup above ~40 k rows, per 16 rows below
    assert _lib.query('sgnn_conv_stats_blocks', 366085) == (366085 + 255) // 256
    assert _lib.query('sgnn_conv_stats_blocks', 1000) == (1000 + 15) // 16
    assert _lib.query('sgnn_conv_stats_blocks', 0) == 0
    cap = _lib.query('sgnn_hash_capacity', 366085)
    assert cap >= 2 * 366085 and cap & (cap - 1) == 0
    assert _lib.query('sgnn_conv_bwd_weight_ws_bytes', 0, 27, 16, 16) == 0
    assert _lib.query('sgnn_conv_bwd_weight_ws_bytes', 1000, 27, 16, 16) == 4 * 27 * 16 * 16 * 4      # 4 row blocks of 256


def test_flat_adam_collect_leaves_no_stale_gradient_behind():
    """capacity-mode backward writes program gradients straight into FlatAdam.flat_g; a later classic step whose
    hierarchy stops early (no gradient for a stage, torch/model.py:211) must not leave that stage's OLD gradient in the
    buffer — a data-parallel all-reduce sums the whole buffer and another rank's "reached" flag would apply it."""
    from sgnn_amd.train import FlatAdam
    torch.manual_seed(0)
    pa = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    pb = [torch.nn.Parameter(torch.randn(2, 2)), torch.nn.Parameter(torch.randn(7))]
    opt = FlatAdam([('a', pa), ('b', pb)], lr=1e-2)
    opt.flat_g[:opt.numel].fill_(123.0)                    # what an earlier step left behind
    for p in pa:
        p.grad = torch.ones_like(p)
    pb[0].grad = None
    pb[1].grad = None
    reached = opt.collect()
    assert reached == [True, False]
    (b0, e0), (b1, e1) = opt.bounds
    assert torch.equal(opt.flat_g[b0:e0], torch.ones(e0 - b0))
    assert torch.equal(opt.flat_g[b1:e1], torch.zeros(e1 - b1)), 'unreached segment keeps a stale gradient'
    # a parameter without a gradient INSIDE a reached segment reads as zero as well
    opt.flat_g[:opt.numel].fill_(7.0)
    pa[1].grad = None
    assert opt.collect() == [True, False]
    assert torch.equal(opt.views_g[1], torch.zeros(5)) and torch.equal(opt.views_g[0], torch.ones(4, 3))
    g = opt.named_gradients(torch.nn.ParameterList(pa + pb))
    assert len(g) == 4 and all(v is not None for v in g.values())


def test_sgnn_tune_calls_the_named_switches_and_rejects_anything_else():
    """SGNN_TUNE is the measurement hook behind scripts/ab_env2.sh: `name=value` pairs call integer switches of the library
    once at load; a name that is not a switch must raise instead of being ignored (a typo would silently A/B nothing)."""
    import subprocess
    import sys
    code = ("from sgnn_amd import _lib; lib = _lib.load(); "
            "print(_lib.tune('conv_one_round', 1), _lib.tune('conv_dw_blocks', 256), _lib.tune('prog_lin_bn', 1))")
    env = dict(os.environ, SGNN_TUNE='conv_one_round=0,conv_dw_blocks=341,prog_lin_bn=0')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ['0', '341', '0']          # the setters return what SGNN_TUNE had installed
    # not a switch; unknown; a *_set_* entry point that takes pointers; a value that is not an integer
    for bad in ('sgnn_conv_fwd=1', 'no_such_switch=1', 'sgnn_prog_set_side_stream=1', 'conv_one_round=on'):
        env = dict(os.environ, SGNN_TUNE=bad)
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=ROOT)
        assert out.returncode != 0 and 'SgnnError' in out.stderr and 'SGNN_TUNE' in out.stderr, (bad, out.stderr[-500:])


def test_hash_capacity_is_monotone_and_keeps_its_load_factors():
    """a table sized from an upper bound must never be smaller than one sized from a live count (n = 65 535 used to
    get 262 144 slots, n = 65 536 only 131 072).  Load factor <= 0.25 below 64 k sites, <= 0.5 everywhere; power of two."""
    from sgnn_amd import _lib
    prev = 0
    for n in list(range(0, 3000, 7)) + list(range(60000, 140000, 251)) + [65535, 65536, 131071, 131072, 131073, 10 ** 6, 10 ** 7]:
        cap = _lib.query('sgnn_hash_capacity', n)
        assert cap & (cap - 1) == 0 and cap >= 1024
        assert cap >= 2 * n and (n >= 65536 or cap >= 4 * n)
    for n in sorted(set(list(range(0, 300000, 997)) + [65535, 65536, 65537, 131072, 131073])):
        cap = _lib.query('sgnn_hash_capacity', n)
        assert cap >= prev, (n, cap, prev)
        prev = cap


def test_tune_table_sets_reads_and_rejects():
    """include/sgnn_hip.h struct sgnn_tune: every field settable by name, booleans normalised, ranges enforced, unknown names
    raise; the struct the header documents is the table the library exports (field count and order)."""
    import ctypes
    import re
    from sgnn_amd import _lib
    lib = _lib.load()
    names = lib.sgnn_tune_names().decode().split(',')
    header = open(os.path.join(ROOT, 'include', 'sgnn_hip.h')).read()
    body = re.sub(r'/\*.*?\*/', '', header[header.index('typedef struct sgnn_tune {'):header.index('} sgnn_tune;')], flags=re.S)
    assert re.findall(r'int64_t\s+(\w+);', body) == names
    cur = ctypes.cast(lib.sgnn_tune_current(), ctypes.POINTER(ctypes.c_int64 * len(names))).contents
    for i, n in enumerate(names):
        assert _lib.tune(n) == cur[i]
    prev = _lib.tune('conv_dw_blocks', 341)
    try:
        assert _lib.tune('conv_dw_blocks') == 341 and cur[names.index('conv_dw_blocks')] == 341
        with pytest.raises(_lib.SgnnError):
            _lib.tune('conv_dw_blocks', 0)
        assert _lib.tune('conv_dw_blocks') == 341
    finally:
        _lib.tune('conv_dw_blocks', prev)
    p = _lib.tune('prog_fusion', 7)
    assert _lib.tune('prog_fusion') == 1
    _lib.tune('prog_fusion', p)
    with pytest.raises(_lib.SgnnError):
        _lib.tune('no_such_switch', 1)
    with pytest.raises(_lib.SgnnError):
        _lib.tune('no_such_switch')


def test_fused_backward_transposition_layout_is_a_conflict_free_bijection():
    """csrc/conv_bwd_fused.hip moves every gathered 64 x 16 tile from the MFMA A-fragment layout (lane (r, q) holds row
    16 m + r, channels 4 q + j) to the B-operand layout of the weight-gradient product (lane (n, kk) holds rows 16 m + 4 kk + i,
    channel n) through a wave-private LDS image: 16 ds_write_b32 in lane order (image (m, j) at m * 296 + img(j) floats) and 4
    ds_read_b128.  Restated here: the mapping returns every element where the product expects it, and the skew {0, 8, 32, 40}
    keeps each of ds_read_b128's four 16-lane groups (MI355X_MICROARCH.md, LDS table) on 64 distinct banks."""
    TS = 296
    img = lambda j: j * 64 + (j & 1) * 8 + (j >> 1) * 32
    lds = {}
    tile = np.arange(64 * 16).reshape(64, 16)            # tile[row][channel] = a unique id
    for lane in range(64):
        r, q = lane & 15, lane >> 4
        for m in range(4):
            for j in range(4):
                addr = m * TS + img(j) + lane             # tw_wr[m * TSTRIDE + fused_img(j)], tw_wr = tw + lane
                assert addr not in lds
                lds[addr] = tile[16 * m + r][4 * q + j]
    assert max(lds) < 4 * TS
    for lane in range(64):
        n, kk = lane & 15, lane >> 4
        base = img(n & 3) + 16 * (n >> 2) + 4 * kk        # rd_off
        for m in range(4):
            assert (m * TS + base) % 4 == 0               # 16-byte aligned ds_read_b128
            for i in range(4):
                assert lds[m * TS + base + i] == tile[16 * m + 4 * kk + i][n]
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for m in range(4):
        for g in groups:
            banks = [(m * TS + img((l & 15) & 3) + 16 * ((l & 15) >> 2) + 4 * (l >> 4) + i) % 64 for l in g for i in range(4)]
            assert len(set(banks)) == 64
