"""Device-resident voxel grids and rulebooks (host-side bookkeeping only).

Counterpart of upstream's `Metadata_3` object that the reference reaches through
`x.metadata.getSpatialLocations(x.spatial_size)` (torch/model.py:380) and that every scn op
shares.  All hashing / rulebook arithmetic happens in libsgnn_hip.so; this module only owns the
torch tensors that back the tables and caches them per (spatial size, filter) like upstream does.
"""
import numpy as np
import os
import threading
from contextlib import nullcontext as _nullcontext

import torch

from .. import _lib
from .._lib import ptr


SIDE_LANE = os.environ.get('SGNN_SIDE_LANE', '1') != '0'     # run dW on a second stream during program backward (sgnn_prog_set_side_stream)
# HIP stream priority of that lane (0 = default, positive = lower than the training stream, negative = higher): the
# weight-gradient kernels are off the critical path, the data-gradient chain they overlap is on it
SIDE_PRIORITY = int(os.environ.get('SGNN_SIDE_PRIORITY', '0'))
# Capacity mode: the stride-2 pyramid of a level (sgnn_down2_chain_tables) and the 3x3x3 rulebooks of its coarse levels
# are issued on a third stream (scratch lane 2) and joined inside sgnn_prog_forward right before the program's first
# Convolution(2,2): they overlap the level's hash / rulebook build and its first convolutions instead of preceding them.
SIDE_PYRAMID = os.environ.get('SGNN_SIDE_PYRAMID', '1') != '0'
PYRAMID_LANE = 2


class _Runtime(object):
    """Per-device scratch: grow-only workspace + an 8-word state block (count, status)."""

    def __init__(self, device):
        self.device = device
        self.ws = torch.empty(1 << 20, dtype=torch.uint8, device=device)
        # [0] count of the last compaction / stride-2 build, [1] status word, [2:] row counts of a pre-issued
        # stride-2 chain (sgnn_down2_chain): one D2H copy returns all of them
        self.state = torch.zeros(8, dtype=torch.int64, device=device)
        self.status32 = self.state[1:2].view(torch.int32)  # two int32 words, first one used
        self.syncs = 0                                     # host read-backs so far (bench / tests)

    def index_volume(self):
        """Persistent dense index volume of the rulebook builder (sgnn_rulebook_subm3_dense): all -1 between calls."""
        if getattr(self, '_volume', None) is None:
            self._volume = torch.full((INDEX_VOLUME_ENTRIES,), -1, dtype=torch.int32, device=self.device)
        return self._volume

    def workspace(self, nbytes):
        if self.ws.numel() < nbytes:
            self.ws = torch.empty(int(nbytes * 1.5) + 256, dtype=torch.uint8, device=self.device)
        return self.ws

    def pyramid_stream(self):
        """Stream of the pyramid lane (see SIDE_PYRAMID); kernels issued there use the scratch of lane PYRAMID_LANE.
        It IS the weight-gradient lane's stream (idle during the forward pass): a third concurrently active stream was
        measured to collide with the training stream on one hardware pipe on some runs — both queues then stall
        40-70 us at every switch (18.8 instead of 6.2 ms per step, profiles/r03t_trace_summary.txt)."""
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device, priority=SIDE_PRIORITY)
            self._side_ws = None
        self._pyr_stream = self._side_stream
        return self._pyr_stream

    def side_lane(self, nbytes):
        """Second stream + private workspace for the weight-gradient lane of sgnn_prog_backward (registered with the
        library whenever it changes; only called between backward calls, when the lane is idle)."""
        if not SIDE_LANE:
            if getattr(self, '_side_on', False):
                _lib.query('sgnn_prog_set_side_stream', None, None, 0)
                self._side_on = False
            return
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device, priority=SIDE_PRIORITY)
            self._side_ws = None
        if self._side_ws is None or self._side_ws.numel() < nbytes or not getattr(self, '_side_on', False):
            if self._side_ws is None or self._side_ws.numel() < nbytes:
                if self._side_ws is not None:
                    self._side_stream.synchronize()    # a deferred join: the lane may still be using the old workspace
                self._side_ws = torch.empty(int(nbytes * 1.5) + 256, dtype=torch.uint8, device=self.device)
            rc = _lib.query('sgnn_prog_set_side_stream', self._side_stream.cuda_stream, self._side_ws.data_ptr(),
                            self._side_ws.numel())
            assert rc == 0
            self._side_on = True

    def read_count(self):
        """One D2H copy: returns the count word and raises on pending input errors."""
        return self.read_counts()[0]

    def check_status(self):
        """Raise now if a kernel flagged an input error (duplicate / out-of-range coordinates) since the last read-back
        (one D2H copy).  GenModel.forward calls it at the end of inference passes, so that the error is attributed to
        the call that caused it; in training it surfaces at the next of the step's five read-backs (with the geometry
        prefetcher, which leaves the training stream without read-backs: when that step's end-of-step event is seen)."""
        self.read_counts()

    def read_counts(self):
        """One D2H copy of the whole state block: [count, status, chain counts...] as Python ints."""
        self.syncs += 1
        host = self.state.cpu().tolist()
        self.raise_status(int(host[1]))
        return host

    def raise_status(self, word, where=None):
        """Turn a copy of the status word into the input error it stands for (and clear the device word).  `where`:
        which step / batch the word belongs to when it is looked at later than it was written (deferred checks)."""
        status = int(word) & 0xFFFFFFFF
        if status:
            self.state[1] = 0
            msgs = []
            if status & 1:
                msgs.append('coordinate outside [0,65535] (batch outside [0,32767]), or — a level whose rulebook is read off the '
                            'dense index volume alone — a site outside the bounds the level declared')
            if status & 2:
                msgs.append('InputLayer(mode=0): duplicate coordinates are a caller error')
            if status & 4 and not (status & 3):
                raise CapacityOverflow('capacity mode: a level produced more rows than its capacity (step discarded)')
            if status & 4:
                msgs.append('capacity overflow')
            raise _lib.SgnnError('; '.join(msgs) + (' [%s]' % where if where else ''))


class CapacityOverflow(_lib.SgnnError):
    """Capacity mode: some level had more rows than the buffers sized for it.  Nothing was written out of bounds and
    sgnn_adam_flat skipped the update; re-run the batch with larger capacities (train.GraphStep does)."""


# exact-mode row counts of the last forward pass, recorded when this is a list: ('enc', n_input, [pyramid rows]) and
# ('gen', kept rows, [pyramid rows]) entries in call order — what scn.capacity.Capacity.from_log sizes itself from
COUNT_LOG = None

_runtimes = {}
_lanes = threading.local()     # .n = scratch lane of this thread (0 = main); see `lane`


class lane(object):
    """`with lane(1):` — runtime() hands out the scratch (workspace, count/status block) of lane 1 inside the block, in
    this thread.  Kernels of different streams must not share them; the caller pairs a lane with ONE stream (geometry
    prefetch: train.GeometryPrefetcher, possibly on a worker thread)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.prev = getattr(_lanes, 'n', 0)
        _lanes.n = self.n

    def __exit__(self, *exc):
        _lanes.n = self.prev


def runtime(device=None):
    idx = device.index if isinstance(device, torch.device) else None
    if idx is None:
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
    ln = getattr(_lanes, 'n', 0)
    rt = _runtimes.get((idx, ln))
    if rt is None:
        _lib.require_gpu()
        rt = _runtimes[(idx, ln)] = _Runtime(torch.device('cuda', idx))
    if idx != torch.cuda.current_device():
        # every entry point launches on the CURRENT device's stream: tensors of another GPU would be touched through
        # a foreign stream (one process per GPU is the supported layout; torch.cuda.set_device selects it)
        raise _lib.SgnnError('sgnn_amd: tensors live on cuda:%d but the current device is cuda:%d — call '
                             'torch.cuda.set_device(%d) (one process per GPU)' % (idx, torch.cuda.current_device(), idx))
    return rt


# 3x3x3 rulebooks of levels with at least DENSE_RULEBOOK_MIN_ROWS rows whose spatial size is known (levels registered
# in a Metadata) are built through a dense index volume instead of hash probes (grid_rules.hip: k_rulebook_subm3_vol;
# identical tables).  Below that the three small launches of the dense path cost more than the probes save.
# SGNN_DENSE_RULEBOOK_MIN_ROWS=-1 switches the path off; SGNN_INDEX_VOLUME_MB sizes the volume (blocks it does not cover
# fall back to the hash inside the kernel).
DENSE_RULEBOOK_MIN_ROWS = int(os.environ.get('SGNN_DENSE_RULEBOOK_MIN_ROWS', '16384'))
if DENSE_RULEBOOK_MIN_ROWS < 0:
    DENSE_RULEBOOK_MIN_ROWS = 1 << 62
INDEX_VOLUME_ENTRIES = int(os.environ.get('SGNN_INDEX_VOLUME_MB', '128')) << 18
# generated levels whose sites are inside the index volume by construction build their rulebook from the volume alone and
# never build a hash grid of their own (Grid.bounds); SGNN_VOLUME_ONLY=0: hash grid + volume with hash fall-back (A/B, parity)
VOLUME_ONLY = os.environ.get('SGNN_VOLUME_ONLY', '1') != '0'


def _round_up(n, m):
    return ((n + m - 1) // m) * m


class Grid(object):
    """Active sites of one resolution level: int32 coords (n,4) [z,y,x,b]; row i <-> site i."""

    def __init__(self, coords32, keys=None, vals=None, cap=0, cnt=None):
        assert coords32.dtype == torch.int32 and coords32.dim() == 2 and coords32.shape[1] == 4
        self.coords = coords32.contiguous()
        self.n = int(coords32.shape[0])
        # capacity mode: `n` is the capacity, the live row count is this device int64[1] (None: n is exact)
        self.cnt = cnt
        self.device = coords32.device
        self.keys, self.vals, self.cap = keys, vals, cap
        self._nbr = None
        self.ready = None      # see Down2.ready
        self.dims = None       # spatial size (z, y, x) of the level once a Metadata registers the grid
        # (B, Z, Y, X): every site satisfies b < B, z < Z, ... BY CONSTRUCTION (generated levels: children of a dense coarse
        # volume; the model passes the bound along with the coordinates) — such a level's rulebook needs no hash grid
        self.bounds = getattr(coords32, '_sgnn_bounds', None)
        self.ld = _round_up(max(self.n, 1), 256)   # table leading dimension (conv kernels: multiple of 256)

    def hash(self):
        if self.keys is None:
            rt = runtime(self.device)
            self.cap = _lib.query('sgnn_hash_capacity', self.n)
            self.keys = torch.empty(self.cap, dtype=torch.int64, device=self.device)
            self.vals = torch.empty(self.cap, dtype=torch.int32, device=self.device)
            _lib.call('sgnn_hash_build', ptr(self.coords), self.n, ptr(self.keys), ptr(self.vals), self.cap,
                      ptr(rt.status32), ptr(self.cnt))
        return self.keys, self.vals, self.cap

    def lookup(self, query32, m_cnt=None):
        """Row of each query site in this grid, or -1 (int32 tensor).  m_cnt: device count of the query rows
        (capacity mode; rows past it are left unwritten)."""
        keys, vals, cap = self.hash()
        m = int(query32.shape[0])
        rows = torch.empty(m, dtype=torch.int32, device=self.device)
        _lib.call('sgnn_hash_lookup', ptr(keys), ptr(vals), cap, ptr(query32), m, ptr(rows), ptr(m_cnt))
        return rows

    def subm_table(self):
        """3x3x3 neighbour table, int32 [27][ld] (cached; one build serves every conv of the level)."""
        if self._nbr is None:
            d, vb = self.dims, self.bounds
            dense = (d is not None and len(d) == 3 and self.n >= DENSE_RULEBOOK_MIN_ROWS
                     and 0 < d[0] * d[1] * d[2] <= INDEX_VOLUME_ENTRIES and max(d) <= 65536)
            if (dense and VOLUME_ONLY and vb is not None and self.keys is None and all(int(vb[1 + a]) <= int(d[a]) for a in range(3))
                    and int(vb[0]) * d[0] * d[1] * d[2] <= INDEX_VOLUME_ENTRIES):
                # every site lies inside the volume by construction: volume only — the level's hash grid is never built
                # (sgnn_hash_build at 517 k sites: 50 us on the training queue between two stages)
                rt = runtime(self.device)
                self._nbr = torch.empty(27 * self.ld, dtype=torch.int32, device=self.device)
                vol = rt.index_volume()
                _lib.call('sgnn_rulebook_subm3_volume', ptr(self.coords), self.n, int(d[0]), int(d[1]), int(d[2]), ptr(vol),
                          vol.numel(), ptr(self._nbr), self.ld, ptr(self.cnt), ptr(rt.status32))
                return self._nbr
            keys, vals, cap = self.hash()
            self._nbr = torch.empty(27 * self.ld, dtype=torch.int32, device=self.device)
            if dense:
                vol = runtime(self.device).index_volume()
                _lib.call('sgnn_rulebook_subm3_dense', ptr(keys), ptr(vals), cap, ptr(self.coords), self.n, int(d[0]),
                          int(d[1]), int(d[2]), ptr(vol), vol.numel(), ptr(self._nbr), self.ld, ptr(self.cnt))
            else:
                _lib.call('sgnn_rulebook_subm3', ptr(keys), ptr(vals), cap, ptr(self.coords), self.n, ptr(self._nbr),
                          self.ld, ptr(self.cnt))
        return self._nbr

    def locations_i64(self):
        out = torch.empty(self.n, 4, dtype=torch.int64, device=self.device)
        _lib.call('sgnn_coords_to_i64', ptr(self.coords), self.n, ptr(out), ptr(self.cnt))
        if self.cnt is not None:
            out._sgnn_cnt = self.cnt
        return out


class Down2(object):
    """Stride-2 rulebook between a fine and a coarse grid.  `ready` (capacity mode, SIDE_PYRAMID): torch.cuda.Event after
    which the tables — and the coarse grid's hash and 3x3x3 rulebook — are complete; they are built on another stream."""
    ready = None

    def __init__(self, fine, coarse, parent, children, ldc, ptable, ldf):
        self.fine, self.coarse = fine, coarse
        self.parent, self.children, self.ldc, self.ptable, self.ldf = parent, children, ldc, ptable, ldf


def build_down2(fine):
    rt = runtime(fine.device)
    nf, dev = fine.n, fine.device
    ccap = _lib.query('sgnn_hash_capacity', nf)
    ckeys = torch.empty(ccap, dtype=torch.int64, device=dev)
    cvals = torch.empty(ccap, dtype=torch.int32, device=dev)
    parent = torch.empty(max(nf, 1), dtype=torch.int32, device=dev)
    ccoords = torch.empty(max(nf, 1), 4, dtype=torch.int32, device=dev)
    wsb = _lib.query('sgnn_down2_ws_bytes', nf)
    ws = rt.workspace(wsb)
    _lib.call('sgnn_rulebook_down2', ptr(fine.coords), nf, ptr(ckeys), ptr(cvals), ccap, ptr(parent), ptr(ccoords),
              ptr(rt.state), ptr(ws), wsb)
    nc = rt.read_count()  # host sync: the coarse row count sizes every downstream buffer
    coarse = Grid(ccoords[:nc], ckeys, cvals, ccap)
    ldc, ldf = coarse.ld, fine.ld
    children = torch.empty(8 * ldc, dtype=torch.int32, device=dev)
    ptable = torch.empty(8 * ldf, dtype=torch.int32, device=dev)  # rows >= nf are never read
    _lib.call('sgnn_down2_tables', ptr(fine.coords), ptr(parent), nf, ptr(children), ldc, nc, ptr(ptable), ldf, None,
              None)
    return Down2(fine, coarse, parent[:nf], children, ldc, ptable, ldf)


CHAIN = True    # False: one stride-2 build + one host read-back per level (the round-1 behaviour)
MAX_CHAIN = 6   # state words available for chain counts


class PendingChain(object):
    """Stride-2 pyramid below a level, issued before the host knows any row count (sgnn_down2_chain).
    `finalize(n0, counts)` turns it into Grid / Down2 objects once the single read-back has happened."""

    def __init__(self, coords_cap, n0, n0_on_device, depth, n0_cnt=None, counts=None, level_caps=None):
        """n0_cnt / counts / level_caps (capacity mode): the device int64[1] holding level 0's row count, the device
        int64[depth] that receives the rows of levels 1..depth (clamped to level_caps, a list of ints) — instead of
        the runtime's state block and a host read-back."""
        dev = coords_cap.device
        rt = runtime(dev)
        cap = int(coords_cap.shape[0])
        depth = min(depth, MAX_CHAIN)
        self.n0_cnt, self.counts, self.level_caps = n0_cnt, counts, level_caps
        self.coords_cap, self.cap, self.depth = coords_cap, cap, depth
        self.ccap = _lib.query('sgnn_hash_capacity', cap)
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.ckeys = [mk(self.ccap, torch.int64) for _ in range(depth)]
        self.cvals = [mk(self.ccap, torch.int32) for _ in range(depth)]
        self.parent = [mk(max(cap, 1), torch.int32) for _ in range(depth)]
        self.ccoords = [mk((max(cap, 1), 4), torch.int32) for _ in range(depth)]
        wsb = _lib.query('sgnn_down2_chain_ws_bytes', cap)
        ws = rt.workspace(wsb)
        arr = lambda ts: np.ascontiguousarray(np.array([t.data_ptr() for t in ts], dtype=np.uint64))
        self._keep = [arr(self.ckeys), arr(self.cvals), arr(self.parent), arr(self.ccoords)]
        if counts is not None:
            # capacity mode: pyramid AND its children / ptable tables in one submission (3 launches per level + 2)
            assert n0_cnt is not None and level_caps is not None and len(level_caps) >= depth and counts.numel() >= depth
            caps = [min(int(c), cap) for c in level_caps[:depth]]
            caps_np = np.ascontiguousarray(np.array(caps, dtype=np.int64))
            ld = [_round_up(max(cap, 1), 256)] + [_round_up(max(c, 1), 256) for c in caps]
            self.children = [mk(8 * ld[l + 1], torch.int32) for l in range(depth)]
            self.ptable = [mk(8 * ld[l], torch.int32) for l in range(depth)]
            self._keep += [arr(self.children), arr(self.ptable)]
            wsb = _lib.query('sgnn_down2_chain_tables_ws_bytes', cap, depth)
            self.side = None
            if SIDE_PYRAMID:
                # every tensor above was allocated on the training stream (its pool, its ordering); only the launches
                # move: the lane waits for what the training stream has issued so far (coordinates, level-0 count)
                self.side = rt.pyramid_stream()
                with lane(PYRAMID_LANE):
                    rt2 = runtime(dev)
                    if rt2.ws.numel() < wsb:
                        self.side.synchronize()        # growing: the old scratch may still be in use over there
                    ws = rt2.workspace(wsb)
                _lib.stamp('fork')
                self.side.wait_stream(torch.cuda.current_stream(dev))
            else:
                ws = rt.workspace(wsb)
            with (torch.cuda.stream(self.side) if self.side is not None else _nullcontext()):
                _lib.stamp('lane<')
                _lib.call('sgnn_down2_chain_tables', ptr(coords_cap), ptr(n0_cnt), cap, depth,
                          self._keep[0].ctypes.data, self._keep[1].ctypes.data, self.ccap, self._keep[2].ctypes.data,
                          self._keep[3].ctypes.data, ptr(counts), caps_np.ctypes.data, self._keep[4].ctypes.data,
                          self._keep[5].ctypes.data, ptr(rt.status32), ptr(ws), wsb)
            return
        _lib.call('sgnn_down2_chain', ptr(coords_cap), 0 if n0_on_device else int(n0),
                  rt.state.data_ptr() if n0_on_device else None, cap, depth, self._keep[0].ctypes.data,
                  self._keep[1].ctypes.data, self.ccap, self._keep[2].ctypes.data, self._keep[3].ctypes.data,
                  rt.state.data_ptr() + 16, None, None, ptr(ws), wsb)

    def finalize(self, n0, host_state):
        """-> (grid of level 0, [Down2 level l -> l+1])."""
        counts = [int(v) for v in host_state[2:2 + self.depth]]
        fine = Grid(self.coords_cap[:n0])
        fine.bounds = getattr(self.coords_cap, '_sgnn_bounds', None)
        grid0, downs = fine, []
        for l in range(self.depth):
            nc = counts[l]
            coarse = Grid(self.ccoords[l][:nc], self.ckeys[l], self.cvals[l], self.ccap)
            downs.append(_down2_tables(fine, coarse, self.parent[l]))
            fine = coarse
        return grid0, downs

    def finalize_capped(self, grid0=None):
        """Capacity mode: Grid / Down2 objects sized by the level capacities, row counts left on the device."""
        fine = grid0 if grid0 is not None else Grid(self.coords_cap, cnt=self.n0_cnt)
        grid0, downs = fine, []
        for l in range(self.depth):
            cap_l = min(int(self.level_caps[l]), self.cap)
            coarse = Grid(self.ccoords[l][:cap_l], self.ckeys[l], self.cvals[l], self.ccap, cnt=self.counts[l:l + 1])
            assert coarse.ld * 8 == self.children[l].numel() and fine.ld * 8 == self.ptable[l].numel()
            downs.append(Down2(fine, coarse, self.parent[l][:fine.n], self.children[l], coarse.ld, self.ptable[l], fine.ld))
            fine = coarse
        if getattr(self, 'side', None) is not None:
            # the coarse levels' 3x3x3 rulebooks on the same lane (hash builder: the dense index volume belongs to the
            # training stream), then the event the consuming program waits for
            for d in downs:
                g = d.coarse
                g._nbr = torch.empty(27 * g.ld, dtype=torch.int32, device=g.device)        # training-stream allocation
            with torch.cuda.stream(self.side):
                for i in range(0, len(downs), _lib.RULEBOOK_MULTI_MAX):
                    gs = [d.coarse for d in downs[i:i + _lib.RULEBOOK_MULTI_MAX]]
                    arr = [np.array(v, dtype=np.int64) for v in (
                        [ptr(g.keys) for g in gs], [ptr(g.vals) for g in gs], [g.cap for g in gs],
                        [ptr(g.coords) for g in gs], [g.n for g in gs], [ptr(g._nbr) for g in gs], [g.ld for g in gs],
                        [ptr(g.cnt) for g in gs])]
                    _lib.call('sgnn_rulebook_subm3_multi', len(gs), *[a.ctypes.data for a in arr])
                _lib.stamp('lane>')
                ev = torch.cuda.Event()
                ev.record(self.side)
            for d in downs:
                d.ready = d.coarse.ready = ev
        return grid0, downs


def _down2_tables(fine, coarse, parent):
    dev = fine.device
    nf, nc = fine.n, coarse.n
    ldc, ldf = coarse.ld, fine.ld
    children = torch.empty(8 * ldc, dtype=torch.int32, device=dev)
    ptable = torch.empty(8 * ldf, dtype=torch.int32, device=dev)  # rows >= nf are never read
    _lib.call('sgnn_down2_tables', ptr(fine.coords), ptr(parent), nf, ptr(children), ldc, nc, ptr(ptable), ldf,
              ptr(fine.cnt), ptr(coarse.cnt))
    return Down2(fine, coarse, parent[:nf], children, ldc, ptable, ldf)


class Metadata(object):
    def __init__(self, dimension=3):
        self.dimension = dimension
        self.grids = {}
        self.down = {}

    @staticmethod
    def key(spatial_size):
        return spatial_size if isinstance(spatial_size, tuple) else tuple(int(s) for s in spatial_size)

    def _register(self, key, grid):
        self.grids[key] = grid
        if grid.dims is None:
            grid.dims = key

    def set_input(self, spatial_size, grid):
        self._register(self.key(spatial_size), grid)

    def adopt(self, spatial_size, grid0, downs):
        """Register a finalized PendingChain: level 0 at `spatial_size`, level l at spatial_size / 2^l."""
        key = self.key(spatial_size)
        self._register(key, grid0)
        for d in downs:
            nxt = tuple(v // 2 for v in key)
            self.down[(key, nxt)] = d
            self._register(nxt, d.coarse)
            key = nxt

    def prebuild(self, spatial_size, depth, capacity=None):
        """Build the stride-2 pyramid (`depth` levels below `spatial_size`) with ONE host read-back instead of one
        per level.  The level-0 grid must be registered already; levels already built are kept.
        capacity = (level capacities, device int64[depth] for the row counts): capacity mode, no read-back at all."""
        key = self.key(spatial_size)
        k, todo = key, 0
        for _ in range(depth):
            if any(v % 2 for v in k):
                break
            nxt = tuple(v // 2 for v in k)
            if (k, nxt) not in self.down:
                todo += 1
            k = nxt
        g0 = self.grids[key]
        if capacity is not None:
            if g0.cnt is None or len(self.down) or todo != depth:
                raise _lib.SgnnError('capacity mode: the level-0 grid needs a device row count and an unbuilt pyramid')
            caps, counts = capacity
            chain = PendingChain(g0.coords, g0.n, True, depth, n0_cnt=g0.cnt, counts=counts, level_caps=caps)
            _, downs = chain.finalize_capped(g0)
            self.adopt(key, g0, downs)
            return
        if not CHAIN or todo < 2 or g0.n == 0 or len(self.down):
            return
        chain = PendingChain(g0.coords, g0.n, False, todo)
        host = runtime(g0.device).read_counts()
        if COUNT_LOG is not None:
            COUNT_LOG.append(('enc', g0.n, [int(v) for v in host[2:2 + chain.depth]]))
        _, downs = chain.finalize(g0.n, host)
        fine = g0
        for d in downs:            # level 0 keeps its Grid object (hash / rulebook caches)
            d.fine = fine
            fine = d.coarse
        self.adopt(key, g0, downs)

    def grid(self, spatial_size):
        return self.grids[self.key(spatial_size)]

    def getSpatialLocations(self, spatial_size):
        """(N,4) int64 [z,y,x,batch] in active-row order (torch/model.py:380)."""
        return self.grid(spatial_size).locations_i64()

    def down2(self, in_size, out_size):
        k = (self.key(in_size), self.key(out_size))
        if k not in self.down:
            d = build_down2(self.grid(in_size))
            if self.key(out_size) not in self.grids:
                self._register(self.key(out_size), d.coarse)
            self.down[k] = d
        return self.down[k]


def join_pyramid_lane(device):
    """The training stream waits for everything issued so far on the side lane's stream — pyramids, dense weight gradients,
    the weight-gradient lane of sgnn_prog_backward (one stream) — no-op if it was never created."""
    rt = runtime(device)
    side = getattr(rt, '_side_stream', None)
    if side is not None:
        _lib.stamp('join<')
        with torch.cuda.stream(side):
            _lib.stamp('lane-end')
        torch.cuda.current_stream(rt.device).wait_stream(side)
        _lib.stamp('join>')


def coords_from_locs(locs, device):
    """Reference-style LongTensor (N,4) [z,y,x,b] (any device) -> device int32 rows."""
    rt = runtime(device)
    cnt = getattr(locs, '_sgnn_cnt', None)       # capacity mode: live row count of `locs` (device int64[1])
    if locs.dtype == torch.int32:
        out = locs.to(device).contiguous()
    else:
        src = locs.to(device=device, dtype=torch.int64).contiguous()
        n = int(src.shape[0])
        out = torch.empty(n, 4, dtype=torch.int32, device=device)
        _lib.call('sgnn_coords_from_i64', ptr(src), n, ptr(out), ptr(rt.status32), ptr(cnt))
    if cnt is not None and out is not locs:
        out._sgnn_cnt = cnt
        for a in ('_sgnn_cnt8', '_sgnn_plan', '_sgnn_children'):
            if hasattr(locs, a):
                setattr(out, a, getattr(locs, a))
    return out
