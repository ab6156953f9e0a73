"""Data files and batch loading — counterpart of torch/data_util.py:63-155 (load_train_file, load_scene,
load_scene_known) and torch/scene_dataloader.py (SceneDataset :39-116, collate :13-36); SURVEY.md §8 row f2.

Two ways in, same results:

  * `load_train_file / load_scene / load_scene_known / SceneDataset / collate` — the reference's names and
    return values, host numpy/torch, for code that wants the reference's loader API.  Parsing is a section
    table from the native `sgnn_io_layout` (bounds-checked, no copies) plus `numpy.frombuffer` views, instead
    of `struct.unpack` of every scalar (data_util.py:74-77 unpacks 3·N python ints per block).
  * `DeviceBatchLoader` — the MI355X path: file images are packed section-by-section into one pinned
    staging buffer, cross PCIe in ONE copy, and are decoded on the GPU (value/voxelsize, |sdf| < truncation
    compaction, xyz -> [z,y,x,b] int64 rows, sparse -> dense target volumes) straight into the collated layout
    the training step consumes.  Dense fp32 volumes never cross PCIe; a reader thread keeps the next batch's
    staging buffer full while the current step runs.

File layout (little endian; writer: datagen's VoxelGrid.h:120-159,199-218 as summarised in SURVEY.md §2):
  header   u64 dimx, dimy, dimz; f32 voxelsize; f32[16] world2grid (row major)
  block    u64 n; u32[n][3] (x,y,z); f32[n] sdf in metres
  .sdf     header, block
  .knw     header, u8[dimz][dimy][dimx]
  .sdfs    header, block input, block target, u64 (= dimx*dimy*dimz), u8 known volume,
           block hierarchy 1/2, block hierarchy 1/4, block hierarchy 1/8
"""
import os
import queue
import threading

import numpy as np
import torch

from . import _lib

KIND_CHUNK, KIND_SCENE, KIND_KNOWN = 0, 1, 2
HEADER_BYTES = 92


class Layout(object):
    """Section table of one file image (see include/sgnn_hip.h, sgnn_io_layout)."""

    def __init__(self, buf, kind):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        out = np.empty(24, dtype=np.int64)
        rc = _lib.query('sgnn_io_layout', buf.ctypes.data, buf.size, kind, out.ctypes.data)
        if rc != 0:
            raise _lib.SgnnError('sgnn_io_layout failed (%d): %s' % (rc, _lib.load().sgnn_last_error().decode()))
        self.buf, self.kind, self.t = buf, kind, out
        self.dimx, self.dimy, self.dimz = int(out[0]), int(out[1]), int(out[2])
        self.voxelsize = np.array([out[3]], dtype=np.int64).astype(np.uint32).view(np.float32)[0]
        self.world2grid = np.frombuffer(buf, '<f4', 16, int(out[4])).reshape(4, 4).copy()

    def block(self, which):
        """(locs_xyz uint32 (n,3) view, values float32 (n,) view); which: 0 input/scene, 1 target, 2.. hierarchy."""
        base = {0: 5, 1: 8}.get(which, 12 + 3 * (which - 2))
        n, ol, ov = (int(v) for v in self.t[base:base + 3])
        return (np.frombuffer(self.buf, '<u4', 3 * n, ol).reshape(n, 3), np.frombuffer(self.buf, '<f4', n, ov))

    def block_bytes(self, which):
        """(count, raw bytes of the (x,y,z) triples, raw bytes of the values) — uint8 views, memcpy-able."""
        base = {0: 5, 1: 8}.get(which, 12 + 3 * (which - 2))
        n, ol, ov = (int(v) for v in self.t[base:base + 3])
        return n, self.buf[ol:ol + 12 * n], self.buf[ov:ov + 4 * n]

    def known_bytes(self):
        o = int(self.t[11])
        return self.buf[o:o + self.dimx * self.dimy * self.dimz]

    def known(self):
        return np.frombuffer(self.buf, np.uint8, self.dimx * self.dimy * self.dimz, int(self.t[11])).reshape(
            self.dimz, self.dimy, self.dimx)


def _read(path):
    return np.fromfile(path, dtype=np.uint8)      # (a page-cache mapping was measured slower: 4 KiB faults)


_scratch = threading.local()


def _read_many(paths):
    """File images of `paths` as slices of one per-thread scratch array that is reused from call to call (fresh
    allocations cost a page fault per 4 KiB; the slices are only valid until the thread's next call)."""
    sizes = [os.path.getsize(p) for p in paths]
    total = sum(_align(n, 64) for n in sizes)
    buf = getattr(_scratch, 'buf', None)
    if buf is None or buf.size < total:
        buf = _scratch.buf = np.empty(total + total // 4, dtype=np.uint8)
    out, off = [], 0
    for p, n in zip(paths, sizes):
        view = buf[off:off + n]
        with open(p, 'rb', buffering=0) as f:
            got = f.readinto(memoryview(view))
        if got != n:
            raise IOError('short read on %s (%d of %d bytes)' % (p, got, n))
        out.append(view)
        off += _align(n, 64)
    return out


def _align(v, a=256):
    return (v + a - 1) // a * a


def _zyx(locs_xyz):
    return np.ascontiguousarray(locs_xyz[:, ::-1]).astype(np.int32)       # data_util.py:77 flip to z,y,x


def sparse_to_dense_np(locs, values, dimx, dimy, dimz, default_val):
    """data_util.py:44-56."""
    nf = 1 if values.ndim == 1 else values.shape[1]
    dense = np.full((dimz, dimy, dimx, nf), default_val, dtype=values.dtype)
    dense[locs[:, 0], locs[:, 1], locs[:, 2], :] = values.reshape(len(locs), nf)
    return dense if nf > 1 else dense.reshape(dimz, dimy, dimx)


def load_train_file(file):
    """data_util.py:63-117: ([input_locs, input_sdfs], target_sdf dense, [dimz,dimy,dimx], world2grid, known,
    hierarchy [1/8, 1/4, 1/2])."""
    lay = Layout(_read(file), KIND_CHUNK)
    vs = lay.voxelsize
    il, iv = lay.block(0)
    tl, tv = lay.block(1)
    target = sparse_to_dense_np(_zyx(tl), (tv / vs)[:, None], lay.dimx, lay.dimy, lay.dimz, -float('inf'))
    hierarchy, factor = [], 2
    for h in range(3):
        hl, hv = lay.block(2 + h)
        hierarchy.append(sparse_to_dense_np(_zyx(hl), (hv / vs)[:, None], lay.dimx // factor, lay.dimy // factor,
                                            lay.dimz // factor, -float('inf')))
        factor *= 2
    hierarchy.reverse()
    return ([_zyx(il), iv / vs], target, [lay.dimz, lay.dimy, lay.dimx], lay.world2grid, lay.known().copy(),
            hierarchy)


def load_scene(file):
    """data_util.py:121-139."""
    lay = Layout(_read(file), KIND_SCENE)
    locs, vals = lay.block(0)
    return [_zyx(locs), vals / lay.voxelsize], [lay.dimz, lay.dimy, lay.dimx], lay.world2grid


def load_scene_known(file):
    """data_util.py:142-155."""
    return Layout(_read(file), KIND_KNOWN).known().copy()


# ---------------------------------------------------------------------------------------------------------
# writers (synthetic data sets, tests): the inverse of the readers above
# ---------------------------------------------------------------------------------------------------------
def _header_bytes(dims_zyx, voxelsize, world2grid):
    dz, dy, dx = (int(d) for d in dims_zyx)
    return (np.array([dx, dy, dz], dtype='<u8').tobytes() + np.float32(voxelsize).tobytes() +
            np.asarray(world2grid, dtype='<f4').reshape(16).tobytes())


def _block_bytes(locs_zyx, vals_metric):
    locs = np.asarray(locs_zyx).reshape(-1, 3)
    xyz = np.ascontiguousarray(locs[:, ::-1]).astype('<u4')
    return (np.array([len(xyz)], dtype='<u8').tobytes() + xyz.tobytes() +
            np.asarray(vals_metric, dtype='<f4').reshape(-1).tobytes())


def write_train_file(path, dims_zyx, voxelsize, world2grid, input_block, target_block, known, hierarchy_blocks):
    """blocks: (locs (n,3) z,y,x ; sdf in metres).  hierarchy_blocks: factor 2, 4, 8 in that order."""
    known = np.ascontiguousarray(known, dtype=np.uint8)
    assert known.shape == tuple(int(d) for d in dims_zyx) and len(hierarchy_blocks) == 3
    with open(path, 'wb') as f:
        f.write(_header_bytes(dims_zyx, voxelsize, world2grid))
        f.write(_block_bytes(*input_block))
        f.write(_block_bytes(*target_block))
        f.write(np.array([known.size], dtype='<u8').tobytes())
        f.write(known.tobytes())
        for blk in hierarchy_blocks:
            f.write(_block_bytes(*blk))


def write_scene(path, dims_zyx, voxelsize, world2grid, block):
    with open(path, 'wb') as f:
        f.write(_header_bytes(dims_zyx, voxelsize, world2grid))
        f.write(_block_bytes(*block))


def write_known(path, dims_zyx, voxelsize, world2grid, known):
    known = np.ascontiguousarray(known, dtype=np.uint8)
    assert known.shape == tuple(int(d) for d in dims_zyx)
    with open(path, 'wb') as f:
        f.write(_header_bytes(dims_zyx, voxelsize, world2grid))
        f.write(known.tobytes())


# ---------------------------------------------------------------------------------------------------------
# reference-shaped host loader
# ---------------------------------------------------------------------------------------------------------
def _padded_dims(dims, num_hierarchy_levels, max_input_height, up_axis=0):
    """scene_dataloader.py:80-87: clamp the up axis, round every axis up to a multiple of 4 * 2^(levels-1)."""
    q = 4 * (2 ** (num_hierarchy_levels - 1))
    d = np.array(dims, dtype=np.int64)
    if max_input_height > 0 and d[up_axis] > max_input_height:
        d[up_axis] = max_input_height
    return ((d + q - 1) // q) * q


def collate(batch):
    """scene_dataloader.py:13-36."""
    locs = torch.cat([torch.cat([x['input'][0], torch.full((x['input'][0].shape[0], 1), b, dtype=torch.long)], 1)
                      for b, x in enumerate(batch)])
    feats = torch.cat([x['input'][1] for x in batch])
    known = torch.stack([x['known'] for x in batch]) if batch[0]['known'] is not None else None
    hierarchy = None
    if batch[0]['hierarchy'] is not None:
        hierarchy = [torch.stack([x['hierarchy'][h] for x in batch]) for h in range(len(batch[0]['hierarchy']))]
    return {'name': [x['name'] for x in batch], 'input': [locs, feats],
            'sdf': torch.stack([x['sdf'] for x in batch]),
            'world2grid': torch.stack([x['world2grid'] for x in batch]), 'known': known, 'hierarchy': hierarchy,
            'orig_dims': torch.stack([x['orig_dims'] for x in batch])}


class SceneDataset(torch.utils.data.Dataset):
    """scene_dataloader.py:39-116 (chunk mode when target_path == '', whole-scene mode otherwise)."""

    def __init__(self, files, input_dim, truncation, num_hierarchy_levels, max_input_height, num_overfit=0,
                 target_path=''):
        assert num_hierarchy_levels <= 4
        self.is_chunks = target_path == ''
        if self.is_chunks:
            self.files = [f for f in files if os.path.isfile(f)]
        else:
            self.files = [(f, os.path.join(target_path, os.path.basename(f))) for f in files
                          if os.path.isfile(f) and os.path.isfile(os.path.join(target_path, os.path.basename(f)))]
        self.input_dim, self.truncation = input_dim, truncation
        self.num_hierarchy_levels, self.max_input_height = num_hierarchy_levels, max_input_height
        self.UP_AXIS = 0
        if num_overfit > 0:
            self.files = self.files * max(1, num_overfit // len(self.files))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        file = self.files[idx]
        if self.is_chunks:
            name = os.path.splitext(os.path.basename(file))[0]
            inputs, targets, dims, world2grid, known, hierarchy = load_train_file(file)
        else:
            name = os.path.splitext(os.path.basename(file[0]))[0]
            inputs, dims, world2grid = load_scene(file[0])
            tgt, dims, world2grid = load_scene(file[1])
            known = load_scene_known(os.path.splitext(file[1])[0] + '.knw')
            targets = sparse_to_dense_np(tgt[0], tgt[1][:, None], dims[2], dims[1], dims[0], -float('inf'))
            hierarchy = None
        orig_dims = torch.LongTensor(targets.shape)
        if not self.is_chunks:
            h = self.max_input_height
            pd = _padded_dims(targets.shape, self.num_hierarchy_levels, h, self.UP_AXIS)
            if h > 0 and targets.shape[self.UP_AXIS] > h:
                keep = inputs[0][:, self.UP_AXIS] < h
                inputs = [inputs[0][keep], inputs[1][keep]]
            padded = np.full(tuple(pd), -float('inf'), dtype=np.float32)
            padded[:min(h, targets.shape[0]), :targets.shape[1], :targets.shape[2]] = targets[:h]
            targets = padded
            kpad = np.full(tuple(pd), 255, dtype=np.uint8)
            kpad[:min(h, known.shape[0]), :known.shape[1], :known.shape[2]] = known[:h]
            known = kpad
        elif self.num_hierarchy_levels < 4:
            hierarchy = hierarchy[4 - self.num_hierarchy_levels:]
        mask = np.abs(inputs[1]) < self.truncation
        sample_in = [torch.from_numpy(inputs[0][mask]).long(), torch.from_numpy(inputs[1][mask][:, None]).float()]
        if hierarchy is not None:
            hierarchy = [torch.from_numpy(g[None]) for g in hierarchy]
        return {'name': name, 'input': sample_in, 'sdf': torch.from_numpy(targets[None]),
                'world2grid': torch.from_numpy(world2grid), 'known': torch.from_numpy(known[None]),
                'hierarchy': hierarchy, 'orig_dims': orig_dims}


# ---------------------------------------------------------------------------------------------------------
# device batch loader
# ---------------------------------------------------------------------------------------------------------
class DeviceBatchLoader(object):
    """Iterates device-resident batches in collate layout, decoded on the GPU from packed file images.

    chunk mode  (target_path == ''): `batch_size` .sdfs files per batch, as SceneDataset + collate.
    scene mode  (target_path given): one (.sdf input, .sdf target, .knw) triple per batch with the reference's
                padding (scene_dataloader.py:80-98); batch_size must be 1 (scenes differ in size).
    """

    def __init__(self, files, batch_size, truncation, num_hierarchy_levels=4, max_input_height=0, target_path='',
                 device=None, prefetch=3, workers=3, drop_last=True):
        _lib.require_gpu()
        self.is_chunks = target_path == ''
        if self.is_chunks:
            self.files = [f for f in files if os.path.isfile(f)]
        else:
            assert batch_size == 1, 'whole scenes are loaded one per batch'
            self.files = [(f, os.path.join(target_path, os.path.basename(f))) for f in files
                          if os.path.isfile(f) and os.path.isfile(os.path.join(target_path, os.path.basename(f)))]
        self.batch_size, self.truncation = batch_size, float(truncation)
        self.levels, self.max_input_height = num_hierarchy_levels, max_input_height
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.prefetch, self.workers, self.drop_last = prefetch, workers, drop_last
        self._ws = None
        self._pinned = queue.Queue()          # staging buffers handed back by the consumer

    def __len__(self):
        n = len(self.files)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # ---- host side: read + pack (runs in the reader thread) ----
    def _stage(self, group):
        """group: list of files of one batch -> (pinned uint8 tensor, plan dict)."""
        lays, names = [], []
        if self.is_chunks:
            for f, img in zip(group, _read_many(group)):
                lays.append([Layout(img, KIND_CHUNK)])
                names.append(os.path.splitext(os.path.basename(f))[0])
        else:
            for f in group:
                imgs = _read_many([f[0], f[1], os.path.splitext(f[1])[0] + '.knw'])
                lays.append([Layout(imgs[0], KIND_SCENE), Layout(imgs[1], KIND_SCENE), Layout(imgs[2], KIND_KNOWN)])
                names.append(os.path.splitext(os.path.basename(f[0]))[0])
        nb = len(group)
        # block lists: name -> per-sample (layout, block index)
        if self.is_chunks:
            blocks = {'input': [(l[0], 0) for l in lays], 'target': [(l[0], 1) for l in lays]}
            for h in range(3):
                blocks['hier%d' % h] = [(l[0], 2 + h) for l in lays]
            known_src = [l[0] for l in lays]
            ref = lays[0][0]
            for l in lays:
                if (l[0].dimx, l[0].dimy, l[0].dimz) != (ref.dimx, ref.dimy, ref.dimz):
                    raise ValueError('chunks of one batch must share their dimensions')
        else:
            blocks = {'input': [(l[0], 0) for l in lays], 'target': [(l[1], 0) for l in lays]}
            known_src = [l[2] for l in lays]
            ref = lays[0][1]
        plan, off = {'blocks': {}, 'nb': nb, 'names': names}, 0
        copies = []                                                  # (dst offset, source uint8 view)
        for key, lst in blocks.items():
            parts = [l.block_bytes(b) for l, b in lst]
            counts = [c for c, _, _ in parts]
            seg = np.zeros(nb + 1, dtype=np.int64)
            seg[1:] = np.cumsum(counts)
            n = int(seg[-1])
            o_locs, o_vals, o_seg = off, _align(off + 12 * n), 0
            o_seg = _align(o_vals + 4 * n)
            off = _align(o_seg + 8 * (nb + 1))
            for s, (_, lo, va) in enumerate(parts):
                copies.append((o_locs + 12 * int(seg[s]), lo))
                copies.append((o_vals + 4 * int(seg[s]), va))
            copies.append((o_seg, seg.view(np.uint8)))
            plan['blocks'][key] = (n, o_locs, o_vals, o_seg)
        vol = ref.dimx * ref.dimy * ref.dimz
        plan['known'] = (off, vol)
        for s, l in enumerate(known_src):
            copies.append((off + s * vol, l.known_bytes()))
        off = _align(off + nb * vol)
        vs = np.array([l[0].voxelsize for l in lays], dtype=np.float32)
        w2g = np.stack([l[0].world2grid for l in lays]).astype(np.float32)
        plan['voxelsize'], plan['world2grid'] = off, _align(off + 4 * nb)
        copies.append((plan['voxelsize'], vs.view(np.uint8)))
        copies.append((plan['world2grid'], w2g.reshape(-1).view(np.uint8)))
        off = _align(plan['world2grid'] + 64 * nb)
        plan['dims'] = (ref.dimz, ref.dimy, ref.dimx)
        plan['in_dims'] = (lays[0][0].dimz, lays[0][0].dimy, lays[0][0].dimx)
        staging = self._staging_buffer(off)
        plan['bytes'] = off
        host = staging.numpy()
        for dst, src in copies:
            host[dst:dst + src.size] = src
        return staging, plan

    def _staging_buffer(self, nbytes):
        staging = None
        try:
            staging = self._pinned.get_nowait()
        except queue.Empty:
            pass
        if staging is None or staging.numel() < nbytes:
            staging = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, pin_memory=True)
        return staging

    # ---- device side ----
    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._ws

    def _decode(self, staging, plan):
        dev = self.device
        raw = staging[:plan['bytes']].to(dev, non_blocking=True)
        nb = plan['nb']

        def sect(off, nbytes, dtype):
            return raw[off:off + nbytes].view(dtype)

        vs = sect(plan['voxelsize'], 4 * nb, torch.float32)
        w2g = sect(plan['world2grid'], 64 * nb, torch.float32).view(nb, 4, 4).clone()
        dz, dy, dx = plan['dims']
        chunk = self.is_chunks
        h = self.max_input_height
        nolimit = 1 << 40
        if chunk:
            out_dims, in_limit, tgt_limit = (dz, dy, dx), nolimit, nolimit
        else:                       # scene_dataloader.py:80-96, expression for expression (h == 0 copies nothing)
            out_dims = tuple(int(v) for v in _padded_dims((dz, dy, dx), self.levels, h))
            in_limit = h if (h > 0 and dz > h) else nolimit
            tgt_limit = max(0, min(h, dz))

        # sparse input: flag -> stable compaction -> rows
        n, o_l, o_v, o_s = plan['blocks']['input']
        p_l, p_v, p_s = raw[o_l:].data_ptr(), raw[o_v:].data_ptr(), raw[o_s:].data_ptr()
        mask = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        sel = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.call('sgnn_io_flag_entries', p_l, p_v, vs.data_ptr(), p_s, nb, n, self.truncation, in_limit,
                  mask.data_ptr())
        wsb = _lib.query('sgnn_compact_ws_bytes', n)
        ws = self._workspace(wsb)
        _lib.call('sgnn_compact_mask', mask.data_ptr(), n, sel.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel())
        locs = torch.empty((n, 4), dtype=torch.int64, device=dev)
        feats = torch.empty((n, 1), dtype=torch.float32, device=dev)
        _lib.call('sgnn_io_emit_entries', p_l, p_v, vs.data_ptr(), p_s, nb, sel.data_ptr(), count.data_ptr(), n,
                  locs.data_ptr(), feats.data_ptr())
        m = int(count.item())                                          # the one read-back of the decode
        locs, feats = locs[:m], feats[:m]

        def dense_of(key, dims):
            n_, ol, ov, os_ = plan['blocks'][key]
            vol = torch.full((nb, 1) + tuple(dims), -float('inf'), dtype=torch.float32, device=dev)
            _lib.call('sgnn_io_scatter_dense', raw[ol:].data_ptr(), raw[ov:].data_ptr(), vs.data_ptr(),
                      raw[os_:].data_ptr(), nb, n_, dims[0], dims[1], dims[2], tgt_limit, vol.data_ptr())
            return vol

        sdf = dense_of('target', out_dims)
        hierarchy = None
        if chunk:
            hierarchy = [dense_of('hier%d' % k, (dz >> (k + 1), dy >> (k + 1), dx >> (k + 1))) for k in range(3)]
            hierarchy.reverse()                                        # data_util.py:116 -> [1/8, 1/4, 1/2]
            if self.levels < 4:
                hierarchy = hierarchy[4 - self.levels:]                # scene_dataloader.py:98-99
        ko, vol = plan['known']
        known = raw[ko:ko + nb * vol].view(nb, 1, dz, dy, dx)
        if not chunk:
            kpad = torch.full((nb, 1) + out_dims, 255, dtype=torch.uint8, device=dev)
            kpad[:, :, :tgt_limit, :dy, :dx] = known[:, :, :tgt_limit]
            known = kpad
        else:
            known = known.clone()
        orig = torch.tensor([[dz, dy, dx]] * nb, dtype=torch.long)
        return {'name': plan['names'], 'input': [locs, feats], 'sdf': sdf, 'world2grid': w2g, 'known': known,
                'hierarchy': hierarchy, 'orig_dims': orig}

    def _groups(self):
        bs = self.batch_size
        for i in range(0, len(self.files), bs):
            g = self.files[i:i + bs]
            if len(g) < bs and self.drop_last:
                break
            yield g

    def __iter__(self):
        """Batches in file order.  `workers` threads read and pack up to `prefetch` batches ahead (numpy's file
        reads and copies release the GIL); decode runs on the caller's thread and current stream."""
        import collections
        from concurrent.futures import ThreadPoolExecutor
        groups = self._groups()
        pending = collections.deque()
        pool = ThreadPoolExecutor(max_workers=max(1, self.workers))

        def submit():
            g = next(groups, None)
            if g is not None:
                pending.append(pool.submit(self._stage, g))

        try:
            for _ in range(max(1, self.prefetch)):
                submit()
            while pending:
                staging, plan = pending.popleft().result()   # re-raises a reader's exception here
                submit()
                batch = self._decode(staging, plan)          # ends with a stream sync (count read-back):
                self._pinned.put(staging)                    # ... so the staging buffer is free again
                yield batch
        finally:
            for f in pending:
                f.cancel()
            pool.shutdown(wait=True)
