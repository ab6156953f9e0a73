"""Mesh extraction on the device — counterpart of torch/marching_cubes/marching_cubes.py (marching_cubes, save_mesh)
and of the compiled module it drives, torch/marching_cubes/marching_cubes.cpp (run_marching_cubes :478-507,
export_marching_cubes :557-572, save_to_ply :509-555); SURVEY.md §8 row f4.

Same names, arguments and results as the reference (vertex order, face indices and every float bit — see
csrc/mc.hip for how), but the volume is classified by one thread per voxel instead of a single-threaded triple loop
and the 1e-5 welding runs as parallel sweeps over a device hash table instead of a std::unordered_map.
The stages are chained here; the only host round trips are the counts needed to size the next buffers.
"""
import os

import numpy as np
import torch

from . import _lib

WELD_THRESH = 0.00001          # marching_cubes.cpp:489,568 merge_close_vertices(results, 0.00001f, true)
MAX_SWEEPS = 100000


def _count(t):
    return int(t.item())


def _compact(mask, n, dev):
    sel = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = _lib.query('sgnn_compact_ws_bytes', n)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    _lib.call('sgnn_compact_mask', _lib.ptr(mask), n, _lib.ptr(sel), _lib.ptr(cnt), _lib.ptr(ws), wsb)
    return sel, _count(cnt)


def triangle_soup(tsdf, colors, isovalue, truncation, thresh):
    """run_marching_cubes_internal (:458-476): (verts (3T,3) f32, vertex colours (3T,3) u8), voxel order."""
    _lib.require_gpu()
    if not tsdf.is_cuda:
        raise _lib.SgnnError('sgnn_amd.marching_cubes runs on the GPU only (got a %s tensor)' % tsdf.device)
    assert tsdf.dim() == 3 and tsdf.dtype == torch.float32
    tsdf = tsdf.contiguous()
    dev = tsdf.device
    d0, d1, d2 = (int(v) for v in tsdf.shape)
    if colors is not None:
        assert colors.shape == (d0, d1, d2, 3) and colors.dtype == torch.uint8
        colors = colors.to(dev).contiguous()
    wsb = _lib.query('sgnn_mc_ws_bytes', d0, d1, d2)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ntri = torch.zeros(1, dtype=torch.int64, device=dev)
    args = (d0, d1, d2, float(isovalue), float(truncation), float(thresh), _lib.ptr(ws), wsb)
    _lib.call('sgnn_mc_count', _lib.ptr(tsdf), *args, _lib.ptr(ntri))
    t = _count(ntri)
    verts = torch.empty((3 * t, 3), dtype=torch.float32, device=dev)
    vcols = torch.empty((3 * t, 3), dtype=torch.uint8, device=dev)
    if t:
        _lib.call('sgnn_mc_emit', _lib.ptr(tsdf), _lib.ptr(colors), *args, _lib.ptr(verts), _lib.ptr(vcols))
    return verts, vcols


def clean_mesh(verts, vcols, thresh=WELD_THRESH):
    """merge_close_vertices(approx=True) + remove_degenerate_faces + remove_duplicate_faces (:266-456) of a
    triangle soup -> (vertices (V,3) f32, colours (V,3) u8, faces (F,3) i32)."""
    dev = verts.device
    nv = int(verts.shape[0])
    ntri = nv // 3
    cap = _lib.query('sgnn_weld_slots', nv)
    cells = torch.empty((max(nv, 1), 3), dtype=torch.int32, device=dev)
    rep = torch.empty(cap, dtype=torch.int32, device=dev)
    first = torch.empty(cap, dtype=torch.int32, device=dev)
    state = torch.empty(cap, dtype=torch.uint8, device=dev)
    _lib.call('sgnn_weld_build', _lib.ptr(verts), nv, float(thresh), _lib.ptr(cells), _lib.ptr(rep), _lib.ptr(first),
              _lib.ptr(state), cap)
    undecided = torch.zeros(1, dtype=torch.int64, device=dev)
    for sweep in range(MAX_SWEEPS):
        _lib.call('sgnn_weld_sweep', _lib.ptr(cells), _lib.ptr(rep), _lib.ptr(first), _lib.ptr(state), cap,
                  _lib.ptr(undecided))
        if sweep % 2 == 1 and _count(undecided) == 0:      # a sweep is cheap: look at the counter every other one
            break
    else:
        raise _lib.SgnnError('vertex welding did not converge')
    creator_of = torch.empty(max(nv, 1), dtype=torch.int32, device=dev)
    is_creator = torch.empty(max(nv, 1), dtype=torch.uint8, device=dev)
    _lib.call('sgnn_weld_lookup', _lib.ptr(cells), nv, _lib.ptr(rep), _lib.ptr(first), _lib.ptr(state), cap,
              _lib.ptr(creator_of), _lib.ptr(is_creator))
    sel, n_new = _compact(is_creator, nv, dev)
    newid = torch.empty(max(nv, 1), dtype=torch.int32, device=dev)
    _lib.call('sgnn_weld_number', _lib.ptr(sel), n_new, _lib.ptr(newid))
    out_v = torch.empty((n_new, 3), dtype=torch.float32, device=dev)
    out_c = torch.empty((n_new, 3), dtype=torch.uint8, device=dev)
    _lib.call('sgnn_take_rows3', _lib.ptr(verts), 4, _lib.ptr(sel), n_new, _lib.ptr(out_v))
    _lib.call('sgnn_take_rows3', _lib.ptr(vcols), 1, _lib.ptr(sel), n_new, _lib.ptr(out_c))
    fcap = _lib.query('sgnn_weld_slots', ntri)
    faces = torch.empty((max(ntri, 1), 3), dtype=torch.int32, device=dev)
    frep = torch.empty(fcap, dtype=torch.int32, device=dev)
    ffirst = torch.empty(fcap, dtype=torch.int32, device=dev)
    keep = torch.empty(max(ntri, 1), dtype=torch.uint8, device=dev)
    _lib.call('sgnn_mesh_faces', _lib.ptr(creator_of), _lib.ptr(newid), ntri, _lib.ptr(faces), _lib.ptr(frep),
              _lib.ptr(ffirst), fcap, _lib.ptr(keep))
    fsel, n_faces = _compact(keep, ntri, dev)
    out_f = torch.empty((n_faces, 3), dtype=torch.int32, device=dev)
    _lib.call('sgnn_take_rows3', _lib.ptr(faces), 4, _lib.ptr(fsel), n_faces, _lib.ptr(out_f))
    return out_v, out_c, out_f


def run_marching_cubes(tsdf, colors, isovalue, truncation, thresh):
    """marching_cubes.cpp:478-507: (vertices (V,3) f32 x,y,z; vertex colours (V,3) u8; faces (F,3) i32), on tsdf's device."""
    verts, vcols = triangle_soup(tsdf, colors, isovalue, truncation, thresh)
    return clean_mesh(verts, vcols)


def save_to_ply(filename, verts, vertcolors, indices):
    """marching_cubes.cpp:509-555, byte for byte: binary little-endian PLY, float xyz + uchar rgb, uchar-count int faces."""
    v = verts.detach().cpu().numpy().astype('<f4').reshape(-1, 3)
    c = vertcolors.detach().cpu().numpy().astype(np.uint8).reshape(-1, 3)
    f = indices.detach().cpu().numpy().astype('<i4').reshape(-1, 3)
    head = ('ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n'
            'property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face %d\n'
            'property list uchar int vertex_indices\nend_header\n' % (len(v), len(f)))
    vrec = np.zeros(len(v), dtype=[('p', '<f4', 3), ('c', 'u1', 3)])
    vrec['p'], vrec['c'] = v, c
    frec = np.zeros(len(f), dtype=[('n', 'u1'), ('i', '<i4', 3)])
    frec['n'], frec['i'] = 3, f
    with open(filename, 'wb') as fh:
        fh.write(head.encode('ascii'))
        fh.write(vrec.tobytes())
        fh.write(frec.tobytes())


def export_marching_cubes(tsdf, colors, isovalue, truncation, thresh, filename):
    """marching_cubes.cpp:557-572."""
    save_to_ply(filename, *run_marching_cubes(tsdf, colors, isovalue, truncation, thresh))


def save_mesh(verts, colors, indices, output_file):
    """marching_cubes.py:9-25 (.obj branch; the .ply branch of the reference needs `plyfile`, so .ply goes through
    save_to_ply, which is what marching_cubes() uses for .ply anyway)."""
    verts, colors, indices = (np.asarray(a.detach().cpu()) if torch.is_tensor(a) else np.asarray(a)
                              for a in (verts, colors, indices))
    if os.path.splitext(output_file)[1] != '.obj':
        return save_to_ply(output_file, torch.from_numpy(verts), torch.from_numpy(colors), torch.from_numpy(indices))
    with open(output_file, 'w') as f:
        for v, c in zip(verts, colors):
            f.write('v %f %f %f %d %d %d\n' % (v[0], v[1], v[2], c[0], c[1], c[2]))
        f.write('g foo\n')
        for ind in indices:
            f.write('f %d %d %d\n' % (ind[0] + 1, ind[1] + 1, ind[2] + 1))
        f.write('g\n')


def marching_cubes(tsdf, colors, isovalue, truncation, thresh, output_filename):
    """marching_cubes.py:27-35.  tsdf may live on the host (as the reference passes it): it is moved to the GPU."""
    if not tsdf.is_cuda:
        _lib.require_gpu()
        tsdf = tsdf.cuda()
    if os.path.splitext(output_filename)[1] == '.ply':
        export_marching_cubes(tsdf, colors, isovalue, truncation, thresh, output_filename)
    else:
        save_mesh(*run_marching_cubes(tsdf, colors, isovalue, truncation, thresh), output_filename)


def dense_from_sparse(locs, vals, dims, device=None):
    """data_util.sparse_to_dense_np(locs, vals, dims[2], dims[1], dims[0], -inf) for one volume, on the device:
    locs (N,3) z,y,x; vals (N,) or (N,1).  Later duplicates win, like numpy's fancy assignment."""
    _lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    locs = torch.as_tensor(locs).to(dev).long()
    vals = torch.as_tensor(vals).to(dev).float().reshape(-1)
    d0, d1, d2 = (int(v) for v in dims)
    dense = torch.full((d0, d1, d2), -float('inf'), dtype=torch.float32, device=dev)
    if len(locs):
        dense.view(-1)[(locs[:, 0] * d1 + locs[:, 1]) * d2 + locs[:, 2]] = vals
    return dense


def write_points_ply(points_xyz, filename):
    """data_util.visualize_points for '.ply' (data_util.py:233-236): a vertex-only PLY of float x, y, z — what
    plyfile.PlyData([PlyElement.describe(verts, 'vertex')]).write() emits (binary little endian)."""
    pts = np.ascontiguousarray(np.asarray(points_xyz, dtype='<f4').reshape(-1, 3))
    with open(filename, 'wb') as f:
        f.write(('ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n'
                 'property float z\nend_header\n' % pts.shape[0]).encode('ascii'))
        f.write(pts.tobytes())


def _np(t):
    return np.asarray(t.detach().cpu() if torch.is_tensor(t) else t)


def save_predictions(output_path, names, inputs, target_for_sdf, target_for_occs, output_sdf, output_occs, world2grids,
                     truncation, thresh=1):
    """data_util.save_predictions (data_util.py:249-284): '<name>input-mesh.ply', '<name>pred-mesh.ply', with targets
    '<name>target-mesh.ply' (the files test_scene.py:98 writes), and — when the per-level occupancies are passed, as
    train.py's visualisation path does — the point clouds '<name>target-<h>.ply' (voxel centres with target
    occupancy 1) and '<name>pred-<h>.ply' (predicted sites), scaled to the finest resolution (data_util.py:252-273)."""
    os.makedirs(output_path, exist_ok=True)
    factors = None
    if output_occs is not None:
        L = len(output_occs)
        factors = [2 ** (L - 1 - h) for h in range(L)]                   # data_util.py:252-256
    in_locs = np.asarray(inputs[0].cpu() if torch.is_tensor(inputs[0]) else inputs[0])
    in_feats = np.asarray(inputs[1].cpu() if torch.is_tensor(inputs[1]) else inputs[1])
    if target_for_sdf is None:
        o0 = output_sdf[0][0]
        o0 = np.asarray(o0.cpu() if torch.is_tensor(o0) else o0)
        dims = np.maximum(np.max(o0, 0), np.max(in_locs, 0)) + 1          # data_util.py:257, the quirk included
    else:
        dims = target_for_sdf.shape[2:]
    trunc = truncation - 0.1
    for k, name in enumerate(names):
        m = in_locs[:, -1] == k
        dense = dense_from_sparse(in_locs[m][:, :-1], in_feats[m], dims[:3])
        marching_cubes(dense, None, 0, trunc, 10, os.path.join(output_path, name + 'input-mesh.ply'))
        if output_occs is not None:
            for h in range(len(output_occs)):
                if target_for_occs is not None:
                    occ = _np(target_for_occs[h][k, 0]) == 1             # visualize_occ_as_points(occ == 1, 0.5, .., 1.5)
                    z, y, x = np.nonzero(occ)                            # z-major raster order, like the reference's loops
                    if len(z):
                        write_points_ply((np.stack([x, y, z], 1) + 0.5) * factors[h],
                                         os.path.join(output_path, '%starget-%d.ply' % (name, h)))
                    else:
                        print('warning: no valid occ points for %s' % os.path.join(output_path, '%starget-%d.ply' % (name, h)))
                if output_occs[h] is not None and output_occs[h][k] is not None:
                    locs = _np(output_occs[h][k])[:, :3]                 # visualize_sparse_locs_as_points: z,y,x -> x,y,z
                    if len(locs):
                        write_points_ply((locs[:, ::-1].astype(np.float32) + 0.5) * factors[h],
                                         os.path.join(output_path, '%spred-%d.ply' % (name, h)))
                    else:
                        print('warning: no valid occ points for %s' % os.path.join(output_path, '%spred-%d.ply' % (name, h)))
        if output_sdf[k] is not None:
            pl, pv = output_sdf[k]
            pl = np.asarray(pl.cpu() if torch.is_tensor(pl) else pl)
            dense = dense_from_sparse(pl[:, :3], pv, dims[:3])
            marching_cubes(dense, None, 0, trunc, 10, os.path.join(output_path, name + 'pred-mesh.ply'))
        if target_for_sdf is not None:
            t = target_for_sdf[k, 0]
            t = t if torch.is_tensor(t) else torch.from_numpy(np.asarray(t))
            marching_cubes(t.float(), None, 0, trunc, 10, os.path.join(output_path, name + 'target-mesh.ply'))
