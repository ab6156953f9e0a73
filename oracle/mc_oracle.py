"""TEST INFRASTRUCTURE — numpy/python restatement of the reference's marching cubes + mesh clean-up
(SURVEY.md §8 row f4).  Only tests/ may import this.  Small volumes only (python loops over triangles).

Follows torch/marching_cubes/marching_cubes.cpp:
  corner values      trilerp :107-131 with get_voxel :66-92 (weights 0.5^3, the reference's accumulation order)
  cube test          extract_isosurface_at_position :159-262 (cube index, pairwise jump test, |d| test, edge table)
  vertex_interp      vertexInterp :133-157
  weld               merge_close_vertices(approx=true) :359-456 with hasNearestNeighborApprox :342-356
  clean faces        remove_degenerate_faces :298-321, remove_duplicate_faces :266-297

Parity status: PINNED — tests/test_oracle_mc.py holds it to tests/golden/mc_expected.npz, which the reference's own
extension (oracle/_ref/marching_cubes_cpp.so, built from /root/reference by oracle/Makefile) produced; where that
module is present the test also re-runs it live.
The triangle table is Bourke's public table in the packed form of sgnn_amd/csrc/mc_table.h (parsed from that file).
"""
import os
import re

import numpy as np

f32 = np.float32
_EA = [2, 4, 1, 0, 5, 7, 6, 3, 2, 4, 1, 0]      # cube edge -> (corner a, corner b) in distArray order (:226-237)
_EB = [4, 1, 0, 2, 7, 6, 3, 5, 5, 7, 6, 3]
_CORNER = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (1, 1, 1)]   # x,y,z side
_BITS = [(2, 1), (4, 2), (1, 4), (0, 8), (5, 16), (7, 32), (6, 64), (3, 128)]                       # :198-205


def _table():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sgnn_amd', 'csrc', 'mc_table.h')
    words = [int(w, 16) for w in re.findall(r'0x([0-9a-f]{16})ull', open(path).read())]
    assert len(words) == 256
    tri = []
    for w in words:
        e = [(w >> (4 * i)) & 0xF for i in range(16)]
        tri.append([v for v in e if v != 0xF])
    return tri


TRI = _table()


def vertex_interp(iso, p1, p2, d1, d2):
    if abs(f32(iso - d1)) < f32(0.00001):
        return p1
    if abs(f32(iso - d2)) < f32(0.00001):
        return p2
    if abs(f32(d1 - d2)) < f32(0.00001):
        return p1
    mu = f32(f32(iso - d1) / f32(d2 - d1))
    return tuple(f32(a + f32(mu * f32(b - a))) for a, b in zip(p1, p2))


def triangle_soup(tsdf, colors, iso, trunc, thresh):
    """-> (verts (3T,3) float32, colours (3T,3) uint8) in the reference's z,y,x voxel order."""
    tsdf = np.asarray(tsdf, dtype=f32)
    iso, trunc, thresh = f32(iso), f32(trunc), f32(thresh)
    d0, d1, d2 = tsdf.shape
    valid = (tsdf != -np.inf) & (np.abs(tsdf) < trunc)
    verts, cols = [], []
    with np.errstate(invalid='ignore', over='ignore'):
        for z in range(1, d0 - 1):
            for y in range(1, d1 - 1):
                for x in range(1, d2 - 1):
                    if not valid[z - 1:z + 2, y - 1:y + 2, x - 1:x + 2].all():
                        continue
                    v = tsdf[z - 1:z + 2, y - 1:y + 2, x - 1:x + 2]
                    dist = []
                    for sx, sy, sz in _CORNER:
                        d = f32(0)
                        for ox, oy, oz in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (1, 1, 1)):
                            d = f32(d + f32(f32(0.125) * v[sz + oz, sy + oy, sx + ox]))
                        dist.append(d)
                    ci = sum(bit for c, bit in _BITS if dist[c] < iso)
                    bad = False
                    for a in dist:
                        for b in dist:
                            if f32(a * b) < 0:
                                bad = bad or f32(abs(a) + abs(b)) > thresh
                            else:
                                bad = bad or abs(f32(a - b)) > thresh
                    if bad or any(abs(a) > thresh for a in dist):
                        continue
                    edges = TRI[ci]
                    mask = 0
                    for e in edges:
                        mask |= 1 << e
                    if mask == 0 or mask == 255:
                        continue
                    pos = [tuple(f32(p + (0.5 if s else -0.5)) for p, s in zip((x, y, z), c)) for c in _CORNER]
                    col = (220, 220, 220) if colors is None else tuple(colors[z, y, x])
                    for e in edges:
                        verts.append(vertex_interp(iso, pos[_EA[e]], pos[_EB[e]], dist[_EA[e]], dist[_EB[e]]))
                        cols.append(col)
    return np.array(verts, dtype=f32).reshape(-1, 3), np.array(cols, dtype=np.uint8).reshape(-1, 3)


def clean_mesh(verts, cols, thresh=f32(0.00001)):
    grid, lookup, new_v, new_c = {}, [], [], []
    with np.errstate(invalid='ignore'):
        cells = (verts / f32(thresh) + f32(0.5) * np.sign(verts).astype(f32)).astype(f32).astype(np.int64)
    for i, c in enumerate(map(tuple, cells)):
        nn = -1
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    if nn < 0:
                        nn = grid.get((c[0] + dx, c[1] + dy, c[2] + dz), -1)
        if nn < 0:
            grid[c] = nn = len(new_v)
            new_v.append(verts[i])
            new_c.append(cols[i])
        lookup.append(nn)
    faces = np.array(lookup, dtype=np.int32).reshape(-1, 3)
    out, seen = [], set()
    for f in faces:
        if f[0] == f[1] or f[0] == f[2] or f[1] == f[2]:
            continue
        k = tuple(sorted(f))
        if k not in seen:
            seen.add(k)
            out.append(f)
    return (np.array(new_v, dtype=f32).reshape(-1, 3), np.array(new_c, dtype=np.uint8).reshape(-1, 3),
            np.array(out, dtype=np.int32).reshape(-1, 3))


def run_marching_cubes(tsdf, colors, iso, trunc, thresh):
    return clean_mesh(*triangle_soup(tsdf, colors, iso, trunc, thresh))
