"""CPU-only: simulated LDS bank conflicts of the tile kernel's A-fragment reads (ds_read_b128, gfx950 lane groups) for slot
assignments (hash order / sorted by row id) x row layouts (64-byte rows, 80-byte rows, XOR quarter swizzle).
  python scripts/sim_lds_banks.py [raster|children]"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import synth
locs = synth.make_batch(4, (64,)*3, cfg=2)['input'][0].numpy()
order = sys.argv[1] if len(sys.argv) > 1 else 'raster'
if order == 'children':
    par = locs[: len(locs)//4]
    offs = np.array([[dz,dy,dx,0] for dz in (0,1) for dy in (0,1) for dx in (0,1)])
    locs = (par[:,None,:]*np.array([2,2,2,1]) + offs[None]).reshape(-1,4)
n = len(locs)
key = lambda l: ((l[:,3]*256 + l[:,0]+1)*256 + l[:,1]+1)*256 + l[:,2]+1
k0 = key(locs); srt = np.argsort(k0); ks = k0[srt]
nbr = np.full((27, n), -1, np.int64)
i = 0
for dz in (-1,0,1):
    for dy in (-1,0,1):
        for dx in (-1,0,1):
            q = key(locs + np.array([dz,dy,dx,0]))
            pos = np.searchsorted(ks, q); pos[pos>=n] = n-1
            hit = ks[pos] == q
            nbr[i, hit] = srt[pos[hit]]; i += 1
print(order, 'n', n, 'valid/row', (nbr>=0).sum()/n)
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
groups += [[l+32 for l in g] for g in groups]
rng = np.random.default_rng(0)
def cycles(slot_of, cnt, tile_rows, RS, swz=None):
    # one wave: rows tile_rows (16), lanes (q, r); per k
    tot = 0; num = 0
    for k in range(27):
        ids = nbr[k, tile_rows]
        sl = np.where(ids >= 0, slot_of[np.maximum(ids,0)], cnt)
        for g in groups:
            addrs = set()
            for l in g:
                r, q = l & 15, l >> 4
                a = sl[r]*RS*4 + q*16
                if swz: a = sl[r]*RS*4 + ((q ^ swz(sl[r])) & 3)*16
                addrs.add(a)
            banks = {}
            for a in addrs:
                b = (a // 16) % 16
                banks[b] = banks.get(b, 0) + 1
            tot += max(banks.values()); num += 1
    return tot / num
res = {}
ntile = n // 128
for t in rng.choice(ntile, 40, replace=False):
    rows = np.arange(t*128, t*128+128)
    u = np.unique(nbr[:, rows]); u = u[u>=0]; cnt = len(u)
    for name, perm in (('hash', rng.permutation(cnt)), ('sorted', np.arange(cnt))):
        slot_of = {}
        so = np.zeros(n, np.int64); so[u] = perm
        for RS, swz, tag in ((16, None, 'rs16'), (20, None, 'rs20'), (16, (lambda s: s >> 2), 'rs16x')):
            c = np.mean([cycles(so, cnt, rows[w*32+m*16: w*32+m*16+16], RS, swz) for w in range(2) for m in range(2)])
            res.setdefault((name, tag), []).append(c)
for k, v in res.items(): print(k, 'avg cycles per lane group (1 = conflict-free): %.2f' % np.mean(v))
