"""CPU ORACLE — test infrastructure, NOT product code.

Restatement of the reference's model composition and loss on torch-CPU + `scn_oracle`:
  GenModel / TSDFEncoder / Refinement / SurfacePrediction   torch/model.py:21-416
  compute_targets / compute_loss and helpers                 torch/loss.py:15-199
  preprocess_sdf_pt                                          torch/data_util.py:151-154
written for torch >= 2 (the reference indexes CPU tensors with device masks, model.py:238,335).
Pinned against the REAL reference files by tests/golden/*.npz (tests/test_oracle_golden.py):
those fixtures were produced by importing /root/reference/torch/model.py and loss.py unmodified
with `scn_oracle` registered as `sparseconvnet` (tests/golden/make_golden.py).
The sparse-op arithmetic underneath is "parity unpinned" (see scn_oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import scn_oracle as scn

UNK_THRESH = 2   # loss.py:10
UNK_ID = -1      # loss.py:13


def _dense_block(cin, cout, k, stride, pad, transposed=False):
    conv = (nn.ConvTranspose3d if transposed else nn.Conv3d)(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)
    return nn.Sequential(conv, nn.BatchNorm3d(cout), nn.ReLU(True))


class SparseEncoderLayer(nn.Module):  # model.py:21-67
    def __init__(self, nf_in, nf, input_sparsetensor, return_sparsetensor, max_data_size):
        nn.Module.__init__(self)
        self.input_sparsetensor, self.return_sparsetensor = input_sparsetensor, return_sparsetensor
        if not input_sparsetensor:
            self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        res = scn.Sequential()
        for _ in range(2):
            res.add(scn.BatchNormReLU(nf)).add(scn.SubmanifoldConvolution(3, nf, nf, 3, False))
        self.p2 = scn.Sequential().add(scn.ConcatTable().add(scn.Identity()).add(res)).add(scn.AddTable())
        self.p2.add(scn.BatchNormReLU(nf))
        self.p3 = scn.Sequential().add(scn.Convolution(3, nf, nf, 2, 2, False)).add(scn.BatchNormReLU(nf))
        if not return_sparsetensor:
            self.p4 = scn.SparseToDense(3, nf)

    def forward(self, x):
        if not self.input_sparsetensor:
            x = self.p0(x)
        skip = self.p2(self.p1(x))
        x = self.p3(skip)
        if self.return_sparsetensor:
            return x, [skip]
        return self.p4(x), [skip, x]


class TSDFEncoder(nn.Module):  # model.py:69-167
    def __init__(self, nf_in, nf_per_level, nf_out, use_skip_sparse, use_skip_dense, input_volume_size):
        nn.Module.__init__(self)
        self.use_skip_sparse, self.use_skip_dense = use_skip_sparse, use_skip_dense
        layers = []
        for lv, nf in enumerate(nf_per_level):
            size = (np.array(input_volume_size) // (lv + 1)).tolist()  # model.py:79 (only level 0 is used)
            layers.append(SparseEncoderLayer(nf_in if lv == 0 else nf_per_level[lv - 1], nf, lv > 0,
                                             lv < len(nf_per_level) - 1, size))
        self.process_sparse = nn.Sequential(*layers)
        nf = nf_per_level[-1]
        nf0, nf1 = nf * 3 // 2, nf * 2
        nf2 = nf1
        self.encode_dense0 = _dense_block(nf, nf0, 4, 2, 1)
        self.encode_dense1 = _dense_block(nf0, nf1, 4, 2, 1)
        self.bottleneck_dense2 = _dense_block(nf1, nf2, 1, 1, 0)
        nf3 = nf2 if not use_skip_dense else nf1 + nf2
        nf4 = nf3 // 2
        self.decode_dense3 = _dense_block(nf3, nf4, 4, 2, 1, True)
        if use_skip_dense:
            nf4 += nf0
        nf5 = nf4 // 2
        self.decode_dense4 = _dense_block(nf4, nf5, 4, 2, 1, True)
        self.final = _dense_block(nf5, nf_out, 1, 1, 0)
        self.occpred = nn.Sequential(nn.Conv3d(nf_out, 1, kernel_size=1, bias=False))
        self.sdfpred = nn.Sequential(nn.Conv3d(nf_out, 1, kernel_size=1, bias=False))

    def forward(self, x):
        skips = []
        for layer in self.process_sparse:
            x, ft = layer(x)
            if self.use_skip_sparse:
                skips.extend(ft)
        enc0 = self.encode_dense0(x)
        enc1 = self.encode_dense1(enc0)
        bott = self.bottleneck_dense2(enc1)
        dec0 = self.decode_dense3(torch.cat([bott, enc1], 1) if self.use_skip_dense else bott)
        x = self.decode_dense4(torch.cat([dec0, enc0], 1) if self.use_skip_dense else dec0)
        x = self.final(x)
        return x, torch.cat([self.occpred(x), self.sdfpred(x)], 1), skips


# Test hooks (parity tests only): MASK_LOG, when a list, receives every generative mask in the order the forward pass
# takes them; FORCED_MASKS, when a non-empty list, supplies them instead of `sigmoid(logit) > 0.5` (consumed front to
# back).  Lets the fp64 evaluation of the model, and the HIP model through its `teacher=` volumes, walk exactly the
# site lists the fp32 oracle decided on, so that 100 % of the sites are comparable.
MASK_LOG = None
FORCED_MASKS = None


def _occupied(logit):  # model.py:233, 322
    if FORCED_MASKS:
        mask = FORCED_MASKS.pop(0)
        assert mask.shape == logit.shape
    else:
        mask = torch.sigmoid(logit) > 0.5
    if MASK_LOG is not None:
        MASK_LOG.append(mask.clone())
    return mask


def expand_children(locs, feats):  # model.py:192-207
    offs = torch.tensor([[dz, dy, dx, 0] for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)], dtype=locs.dtype)
    nxt = locs.unsqueeze(1).repeat(1, 8, 1)
    nxt[:, :, :3] *= 2
    nxt = nxt + offs.unsqueeze(0)
    return nxt.view(-1, 4), feats.unsqueeze(1).repeat(1, 8, 1).view(-1, feats.shape[-1])


class Refinement(nn.Module):  # model.py:169-247
    def __init__(self, nf_in, nf, pass_occ, pass_feats, max_data_size, truncation=3):
        nn.Module.__init__(self)
        self.pass_occ, self.pass_feats = pass_occ, pass_feats
        self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        self.p2 = scn.FullyConvolutionalNet(3, reps=1, nPlanes=[nf, nf, nf], residual_blocks=True)
        self.p3 = scn.BatchNormReLU(nf * 3)
        self.p4 = scn.OutputLayer(3)
        self.n0 = scn.InputLayer(3, max_data_size, mode=0)
        self.n1 = scn.SubmanifoldConvolution(3, nf * 3, nf, 3, False)
        self.n2 = scn.BatchNormReLU(nf)
        self.n3 = scn.OutputLayer(3)
        self.linear = nn.Linear(nf, 1)
        self.linearsdf = nn.Linear(nf, 1)

    def forward(self, x):
        locs_in = x[0]
        if len(locs_in) == 0:
            return [[], []], [[], []]
        f = self.p4(self.p3(self.p2(self.p1(self.p0(x)))))
        locs_unfilt, feats = expand_children(locs_in, f)
        y = self.n3(self.n2(self.n1(self.n0([locs_unfilt, feats]))))
        out = torch.cat([self.linear(y), self.linearsdf(y)], 1)
        mask = _occupied(out[:, 0])
        parts = ([y[mask]] if self.pass_feats else []) + ([out[mask]] if self.pass_occ else [])
        return [locs_unfilt[mask], torch.cat(parts, 1)], [locs_unfilt, out]


class SurfacePrediction(nn.Module):  # model.py:249-272
    def __init__(self, nf_in, nf, nf_out, max_data_size):
        nn.Module.__init__(self)
        self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        self.p2 = scn.FullyConvolutionalNet(3, reps=1, nPlanes=[nf, nf, nf], residual_blocks=True)
        self.p3 = scn.BatchNormReLU(nf * 3)
        self.p4 = scn.OutputLayer(3)
        self.linear = nn.Linear(nf * 3, nf_out)

    def forward(self, x):
        if len(x[0]) == 0:
            return [], []
        return self.linear(self.p4(self.p3(self.p2(self.p1(self.p0(x))))))


class GenModel(nn.Module):  # model.py:276-416
    def __init__(self, encoder_dim, input_dim, input_nf, nf_coarse, nf, num_hierarchy_levels, pass_occ, pass_feats,
                 use_skip_sparse, use_skip_dense, truncation=3):
        nn.Module.__init__(self)
        self.pass_occ, self.pass_feats, self.use_skip_sparse = pass_occ, pass_feats, use_skip_sparse
        L = num_hierarchy_levels
        if not isinstance(input_dim, (list, tuple, np.ndarray)):
            input_dim = [input_dim] * 3
        if L > 2:
            self.nf_per_level = [int(encoder_dim * (1 + float(k) / (L - 2))) for k in range(L - 1)]
        else:
            self.nf_per_level = [encoder_dim] * (L - 1)
        self.encoder = TSDFEncoder(input_nf, self.nf_per_level, nf_coarse, use_skip_sparse, use_skip_dense, input_dim)
        self.refine_sizes = [(np.array(input_dim) // (2 ** k)).tolist() for k in range(L - 1)][::-1]
        self.nf_per_level.append(self.nf_per_level[-1])
        self.refinement = scn.Sequential()
        for h in range(1, L):
            c = (self.nf_per_level[L - h] if use_skip_sparse else 0) + (2 if pass_occ else 0)
            if pass_feats:
                c += nf_coarse if h == 1 else nf
            self.refinement.add(Refinement(c, nf, pass_occ, pass_feats, self.refine_sizes[h - 1], truncation))
        c = (self.nf_per_level[0] if use_skip_sparse else 0) + (2 if pass_occ else 0) + (nf if pass_feats else 0)
        self.surfacepred = SurfacePrediction(c, nf, 1, self.refine_sizes[-1])

    def dense_coarse_to_sparse(self, coarse_feats, coarse_occ):  # model.py:315-336
        B, nf, d0, d1, d2 = coarse_feats.shape
        zz, yy, xx = torch.meshgrid(torch.arange(d0), torch.arange(d1), torch.arange(d2), indexing='ij')
        vox = torch.stack([zz, yy, xx], -1).view(1, -1, 3).repeat(B, 1, 1)
        bcol = torch.arange(B).view(B, 1, 1).repeat(1, d0 * d1 * d2, 1)
        locs_unfilt = torch.cat([vox, bcol], 2).view(-1, 4)
        occ_rows = coarse_occ.permute(0, 2, 3, 4, 1).contiguous().view(-1, 2)
        mask = _occupied(occ_rows[:, 0])
        parts = []
        if self.pass_occ:
            parts.append(occ_rows[mask])
        if self.pass_feats:
            parts.append(coarse_feats.permute(0, 2, 3, 4, 1).contiguous().view(-1, nf)[mask])
        return locs_unfilt[mask], torch.cat(parts, 1), [locs_unfilt, occ_rows]

    @staticmethod
    def concat_skip(x_from, x_to):  # model.py:338-355 (dense indicator volumes -> dictionary join)
        locs_from, locs_to = x_from[0], x_to[0]
        if len(locs_from) == 0 or len(locs_to) == 0:
            return x_to
        rows = scn.Grid(locs_from.numpy()).lookup(locs_to.numpy())
        rows_t = torch.from_numpy(rows)
        got = x_from[1].new_zeros(locs_to.shape[0], x_from[1].shape[1])
        hit = rows_t >= 0
        got = got.index_put((torch.nonzero(hit).view(-1),), x_from[1][rows_t[hit]])
        return [locs_to, torch.cat([x_to[1], got], 1)]

    def update_sizes(self, input_max_dim, refine_max_dim):
        """model.py:357-369 without its array-doubling quirk: spatial sizes are upper
        bounds for the sparse layers, so only the encoder input size changes results (it fixes the dense
        volume).  Level h works at refine*2^h (p0) and refine*2^(h+1) (n0)."""
        inp = np.array(input_max_dim).reshape(-1) * np.ones(3, dtype=np.int64)
        ref = np.array(refine_max_dim).reshape(-1) * np.ones(3, dtype=np.int64)
        self.encoder.process_sparse[0].p0.spatial_size[:] = torch.from_numpy(inp.astype(np.int64))
        for h in range(len(self.refinement)):
            self.refinement[h].p0.spatial_size[:] = torch.from_numpy((ref * 2 ** h).astype(np.int64))
            self.refinement[h].n0.spatial_size[:] = torch.from_numpy((ref * 2 ** (h + 1)).astype(np.int64))
        self.surfacepred.p0.spatial_size[:] = torch.from_numpy((ref * 2 ** len(self.refinement)).astype(np.int64))

    def forward(self, x, loss_weights):  # model.py:371-416
        outputs = []
        xd, out, skips = self.encoder(x)
        if self.use_skip_sparse:
            skips = [[t.metadata.getSpatialLocations(t.spatial_size), t.features] for t in skips]
        locs, feats, out0 = self.dense_coarse_to_sparse(xd, out)
        outputs.append(out0)
        xs = [locs, feats]
        R = len(self.refinement)
        for h in range(R):
            if loss_weights[h + 1] > 0:
                if self.use_skip_sparse:
                    xs = self.concat_skip(skips[R - h], xs)
                xs, occ = self.refinement[h](xs)
                outputs.append(occ)
            else:
                outputs.append([[], []])
        locs = xs[0]
        if loss_weights[-1] > 0:
            if self.use_skip_sparse:
                xs = self.concat_skip(skips[0], xs)
            return [locs, self.surfacepred(xs)], outputs
        return [[], []], outputs


# ---------------------------------------------------------------------------------------------
# loss (torch/loss.py)
# ---------------------------------------------------------------------------------------------
def preprocess_sdf(sdf, truncation):  # data_util.py:151-154 (in place)
    sdf[sdf < -truncation] = -truncation
    sdf[sdf > truncation] = truncation
    return sdf


def compute_targets(target, hierarchy, num_hierarchy_levels, truncation, use_loss_masking, known):  # loss.py:15-32
    L = num_hierarchy_levels
    occs, hier = [None] * L, [None] * L
    target_for_sdf = preprocess_sdf(target, truncation)
    hier[-1] = target.clone()
    occ = (torch.abs(target_for_sdf) < truncation).float()
    if use_loss_masking:
        occ[known >= UNK_THRESH] = UNK_ID
    occs[-1] = occ
    for h in range(L - 2, -1, -1):
        occs[h] = F.max_pool3d(occs[h + 1], kernel_size=2)
        hier[h] = preprocess_sdf(hierarchy[h], truncation)
    return target_for_sdf, occs, hier


def _flat(locs, dims):
    return ((locs[:, 3] * dims[0] + locs[:, 0]) * dims[1] + locs[:, 1]) * dims[2] + locs[:, 2]


def compute_weights_missing_geo(weight_missing_geo, input_locs, target_for_occs, truncation):  # loss.py:35-48
    L = len(target_for_occs)
    weights = [None] * L
    dims = target_for_occs[-1].shape[2:]
    w = torch.ones(target_for_occs[-1].shape, dtype=torch.int32)
    w.view(-1)[_flat(input_locs, dims)] += 1
    w[torch.abs(target_for_occs[-1]) <= truncation] += 3
    weights[-1] = (w == 4).float() * (weight_missing_geo - 1) + 1
    for h in range(L - 2, -1, -1):
        weights[h] = weights[h + 1][:, :, ::2, ::2, ::2].contiguous()
    return weights


def log_transform(sdf):  # loss.py:51-55
    return torch.sign(sdf) * torch.log(torch.abs(sdf) + 1)


def bce_sparse_dense(locs, vals, dense_tgts, weights, use_loss_masking):  # loss.py:58-82
    fl = _flat(locs, dense_tgts.shape[2:])
    pred, tgt = vals.view(-1), dense_tgts.view(-1)[fl]
    w = None if weights is None else weights.view(-1)[fl]
    if use_loss_masking:
        m = tgt != UNK_ID
        pred, tgt = pred[m], tgt[m]
        w = None if w is None else w[m]
    else:
        tgt = torch.where(tgt == UNK_ID, torch.zeros_like(tgt), tgt)
    return F.binary_cross_entropy_with_logits(pred, tgt, weight=w)


def l1_predsurf_sparse_dense(locs, vals, dense_tgts, weights, use_log_transform, use_loss_masking, known):  # loss.py:122-157
    fl = _flat(locs, dense_tgts.shape[2:])
    pred, tgt = vals.view(-1), dense_tgts.view(-1)[fl]
    w = None if weights is None else weights.view(-1)[fl]
    if use_loss_masking:
        m = (known < UNK_THRESH).view(-1)[fl]
        pred, tgt = pred[m], tgt[m]
        w = None if w is None else w[m]
    if use_log_transform:
        pred, tgt = log_transform(pred), log_transform(tgt)
    d = torch.abs(pred - tgt)
    return torch.mean(d * w) if w is not None else torch.mean(d)


def compute_loss(output_sdf, output_occs, target_for_sdf, target_for_occs, target_for_hier, loss_weights, truncation,
                 use_log_transform=True, weight_missing_geo=1, input_locs=None, use_loss_masking=True, known=None):
    """loss.py:160-199, batched=True branch."""
    loss, losses = 0.0, []
    weights = [None] * len(target_for_occs)
    if weight_missing_geo > 1:
        weights = compute_weights_missing_geo(weight_missing_geo, input_locs, target_for_occs, truncation)
    for h in range(len(output_occs)):
        if len(output_occs[h][0]) == 0 or loss_weights[h] == 0:
            losses.append(-1)
            continue
        locs, vals = output_occs[h]
        l_occ = bce_sparse_dense(locs, vals[:, 0], target_for_occs[h], weights[h], use_loss_masking)
        cur_known = None if not use_loss_masking else (target_for_occs[h] == UNK_ID) * UNK_THRESH
        l_sdf = l1_predsurf_sparse_dense(locs, vals[:, 1], target_for_hier[h], weights[h], use_log_transform,
                                         use_loss_masking, cur_known)
        cur = l_occ + l_sdf
        loss = loss + loss_weights[h] * cur
        losses.append(cur.item())
    if len(output_sdf[0]) > 0 and loss_weights[-1] > 0:
        cur = l1_predsurf_sparse_dense(output_sdf[0], output_sdf[1], target_for_sdf, weights[-1], use_log_transform,
                                       use_loss_masking, known)
        loss = loss + loss_weights[-1] * cur
        losses.append(cur.item())
    else:
        losses.append(-1)
    return loss, losses
