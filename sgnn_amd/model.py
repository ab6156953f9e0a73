"""GenModel on the MI355X hot path — the build's counterpart of torch/model.py:276 (GenModel) and its
sub-modules (SparseEncoderLayer :21, TSDFEncoder :69, Refinement :169, SurfacePrediction :249).

Same constructor signature, same forward contract
    forward(x=[locs (N,4) long [z,y,x,b], feats (N,C) float], loss_weights)
        -> ([locs, sdf (M,1)], outputs=[[locs_unfilt_h, (occ,sdf)_h] for h in 0..L-1])
and the same module/attribute names, so a reference checkpoint's state_dict loads unchanged
(SURVEY.md App. B).  What differs is how the generative glue runs:
  * coordinates stay on the device as int32 rows; the reference's CPU nonzero/boolean-mask indexing
    (model.py:195-207, 233-247, 319-336) becomes expand8 / ballot-prefix-sum compaction kernels;
  * concat_skip's two dense int64 indicator volumes (model.py:338-355) become a hash-grid lookup on
    the encoder level's existing grid followed by one fused gather+concat;
  * the dense 8^3 bottleneck (model.py:89-136) stays on torch.nn (MIOpen/rocBLAS): a plain dense
    contraction where the vendor library is the right tool.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import scn
from .scn import functions as F_
from .scn.metadata import coords_from_locs


def _dense_block(cin, cout, k, stride, pad, transposed=False):
    conv = (nn.ConvTranspose3d if transposed else nn.Conv3d)(cin, cout, kernel_size=k, stride=stride, padding=pad,
                                                              bias=False)
    return nn.Sequential(conv, nn.BatchNorm3d(cout), nn.ReLU(True))


class SparseEncoderLayer(nn.Module):
    def __init__(self, nf_in, nf, input_sparsetensor, return_sparsetensor, max_data_size):
        nn.Module.__init__(self)
        self.nf_in, self.nf = nf_in, nf
        self.input_sparsetensor, self.return_sparsetensor = input_sparsetensor, return_sparsetensor
        self.max_data_size = max_data_size
        if not input_sparsetensor:
            self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        body = scn.Sequential()
        for _ in range(2):
            body.add(scn.BatchNormReLU(nf)).add(scn.SubmanifoldConvolution(3, nf, nf, 3, False))
        self.p2 = scn.Sequential().add(scn.ConcatTable().add(scn.Identity()).add(body)).add(scn.AddTable())
        self.p2.add(scn.BatchNormReLU(nf))
        self.p3 = scn.Sequential().add(scn.Convolution(3, nf, nf, 2, 2, False)).add(scn.BatchNormReLU(nf))
        if not return_sparsetensor:
            self.p4 = scn.SparseToDense(3, nf)

    def forward(self, x, batch_size=None):
        if not self.input_sparsetensor:
            x = self.p0(x)
        skip = self.p2(self.p1(x))
        x = self.p3(skip)
        if self.return_sparsetensor:
            return x, [skip]
        return self.p4(x, batch_size), [skip, x]


class TSDFEncoder(nn.Module):
    def __init__(self, nf_in, nf_per_level, nf_out, use_skip_sparse, use_skip_dense, input_volume_size):
        nn.Module.__init__(self)
        assert isinstance(nf_per_level, list)
        self.use_skip_sparse, self.use_skip_dense = use_skip_sparse, use_skip_dense
        layers = []
        for lv, nf in enumerate(nf_per_level):
            size = (np.array(input_volume_size) // (lv + 1)).tolist()  # model.py:79; only level 0's is read
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

    def forward(self, x, batch_size=None):
        skips = []
        for layer in self.process_sparse:
            x, ft = layer(x, batch_size)
            if self.use_skip_sparse:
                skips.extend(ft)
        enc0 = self.encode_dense0(x)
        enc1 = self.encode_dense1(enc0)
        bott = self.bottleneck_dense2(enc1)
        dec0 = self.decode_dense3(torch.cat([bott, enc1], 1) if self.use_skip_dense else bott)
        x = self.decode_dense4(torch.cat([dec0, enc0], 1) if self.use_skip_dense else dec0)
        x = self.final(x)
        # both 1x1 heads in one pass; channel 0 = occupancy logit, 1 = sdf (model.py:163-165)
        out = F.conv3d(x, torch.cat([self.occpred[0].weight, self.sdfpred[0].weight], 0))
        return x, out, skips


class Refinement(nn.Module):
    def __init__(self, nf_in, nf, pass_occ, pass_feats, max_data_size, truncation=3):
        nn.Module.__init__(self)
        self.pass_occ, self.pass_feats = pass_occ, pass_feats
        self.nf_in, self.nf, self.truncation = nf_in, nf, truncation
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
        coords = x[0]
        if len(coords) == 0:
            return [[], []], [[], []]
        f = self.p4(self.p3(self.p2(self.p1(self.p0(x)))))
        # 8-child expansion (model.py:192-207): child row 8i+j, j = 4dz+2dy+dx, features replicated
        coords_next = F_.expand8_coords(self.p0_coords(x))
        feats_next = F_.RepeatRows.apply(f, 8)
        y = self.n3(self.n2(self.n1(self.n0([coords_next, feats_next]))))
        # occupancy + sdf heads as one (nf -> 2) product; column 0 = occ logit, 1 = sdf (model.py:230-231,240)
        out = F.linear(y, torch.cat([self.linear.weight, self.linearsdf.weight], 0),
                       torch.cat([self.linear.bias, self.linearsdf.bias], 0))
        n_all = out.shape[0]
        sel, cnt = F_.compact_sigmoid(out.detach(), 2, n_all)          # stable, == torch boolean indexing order
        locs = F_.gather_coords(coords_next, sel, cnt)
        if self.pass_feats and self.pass_occ:
            feats = F_.ConcatRows.apply(y, sel, out, sel, cnt)          # [feats | occ,sdf] (model.py:242)
        elif self.pass_feats:
            feats = F_.GatherRows.apply(y, sel, cnt)
        else:
            feats = F_.GatherRows.apply(out, sel, cnt)
        return [locs, feats], [F_.coords_to_i64(coords_next), out]

    @staticmethod
    def p0_coords(x):
        return coords_from_locs(x[0], x[1].device)


class SurfacePrediction(nn.Module):
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


class GenModel(nn.Module):
    def __init__(self, encoder_dim, input_dim, input_nf, nf_coarse, nf, num_hierarchy_levels, pass_occ, pass_feats,
                 use_skip_sparse, use_skip_dense, truncation=3):
        nn.Module.__init__(self)
        self.truncation, self.pass_occ, self.pass_feats = truncation, pass_occ, pass_feats
        self.use_skip_sparse = use_skip_sparse
        L = num_hierarchy_levels
        if not isinstance(input_dim, (list, tuple, np.ndarray)):
            input_dim = [input_dim, input_dim, input_dim]
        if L > 2:
            self.nf_per_level = [int(encoder_dim * (1 + float(k) / (L - 2))) for k in range(L - 1)]
        else:
            self.nf_per_level = [encoder_dim] * (L - 1)
        self.encoder = TSDFEncoder(input_nf, self.nf_per_level, nf_coarse, use_skip_sparse, use_skip_dense,
                                   input_volume_size=input_dim)
        self.refine_sizes = [(np.array(input_dim) // (2 ** k)).tolist() for k in range(L - 1)][::-1]
        self.nf_per_level.append(self.nf_per_level[-1])
        self.data_dim = 3
        self.refinement = scn.Sequential()
        for h in range(1, L):
            c = (self.nf_per_level[L - h] if use_skip_sparse else 0) + (2 if pass_occ else 0)
            if pass_feats:
                c += nf_coarse if h == 1 else nf
            self.refinement.add(Refinement(c, nf, pass_occ, pass_feats, self.refine_sizes[h - 1], truncation))
        self.PRED_SURF = True
        c = (self.nf_per_level[0] if use_skip_sparse else 0) + (2 if pass_occ else 0) + (nf if pass_feats else 0)
        self.surfacepred = SurfacePrediction(c, nf, 1, self.refine_sizes[-1])

    # -- generative glue ---------------------------------------------------------------------------------
    def dense_coarse_to_sparse(self, coarse_feats, coarse_occ, truncation=3):
        """model.py:315-336: every coarse voxel is a candidate; keep sigmoid(occ) > 0.5 in raster order."""
        B, nf, d0, d1, d2 = coarse_feats.shape
        coords_all = F_.dense_coords(B, d0, d1, d2, coarse_feats.device)
        occ_rows = F_.DenseToSparseFn.apply(coarse_occ, coords_all)       # == permute(0,2,3,4,1).view(-1,2)
        n_all = occ_rows.shape[0]
        sel, cnt = F_.compact_sigmoid(occ_rows.detach(), 2, n_all)
        locs = F_.gather_coords(coords_all, sel, cnt)
        if self.pass_feats:
            feat_rows = F_.DenseToSparseFn.apply(coarse_feats, coords_all)
        if self.pass_occ and self.pass_feats:
            feats = F_.ConcatRows.apply(occ_rows, sel, feat_rows, sel, cnt)  # [occ,sdf | feats] (model.py:330)
        elif self.pass_occ:
            feats = F_.GatherRows.apply(occ_rows, sel, cnt)
        else:
            feats = F_.GatherRows.apply(feat_rows, sel, cnt)
        return locs, feats, [F_.coords_to_i64(coords_all), occ_rows]

    @staticmethod
    def concat_skip(x_from, x_to, spatial_size=None, batch_size=None):
        """model.py:338-355 as a hash join: x_from = (Grid of the encoder level, its features)."""
        grid_from, feats_from = x_from
        coords_to, feats_to = x_to
        if grid_from.n == 0 or len(coords_to) == 0:
            return x_to
        coords_to = coords_from_locs(coords_to, feats_to.device)
        rows = grid_from.lookup(coords_to)
        return [coords_to, F_.ConcatRows.apply(feats_to, None, feats_from, rows, coords_to.shape[0])]

    def update_sizes(self, input_max_dim, refine_max_dim):
        """model.py:357-369 (called per scene by test_scene.py:78).  Spatial sizes are upper bounds for the
        sparse layers; only the encoder input size changes results (it fixes the dense volume).  The
        reference's in-loop array doubling is not replicated: level h gets refine*2^h
        (p0) and refine*2^(h+1) (n0)."""
        inp = (np.array(input_max_dim).reshape(-1) * np.ones(3, dtype=np.int64)).astype(np.int64)
        ref = (np.array(refine_max_dim).reshape(-1) * np.ones(3, dtype=np.int64)).astype(np.int64)
        self.encoder.process_sparse[0].p0.spatial_size[:] = torch.from_numpy(inp)
        for h in range(len(self.refinement)):
            self.refinement[h].p0.spatial_size[:] = torch.from_numpy(ref * 2 ** h)
            self.refinement[h].n0.spatial_size[:] = torch.from_numpy(ref * 2 ** (h + 1))
        self.surfacepred.p0.spatial_size[:] = torch.from_numpy(ref * 2 ** len(self.refinement))

    # -- forward -----------------------------------------------------------------------------------------
    def forward(self, x, loss_weights, batch_size=None):
        outputs = []
        x = [coords_from_locs(x[0], x[1].device), x[1]]
        xd, out, skips = self.encoder(x, batch_size)
        if self.use_skip_sparse:
            skips = [(t.grid(), t.features) for t in skips]
        locs, feats, out0 = self.dense_coarse_to_sparse(xd, out, truncation=3)
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
        if self.PRED_SURF and loss_weights[-1] > 0:
            if self.use_skip_sparse:
                xs = self.concat_skip(skips[0], xs)
            sdf = self.surfacepred(xs)
            locs_out = F_.coords_to_i64(locs) if len(locs) else locs
            return [locs_out, sdf], outputs
        return [[], []], outputs
