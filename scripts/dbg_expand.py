import sys, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import scn_oracle as oscn, model_oracle as mo, sgnn_amd.scn as scn
from sgnn_amd.scn import functions as F_
from util import random_sites
torch.manual_seed(0)
cin,cout=48,16
locs = random_sites(2, 12, 0.2, 4, surface=True)
f = torch.randn(locs.shape[0], cin)
conv_o = oscn.SubmanifoldConvolution(3, cin, cout, 3, False).double()
locs_c, feats_c = mo.expand_children(locs, f.double())
yo = conv_o(oscn.InputLayer(3,[24]*3,mode=0)([locs_c, feats_c])).features.detach()
fh = f.cuda(); wh = conv_o.weight.detach().float().cuda()
grid = scn.InputLayer(3,[12]*3,mode=0)([locs.cuda(), fh]).grid()
yh = F_.expand_conv(fh, wh, grid).cpu().double()
err = (yh-yo).abs().view(-1,8,cout)
print('n', locs.shape[0], 'per parity max err', err.amax(dim=(0,2)))
print('per row-tile err', err.amax(dim=(1,2))[:20])
# emulate in torch: parent formulation
A,S,ST,PAR = F_.expand_maps(torch.device('cuda'))
wc = (A @ wh.reshape(27,-1)).view(8,8,cin,cout).cpu().double()
nbr = grid.subm_table().view(27, grid.ld)[:, :grid.n].cpu().long()
Sm = S.view(8,8).cpu().long()
ye = torch.zeros(grid.n, 8, cout, dtype=torch.float64)
fd = f.double()
for g in range(8):
    for i in range(8):
        idx = nbr[Sm[g,i]]
        ok = idx>=0
        ye[ok, g] += fd[idx[ok]] @ wc[g,i]
print('emulation vs oracle', (ye.view(-1,cout)-yo).abs().max().item(), ' kernel vs emulation', (ye.view(-1,cout)-yh).abs().max().item())
