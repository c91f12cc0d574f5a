import sys, types; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
import torch.nn.functional as F
from common import *
from oracle import esfnet as oesf, bdcn as obdcn
from egne_amd import synth
DEV='cuda:0'
b = synth.make_batch(2, seed=1234)
bd = bdcn_module()
edge = obdcn.calc_edge(bd.state_dict(), b['img'])
m = esf_module('baseline_edge')
sd = m.state_dict()
# oracle intermediates
x = torch.cat([b['img'], edge], 0)
h = oesf.conv_block(sd, 'enc.head', x, False)
inter = {'head': h}
cur = h
for i in (1,2,3,4):
    s, cur = oesf.down_block(sd, 'enc.down_block%d'%i, cur, 2)
    inter['skip%d'%i] = s; inter['x%d'%i] = cur
s, cur = oesf.down_block(sd, 'enc.bottleneck', cur, 0)
inter['skipb']=s; inter['bott'] = cur
m = m.to(DEV).eval()
args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
with torch.no_grad():
    out = m(*args)
pl = list(m._plans.values())[0]
D = pl.dbg['D']
def piece_nchw(p, C=None):
    C = C or p.C
    return p.buf[..., p.off:p.off+C].permute(0,3,1,2).cpu()
def cmp(name, got, ref):
    print('%-10s err %.3e  scale %.3e' % (name, (got-ref).abs().max().item(), ref.abs().max().item()))
cmp('head', piece_nchw(D[0]['x']), inter['head'])
for i in range(4):
    sk = inter['skip%d'%(i+1)]
    ic = D[i]['out'].C
    cmp('out%d'%i, piece_nchw(D[i]['out']), sk[:, :ic])
    cmp('xin%d'%i, piece_nchw(D[i]['x']), sk[:, ic:])
    cmp('xnext%d'%i, piece_nchw(D[i+1]['x']), inter['x%d'%(i+1)])
cmp('bott', pl.dbg['bott'][..., :153].permute(0,3,1,2).cpu(), inter['bott'])
# decoder
B=2
xb = torch.cat([inter['bott'][:B], inter['bott'][B:]], 1)
hcur = xb
for k, si in zip((4,3,2,1), (4,3,2,1)):
    sk = inter['skip%d'%si][:B]
    hcur = oesf.up_block(sd, 'dec.up_block%d'%k, [sk], hcur)
    cmp('up%d'%k, piece_nchw(pl.dbg['dec.up%d'%k]), hcur)
opr = oesf.conv_block(sd, 'dec.final', hcur, False)
cmp('op', out[0].cpu(), opr)
