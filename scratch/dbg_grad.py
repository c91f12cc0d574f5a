import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np, types
from common import *
from egne_amd import synth
from egne_amd.utils import calc_edge
DEV='cuda:0'
bd = bdcn_module().to(DEV)
for name in ["esf_edge_b2", "esf_edge_b2_absent1", "esf_edge_b2_absent_all"]:
    cfg, variant, kw = ESF_CASES[name]; kw=dict(kw)
    g = gold(name)
    b = synth.make_batch(kw.pop("B"), **kw)
    edge = calc_edge(types.SimpleNamespace(prec=torch.float32, edge_thres=0), b["img"].to(DEV), bd, DEV)
    m = esf_module(cfg, variant).to(DEV).train()
    args = [a.to(DEV) if torch.is_tensor(a) else a for a in batch_args(b, edge)]
    out = m(*args); out[3].sum().backward(); torch.cuda.synchronize()
    params = dict(m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    got = np.array([params[n].grad.double().norm().item() for n in names]); ref = g["grad_l2"]
    rel = np.abs(got-ref)/np.maximum(ref,1e-12)
    order = np.argsort(-rel)[:6]
    print(name, 'loss', out[3].item(), g['t_loss'])
    for i in order: print('   L2 %-34s got %.5e ref %.5e rel %.2e' % (names[i], got[i], ref[i], rel[i]))
    for k in ("elReg.l2.weight", "dec.final.conv2.weight", "enc.head.conv1.weight", "enc.down_block1.conv21.weight", "dec.up_block4.conv11.bias"):
        r = g["grad::"+k]; e = np.abs(params[k].grad.cpu().numpy()-r).max()
        print('   full %-34s err %.3e scale %.3e rel %.2e' % (k, e, np.abs(r).max(), e/np.abs(r).max()))
