"""Where do the f16-storage and fp32-storage plain-f16 plans of the edge network part?  Compares every trunk tensor (stored f16 against
f16(fp32 tensor * scale)) and the 11 outputs."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import egne_amd
from egne_amd import engine, synth
from common import bdcn_module
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bd = bdcn_module().to(DEV)
bd.f16_products = 1
x = torch.cat((synth.make_batch(B, seed=77)["img"],) * 3, 1).to(DEV)
engine.BIG_SPLIT_TAIL = False
res = {}
for on in (True, False):
    engine.F16_STORAGE = on
    bd._plans.clear()
    o = bd(x); torch.cuda.synchronize()
    pl = bd._last_plan
    convs = [(n, a) for (f, a, n) in pl.calls if n.startswith("vgg.conv")]
    tens = {}
    for n, a in convs:
        d = a[0]._obj
        tens[n] = (d.out, d.out_split, d.out_split_scale, d.out_pix_stride, d.Cout_store, d.B * d.Ho * d.Wo, a[2] if len(a) == 4 else a[3])
    res[on] = (o, tens, pl)
o16, t16, p16 = res[True]; o32, t32, p32 = res[False]
import ctypes
def view(ptr, n, dtype):
    # find the keep tensor that owns ptr
    for t in (p16.keep + p32.keep):
        if torch.is_tensor(t) and t.data_ptr() == ptr:
            return t
    return None
for n in t16:
    a, b = t16[n], t32[n]
    ta, tb = view(a[0], 0, 0), view(b[0], 0, 0)
    if ta is None or tb is None:
        print(n, "buffer not found"); continue
    if ta.dtype == torch.float16:
        want = (tb.float() * a[2]).half()
        diff = (ta != want).sum().item()
        print("%-14s f16 scale %g  a_scale(in) f16-plan %g fp32-plan %g  max stored %.1f  elements differing %d of %d" % (n, a[2], a[6], b[6], ta.float().abs().max().item(), diff, ta.numel()))
    else:
        print("%-14s fp32 both: equal %s  a_scale %g / %g" % (n, torch.equal(ta, tb), a[6], b[6]))
for k in range(11):
    print("out", k, torch.equal(o16[k], o32[k]), (o16[k] - o32[k]).abs().max().item())
