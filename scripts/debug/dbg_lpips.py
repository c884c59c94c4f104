import ctypes as C, torch, numpy as np, sys
sys.path.insert(0,'.')
from oracle import grads as OG
from maua_amd.perceptors import LPIPS, LPIPS_TAPS, _ptr_array
from maua_amd import _lib as L
gen = torch.Generator().manual_seed(11)
p = OG.init_vgg_params(OG.VGG16_CFG, 29, generator=gen)
lins = OG.init_lpips_lins(gen)
m = LPIPS(dtype=torch.float32, state_dict=p, lin_state_dict={f"lin{k}.model.1.weight": w.reshape(1, -1, 1, 1) for k, w in enumerate(lins)})
a = torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1
b = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
feats = m.embed(b)
x = m.net._check(a)
for k in range(5):
    taps = (C.c_int * 1)(m.net.op_of(LPIPS_TAPS[k]))
    grad = torch.empty_like(x); dist = torch.empty(2, device='cuda')
    L.check(L.lib().maua_vgg_lpips_grad(m.net._handle(), L.ptr(x), 2, 64, 64, taps, 1, _ptr_array([feats[k]]), (C.c_long*1)(0), _ptr_array([m._lins()[k]]), C.c_float(3.0), L.ptr(grad), L.ptr(dist)))
    with torch.enable_grad():
        xx = a.clone().requires_grad_()
        f0 = OG.vgg_features(p, OG.VGG16_CFG, OG.normalize_img(xx, OG.LPIPS_SHIFT, OG.LPIPS_SCALE), (LPIPS_TAPS[k],))[0]
        f1 = OG.vgg_features(p, OG.VGG16_CFG, OG.normalize_img(b, OG.LPIPS_SHIFT, OG.LPIPS_SCALE), (LPIPS_TAPS[k],))[0]
        d = ((OG.lpips_normalize(f0) - OG.lpips_normalize(f1))**2 * lins[k].reshape(1,-1,1,1)).sum(1).mean((1,2))
        want = torch.autograd.grad(d.sum()*3.0, xx)[0]
    diff = (grad.cpu()-want).abs()
    mx = want.abs().max()
    bad = (diff > 1e-3*mx).nonzero()
    print(k, 'rel', float(diff.max()/mx), 'nbad', len(bad), bad[:6].tolist(), 'dist', dist.tolist(), d.tolist())
# pool argmax agreement: HIP's relu1_2 vs the oracle's
m.net.forward(a)
fh = m.net.features(3).cpu()
fo = OG.vgg_features(p, OG.VGG16_CFG, OG.normalize_img(a, OG.LPIPS_SHIFT, OG.LPIPS_SCALE), (3,))[0]
print('relu1_2 rel', float((fh-fo).abs().max()/fo.abs().max()))
_, ih = torch.nn.functional.max_pool2d(fh, 2, return_indices=True)
vo, io = torch.nn.functional.max_pool2d(fo, 2, return_indices=True)
dis = (ih != io) & (vo > 0)
print('argmax disagreements (positive windows):', int(dis.sum()), dis.nonzero()[:5].tolist())
for (bb, c, y, x) in dis.nonzero()[:3].tolist():
    print('window oracle', fo[bb, c, 2*y:2*y+2, 2*x:2*x+2].flatten().tolist(), 'hip', fh[bb, c, 2*y:2*y+2, 2*x:2*x+2].flatten().tolist())
