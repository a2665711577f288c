import sys, os, math, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from ccvs_amd import ops
n,k,c,h,w=2,3,24,40,64
g = torch.Generator().manual_seed(n * 100 + c)
ctxs = [torch.randn(n, c, h, w, generator=g).cuda() for _ in range(k)]
fo = torch.randn(n * k, 3, h, w, generator=g).cuda()
fo[:, :2] *= 3.0
mult=2.0
packed = ops.backwarp_p8(ctxs, fo, mult)
warped = ops.backwarp(ctxs, fo[:, :2].contiguous(), mult)
want = torch.cat([warped, fo, fo.new_zeros(n * k, 5, h, w)], dim=1)
# exact split of want, as the staging waves do it
hi = want.to(torch.bfloat16); lo = (want - hi.float()).to(torch.bfloat16)
u = packed.data.view(torch.bfloat16).view(n*k, (c+8)//8, 2, h, w, 8)
phi = u[:, :, 0].permute(0,1,4,2,3).reshape(n*k, c+8, h, w); plo = u[:, :, 1].permute(0,1,4,2,3).reshape(n*k, c+8, h, w)
print("hi equal", torch.equal(phi, hi), "lo equal", torch.equal(plo, lo), "mismatch hi", (phi != hi).sum().item(), "lo", (plo != lo).sum().item())
wt = torch.randn(128, c + 8, 3, 3, generator=g).cuda(); wt[:, c+3:] = 0
b = torch.randn(128, generator=g).cuda(); pre = torch.randn(n, 128, h, w, generator=g).cuda()
pk = ops.pack_conv_weight(wt)
a = ops.conv2d(want, pk, b, 128, 3, pad=1, act=True, pre=pre, pre_div=k)
bb = ops.conv2d(packed, pk, b, 128, 3, pad=1, act=True, pre=pre, pre_div=k)
print("conv equal", torch.equal(a, bb), "max diff", (a-bb).abs().max().item(), "n diff", (a != bb).sum().item(), "of", a.numel())
a2 = ops.conv2d(want, pk, b, 128, 3, pad=1, act=True)
b2 = ops.conv2d(packed, pk, b, 128, 3, pad=1, act=True)
print("no pre: conv equal", torch.equal(a2, b2), (a2 != b2).sum().item())
# packed input made by an identity conv from want (the path the chain test uses)
eye = torch.eye(c+8, device="cuda").view(c+8, c+8, 1, 1) * ((c+8) ** 0.5)
p2 = ops.conv2d(want, ops.pack_conv_weight(eye), None, c+8, 1, out_p8=True)
b3 = ops.conv2d(p2, pk, b, 128, 3, pad=1, act=True)
print("identity-packed: equal to fp32", torch.equal(a2, b3), "equal to warp-packed", torch.equal(b2, b3))
d = (a2 != b2).nonzero()
print(d[:5].tolist(), a2[a2 != b2][:5].tolist(), b2[a2 != b2][:5].tolist())
