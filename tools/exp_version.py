import torch
p = torch.randn(8, 8, device="cuda", requires_grad=True)
for mode in ({}, {"fused": True}, {"foreach": False}):
    opt = torch.optim.Adam([p], lr=0.1, **mode)
    v0 = p._version
    p.grad = torch.ones_like(p)
    opt.step()
    print(mode, "version", v0, "->", p._version)
v0 = p._version; p.data.mul_(2.0); print("p.data.mul_", v0, "->", p._version)
with torch.no_grad():
    v0 = p._version; p.mul_(2.0); print("no_grad p.mul_", v0, "->", p._version)
