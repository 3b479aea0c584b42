"""Does torch's fused=True Adam follow the default (foreach) Adam on this build?  Grad patterns of the trainer:
dense tiny gradients, rows with exactly zero gradient (weight decay only), 0-dim parameters."""
import torch
torch.manual_seed(0)
dev = "cuda"
ps = [torch.randn(16, 16, 64, device=dev) * 0.1, torch.zeros((), device=dev), torch.randn(256, 64, device=dev) * 0.1, torch.zeros(256, device=dev), torch.zeros((), device=dev)]
def grads(kind):
    gs = [torch.randn_like(p) * (1e-3 if p.ndim else 1.0) for p in ps]
    if kind == "sparse":
        gs[0][:, ::2] = 0; gs[2][::3] = 0; gs[3][::2] = 0
    if kind == "tiny":
        gs = [g * 1e-6 for g in gs]
    return gs
for kind in ("dense", "sparse", "tiny"):
    torch.manual_seed(1)
    gl = [grads(kind) for _ in range(3)]
    res = []
    for mode in ({}, {"fused": True}, {"foreach": False}):
        params = [p.clone().requires_grad_(True) for p in ps]
        opt = torch.optim.Adam(params, lr=0.005, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, **mode)
        sch = torch.optim.lr_scheduler.StepLR(opt, step_size=6 / 4, gamma=0.5)
        for it in range(3):
            for p, g in zip(params, gl[it]):
                p.grad = g.clone()
            opt.step(); opt.zero_grad(); sch.step()
        res.append([p.detach().clone() for p in params])
    for i, (a, b, c) in enumerate(zip(*res)):
        print(kind, i, tuple(a.shape), "fused-vs-foreach %.3e  single-vs-foreach %.3e  max|p| %.3f" % (float((a - b).abs().max()), float((a - c).abs().max()), float(a.abs().max())))
