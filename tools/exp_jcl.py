"""JointCodebookLoss forward+backward at a realistic size: HIP-kernel module vs the same mathematics written with
torch ops (the reference's op sequence, quantization/prediction.py:38-81) on the same device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from quantization_amd import JointCodebookLoss

def torch_ops(m, pred, idx):
    N, K, H = m.num_codebooks, m.codebook_size, m.hidden_channels
    idx = idx.to(torch.int64)
    first = idx[:, :-1].clamp(min=0) + torch.arange(0, (N - 1) * K, step=K, device=idx.device)
    emb = F.embedding(first, m.codebook_embedding.weight) * (0.5 * ((H / N) ** 0.5))
    hp = m.linear1(pred)
    a = torch.relu(torch.cumsum(torch.cat((hp.unsqueeze(1), emb), dim=1), dim=1))
    lp = torch.matmul(a.transpose(0, 1), m.linear2_weight.transpose(1, 2)).transpose(0, 1)
    lp = lp + torch.matmul(pred, m.linear2b_weight.transpose(1, 2)).transpose(0, 1)
    lp = lp + m.linear2_bias
    return F.cross_entropy(lp.reshape(-1, K), idx.reshape(-1), ignore_index=-100, reduction="sum")

torch.manual_seed(0)
B, P, N, H, K = 4096, 512, 8, 512, 256
for ckpt in (True, False):
    m = JointCodebookLoss(P, N, H, K, checkpoint=ckpt).cuda()
    pred = torch.randn(B, P, device="cuda", requires_grad=True)
    idx = torch.randint(0, K, (B, N), device="cuda")
    idx[::9] = -100
    def run(fn):
        m.zero_grad(); pred.grad = None
        loss = fn(); loss.backward(); return loss
    la = run(lambda: m(pred, idx)); ga = pred.grad.clone(); gw = m.linear2_weight.grad.clone()
    lb = run(lambda: torch_ops(m, pred, idx))
    print("checkpoint", ckpt, "loss", float(la), float(lb), "rel grad diff", float((ga - pred.grad).abs().max() / pred.grad.abs().max()),
          float((gw - m.linear2_weight.grad).abs().max() / gw.abs().max()))
    for name, fn in (("hip kernels", lambda: m(pred, idx)), ("torch ops  ", lambda: torch_ops(m, pred, idx))):
        for _ in range(3): run(fn)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): run(fn)
        torch.cuda.synchronize()
        print("   ", name, "fwd+bwd ms %.3f" % ((time.perf_counter() - t) / 20 * 1e3), " peak MB %.0f" % (torch.cuda.max_memory_allocated() / 2**20))
        torch.cuda.reset_peak_memory_stats()
