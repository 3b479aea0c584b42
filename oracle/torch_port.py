"""torch-CPU restatement of the reference's encode op sequence, used ONLY as the
`cpu_baseline` leg of bench.py (and checked against the fixtures in tests/).

TEST/BENCH INFRASTRUCTURE: never imported by quantization_amd/.

It performs the same torch operations, in the same order and with the same
materialised intermediates, as Quantizer._compute_indexes / _refine_indexes
(/root/reference/quantization/quantization.py:281-547): linear + argmax, gather of the
old centers, the cross-term matmul, full sorts with truncation, gather/sub/add of
the (B, N, K_cutoff, dim) delta tensors and the per-vector bmm of every combine --
so its throughput on the host cores stands in for the reference's CPU encode, which
cannot travel to the GPU box.  Written from SURVEY.md Appendix A, not from the
reference's source text.
"""
import torch


def k_cutoff(K: int, L: int) -> int:
    kc = 8 if K <= 16 else 16
    while L >= 4:
        L //= 4
        kc *= 2
    return min(kc, 128)


class TorchPortQuantizer:
    def __init__(self, state):
        t = lambda k: torch.as_tensor(state[k], dtype=torch.float32)
        self.centers = t("centers")
        self.N, self.K, self.D = self.centers.shape
        self.weight, self.bias = t("to_logits.weight"), t("to_logits.bias")
        self.centers_scale, self.logits_scale = t("centers_scale"), t("logits_scale")
        # the two scale factors: this host's torch exp, or the ones a fixture pins (tests/golden/fixtures.PinnedState: torch's fp32
        # exp differs in the last bit between CPUs, and the fixtures' codes belong to the generating machine's)
        pin = getattr(state, "scales_exp", None)
        self.cscale = torch.tensor(pin[0], dtype=torch.float32) if pin is not None else (self.centers_scale * 10.0).exp()
        self.lscale = torch.tensor(pin[1], dtype=torch.float32) if pin is not None else (self.logits_scale * 10.0).exp()

    def scaled_centers(self):
        return self.cscale * self.centers

    def compute_indexes(self, x: torch.Tensor, iters: int) -> torch.Tensor:
        B = x.shape[0]
        sx = self.lscale * x
        logits = torch.nn.functional.linear(sx, self.weight, self.bias).reshape(B, self.N, self.K)
        idx = logits.argmax(dim=-1)
        for _ in range(iters):
            idx = self.refine(x, idx)
        return idx

    def refine(self, x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        N, K, D = self.N, self.K, self.D
        B = x.shape[0]
        C = self.scaled_centers().unsqueeze(0)                               # (1,N,K,D)
        pick = idx.reshape(B, N, 1, 1).expand(B, N, 1, D)
        old = torch.gather(C.expand(B, N, K, D), 2, pick)                    # (B,N,1,D)
        xerr = old.sum(dim=1, keepdim=True) - x.reshape(B, 1, 1, D)          # (B,1,1,D)
        E = (xerr ** 2).sum(dim=-1)                                          # (B,1,1)
        xrem = xerr - old                                                    # (B,N,1,D)
        R = (xrem ** 2).sum(dim=-1)                                          # (B,N,1)
        Q = (C ** 2).sum(dim=-1)                                             # (1,N,K)
        cross = torch.matmul(C, xrem.permute(2, 1, 3, 0)).squeeze(0).permute(2, 0, 1)   # (B,N,K)
        score = R + Q + 2 * cross
        tuples = torch.arange(K).reshape(1, 1, K, 1).expand(B, N, K, 1)
        G, Kg, L = N, K, 1
        deltas, even, odd, prevK = None, None, None, None
        while True:
            kc = k_cutoff(K, L)
            if G == 1 and Kg == 1:
                return tuples.reshape(B, N)
            if Kg > kc or G == 1:
                keep = 1 if G == 1 else kc
                order = torch.sort(score, dim=2)[1][:, :, :keep]
                score = torch.gather(score, 2, order)
                o4 = order.unsqueeze(-1)
                tuples = torch.gather(tuples, 2, o4.expand(B, G, keep, L))
                if even is None:   # first truncation: deltas are (chosen center - old center)
                    deltas = torch.gather(C.expand(B, G, K, D), 2, o4.expand(B, G, keep, D)) - old
                else:              # later: sums of the two halves' deltas
                    deltas = (torch.gather(even, 2, (o4 // prevK).expand(B, G, keep, D)) +
                              torch.gather(odd, 2, (o4 % prevK).expand(B, G, keep, D)))
                Kg = keep
            else:
                even, odd = deltas[:, 0::2], deltas[:, 1::2]
                G2 = G // 2
                te = tuples[:, 0::2].unsqueeze(3).expand(B, G2, Kg, Kg, L).reshape(B, G2, Kg * Kg, L)
                to = tuples[:, 1::2].unsqueeze(2).expand(B, G2, Kg, Kg, L).reshape(B, G2, Kg * Kg, L)
                tuples = torch.cat((te, to), dim=3)
                se, so = score[:, 0::2].unsqueeze(3), score[:, 1::2].unsqueeze(2)
                score = ((se + so).reshape(B, G2, Kg * Kg) - E +
                         2 * torch.matmul(even, odd.transpose(2, 3)).reshape(B, G2, Kg * Kg))
                prevK = Kg
                G, Kg, L = G2, Kg * Kg, 2 * L

    def encode(self, x, iters=5, chunk=256):
        x = torch.as_tensor(x, dtype=torch.float32).reshape(-1, self.D)
        out = []
        with torch.no_grad():
            for lo in range(0, x.shape[0], chunk):
                out.append(self.compute_indexes(x[lo:lo + chunk], iters))
        idx = torch.cat(out) if out else torch.zeros((0, self.N), dtype=torch.int64)
        return idx.to(torch.uint8)
