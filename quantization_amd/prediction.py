"""MI355X-native `quantization.JointCodebookLoss` (reference: quantization/prediction.py:9-172).

Same constructor, parameter names, shapes and initialisation, so a reference state_dict loads unchanged.
The module predicts codebook n of a frame from a predictor vector and the entries chosen in codebooks
0..n-1, and returns the cross-entropy of the true codes.  On the HIP device the work is:

    hp   = linear1(predictor)                                   rocBLAS (torch.addmm)
    A    = relu(cumsum([hp, scale * embedding(idx[:, :-1])]))    mcq_jcl_prefix_fwd   -> [N][B][H]
    Z    = bias + A @ W2^T + predictor @ W2b^T   per codebook     rocBLAS (torch.baddbmm) -> [N][B][K]
    loss = cross_entropy(Z, idx, ignore negative)                mcq_loss_fwd on [N*B][K]

and a hand-derived backward (mcq_loss_bwd, mcq_jcl_prefix_bwd, mcq_scatter_rows + library GEMMs).
`checkpoint=True` (the reference's default) keeps only the inputs and recomputes A and Z in backward.
There is no CPU path: like the rest of the package this fails loudly off the HIP device.
"""
import torch
from torch import Tensor, nn

from . import _lib


def _check_hip(t: Tensor, what: str):
    if not t.is_cuda:
        raise _lib.McqError(f"quantization_amd.JointCodebookLoss: {what} must live on the HIP device (no CPU fallback)")


def _forward_kernels(pred2d, idx2d, w1, b1, emb, w2, w2b, bias2):
    """-> (hp, A [N][B][H], Z [N][B][K], idxT [N][B], lse [N*B], chosen_sum (1,), count (1, K))"""
    L = _lib.lib()
    B, N = idx2d.shape
    K, H = w2.shape[1], w2.shape[2]
    dev = pred2d.device
    f32 = dict(dtype=torch.float32, device=dev)
    for t_ in (pred2d, w1, emb, w2, w2b, bias2):     # raw pointers go to fp32 kernels: no autocast / half tensors here
        assert t_.dtype == torch.float32, "JointCodebookLoss kernels are fp32 (autocast is disabled inside the Function)"
    hp = torch.addmm(b1, pred2d, w1.t()) if b1 is not None else torch.mm(pred2d, w1.t())
    assert hp.dtype == torch.float32 and hp.is_contiguous()
    A = torch.empty((N, B, H), **f32)
    scale = 0.5 * ((H / N) ** 0.5)                                            # prediction.py:52
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.mcq_jcl_prefix_fwd(hp.data_ptr(), emb.data_ptr(), idx2d.data_ptr(), B, N, K, H, scale, A.data_ptr(), st),
                   "mcq_jcl_prefix_fwd")
    Z = torch.baddbmm(bias2.unsqueeze(1), A, w2.transpose(1, 2))                # (N, B, K)   :67-77
    Z = torch.baddbmm(Z, pred2d.unsqueeze(0).expand(N, B, pred2d.shape[1]), w2b.transpose(1, 2))
    assert Z.dtype == torch.float32 and Z.is_contiguous()
    idxT = idx2d.t().contiguous()
    lse = torch.empty((N * B,), **f32)
    chosen = torch.empty((1,), **f32)
    prob_sum = torch.empty((1, K), **f32)
    count = torch.empty((1, K), **f32)
    ws = torch.empty(L.mcq_loss_workspace_bytes(N * B, 1, K), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.mcq_loss_fwd(Z.data_ptr(), idxT.data_ptr(), N * B, 1, K, lse.data_ptr(), chosen.data_ptr(),
                                  prob_sum.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(), st), "mcq_loss_fwd")
    return hp, A, Z, idxT, lse, chosen, count, scale


class _JointCodebookLossFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, pred2d, idx2d, w1, b1, emb, w2, w2b, bias2, reduction, keep):
        hp, A, Z, idxT, lse, chosen, count, scale = _forward_kernels(pred2d, idx2d, w1, b1, emb, w2, w2b, bias2)
        nvalid = count.sum()
        if reduction == "sum":
            loss = -chosen[0]
        elif reduction == "mean":
            loss = -chosen[0] / nvalid
        elif reduction == "none":
            # per (frame, codebook) losses in the reference's order (logprobs.reshape(-1, K): frame-major), 0 where ignored
            N_, B_ = idxT.shape
            tgt = idxT.clamp(min=0).unsqueeze(2)
            rows = lse.view(N_, B_) - torch.gather(Z, 2, tgt).squeeze(2)
            loss = torch.where(idxT >= 0, rows, torch.zeros_like(rows)).t().reshape(-1)
        else:
            raise ValueError(f"reduction {reduction!r}: expected 'sum', 'mean' or 'none'")
        ctx.reduction, ctx.scale, ctx.keep = reduction, scale, keep
        ctx.has_b1 = b1 is not None
        saved = [pred2d, idx2d, w1, b1 if b1 is not None else pred2d.new_empty(0), emb, w2, w2b, bias2, nvalid]
        if keep:                     # checkpoint=False: keep the activations
            saved += [A, Z, idxT, lse]
        ctx.save_for_backward(*saved)
        return loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_out):
        saved = ctx.saved_tensors
        pred2d, idx2d, w1, b1, emb, w2, w2b, bias2, nvalid = saved[:9]
        b1 = b1 if ctx.has_b1 else None
        if ctx.keep:
            A, Z, idxT, lse = saved[9:]
        else:                        # checkpoint=True: recompute them (prediction.py:166-170)
            _, A, Z, idxT, lse, _, _, _ = _forward_kernels(pred2d, idx2d, w1, b1, emb, w2, w2b, bias2)
        L = _lib.lib()
        B, N = idx2d.shape
        K, H = w2.shape[1], w2.shape[2]
        dev = pred2d.device
        f32 = dict(dtype=torch.float32, device=dev)
        G = torch.empty((N, B, K), **f32)                                         # dL/dZ
        gc = ((-g_out / nvalid) if ctx.reduction == "mean" else
              (-g_out if ctx.reduction == "sum" else -torch.ones((), **f32))).to(torch.float32).reshape(1).contiguous()
        zeros = torch.zeros((1, K), **f32)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.mcq_loss_bwd(Z.data_ptr(), idxT.data_ptr(), lse.data_ptr(), N * B, 1, K, gc.data_ptr(),
                                      zeros.data_ptr(), G.data_ptr(), st), "mcq_loss_bwd")
        if ctx.reduction == "none":      # unit-weight row gradients from the kernel, weighted by the upstream vector
            G = G * g_out.to(torch.float32).reshape(B, N).t().unsqueeze(2)
        g_bias2 = G.sum(dim=1)
        Gt = G.transpose(1, 2)                                                      # (N, K, B)
        g_w2 = torch.bmm(Gt, A)                                                     # (N, K, H)
        g_w2b = torch.matmul(Gt, pred2d)                                            # (N, K, P)
        gA = torch.bmm(G, w2)                                                       # (N, B, H)
        g_pred = torch.einsum("nbk,nkp->bp", G, w2b)
        g_hp = torch.empty((B, H), **f32)
        gE = torch.empty((N - 1, B, H), **f32)
        g_emb = torch.empty(((N - 1) * K, H), **f32)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.mcq_jcl_prefix_bwd(A.data_ptr(), gA.contiguous().data_ptr(), B, N, H, ctx.scale, g_hp.data_ptr(),
                                            gE.data_ptr(), st), "mcq_jcl_prefix_bwd")
            _lib.check(L.mcq_scatter_rows(gE.data_ptr(), H, B * H, idx2d.data_ptr(), N, B, N - 1, K, H, g_emb.data_ptr(), st),
                       "mcq_scatter_rows")
        g_w1 = torch.mm(g_hp.t(), pred2d)
        g_b1 = g_hp.sum(dim=0) if ctx.has_b1 else None
        g_pred = g_pred + torch.mm(g_hp, w1)
        return g_pred, None, g_w1, g_b1, g_emb, g_w2, g_w2b, g_bias2, None, None


class JointCodebookLoss(nn.Module):
    """Drop-in for quantization.JointCodebookLoss (prediction.py:86-172): same arguments, parameters and state_dict."""

    def __init__(self, predictor_channels: int, num_codebooks: int, hidden_channels: int = 512, codebook_size: int = 256,
                 reduction: str = "sum", ignore_index: int = -100, checkpoint: bool = True):
        super().__init__()
        assert num_codebooks > 1                                                   # prediction.py:130
        assert 16 <= codebook_size <= 256 and (codebook_size & (codebook_size - 1)) == 0, \
            "the cross-entropy kernels cover power-of-two codebook sizes in [16, 256]"
        self.num_codebooks = num_codebooks
        self.codebook_size = codebook_size
        self.hidden_channels = hidden_channels
        self.ignore_index = ignore_index
        self.reduction = reduction
        self.checkpoint = checkpoint
        self.linear1 = nn.Linear(predictor_channels, hidden_channels)
        self.codebook_embedding = nn.Embedding(
            (num_codebooks - 1) * codebook_size, hidden_channels,
            _weight=torch.randn((num_codebooks - 1) * codebook_size, hidden_channels) * (hidden_channels ** -0.5))
        self.linear2_weight = nn.Parameter(torch.randn(num_codebooks, codebook_size, hidden_channels) * (hidden_channels ** -0.5))
        self.linear2b_weight = nn.Parameter(torch.randn(num_codebooks, codebook_size, predictor_channels) * (predictor_channels ** -0.5))
        self.linear2_bias = nn.Parameter(torch.zeros(num_codebooks, codebook_size))

    def forward(self, predictor: Tensor, codebook_indexes: Tensor) -> Tensor:
        """predictor (*, predictor_channels), codebook_indexes (*, num_codebooks) integers with negative values
        on padding frames -> the cross-entropy (total negated log-probability for reduction='sum')."""
        _check_hip(predictor, "predictor")
        _check_hip(self.linear2_weight, "the module")
        assert list(predictor.shape[:-1]) == list(codebook_indexes.shape[:-1])     # prediction.py:41
        # deviation from F.cross_entropy(ignore_index=...): EVERY negative target is ignored (the reference's callers pad
        # with -100; any other negative index is an error there), and out-of-range codes are not diagnosed
        assert self.ignore_index < 0, "targets are ignored by sign (every negative index), as the reference's callers use it"
        pred2d = predictor.reshape(-1, predictor.shape[-1]).to(torch.float32).contiguous()
        idx2d = codebook_indexes.reshape(-1, codebook_indexes.shape[-1]).to(device=pred2d.device, dtype=torch.int64).contiguous()
        assert idx2d.shape[1] == self.num_codebooks
        return _JointCodebookLossFn.apply(pred2d, idx2d, self.linear1.weight, self.linear1.bias, self.codebook_embedding.weight,
                                          self.linear2_weight, self.linear2b_weight, self.linear2_bias, self.reduction,
                                          not self.checkpoint)
