"""Host-side mirror of `quantization.QuantizerTrainer`
(/root/reference/quantization/quantization.py:577-742)."""
import logging
import math
import os
import random
import time

import torch

from . import _lib
from .quantizer import Quantizer


class _FlatAdam(torch.optim.Optimizer):
    """The reference's optimizer (torch.optim.Adam with weight decay, quantization.py:722-727) as ONE kernel
    (mcq_adam_step) over a flat bucket: the parameters' `.data` and `.grad` are views of `flat_p` / `flat_g`, the two
    moments are flat too.  A torch Optimizer subclass, so the reference's StepLR drives its learning rate unchanged and
    optimizer step hooks fire.

    The moments and the step count live in `self.state` (under the first parameter, as flat tensors over the WHOLE
    bucket), so `state_dict()` / `load_state_dict()` carry them like torch.optim.Adam's: a resumed run continues with
    the same bias correction.  `zero_grad` always zeroes in place (the gradients must stay views of the bucket;
    `set_to_none` is ignored)."""

    def __init__(self, params, flat_p, flat_g, lr, betas, eps, weight_decay):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_p, self.flat_g = flat_p, flat_g
        self._anchor = params[0]
        self.state[self._anchor] = dict(step=0, exp_avg=torch.zeros_like(flat_p), exp_avg_sq=torch.zeros_like(flat_p))

    # views of the optimizer state (load_state_dict replaces the tensors: always read through self.state)
    @property
    def exp_avg(self):
        return self.state[self._anchor]["exp_avg"]

    @property
    def exp_avg_sq(self):
        return self.state[self._anchor]["exp_avg_sq"]

    @property
    def t(self) -> int:
        return int(self.state[self._anchor]["step"])

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self.state[self._anchor]
        n = self.flat_p.numel()
        for k in ("exp_avg", "exp_avg_sq"):
            v = st[k]
            assert v.numel() == n, f"_FlatAdam.load_state_dict: {k} has {v.numel()} elements, the bucket {n}"
            # an own copy (torch's load_state_dict keeps a tensor that already has the right device and dtype as it is:
            # the moments would alias the state_dict's, i.e. those of the optimizer it came from)
            st[k] = v.to(device=self.flat_p.device, dtype=torch.float32).reshape(n).clone()
        st["step"] = int(st["step"])

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        st = self.state[self._anchor]
        st["step"] = t = int(st["step"]) + 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** t
        bc2_sqrt = math.sqrt(1.0 - b2 ** t)
        dev = self.flat_p.device
        with torch.cuda.device(dev):
            rc = _lib.lib().mcq_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), st["exp_avg"].data_ptr(),
                                          st["exp_avg_sq"].data_ptr(), self.flat_p.numel(), float(g["lr"]), float(b1), float(b2),
                                          float(g["eps"]), float(g["weight_decay"]), float(bc1), float(bc2_sqrt),
                                          torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "mcq_adam_step")

    def zero_grad(self, set_to_none: bool = True):
        self.flat_g.zero_()          # the gradients stay views of the bucket


class QuantizerTrainer(object):
    """Two-phase trainer: codebook_size 16 with 2*bytes_per_frame codebooks for
    `phase_one_iters` steps, then the product quantizer (codebook_size 256,
    bytes_per_frame codebooks) for `phase_two_iters` more.  quantization.py:578-631.

    Extension (not in the reference, which is single-process): `process_group`.
    When given (or when torch.distributed is initialised and `data_parallel=True`),
    every rank steps on its own shard of the batch and the gradients are summed with
    one all-reduce per step so that all ranks hold the parameters a single process
    would have computed on the concatenated batch (multi-GPU)."""

    def __init__(self, dim: int, bytes_per_frame: int, device: torch.device, phase_one_iters: int = 10000,
                 phase_two_iters: int = 10000, lr: float = 0.005, process_group=None, data_parallel: bool = False,
                 force_collectives: bool = False):
        super().__init__()
        assert bytes_per_frame in [1, 2, 4, 8, 16, 32]                   # quantization.py:614
        self.phase_one_iters = phase_one_iters
        self.phase_two_iters = phase_two_iters
        self.cur_iter = 0
        self.lr = lr
        self.two_iter_prob = 0.5
        self.entropy_scale = 0.01                                           # quantization.py:682
        # HIP tensors: loss + gradients from the kernels directly, no autograd (MCQ_TRAINER_FUSED=0: tuning hook)
        self.fused_step = os.environ.get("MCQ_TRAINER_FUSED", "1") != "0"
        self.quantizer = Quantizer(dim=dim, codebook_size=16, num_codebooks=bytes_per_frame * 2).to(device)
        self.start_time = time.time()
        self.process_group = process_group
        self.data_parallel = data_parallel or process_group is not None
        # test hook: issue the step's collectives even in a group of ONE rank (sums over one rank are identities), so the
        # RCCL call pattern -- asynchronous all-reduce on a slice of the bucket, the rest behind it -- runs on a 1-GPU box
        self.force_collectives = force_collectives
        self.overlap_all_reduce = os.environ.get("MCQ_TRAINER_OVERLAP", "1") != "0"    # tuning hook: one collective per step
        self._pending = None
        self._grads_dirty = False       # the gradient bucket holds a fused step's (never cleared) gradients
        if self.data_parallel:
            self._broadcast_parameters()
        self._init_optimizer()

    # ------------------------------------------------------------ data parallel
    def _dist(self):
        import torch.distributed as dist
        return dist

    def _world(self) -> int:
        if not self.data_parallel:
            return 1
        dist = self._dist()
        return dist.get_world_size(self.process_group) if dist.is_initialized() else 1

    def _collective(self) -> bool:
        """Does a step exchange anything?  (more than one rank, or the single-rank test hook)"""
        if not self.data_parallel:
            return False
        return self._world() > 1 or (self.force_collectives and self._dist().is_initialized())

    def _broadcast_parameters(self):
        """All ranks start from rank 0's random initialisation."""
        if not self._collective():
            return
        dist = self._dist()
        src = dist.get_global_rank(self.process_group, 0) if self.process_group is not None else 0
        for t in list(self.quantizer.parameters()) + list(self.quantizer.buffers()):
            dist.broadcast(t.data, src=src, group=self.process_group)
        self.quantizer.id_str = bytes(self.quantizer.id_buf.tolist()).decode("utf-8")

    def _all_reduce_flat(self, tensors):
        """One flat-bucket sum all-reduce (RCCL over xGMI on the GPU node; gloo in CPU tests)."""
        dist = self._dist()
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].reshape(t.shape))
            off += n

    # ----------------------------------------------------------------- API
    def done(self) -> bool:
        ans = self.cur_iter > self.phase_one_iters + self.phase_two_iters   # quantization.py:633-639
        if ans:
            elapsed_time = time.time() - self.start_time
            logging.info(f"Elapsed time, training model of dim={self.quantizer.dim}, "
                         f"num_codebooks={self.quantizer.num_codebooks}, "
                         f"codebook_size={self.quantizer.codebook_size}, is: {elapsed_time:.2f} seconds.")
        return ans

    def step(self, x: torch.Tensor) -> None:
        """One optimisation step on frames x (*, dim).  quantization.py:641-719."""
        x = x.reshape(-1, self.quantizer.dim)
        num_iters = 2 if random.random() < self.two_iter_prob else 1          # quantization.py:651
        fused = self.fused_step and x.is_cuda and not x.requires_grad and x.shape[0] > 0
        if not fused and self._grads_dirty:
            # the previous step was a fused one (its kernels OVERWRITE the gradient bucket and nothing clears it afterwards);
            # autograd ACCUMULATES into .grad, so the bucket has to be cleared before this backward
            self.optim.zero_grad()
            self._grads_dirty = False
        if fused:
            try:
                losses = self._fused_loss_and_grads(x, num_iters)
            except BaseException:
                self._drop_pending()        # a collective started by the backward must not pair with a later step's
                raise
            self._grads_dirty = self._flat is not None
        elif self._collective():
            losses = self._dp_losses(x, num_iters)
        else:
            losses = self.quantizer.compute_loss(x, num_iters)
        reconstruction_loss, logprob_loss, logits_entropy_loss, index_entropy_loss = losses

        if self.cur_iter % 200 == 0:                                        # quantization.py:656-671
            det_losses = [float("%.3f" % self.quantizer.compute_loss(x, j)[0].item()) for j in range(6)]
            phase = 1 if self.cur_iter <= self.phase_one_iters else 2
            i = self.cur_iter - self.phase_one_iters if phase > 1 else self.cur_iter
            logging.info(f"phase={phase}/2, iter={i}, "
                         f"dim,nc,csz={self.quantizer.dim},{self.quantizer.num_codebooks},"
                         f"{self.quantizer.codebook_size}, loss_per_iter={det_losses}, "
                         f"logprob_loss={logprob_loss.item():.3f}, "
                         f"logits_entropy_loss={logits_entropy_loss.item():.3f}, "
                         f"index_entropy_loss={index_entropy_loss.item():.3f}")
        if self.cur_iter % 2000 == 0 and self.cur_iter > 0:                 # quantization.py:673-675
            logging.info(f"correlations = {self.quantizer.compute_codebook_correlations()}")

        self._last_losses = tuple(v.detach() for v in losses)   # floats on demand: no device sync per step
        if not fused:
            tot_loss = reconstruction_loss + logprob_loss + logits_entropy_loss * self.entropy_scale   # quantization.py:682-683
            tot_loss.backward()
        if self._collective():
            if self._flat is not None:      # collectives on the bucket the gradients already live in
                dist = self._dist()
                pending, self._pending = self._pending, None
                if pending is not None:     # the centers' part went out during the backward (see _fused_loss_and_grads)
                    work, n_done = pending
                    dist.all_reduce(self._flat[1][n_done:], op=dist.ReduceOp.SUM, group=self.process_group)
                    work.wait()
                else:
                    dist.all_reduce(self._flat[1], op=dist.ReduceOp.SUM, group=self.process_group)
            else:
                self._all_reduce_flat([p.grad for p in self.quantizer.parameters()])
        self.optim.step()
        if not (fused and self._flat is not None):
            self.optim.zero_grad()          # the fused step overwrites every gradient: nothing to clear
            self._grads_dirty = False
        self.scheduler.step()
        if self.cur_iter == self.phase_one_iters:                           # quantization.py:717-718
            self._begin_second_phase()
        self.cur_iter += 1

    def _drop_pending(self):
        pending, self._pending = self._pending, None
        if pending is not None:
            try:
                pending[0].wait()
            except Exception:
                pass

    def _fused_loss_and_grads(self, x, num_iters):
        """The step's loss AND its parameter gradients without autograd (HIP device): forward kernels ->
        batch sums (mcq_loss_head; all-reduced across ranks in data-parallel training) -> mcq_loss_tail (the four
        losses and the upstream gradients of total = rel + logprob + entropy_scale * logits_entropy, :682-683) ->
        backward kernels on the local shard, writing straight into the flat gradient bucket.  Same mathematics as
        compute_loss + backward() (tested against it)."""
        from .quantizer import _loss_backward_kernels, _loss_forward_kernels
        q = self.quantizer
        q._check_domain()
        N, K = q.num_codebooks, q.codebook_size
        B, dev = x.shape[0], x.device
        L = _lib.lib()
        with torch.enable_grad():
            blob = q._prepared()                    # training flavour: scale factors stay on the device
        with torch.no_grad():
            # head (4) | prob_sum (N*K) | count (N*K): one buffer, so the data-parallel forward exchange is one all-reduce
            stats = torch.empty(4 + 2 * N * K, dtype=torch.float32, device=dev)
            head, prob_sum, count = stats[:4], stats[4:4 + N * K].view(N, K), stats[4 + N * K:].view(N, K)
            st_ = _loss_forward_kernels(q, x, num_iters, blob, q._lscale_exp, q._scale_flags, prob_sum, count)
            out = torch.empty(6 + N * K, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                st = torch.cuda.current_stream(dev).cuda_stream
                if self._collective():
                    _lib.check(L.mcq_loss_head(st_.parts[0].data_ptr(), st_.parts[1].data_ptr(), st_.parts.shape[1],
                                               st_.chosen_n.data_ptr(), N, float(B), head.data_ptr(), st), "mcq_loss_head")
                    dist = self._dist()
                    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.process_group)
                    _lib.check(L.mcq_loss_tail(head.data_ptr(), prob_sum.data_ptr(), count.data_ptr(), N, K, self.entropy_scale,
                                               out.data_ptr(), out[4:].data_ptr(), out[6:].data_ptr(), st), "mcq_loss_tail")
                else:       # nothing to exchange between the sums and the losses: one launch
                    _lib.check(L.mcq_loss_head_tail(st_.parts[0].data_ptr(), st_.parts[1].data_ptr(), st_.parts.shape[1],
                                                    st_.chosen_n.data_ptr(), N, float(B), head.data_ptr(), prob_sum.data_ptr(),
                                                    count.data_ptr(), K, self.entropy_scale, out.data_ptr(), out[4:].data_ptr(),
                                                    out[6:].data_ptr(), st), "mcq_loss_head_tail")
            names = ("centers", "centers_scale", "to_logits.weight", "to_logits.bias", "logits_scale")
            params = (q.centers, q.centers_scale, q.to_logits.weight, q.to_logits.bias, q.logits_scale)
            views = None
            hook = None
            if self._flat is not None:
                views = {n_: p_.grad for n_, p_ in zip(names, params)}
                if self._collective() and self.overlap_all_reduce:
                    # two gradient buckets: the centers' (first in the flat bucket, complete after the scatter kernel) is
                    # all-reduced while the classifier's backward (softmax backward, weight-gradient GEMM) still runs
                    def hook(g_centers):
                        assert g_centers.data_ptr() == self._flat[1].data_ptr()
                        dist = self._dist()
                        n = self._flat_offs[1]
                        self._pending = (dist.all_reduce(self._flat[1][:n], op=dist.ReduceOp.SUM, group=self.process_group,
                                                         async_op=True), n)
            grads = _loss_backward_kernels(q, st_, out[4], out[5], out[6:].view(N, K), q.centers, q.centers_scale,
                                           q.to_logits.bias, q.logits_scale, out=views, scales=getattr(q, "_scales_dev", None),
                                           after_centers=hook)
            if views is None:
                for p_, g_ in zip(params, grads):
                    p_.grad = g_.reshape(p_.shape)
        return out[0], out[1], out[2], out[3]

    def _dp_losses(self, x, num_iters):
        """compute_loss on this rank's shard, arranged so that SUMMING the ranks' gradients gives
        the gradient of the loss on the concatenated batch (quantization.py:211-242 on B_total
        frames).  Every term of that loss is a ratio or a function of batch sums, so the sums
        are all-reduced in the forward pass and re-enter the graph as constants plus the local
        term (gradient of the global sum w.r.t. local parameters uses the local part only)."""
        import math
        q = self.quantizer
        B = x.shape[0]
        N, K = q.num_codebooks, q.codebook_size
        num_local, den_local, chosen_local, probs_local, counts_local = q._loss_sums(x, num_iters)
        stats = [num_local.detach().clone().reshape(1), den_local.detach().clone().reshape(1),
                 chosen_local.detach().clone().reshape(1), probs_local.detach().clone(),
                 counts_local.clone(), torch.tensor([float(B)], device=x.device)]
        self._all_reduce_flat(stats)
        num_g, den_g, chosen_g, probs_g, counts_g, Bg = stats
        Bt = Bg.item()

        def with_local_grad(global_value, local):   # value = global, gradient = d(local)
            return global_value + (local - local.detach())

        rel = with_local_grad(num_g.squeeze(0), num_local) / (den_g.squeeze(0) + 1.0e-20)
        logprob_loss = -with_local_grad(chosen_g.squeeze(0), chosen_local) / (Bt * N)
        probs = with_local_grad(probs_g, probs_local) / Bt + 1.0e-20
        logits_entropy = -(probs * probs.log()).sum(dim=1).mean()
        avg_counts = counts_g / Bt + 1.0e-20
        index_entropy = -(avg_counts * avg_counts.log()).sum(dim=1).mean()
        ref_entropy = math.log(K)
        return (rel, logprob_loss, (ref_entropy - logits_entropy) / ref_entropy,
                (ref_entropy - index_entropy) / ref_entropy)

    @property
    def last_losses(self):
        """(rel_reconstruction, logprob, logits_entropy, index_entropy) losses of the last step, as floats."""
        return tuple(float(v) for v in self._last_losses)

    def _flatten_parameters(self):
        """Move the quantizer's parameters into one flat fp32 bucket (and their gradients into another): `.data` and
        `.grad` become views, so the backward kernels write where Adam reads and the data-parallel all-reduce runs on
        the bucket in place.  Offsets are multiples of 16 bytes."""
        ps = list(self.quantizer.parameters())
        offs, off = [], 0
        for p in ps:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        dev = ps[0].device
        flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(ps, offs):
                v = flat_p[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = flat_g[o:o + p.numel()].view(p.shape)
        self.quantizer.invalidate_cache()
        self._flat_offs = offs + [off]
        assert ps[0] is self.quantizer.centers and offs[0] == 0
        return flat_p, flat_g

    def _init_optimizer(self):
        # quantization.py:722-730
        params = list(self.quantizer.parameters())
        on_gpu = all(p.is_cuda and p.dtype == torch.float32 for p in params)
        adam = dict(lr=self.lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=1.0e-06)
        self._flat = None
        if on_gpu and os.environ.get("MCQ_TRAINER_FUSED_ADAM", "1") != "0":
            # the reference's Adam as one kernel over a flat parameter / gradient bucket (same update to 3e-8)
            self._flat = self._flatten_parameters()
            self.optim = _FlatAdam(params, self._flat[0], self._flat[1], **adam)
        else:
            self.optim = torch.optim.Adam(params, **adam)
        self.scheduler = torch.optim.lr_scheduler.StepLR(
            self.optim, step_size=(self.phase_one_iters if self.cur_iter == 0 else self.phase_two_iters) / 4,
            gamma=0.5)

    def _begin_second_phase(self):
        # quantization.py:732-738
        self.quantizer = self.quantizer.get_product_quantizer()
        if self._collective():
            # the product quantizer is a deterministic function of (identical) parameters; only
            # its fresh random id differs between ranks
            dist = self._dist()
            src = dist.get_global_rank(self.process_group, 0) if self.process_group is not None else 0
            dist.broadcast(self.quantizer.id_buf, src=src, group=self.process_group)
            self.quantizer.id_str = bytes(self.quantizer.id_buf.tolist()).decode("utf-8")
        self.lr *= 0.5
        self._init_optimizer()

    def get_quantizer(self) -> Quantizer:
        assert self.cur_iter >= self.phase_one_iters + self.phase_two_iters   # quantization.py:740-742
        return self.quantizer
