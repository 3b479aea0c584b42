"""Host-side mirror of `quantization.Quantizer` (reference:
/root/reference/quantization/quantization.py:16-573) over the MI355X kernels.

Same constructor, parameters, state-dict layout and method names, so a reference
checkpoint loads unchanged and callers switch by changing the import.  The index
search (`encode`, `_compute_indexes`) and the inference `decode` run in
libmcq_hip.so (include/mcq.h); torch is used for device memory, streams and the
differentiable parts of `compute_loss`.  There is no CPU fallback.
"""
import binascii
import ctypes
import math
import os
import weakref

import torch
from torch import Tensor, nn

from . import _lib


# The derived device state (_prepared) is cached per parameter version.  torch's fused optimizers
# (Adam(fused=True) ...) update parameters WITHOUT bumping Tensor._version (measured: tools/exp_version.py),
# so every optimizer step, of any optimizer, also advances this epoch, which is part of the cache key.
# Only optimizers that hold a Quantizer parameter count: a frozen quantizer used inside the training loop of
# another model (codes as targets for JointCodebookLoss) keeps its cached state across that model's steps.
_param_epoch = [0]
_quantizer_params = {}                      # id(parameter) -> weakref(parameter), filled by Quantizer._prepared
_optimizer_hits = weakref.WeakKeyDictionary()   # optimizer -> (number of parameters seen, holds a quantizer parameter)


def _note_optimizer_step(optimizer, *_args, **_kwargs):
    try:
        nparams = sum(len(g["params"]) for g in optimizer.param_groups)
        known = _optimizer_hits.get(optimizer)
        if known is None or known[0] != nparams or known[2] != len(_quantizer_params):
            hit = False
            for g in optimizer.param_groups:
                for q in g["params"]:
                    r = _quantizer_params.get(id(q))
                    if r is not None and r() is q:
                        hit = True
                        break
                if hit:
                    break
            known = (nparams, hit, len(_quantizer_params))
            _optimizer_hits[optimizer] = known
        if known[1]:
            _param_epoch[0] += 1
    except Exception:                       # an exotic optimizer object: stay on the safe side
        _param_epoch[0] += 1


_hook_installed = [False]


def _ensure_optimizer_hook():
    """Installed on first use by a quantizer that has trainable parameters (not at import): a program that only
    encodes / decodes with frozen quantizers never gets a process-wide optimizer hook."""
    if not _hook_installed[0]:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(_note_optimizer_step)
        _hook_installed[0] = True


class _PreparedState:
    """The derived device state of one parameter version and everything that belongs to it: the scale factors it
    was built with travel WITH the blob, so a cache hit can never pair a blob with another build's factors."""
    __slots__ = ("key", "blob", "flavour", "stream", "event", "scales_dev", "scale_flags", "lscale_exp", "cscale_exp")


def _is_pow2(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


def _scale_exp(scale: Tensor, speed: float) -> float:
    """exp(scale * speed) as the reference forms it on the host (fp32 multiply, fp32
    exp; quantization.py:78, :278) -- evaluated on the CPU as the reference evaluates it there.  (Not the same bits on every
    machine: see Quantizer.pin_scale_factors.)"""
    return float((scale.detach().to("cpu", torch.float32) * speed).exp())


class Quantizer(nn.Module):
    """Trainable direct-sum (multi-codebook) vector quantizer; quantization.py:16-55."""

    def __init__(self, dim: int, codebook_size: int, num_codebooks: int):
        super().__init__()
        assert _is_pow2(codebook_size)      # quantization.py:35
        assert _is_pow2(num_codebooks)      # quantization.py:36
        self.dim = dim
        self.codebook_size = codebook_size
        self.num_codebooks = num_codebooks
        self.to_logits = nn.Linear(dim, codebook_size * num_codebooks)
        # centers start as a copy of the classifier weights (quantization.py:41-42)
        self.centers = nn.Parameter(
            self.to_logits.weight.detach().clone().reshape(num_codebooks, codebook_size, dim))
        self.logits_scale = nn.Parameter(torch.zeros(()))
        self.centers_scale = nn.Parameter(torch.zeros(()))
        self.scale_speed = 10.0
        id_bytes = binascii.b2a_hex(os.urandom(4))           # quantization.py:53-55
        self.id_str = id_bytes.decode("utf-8")
        self.register_buffer("id_buf", torch.tensor(list(id_bytes), dtype=torch.uint8))
        # opt-in (not in the reference): vectors whose indexes a refinement pass leaves unchanged are
        # final (the pass is a deterministic map) and skip the remaining passes; same codes, less work
        self.skip_fixed_points = False
        self._pinned_scales = None   # pin_scale_factors()
        self._prep = None       # (key, device buffer) of derived state for the kernels
        self._ws = None         # cached encode workspace (device uint8 tensor)

    def __getstate__(self):
        # derived device state and scratch are rebuilt on demand: keep them out of pickles / deepcopies
        st = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        st = dict(st)
        st["_prep"] = None
        st["_ws"] = None
        st.pop("_host_stage", None)
        return st

    # the scale factors of the CURRENT derived state (set by _prepared, dropped with it)
    def _current_prep(self):
        if self._prep is None:      # (after invalidate_cache / load_state_dict / unpickling: _prepared() has to run first)
            raise RuntimeError("no derived state: call Quantizer._prepared() before reading its scale factors")
        return self._prep

    @property
    def _scale_flags(self) -> int:
        return self._current_prep().scale_flags

    @property
    def _lscale_exp(self) -> float:
        return self._current_prep().lscale_exp

    @property
    def _cscale_exp(self) -> float:
        return self._current_prep().cscale_exp

    @property
    def _scales_dev(self):
        """device float[2] {exp(speed*centers_scale), exp(speed*logits_scale)} of the current derived state, or None
        when that state was built with host-formed factors (the trainer's backward then forms them itself)."""
        return self._prep.scales_dev if self._prep is not None else None

    # ------------------------------------------------------------ bookkeeping
    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self.id_str = bytes(self.id_buf.tolist()).decode("utf-8")   # quantization.py:57-59
        self._prep = None
        return ret

    def pin_scale_factors(self, cscale_exp=None, lscale_exp=None) -> None:
        """Use THESE fp32 values for exp(10 * centers_scale) and exp(10 * logits_scale) in inference (autograd off) instead of
        forming them with the host's exp; call with no arguments to unpin.  Why: torch's fp32 exp is not the same function on
        every CPU -- the same argument gave 0x40c6eb47 on a Xeon and 0x40c6eb46 on an EPYC (round 6) -- so the reference's own codes for
        near-tie vectors depend on the machine it ran on, and so do this module's.  A caller that needs codes bit-identical to
        another machine's (the test fixtures captured from the reference; an index built elsewhere) passes that machine's
        factors.  Not used by the training flavour of the derived state (the scales are read on the device there)."""
        if cscale_exp is None and lscale_exp is None:
            self._pinned_scales = None
        else:
            assert cscale_exp is not None and lscale_exp is not None
            # (the pin belongs to the scale parameters as they are NOW: once they change -- an optimizer step, load_state_dict --
            # the factors are formed from the new values again)
            self._pinned_scales = (float(cscale_exp), float(lscale_exp), float(self.centers_scale.detach()), float(self.logits_scale.detach()))
        self._prep = None

    def invalidate_cache(self) -> None:
        """Drop the cached derived state.  Needed only after parameters were modified in a way torch does not
        version (in-place edits through `.data`); optimizer steps and load_state_dict are tracked."""
        self._prep = None

    def get_id(self) -> str:
        return self.id_str

    def show_init_invocation(self) -> str:
        return (f"quantization.Quantizer(dim={self.dim}, codebook_size={self.codebook_size}, "
                f"num_codebooks={self.num_codebooks})")

    def get_centers(self) -> Tensor:
        return (self.centers_scale * self.scale_speed).exp() * self.centers   # quantization.py:77-79

    def get_data_mean(self) -> Tensor:
        return self.get_centers().mean(dim=1).sum(dim=0).detach()             # quantization.py:67-75

    # --------------------------------------------------------- derived state
    def _prepared(self, any_flavour: bool = False) -> Tensor:
        """Device blob consumed by mcq_encode / mcq_decode; rebuilt only when a parameter changed.

        Inference (autograd off): exp(10*scale) is formed on the host exactly as the reference does on
        CPU (parity with its fixtures).  Training (autograd recording): the scales are read on the
        device (mcq_prepare_dev) so that a training loop never synchronises with the host.  A blob of
        the training flavour is never used for an inference search (its scale factors may differ from
        the host's by an ulp); decode (`any_flavour`) takes whichever is current, and builds a state of its own kind when
        there is none: the scaled centers only ("decode"), which no search accepts."""
        ps = (self.centers, self.centers_scale, self.logits_scale, self.to_logits.weight, self.to_logits.bias)
        trainable = any(p.requires_grad for p in ps)
        if trainable:
            _ensure_optimizer_hook()
        training = torch.is_grad_enabled() and trainable
        for p in ps:
            if id(p) not in _quantizer_params or _quantizer_params[id(p)]() is not p:
                if len(_quantizer_params) > 4096:     # drop entries of collected modules
                    for k_ in [k_ for k_, r_ in _quantizer_params.items() if r_() is None]:
                        del _quantizer_params[k_]
                _quantizer_params[id(p)] = weakref.ref(p)
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps) + (_param_epoch[0],)
        pr = self._prep
        if pr is not None and pr.key == key and (any_flavour or pr.flavour == "host" or (training and pr.flavour != "decode")):
            cur = torch.cuda.current_stream(pr.blob.device)
            if cur.cuda_stream != pr.stream:
                cur.wait_event(pr.event)           # built (asynchronously) on another stream: order this one after it
                pr.blob.record_stream(cur)         # and keep the allocator from recycling it under this stream's kernels
                if pr.scales_dev is not None:
                    pr.scales_dev.record_stream(cur)
            return pr.blob
        on_device = training
        # decode with nothing cached (a quantizer that is only ever decoded with): the scaled centers alone -- no limb
        # planes, no Gram matrix (16 MB at 8 x 256, 1 GB at 64 x 256)
        decode_only = any_flavour and not training
        dev = self.centers.device
        if dev.type != "cuda":
            raise _lib.McqError("quantization_amd.Quantizer runs on a HIP device only: move the module with "
                                ".to('cuda') (the CPU oracle under oracle/ is test infrastructure)")
        L = _lib.lib()
        N, K, D = self.num_codebooks, self.codebook_size, self.dim
        blob = torch.empty(L.mcq_prepared_decode_bytes(N, K, D) if decode_only else L.mcq_prepared_bytes(N, K, D),
                           dtype=torch.uint8, device=dev)
        centers = self.centers.detach().to(torch.float32).contiguous()
        weight = self.to_logits.weight.detach().to(torch.float32).contiguous()
        bias = self.to_logits.bias.detach().to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            if on_device:
                scales = torch.empty(2, dtype=torch.float32, device=dev)     # {exp(speed*centers_scale), exp(speed*logits_scale)}
                cs, ls = self.centers_scale.detach(), self.logits_scale.detach()
                assert cs.dtype == torch.float32 and ls.dtype == torch.float32
                scale_flags = 2             # MCQ_ENCODE_LSCALE_FROM_PREPARED
                cscale_exp = lscale_exp = 1.0
                # exp(speed * scale) is formed by the first kernel of the chain (no launch of its own)
                rc = L.mcq_prepare_params(centers.data_ptr(), cs.data_ptr(), ls.data_ptr(), self.scale_speed, weight.data_ptr(),
                                          bias.data_ptr(), N, K, D, blob.data_ptr(), scales.data_ptr(), st)
            else:
                both = torch.stack([self.centers_scale.detach(), self.logits_scale.detach()]).to(torch.float32)
                both = both.to("cpu")       # both scalars in one device->host copy; exp on the host
                scales = None               # host-formed factors: no device copy belongs to this state
                scale_flags = 0
                pin = self._pinned_scales
                if pin is not None and (float(both[0]), float(both[1])) == pin[2:]:
                    cscale_exp, lscale_exp = pin[0], pin[1]
                else:
                    cscale_exp = _scale_exp(both[0], self.scale_speed)
                    lscale_exp = _scale_exp(both[1], self.scale_speed)
                rc = L.mcq_prepare(centers.data_ptr(), cscale_exp, None if decode_only else weight.data_ptr(),
                                   None if decode_only else bias.data_ptr(), N, K, D, blob.data_ptr(), st)
        _lib.check(rc, "mcq_prepare")
        # the inputs above may be temporaries: the stream orders their reuse after the kernel
        with torch.cuda.device(dev):
            ev = torch.cuda.Event()
            cur = torch.cuda.current_stream(dev)
            ev.record(cur)
        pr = _PreparedState()
        pr.key, pr.blob, pr.stream, pr.event = key, blob, cur.cuda_stream, ev
        pr.flavour = "device" if on_device else ("decode" if decode_only else "host")
        pr.scales_dev, pr.scale_flags, pr.lscale_exp, pr.cscale_exp = scales, scale_flags, lscale_exp, cscale_exp
        self._prep = pr
        return blob

    def _check_domain(self):
        assert 16 <= self.codebook_size <= 1024, (
            "the index search needs 16 <= codebook_size <= 1024 (the reference itself fails below 16, "
            "quantization.py:506; byte codes need <= 256, :271: larger codebooks encode with as_bytes=False)")
        assert self.num_codebooks <= 64, "num_codebooks <= 64 (QuantizerTrainer produces at most 64: quantization.py:614)"
        assert self.num_codebooks * self.codebook_size <= 16384, "num_codebooks * codebook_size <= 16384 (the Gram matrix: 1 GB)"
        assert self.dim <= 16384, "dim <= 16384 (the i32 accumulators of the fixed-point products, include/mcq.h)"

    def _workspace(self, B: int, dev) -> Tensor:
        """Scratch of the search, one buffer per (device, stream): encodes issued on different streams never share it."""
        L = _lib.lib()
        need = L.mcq_encode_workspace_bytes(B, self.num_codebooks, self.codebook_size, self.dim)
        if not isinstance(self._ws, dict):
            self._ws = {}
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            if len(self._ws) >= 8:
                self._ws.clear()
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        return ws

    def _search(self, x2d: Tensor, iters: int, as_bytes: bool) -> Tensor:
        """x2d (B, dim) on the HIP device -> uint8 codes or int64 indexes via mcq_encode."""
        self._check_domain()
        if not x2d.is_cuda:
            raise _lib.McqError("quantization_amd: the index search runs on a HIP device tensor only "
                                "(no CPU fallback)")
        L = _lib.lib()
        N, K, D = self.num_codebooks, self.codebook_size, self.dim
        # fp16 frames (what the reference's read_hdf5_data yields, quantization.py:798) are consumed as they
        # are: the kernels widen them in the load path (MCQ_ENCODE_X_FP16), same codes as for x.float()
        x_fp16 = x2d.dtype == torch.float16
        x2d = x2d.detach().contiguous() if x_fp16 else x2d.detach().to(torch.float32).contiguous()
        B = x2d.shape[0]
        dev = x2d.device
        with torch.no_grad():       # the public search always uses host-formed scale factors (the reference's, bit for bit),
            blob = self._prepared()  # whatever the caller's autograd mode; the trainer's own path keeps them on the device
        pack = 2 if (as_bytes and K == 16 and N >= 2) else 1
        if as_bytes:
            out = torch.empty((B, N // pack), dtype=torch.uint8, device=dev)
        else:
            out = torch.empty((B, N), dtype=torch.int64, device=dev)
        if B == 0:
            return out
        ws = self._workspace(B, dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            rc = L.mcq_encode_ex(x2d.data_ptr(), B, blob.data_ptr(), self._lscale_exp, N, K, D, int(iters),
                                 out.data_ptr() if as_bytes else None, None if as_bytes else out.data_ptr(),
                                 ws.data_ptr(), ws.numel(), st,
                                 (1 if getattr(self, "skip_fixed_points", False) else 0) | self._scale_flags |
                                 (4 if x_fp16 else 0))
        _lib.check(rc, "mcq_encode")
        return out

    # ------------------------------------------------------------ public API
    def encode(self, x: Tensor, refine_indexes_iters: int = 5, as_bytes: bool = True) -> Tensor:
        """x (*, dim) -> uint8 (*, num_codebooks) [packed two per byte when codebook_size == 16],
        or int64 (*, num_codebooks) with as_bytes=False.  quantization.py:244-275."""
        x2d = x.reshape(-1, self.dim)
        if as_bytes:
            assert self.codebook_size <= 256                              # quantization.py:271
        codes = self._search(x2d, refine_indexes_iters, as_bytes)
        return codes.reshape(*x.shape[:-1], codes.shape[-1])

    def encode_from_host(self, x: Tensor, refine_indexes_iters: int = 5, as_bytes: bool = True,
                         chunk: int = 32768) -> Tensor:
        """encode() for a batch that lives in HOST memory (not in the reference: its callers move data
        themselves).  The batch is cut into chunks; the host->device copy of chunk i+1 runs on a copy
        stream while chunk i is encoded, and the codes come back asynchronously, so PCIe time (about
        9 % of the encode at dim 512) hides behind the kernels.  Returns CPU codes, same values as
        encode(x.to(device)).cpu()."""
        dev = self.centers.device
        if dev.type != "cuda":
            raise _lib.McqError("quantization_amd.Quantizer runs on a HIP device only")
        x2d = x.reshape(-1, self.dim)
        if x2d.dtype != torch.float16:       # fp16 frames cross PCIe as they are (half the bytes)
            x2d = x2d.to(torch.float32)
        if not x2d.is_pinned():
            x2d = x2d.contiguous().pin_memory()
        B = x2d.shape[0]
        pack = 2 if (as_bytes and self.codebook_size == 16 and self.num_codebooks >= 2) else 1
        width = self.num_codebooks // pack
        out = torch.empty((B, width), dtype=torch.uint8 if as_bytes else torch.int64, pin_memory=True)
        compute = torch.cuda.current_stream(dev)
        rows = min(chunk, max(B, 1))
        stage = getattr(self, "_host_stage", None)
        if stage is None or stage[0].device != dev or stage[0].shape[1] < rows or stage[0].dtype != x2d.dtype:
            stage = (torch.empty((2, rows, self.dim), dtype=x2d.dtype, device=dev), torch.cuda.Stream(dev))
            self._host_stage = stage
        bufs, copy = [stage[0][0], stage[0][1]], stage[1]
        copy.wait_stream(compute)       # earlier users of the staging buffers are done
        ready = [torch.cuda.Event() for _ in range(2)]      # chunk landed in bufs[i]
        freed = [torch.cuda.Event() for _ in range(2)]      # bufs[i] consumed by the encode
        results = []
        for ci, lo in enumerate(range(0, B, chunk)):
            hi = min(lo + chunk, B)
            slot = ci & 1
            with torch.cuda.stream(copy):
                if ci >= 2:
                    copy.wait_event(freed[slot])
                bufs[slot][:hi - lo].copy_(x2d[lo:hi], non_blocking=True)
                ready[slot].record(copy)
            compute.wait_event(ready[slot])
            codes = self.encode(bufs[slot][:hi - lo], refine_indexes_iters, as_bytes)
            freed[slot].record(compute)
            out[lo:hi].copy_(codes, non_blocking=True)
            results.append(codes)      # keep the device tensors alive until the copies are done
        compute.synchronize()
        del results
        return out.reshape(*x.shape[:-1], width)

    def _compute_indexes(self, x: Tensor, refine_indexes_iters: int = 3) -> Tensor:
        """x (B, dim) -> int64 (B, num_codebooks).  quantization.py:281-305."""
        assert x.ndim == 2 and x.shape[1] == self.dim                     # quantization.py:293
        return self._search(x, refine_indexes_iters, as_bytes=False)

    def _refine_indexes(self, x: Tensor, indexes: Tensor) -> Tensor:
        """One refinement pass from the given indexes: x (B, dim), indexes (B, num_codebooks) ->
        int64 (B, num_codebooks).  quantization.py:308-547."""
        self._check_domain()
        if not x.is_cuda:
            raise _lib.McqError("quantization_amd: the index search runs on a HIP device tensor only")
        L = _lib.lib()
        N, K, D = self.num_codebooks, self.codebook_size, self.dim
        x2d = x.detach().to(torch.float32).contiguous()
        B = x2d.shape[0]
        idx = indexes.to(device=x2d.device, dtype=torch.int64).contiguous()
        assert idx.shape == (B, N)
        out = torch.empty_like(idx)
        if B == 0:
            return out
        with torch.no_grad():
            blob = self._prepared()
        ws = self._workspace(B, x2d.device)
        with torch.cuda.device(x2d.device):
            st = torch.cuda.current_stream(x2d.device).cuda_stream
            rc = L.mcq_refine_indexes(x2d.data_ptr(), B, blob.data_ptr(), N, K, D, 1, idx.data_ptr(), out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "mcq_refine_indexes")
        return out

    def _logits(self, x: Tensor) -> Tensor:
        x = (self.logits_scale * self.scale_speed).exp() * x              # quantization.py:277-279
        return self.to_logits(x)

    def _maybe_separate_indexes(self, indexes: Tensor) -> Tensor:
        """Undo the byte packing of encode(); quantization.py:551-573 (torch ops; the kernel
        path of decode() unpacks in-kernel)."""
        B = indexes.shape[0]
        if indexes.shape[-1] != self.num_codebooks:
            n = indexes.shape[-1]
            rep = self.num_codebooks // n
            assert rep in [2, 4, 8, 16] and self.num_codebooks == n * rep   # quantization.py:566
            div = self.codebook_size ** torch.arange(rep, device=indexes.device)
            indexes = (indexes.unsqueeze(2).expand(B, n, rep) // div) % self.codebook_size
            indexes = indexes.reshape(B, self.num_codebooks)
        assert indexes.shape == (B, self.num_codebooks)
        return indexes

    def decode(self, indexes: Tensor) -> Tensor:
        """codes (*, num_codebooks) or packed (*, num_codebooks / r), any integer dtype ->
        fp32 (*, dim) sum of the chosen scaled centers.  quantization.py:117-148.
        Differentiable w.r.t. centers / centers_scale when autograd is recording."""
        lead = indexes.shape[:-1]
        per_row = indexes.shape[-1]
        flat = indexes.reshape(-1, per_row)
        rep = self.num_codebooks // per_row if per_row else 0
        assert per_row * rep == self.num_codebooks and rep in [1, 2, 4, 8, 16]   # quantization.py:566
        needs_grad = torch.is_grad_enabled() and (self.centers.requires_grad or self.centers_scale.requires_grad)
        if needs_grad:
            return _DecodeFn.apply(self, flat, self.centers, self.centers_scale).reshape(*lead, self.dim)
        return self._decode_kernel(flat).reshape(*lead, self.dim)

    def _decode_kernel(self, flat: Tensor) -> Tensor:
        self._check_domain()
        if not flat.is_cuda:
            raise _lib.McqError("quantization_amd: decode runs on a HIP device tensor only (no CPU fallback)")
        L = _lib.lib()
        N, K, D = self.num_codebooks, self.codebook_size, self.dim
        if flat.dtype not in (torch.uint8, torch.int64):
            flat = flat.to(torch.int64)
        flat = flat.contiguous()
        B, per_row = flat.shape
        out = torch.empty((B, D), dtype=torch.float32, device=flat.device)
        if B == 0:
            return out
        if os.environ.get("MCQ_CHECK_CODES") == "1" and flat.numel():
            # debugging aid (synchronises): the kernels mask digits with K - 1 where the reference's gather would
            # raise on an out-of-range index (quantization.py:142)
            assert int(flat.min()) >= 0 and int(flat.max()) < K ** (N // per_row), "codes outside [0, codebook_size)"
        if flat.dtype == torch.int64 and per_row == N and N in (4, 8, 16) and B >= 16384 and K <= 256:
            # unpacked int64 indexes of a large batch: as bytes they take the block-staged LDS-resident kernel (the kernels
            # mask a digit with K - 1 either way, and K <= 256: the low byte carries the same digit)
            flat = flat.to(torch.uint8)
        blob = self._prepared(any_flavour=True)
        with torch.cuda.device(flat.device):
            st = torch.cuda.current_stream(flat.device).cuda_stream
            rc = L.mcq_decode(flat.data_ptr(), 1 if flat.dtype == torch.uint8 else 8, per_row, B, blob.data_ptr(),
                              N, K, D, out.data_ptr(), st)
        _lib.check(rc, "mcq_decode")
        return out

    def logits_kernel(self, x: Tensor) -> Tensor:
        """Logits as the index-search kernel forms them (test hook; mcq_logits)."""
        L = _lib.lib()
        x2d = x.reshape(-1, self.dim).detach().to(torch.float32).contiguous()
        N, K, D = self.num_codebooks, self.codebook_size, self.dim
        out = torch.empty((x2d.shape[0], N * K), dtype=torch.float32, device=x2d.device)
        with torch.no_grad():      # host-side scale factors (mcq_logits takes lscale by value)
            blob = self._prepared()
        with torch.cuda.device(x2d.device):
            st = torch.cuda.current_stream(x2d.device).cuda_stream
            ws = torch.empty(L.mcq_logits_workspace_bytes(x2d.shape[0], N, D), dtype=torch.uint8, device=x2d.device)
            rc = L.mcq_logits(x2d.data_ptr(), x2d.shape[0], blob.data_ptr(), self._lscale_exp, N, K, D,
                              out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "mcq_logits")
        return out

    # -------------------------------------------------------------- training
    def _loss_sums(self, x: Tensor, refine_indexes_iters: int):
        """The batch SUMS every term of compute_loss is made of (quantization.py:211-242):
        (sum of squared reconstruction error, sum of squared (x - data mean), sum of the chosen log-probs,
        per-(codebook, entry) sum of softmax probabilities (N, K), per-(codebook, entry) count of chosen
        indexes (N, K)).  compute_loss forms the means / ratios from them; the data-parallel trainer
        all-reduces them first.  On a HIP tensor the whole thing is five kernels forward (logits + argmax,
        refinement, reconstruction, softmax statistics) and a hand-derived backward (_LossSumsFn)."""
        B = x.shape[0]
        N, K = self.num_codebooks, self.codebook_size
        if x.is_cuda and not x.requires_grad and B > 0 and K <= 256:      # (the fused kernels: what QuantizerTrainer runs)
            self._check_domain()
            blob = self._prepared()          # here, not inside the Function: autograd mode decides the flavour
            return _LossSumsFn.apply(self, x, int(refine_indexes_iters), blob, self._lscale_exp, self._scale_flags,
                                     self.centers, self.centers_scale, self.to_logits.weight, self.to_logits.bias,
                                     self.logits_scale)[:5]
        # the same sums with the reference's own torch op sequence (x that requires grad; codebooks of more than 256 entries:
        # the index search and decode are the HIP kernels, the rest torch ops; the CPU test harness)
        indexes = self._compute_indexes(x, refine_indexes_iters)
        x_approx = self.decode(indexes)
        num = ((x_approx - x) ** 2).sum()
        den = ((x - self.get_data_mean()) ** 2).sum()
        logprobs = self._logits(x).reshape(B, N, K).log_softmax(dim=2)
        chosen = torch.gather(logprobs, dim=2, index=indexes.unsqueeze(2)).sum()
        prob_sum = logprobs.exp().sum(dim=0)
        count = torch.zeros(N, K, device=x.device)
        count.scatter_add_(1, indexes.t().contiguous(), torch.ones(N, B, device=x.device))
        return num, den, chosen, prob_sum, count

    def compute_loss(self, x: Tensor, refine_indexes_iters: int = 0):
        """(rel_reconstruction_loss, logprob_loss, logits_entropy_loss, index_entropy_loss);
        quantization.py:184-242."""
        x = x.reshape(-1, self.dim)
        B = x.shape[0]
        N, K = self.num_codebooks, self.codebook_size
        num, den, chosen, prob_sum, count = self._loss_sums(x, refine_indexes_iters)
        rel_reconstruction_loss = num / (den + 1.0e-20)
        logprob_loss = -chosen / (B * N)

        avg_counts = count / B + 1.0e-20
        index_entropy = -(avg_counts * avg_counts.log()).sum(dim=1).mean()
        probs = prob_sum / B + 1.0e-20
        logits_entropy = -(probs * probs.log()).sum(dim=1).mean()
        ref_entropy = math.log(K)
        logits_entropy_loss = (ref_entropy - logits_entropy) / ref_entropy
        index_entropy_loss = (ref_entropy - index_entropy) / ref_entropy
        return rel_reconstruction_loss, logprob_loss, logits_entropy_loss, index_entropy_loss

    def compute_codebook_correlations(self) -> Tensor:
        """(N, N) subspace-sharing diagnostic; quantization.py:150-181."""
        centers = self.get_centers().detach()
        centers = centers - centers.mean(dim=1, keepdim=True)
        var = torch.matmul(centers.transpose(1, 2), centers).reshape(self.num_codebooks, self.dim * self.dim)
        cross = torch.matmul(var, var.t())
        norm = cross.diag() ** -0.5
        return cross * (norm.unsqueeze(0) * norm.unsqueeze(1))

    def get_product_quantizer(self) -> "Quantizer":
        """codebook_size**2 entries, half the codebooks: entry k1*K + k2 of new codebook c is
        the sum of entry k1 of codebook 2c and entry k2 of codebook 2c+1, for the centers and
        for the classifier rows and biases alike.  quantization.py:81-112."""
        K, N, D = self.codebook_size, self.num_codebooks, self.dim
        ans = Quantizer(D, K * K, N // 2).to(self.centers.device)
        with torch.no_grad():
            ans.logits_scale.fill_(self.logits_scale.item())
            ans.centers_scale.fill_(self.centers_scale.item())
            ans.scale_speed = self.scale_speed

            def pair_sum(t):  # t: (N, K, ...) -> (N/2, K*K, ...), k_out = k1*K + k2
                even, odd = t[0::2], t[1::2]
                s = even.unsqueeze(2) + odd.unsqueeze(1)
                return s.reshape(N // 2, K * K, *t.shape[2:])

            ans.to_logits.weight.copy_(pair_sum(self.to_logits.weight.reshape(N, K, D)).reshape(-1, D))
            ans.to_logits.bias.copy_(pair_sum(self.to_logits.bias.reshape(N, K)).reshape(-1))
            ans.centers.copy_(pair_sum(self.centers))
        return ans


class _DecodeFn(torch.autograd.Function):
    """decode() under autograd: forward in the HIP kernel, backward = the scatter-add of
    d(out) into the chosen rows (what torch.gather/sum differentiate to, quantization.py:142-147)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, flat, centers, centers_scale):
        out = module._decode_kernel(flat)
        idx = module._maybe_separate_indexes(flat.to(torch.int64))
        ctx.module = module
        ctx.save_for_backward(idx, centers, centers_scale)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, centers, centers_scale = ctx.saved_tensors
        m = ctx.module
        N, K, D = centers.shape
        scale = (centers_scale.detach() * m.scale_speed).exp()
        # d/d(scaled centers): rows receive the sum of grad_out over the vectors that chose them
        if grad_out.is_cuda and grad_out.dtype == torch.float32:
            # deterministic HIP kernel (fixed summation order; index_add_ on the device uses atomics)
            go = grad_out.contiguous()
            g = torch.empty((N, K, D), dtype=torch.float32, device=go.device)
            with torch.cuda.device(go.device):
                rc = _lib.lib().mcq_decode_backward(go.data_ptr(), idx.contiguous().data_ptr(), go.shape[0], N, K, D,
                                                    g.data_ptr(), torch.cuda.current_stream(go.device).cuda_stream)
            _lib.check(rc, "mcq_decode_backward")
        else:
            g = torch.zeros(N * K, D, dtype=grad_out.dtype, device=grad_out.device)
            rows = (idx + torch.arange(N, device=idx.device) * K).reshape(-1)
            g.index_add_(0, rows, grad_out.unsqueeze(1).expand(-1, N, -1).reshape(-1, D))
            g = g.reshape(N, K, D)
        g_centers = g * scale
        g_scale = (g * centers.detach()).sum() * scale * m.scale_speed
        return None, None, g_centers, g_scale


class _LossState:
    """What the forward loss kernels leave behind for the backward ones."""
    __slots__ = ("xf", "idx", "codes", "err", "logits", "lse", "parts", "chosen_n", "prob_sum", "count")


def _loss_forward_kernels(module, x, iters, blob, lscale_exp, flags, prob_sum=None, count=None) -> _LossState:
    """mcq_logits_refine (one product gives the logits AND the initial indexes, the passes follow), mcq_recon_fwd,
    mcq_loss_fwd on x (B, dim) fp32/fp16 on the HIP device."""
    L = _lib.lib()
    N, K, D = module.num_codebooks, module.codebook_size, module.dim
    x_fp16 = x.dtype == torch.float16
    xk = x.detach().contiguous() if x_fp16 else x.detach().to(torch.float32).contiguous()
    B, dev = xk.shape[0], xk.device
    f32 = dict(dtype=torch.float32, device=dev)
    st_ = _LossState()
    st_.logits = torch.empty((B, N * K), **f32)
    st_.idx = torch.empty((B, N), dtype=torch.int64, device=dev)
    st_.codes = torch.empty((B, N), dtype=torch.uint8, device=dev)     # the same indexes as bytes (the scatter scans these)
    ws = module._workspace(B, dev)
    st_.xf = xk.float() if x_fp16 else xk
    # get_data_mean() (:67-75) of the scaled centers: formed by mcq_prepare inside the prepared blob
    Dp = L.mcq_padded_dim(D)
    moff = L.mcq_prepared_mean_offset(N, K, D)
    mean = blob[moff:moff + 4 * Dp].view(torch.float32)[:D]
    st_.err = torch.empty((B, D), **f32)
    st_.parts = torch.empty((2, (B + 3) // 4), **f32)
    st_.lse = torch.empty((B, N), **f32)
    st_.chosen_n = torch.empty((N,), **f32)
    st_.prob_sum = prob_sum if prob_sum is not None else torch.empty((N, K), **f32)
    st_.count = count if count is not None else torch.empty((N, K), **f32)
    lws = torch.empty(L.mcq_loss_workspace_bytes(B, N, K), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.mcq_logits_refine_codes(xk.data_ptr(), B, blob.data_ptr(), lscale_exp, N, K, D, iters,
                                             st_.logits.data_ptr(), st_.idx.data_ptr(), st_.codes.data_ptr(), ws.data_ptr(),
                                             ws.numel(), st, flags | (4 if x_fp16 else 0)), "mcq_logits_refine_codes")
        _lib.check(L.mcq_recon_fwd(st_.xf.data_ptr(), st_.idx.data_ptr(), B, blob.data_ptr(), mean.data_ptr(), N, K, D,
                                   st_.err.data_ptr(), st_.parts[0].data_ptr(), st_.parts[1].data_ptr(), st), "mcq_recon_fwd")
        _lib.check(L.mcq_loss_fwd(st_.logits.data_ptr(), st_.idx.data_ptr(), B, N, K, st_.lse.data_ptr(),
                                  st_.chosen_n.data_ptr(), st_.prob_sum.data_ptr(), st_.count.data_ptr(), lws.data_ptr(),
                                  lws.numel(), st), "mcq_loss_fwd")
    return st_


def _loss_backward_kernels(module, st_: _LossState, g_num, g_chosen, g_prob, centers, centers_scale, bias, logits_scale,
                           out=None, scales=None, after_centers=None):
    """Gradients of  g_num * sum err^2 + g_chosen * sum chosen + <g_prob, prob_sum>  w.r.t. (centers, centers_scale,
    to_logits.weight, to_logits.bias, logits_scale); g_* are device tensors (or None).  Derivation:
      d sum err^2 / d(scaled centers) = 2 * scatter-add of err  (mcq_decode_backward_u8_ex: rows scaled by
        f = exp(speed*centers_scale) * 2 * g_num in its epilogue, <unscaled rows, centers> left as per-wave partials);
      logits = s (x W^T) + b with s = exp(speed * logits_scale):  dW = s G^T x,  db = sum_b G  (mcq_weight_grad, fp32 MFMA),
      d logits_scale = speed * <G, logits - b>  (per-wave partials of mcq_loss_bwd_ex),  G = d/d logits;
      mcq_grad_tail reduces the two partial arrays in a fixed order.
    `out`: optional dict name -> preallocated tensor (the trainer's flat gradient bucket); `scales`: optional device
    float[2] {exp(speed*centers_scale), exp(speed*logits_scale)} (the trainer's prepared state holds it);
    `after_centers`: optional callable, invoked once the centers' gradient is enqueued and before the classifier's
    kernels (the data-parallel trainer starts that bucket's all-reduce there, so it overlaps the rest of the backward)."""
    L = _lib.lib()
    N, K, D = centers.shape
    B, dev = st_.xf.shape[0], st_.xf.device
    f32 = dict(dtype=torch.float32, device=dev)
    speed = float(module.scale_speed)
    out = out or {}

    def buf(name, shape):
        t = out.get(name)
        if t is None:
            t = torch.empty(shape, **f32)
        assert t.dtype == torch.float32 and t.is_contiguous()
        return t

    if scales is None:
        scales = torch.stack([(centers_scale.detach() * speed).exp(), (logits_scale.detach() * speed).exp()]).to(torch.float32)
    assert scales.dtype == torch.float32 and st_.xf.dtype == torch.float32 and st_.err.dtype == torch.float32
    g_centers = g_cscale = g_weight = g_bias = g_lscale = None
    part_c = part_l = None
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        if g_num is not None:
            g_centers, g_cscale = buf("centers", (N, K, D)), buf("centers_scale", ())
            gn = g_num.detach().to(torch.float32).reshape(1)
            cw = centers.detach()
            assert cw.dtype == torch.float32 and cw.is_contiguous()
            codes = st_.codes                       # the scatter is bound by scanning the index column: 1 byte, not 8
            part_c = torch.empty(L.mcq_decode_backward_waves(N, K, D), **f32)
            _lib.check(L.mcq_decode_backward_u8_ex(st_.err.data_ptr(), codes.data_ptr(), B, N, K, D, g_centers.data_ptr(),
                                                   scales.data_ptr(), gn.data_ptr(), 2.0, cw.data_ptr(), part_c.data_ptr(), st),
                       "mcq_decode_backward_u8_ex")
            if after_centers is not None:
                after_centers(g_centers)
        if g_chosen is not None or g_prob is not None:
            g_weight, g_bias, g_lscale = buf("to_logits.weight", (N * K, D)), buf("to_logits.bias", (N * K,)), buf("logits_scale", ())
            gc = (g_chosen if g_chosen is not None else torch.zeros((), **f32)).detach().to(torch.float32).reshape(1).contiguous()
            gp = (g_prob if g_prob is not None else torch.zeros((N, K), **f32)).detach().to(torch.float32).contiguous()
            bw = bias.detach()
            assert bw.dtype == torch.float32 and bw.is_contiguous()
            G = torch.empty((B, N * K), **f32)
            part_l = torch.empty(L.mcq_loss_bwd_waves(B, N, K), **f32)
            _lib.check(L.mcq_loss_bwd_ex(st_.logits.data_ptr(), st_.idx.data_ptr(), st_.lse.data_ptr(), B, N, K, gc.data_ptr(),
                                         gp.data_ptr(), G.data_ptr(), bw.data_ptr(), part_l.data_ptr(), st), "mcq_loss_bwd_ex")
            wws = torch.empty(L.mcq_weight_grad_workspace_bytes(B, N * K, D), dtype=torch.uint8, device=dev)
            _lib.check(L.mcq_weight_grad(G.data_ptr(), st_.xf.data_ptr(), B, N * K, D, scales[1:].data_ptr(), g_weight.data_ptr(),
                                         g_bias.data_ptr(), wws.data_ptr(), wws.numel(), st), "mcq_weight_grad")
        if part_c is not None or part_l is not None:
            _lib.check(L.mcq_grad_tail(part_c.data_ptr() if part_c is not None else None, part_c.numel() if part_c is not None else 0,
                                       scales.data_ptr(), gn.data_ptr() if part_c is not None else None, 2.0,
                                       part_l.data_ptr() if part_l is not None else None, part_l.numel() if part_l is not None else 0,
                                       speed, g_cscale.data_ptr() if g_cscale is not None else None,
                                       g_lscale.data_ptr() if g_lscale is not None else None, st), "mcq_grad_tail")
    return g_centers, g_cscale, g_weight, g_bias, g_lscale


class _LossSumsFn(torch.autograd.Function):
    """Quantizer._loss_sums on the HIP device under autograd: the forward kernels of _loss_forward_kernels,
    the hand-derived backward of _loss_backward_kernels."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, x, iters, blob, lscale_exp, flags, centers, centers_scale, weight, bias, logits_scale):
        st_ = _loss_forward_kernels(module, x, iters, blob, lscale_exp, flags)
        sums = st_.parts.sum(dim=1)
        num, den, chosen = sums[0], sums[1], st_.chosen_n.sum()
        ctx.module = module
        ctx.st = st_
        ctx.save_for_backward(centers, centers_scale, bias, logits_scale)
        ctx.mark_non_differentiable(den, st_.count, st_.idx)
        return num, den, chosen, st_.prob_sum, st_.count, st_.idx

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_num, g_den, g_chosen, g_prob, g_count, g_idx):
        centers, centers_scale, bias, logits_scale = ctx.saved_tensors
        grads = _loss_backward_kernels(ctx.module, ctx.st, g_num, g_chosen, g_prob, centers, centers_scale, bias,
                                       logits_scale)
        return (None, None, None, None, None, None) + tuple(grads)
