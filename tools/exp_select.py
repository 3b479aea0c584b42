"""The wave-level selection on its own (mcq_test_select: one wave per problem, scores from memory): time per launch for the three
shapes the passes use -- 16 of 256 (stage 0, level 0), 32 of 256 (level 1), 32 of 1,024 (level 2 of 16 codebooks) -- and a check
against numpy's stable argsort."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantization_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for per_lane, cnt, cases in ((4, 16, 262144), (4, 32, 131072), (16, 32, 65536)):
    M = 64 * per_lane
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    scores = (torch.rand(cases, M, device=dev, generator=g) * 50 + 400).contiguous()
    ov = torch.empty(cases, 64, device=dev)
    op = torch.empty(cases, 64, device=dev, dtype=torch.int32)
    for _ in range(3):
        assert L.mcq_test_select(scores.data_ptr(), cases, per_lane, cnt, ov.data_ptr(), op.data_ptr(), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.mcq_test_select(scores.data_ptr(), cases, per_lane, cnt, ov.data_ptr(), op.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    want = np.sort(np.argsort(scores[:512].cpu().numpy(), axis=1, kind="stable")[:, :cnt], axis=1)
    ok = np.array_equal(np.sort(op[:512, :cnt].cpu().numpy(), axis=1), want)      # (the set: builds before round 6 list it by value)
    print(f"{cnt:3d} of {M:5d}: {ms * 1e3:8.1f} us per launch of {cases} selections = {ms * 1e6 / cases:6.2f} ns each; correct: {ok}")
