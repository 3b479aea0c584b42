"""ms per QuantizerTrainer.step in both phases (bench.py's trainer_leg) -- run once per library build by tools/ab_lib.sh-style
loops: MCQ_ALLOW_LIB_PATH=1 MCQ_LIB_PATH=... python tools/ab_trainer.py [batch] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ms, _ = bench.trainer_leg(torch.device("cuda", 0), 512, 8, batch, iters)
print(os.path.basename(os.environ.get("MCQ_LIB_PATH", "libmcq_hip.so")), batch, ms)
