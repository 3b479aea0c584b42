"""ms per QuantizerTrainer.step in both phases at config E's shape (dim 512, 8 bytes, 4,096 frames) -- bench.py's trainer_leg -- for
the environment it is started with (same-box A/B of one hook: run it once per value)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
for rep in range(3):
    ms, _ = bench.trainer_leg(dev, 512, 8, 4096, 60)
    print(os.environ.get("AB_TAG", ""), ms)
