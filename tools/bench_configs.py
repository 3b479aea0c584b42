"""Secondary measurements on one MI355X: other BASELINE configs and a trainer step."""
import sys, os, time, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from quantization_amd import synthetic as gen
from quantization_amd import Quantizer, QuantizerTrainer

def load(D, K, N, seed=103):
    st = gen.synthetic_state(seed, D, K, N)
    q = Quantizer(D, K, N); sd = q.state_dict()
    for k, v in st.items(): sd[k] = torch.from_numpy(np.asarray(v))
    q.load_state_dict(sd); return q.cuda()

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

res = {}
with torch.no_grad():
    for name, D, K, N, B in [("A_d256_n4", 256, 256, 4, 65536), ("B_d512_n8", 512, 256, 8, 65536), ("D_d1024_n16", 1024, 256, 16, 65536),
                             ("B_1M_shard", 512, 256, 8, 1048576), ("k16_n16_d512", 512, 16, 16, 65536)]:
        q = load(D, K, N)
        x = torch.randn(B, D, device="cuda")
        dt = timeit(lambda: q.encode(x, 5), reps=2)
        codes = q.encode(x, 5)
        dd = timeit(lambda: q.decode(codes), reps=5)
        res[name] = {"encode_Mvec_s": round(B / dt / 1e6, 3), "encode_ms": round(dt * 1e3, 2), "decode_Gvec_s": round(B / dd / 1e9, 3),
                     "decode_GBps": round(B * (codes.shape[1] + 4 * D) / dd / 1e9, 1)}
        print(name, res[name], flush=True)
        del q, x
# trainer step (BASELINE config E shape, single GPU): dim 512, 8 bytes, batch 4096
torch.manual_seed(0); random.seed(0)
tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=torch.device("cuda"), phase_one_iters=30, phase_two_iters=30)
x = torch.randn(4096, 512, device="cuda")
tr.step(x); torch.cuda.synchronize()   # includes the 6 diagnostic losses of iteration 0
times = []
while not tr.done():
    torch.cuda.synchronize(); t = time.perf_counter()
    tr.step(x); torch.cuda.synchronize()
    times.append((tr.cur_iter, tr.quantizer.codebook_size, time.perf_counter() - t))
p1 = [t for i, k, t in times if k == 16][2:]
p2 = [t for i, k, t in times if k == 256][2:]
res["trainer_step_ms"] = {"phase1_K16_N16": round(1e3 * float(np.median(p1)), 3), "phase2_K256_N8": round(1e3 * float(np.median(p2)), 3), "batch": 4096}
print(res["trainer_step_ms"])
# where does a phase-2 step go?
q = tr.quantizer
def search(): q._compute_indexes(x, 2)
print("search(2 iters) ms", round(1e3 * timeit(search, 5), 3))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)

# trainer throughput without a host sync per step (what a training loop sees)
torch.manual_seed(0); random.seed(0)
tr = QuantizerTrainer(dim=512, bytes_per_frame=8, device=torch.device("cuda"), phase_one_iters=400, phase_two_iters=400)
x = torch.randn(4096, 512, device="cuda")
for _ in range(5): tr.step(x)
res2 = {}
for phase, until in (("phase1", 300), ("phase2", 700)):
    while tr.cur_iter < until - 100: tr.step(x)
    torch.cuda.synchronize(); t = time.perf_counter(); n0 = tr.cur_iter
    while tr.cur_iter < until: tr.step(x)
    torch.cuda.synchronize()
    res2[phase] = round(1e3 * (time.perf_counter() - t) / (tr.cur_iter - n0), 3)
print("trainer ms/step, free-running:", res2)
