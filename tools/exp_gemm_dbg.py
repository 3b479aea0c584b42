import sys, os, ctypes, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dbg in ("", "1", "2"):
    env = dict(os.environ)
    if dbg:
        env["MCQ_LIB_OVERRIDE"] = os.path.join(ROOT, "quantization_amd", "lib", f"libmcq_dbg{dbg}.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    d = json.loads(out)
    print("dbg", dbg or "0", [(k, v["avg_ms"], v["tflops"]) for k, v in d["kernels"].items() if "gemm" in k or "logits" in k])
