"""`python bench.py --gpus N` starts its own ranks (no torchrun wrapper): the self-launch is exercised on the CPU with
`--rendezvous-only` (every rank joins a gloo group on 127.0.0.1 and reports in; no device, no compute), and the helpers
that price the dominant kernel are checked on numbers."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_2_launches_two_ranks_without_a_wrapper():
    r = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    got = json.loads(line)
    assert got["rendezvous"] == 2 and got["ranks"] == [0, 1] and got["pids"] == 2 and got["backend"] == "gloo"


def test_a_wrong_world_size_under_a_launcher_is_an_error():
    # under an external launcher the ranks are the launcher's: --gpus must agree with WORLD_SIZE
    r = _run(["--gpus", "2", "--backend", "gloo", "--rendezvous-only"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_roofline_of_prices_a_table_kernel_against_the_l2_and_a_product_against_the_matrix_cores():
    sys.path.insert(0, ROOT)
    import bench
    work = bench.kernel_work(65536, 512, 8, 256)
    kern = {"avg_ms": 0.35}
    pmc = {"traffic_bytes": 460_000_000, "l2_read_request_bytes": 9_210_000_000}
    r = bench.roofline_of("level1_combines_and_tables", kern, work["level1_combines_and_tables"], pmc, "test")
    assert r["bound"] == "l2" and abs(r["achieved"] - 9.21e9 / 0.35e-3 / 1e9) < 1 and r["peak"] == bench.PEAK_L2_GBPS
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["useful_frac"] < r["frac"]
    assert r["hbm"]["peak"] == bench.PEAK_HBM_GBPS and r["traffic"] == 460_000_000
    r = bench.roofline_of("level1_combines_and_tables", kern, work["level1_combines_and_tables"], None, None)
    assert r["bound"] == "hbm" and r["traffic"] is None
    r = bench.roofline_of("xc_product", {"avg_ms": 0.5}, work["xc_product"], None, None)
    assert r["bound"] == "mfma" and 0.4 < r["frac"] < 0.7
