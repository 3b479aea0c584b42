"""Invariants of hand-counted waits, checked on the gfx950 ISA the compiler emits (no GPU needed: hipcc cross-compiles).

k_decode_blk waits for the codes of its next block with `s_waitcnt vmcnt(TRIPS)`: the LDS-DMA piece was requested before
the block's TRIPS stores, and loads and stores share one in-order counter on gfx9.  The wait is right only if the main
loop issues EXACTLY TRIPS vector-memory instructions after the DMA request -- one global_store_dwordx4 per trip, no
compiler-made loads, no scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "mcq.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-w", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "quantization_amd", "csrc", "mcq_api.hip"),
                           "-o", str(out)])
    return open(out).read().split("\n")


@pytest.mark.parametrize("N,LPV", [(4, 4), (4, 2), (8, 4), (8, 2), (16, 4), (16, 2)])
def test_decode_blk_counted_wait_matches_the_stores_of_a_block(isa, N, LPV):
    trips = 16 * LPV // N
    pref = f"_ZN3mcq12k_decode_blkILi{N}ELi{LPV}EEE"
    start = next(i for i, l in enumerate(isa) if l.startswith(pref) and ":" in l)
    end = next(i for i in range(start, len(isa)) if "s_endpgm" in isa[i])
    body = isa[start:end]
    assert not any("scratch_" in l for l in body)
    waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\(\d+\)", l) and any("s_barrier" in x for x in body[i:i + 3])
             and "vmcnt(0)" not in l]
    assert len(waits) == 1 and f"vmcnt({trips})" in body[waits[0]]
    dmas = [i for i, l in enumerate(body) if "global_load_lds_dwordx4" in l]
    stores = [i for i, l in enumerate(body) if "global_store_dword" in l]
    loads = [i for i, l in enumerate(body) if re.search(r"global_load_dword|buffer_load|flat_load", l) and "lds" not in l]
    # DMA requests: the first block's codes, the row slices (prologue), the next block's codes (main loop)
    assert len(dmas) == 3
    loop_dma = dmas[-1]
    # the main loop's stores follow its DMA request; the partial-block path (one load, one store) comes after them
    main_stores = [i for i in stores if loop_dma < i and (not loads or i < loads[0])]
    assert len(main_stores) == trips and all("global_store_dwordx4" in body[i] for i in main_stores)
    assert len(loads) == 1 and len(stores) == trips + 1 and loads[0] > main_stores[-1]


@pytest.mark.parametrize("mode", [0, 1])
def test_product_kernel_steps_request_four_pieces_and_wait_for_all_but_eight(isa, mode):
    """k_fgemm (mcq_fix_kernels.h): every ring stage is 4 LDS-DMA pieces per wave, requested through inline asm that writes
    m0 behind the compiler's back, and the sync of a step is `s_waitcnt vmcnt(8)` = "only my pieces of the two later stages
    outstanding".  That count is right only if the main loop issues nothing else on the vector-memory counter (no spill, no
    compiler-made load between the pieces), and m0 may be written by nothing but those asm statements."""
    pref = f"_ZN3mcq7k_fgemmILi{mode}EEE"
    start = next(i for i, l in enumerate(isa) if l.startswith(pref) and ":" in l)
    end = next(i for i in range(start, len(isa)) if "s_endpgm" in isa[i])
    body = [l.split(";")[0].strip() for l in isa[start:end]]
    body = [l for l in body if l]
    # nothing else that reads or writes m0: register-indexed moves, GWS, the LDS-DMA builtin's own bookkeeping
    assert not any(re.search(r"s_set_gpr_idx|v_movrel|ds_gws|s_movrel", l) for l in body)
    m0_writes = [l for l in body if re.search(r"\bm0\b", l) and not l.startswith("s_mov_b32 m0,")]
    assert not m0_writes, m0_writes[:3]
    n_m0 = sum(1 for l in body if l.startswith("s_mov_b32 m0,"))
    n_dma = sum(1 for l in body if l.startswith("global_load_lds_dword"))
    assert n_m0 == n_dma and n_dma >= 16 + 8          # prologue (info + four stages) and the two unrolled steps
    # the main loop: between two consecutive step barriers that are preceded by the counted wait there are exactly four
    # pieces, twenty MFMAs, and no other vector-memory instruction
    syncs = [i for i, l in enumerate(body) if l == "s_barrier" and any("vmcnt(8)" in x for x in body[max(0, i - 4):i])]
    assert len(syncs) == 2, len(syncs)
    seg = body[syncs[0]:syncs[1]]
    assert sum(1 for l in seg if l.startswith("global_load_lds_dwordx4")) == 4
    assert sum(1 for l in seg if l.startswith("v_mfma_i32_32x32x32_i8")) >= 17      # (a few are hoisted above the barrier)
    assert not any(re.search(r"^(global_load_dword|global_store|buffer_|scratch_|flat_)", l) and "lds" not in l for l in seg)
    assert sum(1 for l in body if l.startswith("v_mfma_i32_32x32x32_i8")) == 40
