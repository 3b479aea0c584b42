#!/bin/bash
# Per-kernel times of the pair stages under the MCQ_PAIR_ABL timing ablations.  The ablated kernels return WRONG
# codes by design, so they are compiled only with -DMCQ_ABLATE: this script builds that variant over the product
# library, runs the bench per ablation, and rebuilds the product library afterwards.
set -e
cd "$(dirname "$0")/.."
LIB=quantization_amd/lib/libmcq_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DMCQ_ABLATE quantization_amd/csrc/mcq_api.hip -o $LIB
trap '/opt/rocm/bin/hipcc $FLAGS quantization_amd/csrc/mcq_api.hip -o $LIB' EXIT
for a in ${ABLS:-0 1 2 3 4 5 6}; do
  export MCQ_PAIR_ABL=$a
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['kernels']
print('ABL', os.environ['MCQ_PAIR_ABL'], ' '.join('%s %.3f' % (n, k[n]['avg_ms']) for n in k if n.startswith('pair')))"
done
