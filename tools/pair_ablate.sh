#!/bin/bash
# per-kernel times of the pair stages under the MCQ_PAIR_ABL timing ablations (results are wrong by design)
for a in ${ABLS:-0 1 2 3 4 5 6}; do
  export MCQ_PAIR_ABL=$a
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['kernels']
print('ABL', os.environ['MCQ_PAIR_ABL'], ' '.join('%s %.3f' % (n, k[n]['avg_ms']) for n in k if n.startswith('pair')))"
done
