import json, glob, sys
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d = json.load(open(f)); k = d['kernels']
        print(f.split('/')[-1], round(d['value'] / 1e6, 3), d['parity']['bit_exact'], *[k[n]['avg_ms'] for n in ('logits_product_argmax', 'xc_product', 'stage0_tables', 'combine_level0', 'combine_level1', 'tables_level1', 'combine_level2')])
    except Exception as e:
        print(f, 'ERR', e)
