"""Per-kernel rocprofv3 averages of tools/ab_kernels.sh: the build in the tree against quantization_amd/lib/libmcq_prev.so (a build of another
commit: git stash / hipcc -o ../lib/libmcq_prev.so / git stash pop), interleaved twice in one gpurun call."""
import csv,glob
def load(d):
    f=glob.glob('/root/repo/gpurun_out/%s/*/*kernel_stats.csv'%d)[0]
    return {r['Name'].split('(')[0].replace('void mcq::','').replace('mcq::',''):(float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f)) if 'mcq' in r['Name']}
import os
runs={k:load(k) for k in sorted(d for d in os.listdir('/root/repo/gpurun_out') if d.startswith('abk_'))}
tot={r:0 for r in runs}
for k in sorted(set().union(*[set(v) for v in runs.values()])):
    if 'k_tf' in k or 'fgemm' in k or 'fix_rows<' in k:
        print('%-28s' % k[:28], '  '.join('%s %7.1f' % (r[4:], runs[r].get(k, float('nan'))) for r in runs))
        for r in runs: tot[r]+=runs[r].get(k,0)*(5 if k.startswith(('k_tf_comb','k_tf_level1','k_tf_pair0','k_tf_stage0')) else 1)
print({k:round(v,1) for k,v in tot.items()})
import statistics
print('mean now %.1f prev %.1f' % (statistics.mean(v for k,v in tot.items() if 'now' in k), statistics.mean(v for k,v in tot.items() if 'prev' in k)))
