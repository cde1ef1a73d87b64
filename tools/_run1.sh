cd /root/repo
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=r03e
tools/prof_bench.sh ${tag} || exit 1
tools/prof_bench.sh ${tag}_config2 --config 2 || exit 1
tools/prof_bench.sh ${tag}_config4 --config 4 || exit 1
tools/trace_chain.sh ${tag}_8 8 || exit 1
tools/trace_chain.sh ${tag}_4 4 || exit 1
tools/trace_chain.sh ${tag}_1 1 || exit 1
for c in 3 2 4 5 1; do
  timeout -k 10 400 python bench.py --config $c > $O/${tag}_bench_config$c.json 2> $O/${tag}_bench_config$c.err || { tail -5 $O/${tag}_bench_config$c.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/${tag}_bench_config$c.json')); r=d['roofline']; print('config $c', d['value'], d['unit'], r['kernel'], r['kernel_ms'], 'frac', r['frac'], 'valu', (d.get('roofline_valu') or {}).get('frac'), (d.get('roofline_valu') or {}).get('frac_with_scalar_instructions'))"
done
