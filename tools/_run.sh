set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python bench.py > gpurun_out/r05/bench_a.json 2> gpurun_out/r05/bench_a.err; tail -3 gpurun_out/r05/bench_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_a.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step'],'solo',d['solo_latency_ms'])
print('host',d['end_to_end_host_scalars']['ms'],d['end_to_end_host_scalars']['pipelined_ms'])
print('dropin',{k:v for k,v in d['dropin'].items() if 'note' not in k})
print('cold',d['cold_path']['first_10_opens_ms'],d['cold_path']['fold_table_status_after_each'])
o=d['pcdl_open_check']; print('open+check',o['ms'],'without',o['without_fold_table_ms'],'check',o['check_alone_ms'],'host poly',o['end_to_end_host_polynomial_ms'])
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['ns_per_field_product'],d['cpu_baseline']['build'])
print('asdl',{k:v for k,v in d['asdl_chain'].items() if isinstance(v,(int,float))})
PY
