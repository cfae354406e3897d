mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/pytest_gpu_2.txt 2>&1; tail -5 gpurun_out/r05/pytest_gpu_2.txt
HALO_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 50 --host-steps 0 --var-steps 0 --asdl-steps 0 --cpu-msms 0 --fr-reps 0 --concurrent-opens 0 > gpurun_out/r05/bench_rccl_one_rank.json 2> gpurun_out/r05/bench_rccl_one_rank.err; tail -3 gpurun_out/r05/bench_rccl_one_rank.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_rccl_one_rank.json').read().strip().splitlines()[-1])
print({k:v for k,v in d.get('pcdl_open_check_collective_path',{}).items() if k!='note'})
PY
