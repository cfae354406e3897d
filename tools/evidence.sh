#!/bin/bash
# Round evidence on the GPU box: parity tests, bench line, rocprofv3 kernel stats (4 launches in flight, 1 in
# flight, with open+check), PMC traffic passes, N>1 rehearsals (gloo ranks sharing the one GPU), ASDL chain.
# Usage (from the repo root, on the GPU box): bash tools/evidence.sh [A|B] ; everything lands in gpurun_out/ev/
# (A = tests, bench line, kernel statistics and PMC passes; B = n = 2^24, the N > 1 rehearsals and the ASDL chain: two calls
# where one would not fit a gpurun time limit; no argument = both)
set -o pipefail
PART=${1:-all}
OUT=gpurun_out/ev
if [ "$PART" != B ]; then rm -rf $OUT; fi
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "== $1" | tee -a $OUT/progress.txt; }
if [ "$PART" != B ]; then
step pytest;  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -5 $OUT/pytest_gpu.txt; exit 1; }
tail -1 $OUT/pytest_gpu.txt
step bench;   timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
step prof_d4; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_d4 -- python bench.py --steps 40 --warmup 4 --cpu-msms 0 --open-steps 0 --asdl-steps 0 --host-steps 0 --fr-reps 0 --var-steps 0 --min-seconds 0 > $OUT/prof_d4.log 2>&1 || exit 1
step prof_d1; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_d1 -- python bench.py --steps 40 --warmup 4 --depth 1 --cpu-msms 0 --open-steps 0 --asdl-steps 0 --host-steps 0 --fr-reps 0 --var-steps 0 --min-seconds 0 > $OUT/prof_d1.log 2>&1 || exit 1
step prof_open; FOLD_TABLE=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_open -- python3 tools/open_loop.py 20 5 > $OUT/prof_open.log 2>&1 || exit 1
python tools/trace_timeline.py $(find $OUT/prof_open -name "*kernel_trace.csv" | head -1) > $OUT/open_timeline.txt 2>&1
step pmc_fetch; timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python tools/pipe_loop.py 20 1 6 > $OUT/pmc_fetch.log 2>&1 || exit 1
step pmc_write; timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python tools/pipe_loop.py 20 1 6 > $OUT/pmc_write.log 2>&1 || exit 1
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json || exit 1
# the same two passes for the table-free (variable-base) pipeline: halo_set_table_mode(ctx, 0)
step pmc_fetch_gen; TABLE_MODE=0 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_gen -- python tools/pipe_loop.py 20 1 6 > $OUT/pmc_fetch_gen.log 2>&1 || exit 1
step pmc_write_gen; TABLE_MODE=0 timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_gen -- python tools/pipe_loop.py 20 1 6 > $OUT/pmc_write_gen.log 2>&1 || exit 1
python tools/pmc_summary.py $OUT/pmc_fetch_gen $OUT/pmc_write_gen "TABLE_MODE=0 python tools/pipe_loop.py 20 1 6" > $OUT/pmc_traffic_general.json || exit 1
step prof_d1_gen; TABLE_MODE=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_d1_gen -- python tools/pipe_loop.py 20 1 40 > $OUT/prof_d1_gen.log 2>&1 || exit 1
# what the SIMDs do during the integer kernels: issue slots used, clock held, parked wave cycles (tools/sq_summary.py)
step pmc_sq; timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python tools/pipe_loop.py 20 1 6 > $OUT/pmc_sq.log 2>&1 || exit 1
python tools/sq_summary.py $OUT/pmc_sq "python tools/pipe_loop.py 20 1 6" "one MSM in flight (n = 2^20)" > $OUT/sq_msm.json || exit 1
# the same counters over the kernels of an open + check loop (fold table in place from the second open on)
step pmc_sq_open; FOLD_TABLE=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_open -- python3 tools/open_loop.py 20 4 > $OUT/pmc_sq_open.log 2>&1 || exit 1
python tools/sq_summary.py $OUT/pmc_sq_open "FOLD_TABLE=1 python3 tools/open_loop.py 20 4" "four pcdl::open + check pairs at n = 2^20, one at a time, fold table requested at the first open" > $OUT/sq_open.json || exit 1
# the bandwidth-side Fr kernels alone (K4-K9): steady-state durations and HBM-side bytes, each launch back to back
step fr_trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fr_trace -- python3 tools/fr_kernels.py 20 20 > $OUT/fr_trace.log 2>&1 || exit 1
step fr_fetch; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fr_fetch -- python3 tools/fr_kernels.py 20 6 > $OUT/fr_fetch.log 2>&1 || exit 1
step fr_write; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/fr_write -- python3 tools/fr_kernels.py 20 6 > $OUT/fr_write.log 2>&1 || exit 1
python tools/pmc_summary.py $OUT/fr_fetch $OUT/fr_write "python3 tools/fr_kernels.py 20 6" > $OUT/pmc_fr.json || exit 1
python tools/fr_kernels.py 20 20 > $OUT/fr_kernels_events.json 2>/dev/null || exit 1
( cd tools && ./fr29_bench > ../$OUT/fr29_bench.txt 2>&1 ) || true
# HBM-side bytes of the kernels of an open (the comb-table fold among them)
step open_fetch; FOLD_TABLE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/open_fetch -- python3 tools/open_loop.py 20 4 > $OUT/open_fetch.log 2>&1 || exit 1
step open_write; FOLD_TABLE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/open_write -- python3 tools/open_loop.py 20 4 > $OUT/open_write.log 2>&1 || exit 1
python tools/pmc_summary.py $OUT/open_fetch $OUT/open_write "FOLD_TABLE=1 python3 tools/open_loop.py 20 4" > $OUT/pmc_open_kernels.json || exit 1
# halo_msm from pageable host memory: one copy + one launch sequence against the default stretches (HALO_HOST_SPLIT)
step host_msm; for cfg in 16 4,12; do HALO_HOST_SPLIT=$cfg timeout -k 10 120 python tools/host_msm.py 20 30 2>/dev/null | grep lg= >> $OUT/host_msm.txt; done
fi
if [ "$PART" = A ]; then step done_A; exit 0; fi
# BASELINE config 5's size on the one GPU: n = 2^24, the bench line (rate with four in flight, solo latency, roofline), and the
# same MSM through the multi-device context with the one GPU standing in for eight
step bench_2_24; timeout -k 10 500 python bench.py --log-n 24 --steps 24 --warmup 2 --cpu-msms 0 --open-steps 0 --asdl-steps 0 --host-steps 0 --fr-reps 0 --var-steps 0 --min-seconds 0.5 > $OUT/bench_2_24.json 2> $OUT/bench_2_24.err || exit 1
step oneproc8_2_24; timeout -k 10 500 python bench.py --log-n 24 --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 16 --warmup 2 --min-seconds 0.3 > $OUT/bench_oneproc8_2_24.json 2> $OUT/bench_oneproc8_2_24.err || exit 1
# multi-device context from ONE process (halo_ctx_create_urs_multi), the one GPU of this box standing in for every device
for N in 2 4 8; do
  step oneproc$N; timeout -k 10 300 python bench.py --gpus $N --devices $(python -c "print(','.join(['0']*$N))") --steps 64 --min-seconds 0.3 > $OUT/bench_oneproc$N.json 2> $OUT/bench_oneproc$N.err || exit 1
done
# the N > 1 path with its collectives on RCCL, as far as one GPU goes: ONE rank that still runs every all-gather
step rccl1; HALO_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29540 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 400 python bench.py --steps 64 --warmup 8 --cpu-msms 0 --asdl-steps 0 --host-steps 0 --fr-reps 0 --min-seconds 0.3 2> $OUT/bench_rccl1.err | grep '^{' > $OUT/bench_rccl1.json || exit 1
export HALO_BENCH_BACKEND=gloo
for N in 2 4; do
  step gloo$N; timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 64 --warmup 8 2> $OUT/bench_gloo$N.err | grep '^{' > $OUT/bench_gloo$N.json || exit 1
done
# (the default of N > 1 is index blocks through the table plan; the other cut, Pippenger windows, explicitly)
step gloo2_window; timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 64 --warmup 8 --shard window 2> $OUT/bench_gloo2_window.err | grep '^{' > $OUT/bench_gloo2_window.json || exit 1
# BASELINE config 5 shape: n = 2^24 in index shards (two ranks sharing this GPU: 2^23 points each)
step gloo2_index_2_24; timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 2 --log-n 24 --steps 12 --warmup 2 --shard index --min-seconds 0 2> $OUT/bench_gloo2_index_2_24.err | grep '^{' > $OUT/bench_gloo2_index_2_24.json || exit 1
unset HALO_BENCH_BACKEND
step asdl64;  timeout -k 10 600 python tools/time_acc.py 20 64 > $OUT/asdl64.json 2> $OUT/asdl64.err || exit 1
# keep only the summaries (traces are large)
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete
step done
