mkdir -p gpurun_out/r05
{ echo "== solo MSM 2^20 (tools/solo_msm.py 20 40)"; bash tools/ab.sh "tools/solo_msm.py 20 40" base pf2
  echo "== pipelined (tools/pipe_loop.py 20 4)"; bash tools/ab.sh "tools/pipe_loop.py 20 4" base pf2
  echo "== open (tools/open_loop.py 20 15)"; FOLD_TABLE=1 bash tools/ab.sh "tools/open_loop.py 20 15" base pf2; } > gpurun_out/r05/ab_reduce_rc_prefetch.txt 2>&1
cat gpurun_out/r05/ab_reduce_rc_prefetch.txt
