timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_scalar_msm or msm_2_20" 2>&1 | tail -15
