#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or golden or full_size or repacked" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -5 gpurun_out/pytest_exp.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_exp.log 2>&1; tail -1 gpurun_out/bench_exp.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'], 'gemm', d['roofline']['us_per_launch'], d['roofline']['frac']); print({k:(v['gemm_b512_us'], v['decode_b1_us_cache_warm']) for k,v in d['formats'].items()})"
