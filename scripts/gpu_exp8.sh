#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "repacked or gemm_parity_auto or golden or fused or decode" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -5 gpurun_out/pytest_exp.log
for cfg in 0 1 2 3; do echo "== CDNA4_FUSED_CFG=$cfg"; CDNA4_FUSED_CFG=$cfg timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); dd=d['decode']; print('fused cold', dd['us_per_step'], 'warm', dd['us_per_step_cache_warm'], 'wall', dd['us_per_step_host_wall']); print({k:(v['gemm_b512_us'], v['decode_b1_us_cache_warm']) for k,v in d['formats'].items()})"; done
CDNA4_FUSED_CFG=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fused or decode or gemv_parity" 2>&1 | tail -2
