#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "repacked or gemm_parity_auto or golden or full_size" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -8 gpurun_out/pytest_exp.log
for v in "A=1" "CDNA4_NO_STAGED=1"; do echo "== $v"; env $v timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:(v['gemm_b512_us']) for k,v in d['formats'].items()})"; done
