#!/bin/bash
# experiment session: GEMM scheduling variants + fused decode GEMV
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench.txt
GB_VARIANTS="23,119,151,407,663,919" timeout 180 tools/microbench/gemm_bench 4096 4096 512 407,919 >> gpurun_out/gemm_bench.txt 2>&1
GB_VARIANTS="23,119,407,919" timeout 120 tools/microbench/gemm_bench 8192 4096 512 2>&1 | grep -E "^variant .* splitk 1" >> gpurun_out/gemm_bench.txt
cat gpurun_out/gemm_bench.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemv or fused or decode or golden or gemm_variants" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -5 gpurun_out/pytest_exp.log
for v in "A=1" "CDNA4_FUSE_NOPREFETCH=1" "CDNA4_NO_FUSE=1"; do echo "== $v"; env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['us_per_launch'], json.dumps(d['decode']))"; done
