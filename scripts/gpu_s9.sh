#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for shape in "4096 4096 512" "4096 11008 512" "32768 8192 512"; do GB_VARIANTS=0,4119 GB_SPLITKS=0 GB_ROUNDS=4 timeout 200 tools/microbench/gemm_bench $shape 2>&1 | grep -E "variant|M=" ; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_ops.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/s9_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s9_pytest.log | head -20; tail -2 gpurun_out/s9_pytest.log
timeout 200 python bench.py --lean --steps 300 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench lean: step_us %.2f gemm_us %.2f value %.1f' % (d['ms_per_step']*1e3, d['roofline']['us_per_launch'], d['value']))"
