#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_widening.py -m gpu -q --tb=short -p no:cacheprovider -k "int8_matrix_cores or more_formats_prefill_gemm or iq4_nl_reencoding or small_batch or stock_harness" > gpurun_out/s18_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s18_pytest.log | head; tail -2 gpurun_out/s18_pytest.log
for shape in "4096 14336" "4096 4096"; do
  timeout 100 python scripts/batch_rows.py q6_K $shape | tail -1
  CDNA4_NO_MMQ=1 timeout 100 python scripts/batch_rows.py q6_K $shape | tail -1
done
