#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for shape in "4096 4096 512" "4096 11008 512" "32768 8192 512"; do GB_VARIANTS=0,4119 GB_SPLITKS=0 GB_ROUNDS=4 timeout 200 tools/microbench/gemm_bench $shape 2>&1 | grep -E "variant|M=" ; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "int8_matrix or small_batch or mul_mat_id_parity" > gpurun_out/s8_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s8_pytest.log | head -20; tail -2 gpurun_out/s8_pytest.log
timeout 400 python - <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
from ggml_amd import native; native.lib()
print("batch_sweep", json.dumps(bench.batch_sweep(dev, 100)))
PY
CDNA4_NO_MMQ=1 timeout 400 python - <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
from ggml_amd import native; native.lib()
print("batch_sweep NO_MMQ", json.dumps(bench.batch_sweep(dev, 100)))
PY
