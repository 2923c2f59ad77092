#!/bin/bash
# round 3, session 4: IQ4_XS route-vs-data diagnosis, GGUF end to end, decode timings through bench.py's own harness
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python scripts/gpu_diag_iq4xs.py > gpurun_out/s4_iq4xs.log 2>&1; echo "iq4xs rc=$?"; grep "^{" gpurun_out/s4_iq4xs.log | cut -c1-600; tail -3 gpurun_out/s4_iq4xs.log | grep -v "^{" | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gguf.py tests/test_gpu_gpt2.py -m gpu -q --tb=short -p no:cacheprovider -k "c5_legs or gguf" > gpurun_out/s4_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|^E " gpurun_out/s4_pytest.log | head -20; tail -2 gpurun_out/s4_pytest.log
grep -h "gguf" gpurun_out/parity_report.jsonl gpurun_out/gpt2_parity.jsonl 2>/dev/null | tail -3
timeout 300 python - <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
from ggml_amd import native; native.lib()
print(json.dumps(bench.decode_rows(dev, 200)))
PY
