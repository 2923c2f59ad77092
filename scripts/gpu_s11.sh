#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "int8_matrix or quantize or small_batch" > gpurun_out/s11_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s11_pytest.log | head; tail -2 gpurun_out/s11_pytest.log
for e in "CDNA4_NO_MMQ=1" "CDNA4_MMQ_MINB=2 CDNA4_MMQ_MAXB=64"; do
env $e timeout 400 python - <<'PY'
import json, os, torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
from ggml_amd import native; native.lib()
print({k: os.environ.get(k) for k in ("CDNA4_NO_MMQ", "CDNA4_MMQ_MAXB")}, "batch_sweep", json.dumps(bench.batch_sweep(dev, 100)))
PY
done
timeout 200 python bench.py --lean --steps 300 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench lean: step_us %.2f gemm_us %.2f value %.1f' % (d['ms_per_step']*1e3, d['roofline']['us_per_launch'], d['value']))"
