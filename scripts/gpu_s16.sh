#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_widening.py -m gpu -q --tb=short -p no:cacheprovider -k "more_formats_prefill_gemm or iq4_nl_reencoding or q4_1_q5_1_iq4_nl_prefill or gpt2" > gpurun_out/s16_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s16_pytest.log | head; tail -2 gpurun_out/s16_pytest.log
for cfg in 1 3 1 3; do echo -n "CDNA4_FUSED_CFG=$cfg: "; CDNA4_FUSED_CFG=$cfg timeout 100 python scripts/decode_loop.py 4096 14336 600 | tail -1; done
bash scripts/gpu_prof_fa.sh
