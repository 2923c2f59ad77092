#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'], d['roofline'].get('traffic'))
print(json.dumps(d.get('mul_mat_id'), indent=1))
print(json.dumps(d.get('stock_test_backend_ops_perf', {}).get('q4_K_cpu_backend'), indent=1)[:1500])
print({k: (v if not isinstance(v, dict) else '...') for k, v in d.items() if k not in ('roofline','cpu_baseline','decode','shapes','formats','stock_test_backend_ops_perf','mul_mat_id','config')})
PY
