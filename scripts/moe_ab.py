"""grouped MUL_MAT_ID prefill (8 experts x 2 used x 512 tokens x 4096^2 Q4_K) under the environment of the process: bench.moe_row, printed with a tag"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
from ggml_amd import native
native.lib()
r = bench.moe_row(dev, 150)
print(json.dumps({"tag": os.environ.get("AB_TAG", ""), "prefill_us": r["prefill_512_tokens"]["us_per_call"], "decode_us": r["decode_1_token"]["us_per_call"]}), flush=True)
