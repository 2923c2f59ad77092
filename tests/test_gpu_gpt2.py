"""-m gpu, BASELINE config #4: the reference's own examples/gpt-2 graph (main-backend.cpp, included unmodified by
oracle/gpt2_harness.cpp) on a synthetic 117M-shaped model quantized to Q4_0 by the reference's gpt-2-quantize,
evaluated on the reference CPU backend and on our plug-in; logits must agree to rel-L2 <= 1e-3 at every step
(prompt batch -> MFMA GEMM path, single-token steps -> int8-dot GEMV path)."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import torch
import refutil as R

pytestmark = pytest.mark.gpu
REF = R.REF_DIR
PLUGIN = os.path.join(R.ROOT, "ggml_amd", "lib", "libggml-cdna4.so")
N_VOCAB = 50257


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    for f in ("gpt-2-quantize", "gpt2_harness"):
        if not os.path.exists(os.path.join(REF, f)):
            pytest.fail("prebuilt oracle/_ref/%s missing from the snapshot" % f)
    d = tmp_path_factory.mktemp("gpt2")
    f32, q4 = str(d / "f32.bin"), str(d / "q4_0.bin")
    subprocess.run([sys.executable, os.path.join(R.ROOT, "tools", "make_synth_gpt2.py"), f32], check=True, timeout=600)
    subprocess.run([os.path.join(REF, "gpt-2-quantize"), f32, q4, "q4_0"], check=True, timeout=600, capture_output=True)
    os.remove(f32)
    return q4, str(d)


def _run(model_path, backend, out, n_prompt, n_decode):
    r = subprocess.run([os.path.join(REF, "gpt2_harness"), model_path, backend, PLUGIN if backend != "CPU" else "-", out, str(n_prompt), str(n_decode), "16"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), np.fromfile(out, np.float32).reshape(-1, N_VOCAB)


@pytest.mark.parametrize("n_prompt,n_decode", [(8, 6), (64, 4), (200, 2)])
def test_gpt2_logits_match_cpu_backend(model, n_prompt, n_decode):
    q4, d = model
    tc, lc = _run(q4, "CPU", os.path.join(d, "cpu.bin"), n_prompt, n_decode)
    tg, lg = _run(q4, "CDNA40", os.path.join(d, "gpu.bin"), n_prompt, n_decode)
    assert "CDNA4" in tg["backend"] and lc.shape == lg.shape == (1 + n_decode, N_VOCAB)
    errs = [R.rel_l2(lg[i], lc[i]) for i in range(lc.shape[0])]
    os.makedirs(os.path.join(R.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(R.ROOT, "gpurun_out", "gpt2_parity.jsonl"), "a") as f:
        f.write(json.dumps({"n_prompt": n_prompt, "n_decode": n_decode, "rel_l2_per_step": errs, "cpu": tc, "gpu": tg,
                            "argmax_agree": [int(np.argmax(lg[i]) == np.argmax(lc[i])) for i in range(lc.shape[0])]}) + "\n")
    assert np.isfinite(lg).all()
    assert max(errs) < 1e-3, errs
