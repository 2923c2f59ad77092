"""batch sweep on DEVICE time: us per ggml_cdna4_mul_mat call (activation quantizer included) for 2 .. 64 activation rows, measured by
replaying a HIP graph of 40 calls (no host time between launches), per weight format and matrix, under three routings — AUTO, int8 matrix-core kernel off
(CDNA4_NO_MMQ=1: GEMV units up to 8 rows, fp16 GEMM above), int8 matrix-core kernel forced for 2 .. 64 rows (CDNA4_MMQ_MINB=2 CDNA4_MMQ_MAXB=64).  One child process
per routing (the knobs are read once).  python scripts/gpu_batch_sweep.py > profiles/r04/batch_sweep.txt"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = (2, 3, 4, 6, 8, 9, 12, 16, 24, 32, 48, 64)
SHAPES = ((4096, 4096), (4096, 14336), (3072, 768))
TYPES = (12, 14, 2, 8, 13)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import bench as B
    from ggml_amd import native, ops
    native.lib(); dev = torch.device("cuda", 0)
    out = {}
    for t in TYPES:
        for (m, k) in SHAPES:
            a = ops.QTensor.from_host_bytes(t, k, m, B.synth_blocks(t, m, k, 7), device=dev)
            row = {}
            for b in BS:
                x = torch.from_numpy(np.random.default_rng(b).uniform(-1, 1, (b, k)).astype(np.float32)).to(dev)
                y = torch.empty((b, m), dtype=torch.float32, device=dev)
                for _ in range(5): ops.mul_mat(a, x, out=y)
                torch.cuda.synchronize()
                s2 = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(s2):
                    for _ in range(3): ops.mul_mat(a, x, out=y)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s2):
                    for _ in range(40): ops.mul_mat(a, x, out=y)
                row[b] = round(B.events_us(g.replay, 8, 3) / 40, 2)
                del g
            out["%s %dx%d" % (B.TYPE_NAME[t], m, k)] = row
    print(json.dumps(out))
    sys.exit(0)
res = {}
for name, env in (("auto", {}), ("no_mmq", {"CDNA4_NO_MMQ": "1"}), ("mmq_2_64", {"CDNA4_MMQ_MINB": "2", "CDNA4_MMQ_MAXB": "64"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    res[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-600:]}
print("# us per ggml_cdna4_mul_mat call (quantizer included), HIP-graph replay of 40 calls; rows = activation rows; columns: AUTO | int8 matrix-core kernel off | forced")
for key in res["auto"]:
    print(key)
    for b in BS:
        cells = [res[n].get(key, {}).get(str(b)) for n in ("auto", "no_mmq", "mmq_2_64")]
        print("  %3d rows: %8s | %8s | %8s" % (b, *cells))
print(json.dumps(res))
