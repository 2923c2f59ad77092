"""decode: the four wave / row configurations of k_gemv_q_fused (CDNA4_FUSED_CFG 0 = 4 waves x 1 row, 1 = 8 x 2, 2 = 4 x 2, 3 = 8 x 1) per matrix, cold and cache-warm, HIP-graph replay"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import bench as B
    from ggml_amd import native, ops
    L = native.lib(); dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for t in (12, 14, 2):
        for (m, k) in ((4096, 4096), (4096, 14336), (4096, 11008), (4096, 8192), (14336, 4096), (11008, 4096)):
            w = B.synth_blocks(t, m, k, 7)
            a = ops.QTensor.from_host_bytes(t, k, m, w, device=dev)
            mat_b = a.row_bytes * m
            ncopy = max(2, int(600e6 // mat_b))
            big = a.data.reshape(-1).repeat(ncopy)
            x1 = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (1, k)).astype(np.float32)).to(dev)
            y1 = torch.empty((1, m), dtype=torch.float32, device=dev)
            ws = torch.empty(L.ggml_cdna4_mul_mat_workspace_size(t, k, 1), dtype=torch.uint8, device=dev)
            cnt = [0]
            def cold():
                i = cnt[0] % ncopy; cnt[0] += 1
                native.check(L.ggml_cdna4_mul_mat(t, big.data_ptr() + i * mat_b, a.row_bytes, x1.data_ptr(), k, y1.data_ptr(), m, m, k, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, st))
            def warm():
                native.check(L.ggml_cdna4_mul_mat(t, big.data_ptr(), a.row_bytes, x1.data_ptr(), k, y1.data_ptr(), m, m, k, 1, ws.data_ptr(), ws.numel(), 0, 0, 0, st))
            out["%s %dx%d" % (B.TYPE_NAME[t], m, k)] = (round(B.events_us(cold, 300, 20), 2), round(B.events_us(warm, 300, 20), 2))
            del big
    print(json.dumps(out))
    sys.exit(0)
res = {}
for cfg in ("1", "3", "0", "2"):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, CDNA4_FUSED_CFG=cfg), capture_output=True, text=True, timeout=600)
    res[cfg] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
print("# us per decode step (cold / cache-warm), CDNA4_FUSED_CFG: 1 = 8 waves x 2 rows (default at M >= 4096), 3 = 8 x 1, 0 = 4 x 1, 2 = 4 x 2")
for key in res["1"]:
    print("%-18s" % key, {c: res[c].get(key) for c in res})
