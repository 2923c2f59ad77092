"""decode A/B on hardware: k_gemv_q_fused register form vs its DMA form (CDNA4_DECODE_DMA), cold (rotating copies of W > Infinity Cache) and cache-warm,
HIP events.  One child process per setting (the knob is read once).  python scripts/gpu_decode_ab.py"""
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
    for t in (12, 13):
        for (m, k) in ((4096, 4096), (4096, 14336), (4096, 11008), (4096, 8192), (11008, 4096)):
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
            c, wm = B.events_us(cold, 300, 20), B.events_us(warm, 300, 20)
            warm(); torch.cuda.synchronize()
            out["%s %dx%d" % (B.TYPE_NAME[t], m, k)] = {"cold_us": round(c, 3), "warm_us": round(wm, 3), "GBps_cold": round((mat_b + 4 * k + 4 * m) / c / 1e3, 1), "y_sum": float(y1.double().sum().item())}
            del big
    print(json.dumps(out))
    sys.exit(0)
res = {}
for dma in ("0", "default", "1"):
    env = dict(os.environ)
    if dma != "default":
        env["CDNA4_DECODE_DMA"] = dma
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=900)
    res[dma] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-800:]
for k in res["0"]:
    print(k, {d: (res[d][k]["cold_us"], res[d][k]["warm_us"]) for d in res if isinstance(res[d], dict)}, "same result:", len({res[d][k]["y_sum"] for d in res if isinstance(res[d], dict)}) == 1)
print(json.dumps(res))
