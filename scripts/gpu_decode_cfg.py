"""decode A/B on hardware: the one-launch decode kernel k_gemv_q_fused under CDNA4_FUSED_CFG (1 = 8 waves x 2 rows, 3 = 8 x 1, 4 = 16 x 1; default = the launcher's choice),
cold (rotating copies of W beyond the Infinity Cache) and cache-warm, HIP events.  One child process per setting (the knob is read once), settings alternating.
python scripts/gpu_decode_cfg.py [types] e.g. 12,14"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((4096, 4096), (4096, 14336), (4096, 11008), (4096, 8192), (14336, 4096), (11008, 4096))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import bench as B
    from ggml_amd import native, ops
    L = native.lib(); dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for t in [int(v) for v in sys.argv[2].split(",")]:
        for (m, k) in SHAPES:
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
            c, wm = min(B.events_us(cold, 300, 20) for _ in range(2)), min(B.events_us(warm, 300, 20) for _ in range(2))
            warm(); torch.cuda.synchronize()
            out["%s %dx%d" % (B.TYPE_NAME[t], m, k)] = {"cold_us": round(c, 2), "warm_us": round(wm, 2), "y_bits": int(y1.view(torch.int32).to(torch.int64).sum().item())}
            del big
    print(json.dumps(out))
    sys.exit(0)
types = sys.argv[1] if len(sys.argv) > 1 else "12"
cfgs = os.environ.get("CFGS", "default,1,3,4").split(",")
res = {}
for rep in range(2):
    for cfg in cfgs:
        env = dict(os.environ)
        if cfg != "default":
            env["CDNA4_FUSED_CFG"] = cfg
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", types], env=env, capture_output=True, text=True, timeout=150)
            res[(cfg, rep)] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-400:]
        except subprocess.TimeoutExpired:
            res[(cfg, rep)] = "timeout"
ok = [k for k in res if isinstance(res[k], dict)]
for name in res[ok[0]]:
    print("%-18s" % name, "  ".join("%s: %s" % (c, "/".join("%.2f|%.2f" % (res[(c, r)][name]["cold_us"], res[(c, r)][name]["warm_us"]) for r in range(2) if isinstance(res.get((c, r)), dict))) for c in cfgs),
          " same bits:", len({res[k][name]["y_bits"] for k in ok}) == 1)
for k in res:
    if not isinstance(res[k], dict): print(k, res[k])
