"""hardware diagnosis #3: WHICH ingredient of k_convert_iq4_xs_q6_K2 makes its output differ from the same source run on the CPU (session 1 of round 4:
a few rows per conversion, always the top byte of the dword st32(da + 64 n + 32 + l, ..) — low nibble = a valid but WRONG codebook entry of code 3 of j = 1).
Runs the conversion alone, N times per variant (CDNA4_DIAG_CONV, convert_w.hip), each in its own process, and counts the bytes that differ."""
import os, sys, json, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refutil as R
    from ggml_amd import ops, native
    T, m, k = R.IQ4_XS, 4096, 4096
    L = native.lib()
    w = R.random_weights(T, m, k, seed=5 * m + k)
    a = ops.QTensor.from_host_bytes(T, k, m, w)
    cw = np.fromfile(sys.argv[2], np.uint8)
    rowb = cw.size // m
    buf = torch.empty(cw.size, dtype=torch.uint8, device="cuda")
    out = []
    for rep in range(int(sys.argv[3])):
        buf.fill_(0xEE if rep % 2 else 0x11)
        native.check(L.ggml_cdna4_convert_weights(int(T), a.data.data_ptr(), a.row_bytes, m, k, buf.data_ptr(), ops._stream(buf.device)))
        g = buf.cpu().numpy()
        bad = np.nonzero(g != cw)[0]
        out.append(int(bad.size))
        if bad.size and rep < 3:
            o = bad % rowb
            print(json.dumps({"var": os.environ.get("CDNA4_DIAG_CONV"), "rep": rep, "bad": int(bad.size), "offsets_mod_210": sorted(set((o % 210).tolist()))[:24], "stale_fill": int((g[bad] == (0xEE if rep % 2 else 0x11)).sum())}), flush=True)
    print(json.dumps({"var": os.environ.get("CDNA4_DIAG_CONV"), "bad_bytes_per_rep": out}), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refutil as R
m, k = 4096, 4096
w = R.random_weights(R.IQ4_XS, m, k, seed=5 * m + k)
with tempfile.TemporaryDirectory() as td:
    w.tofile(os.path.join(td, "w.bin"))
    r = subprocess.run([os.path.join(ROOT, "tools", "emul", "convert_emul"), str(int(R.IQ4_XS)), str(m), str(k), os.path.join(td, "w.bin"), os.path.join(td, "o.bin")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    for var in (sys.argv[1:] or ["0", "1", "2", "4", "3", "7"]):
        env = dict(os.environ, CDNA4_DIAG_CONV=var, CDNA4_DIAG_CONVERT_ANY="1")
        rr = subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(td, "o.bin"), "12"], env=env, capture_output=True, text=True, timeout=600)
        print(rr.stdout.strip() or rr.stderr[-400:], flush=True)
