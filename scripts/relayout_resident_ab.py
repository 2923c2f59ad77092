"""Q4_0 / Q8_0 prefill (AB_TYPE=2 | 8) with and without a resident Q4_0R / Q8_0R image, on ONE box, alternating: the per-call route (k_gemm_kq_w12 with loader waves that
re-lay the 18- / 34-byte blocks) against k_gemm_kq_t64 / k_gemm_r8 on the image.  One JSON line per shape: GEMM-only time per call (HIP-graph replay, activations pre-quantized: ggml_cdna4_mul_mat_prepared), the
routes, the relative difference of the two results, and the oracle-independent check that the image route equals the per-call route within 1e-5."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

Q4_0 = int(os.environ.get("AB_TYPE", "2"))        # (the name stays: 2 = Q4_0, 8 = Q8_0)
SHAPES = [(4096, 4096, 512), (4096, 14336, 512), (14336, 4096, 512), (4096, 4096, 128), (8192, 8192, 2048), (4096, 4096, 2048), (16384, 4096, 1024)]


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from ggml_amd import native
    L = native.lib()
    shapes = SHAPES if len(sys.argv) < 2 else [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
    for (m, k, b) in shapes:
        w = bench.synth_blocks(Q4_0, m, k, 1234)
        x = np.random.default_rng(4321).uniform(-1, 1, (b, k)).astype(np.float32)
        h = bench.Hot(dev, Q4_0, w, m, k, x)
        img = torch.empty(L.ggml_cdna4_resident_image_size(Q4_0, m, k), dtype=torch.uint8, device=dev)
        wp, rb = h.a.data.data_ptr(), h.a.row_bytes

        def call():
            h.stream = torch.cuda.current_stream(dev).cuda_stream
            h.gemm_only()
        out = {"type": Q4_0, "shape": "%dx%dx%d" % (m, k, b)}
        h.step(); torch.cuda.synchronize()
        y_percall = h.y.clone()
        res = {"percall": [], "resident": []}
        for rep in range(3):                                             # alternate on the same box
            out["route_percall"] = L.ggml_cdna4_mul_mat_route_of(Q4_0, wp, rb, m, k, b)
            res["percall"].append(bench.graph_us(dev, call, 40))
            native.check(L.ggml_cdna4_resident_image_register(Q4_0, wp, rb, m, k, img.data_ptr(), 1 if rep == 0 else 0, None))
            out["route_resident"] = L.ggml_cdna4_mul_mat_route_of(Q4_0, wp, rb, m, k, b)
            res["resident"].append(bench.graph_us(dev, call, 40))
            if rep == 0:
                h.stream = torch.cuda.current_stream(dev).cuda_stream
                h.step(); torch.cuda.synchronize()
                d = (h.y.double() - y_percall.double())
                out["rel_l2_resident_vs_percall"] = float(d.norm() / y_percall.double().norm())
            L.ggml_cdna4_resident_image_unregister(wp)
        fl = 2.0 * m * k * b
        out.update({"percall_us": round(min(res["percall"]), 2), "resident_us": round(min(res["resident"]), 2),
                    "percall_tflops": round(fl / min(res["percall"]) / 1e6, 1), "resident_tflops": round(fl / min(res["resident"]) / 1e6, 1),
                    "all_us": {k_: [round(v, 2) for v in vs] for k_, vs in res.items()}})
        print(json.dumps(out), flush=True)
        del h, img


if __name__ == "__main__":
    main()
