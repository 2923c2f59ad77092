"""A/B of the whole MUL_MAT step (ggml_cdna4_mul_mat, path GEMM) under whatever environment the process was started with — run once per setting
(CDNA4_NO_FUSEQ=1: quantizer launch + GEMM launch; default: the one-launch step; CDNA4_FQ_TICKETED=1: one launch with the ticketed split) and alternate the
settings on ONE box.  Prints one JSON line per shape: HIP-event loop time per call, HIP-graph replay time per call (no host between launches), the route id."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

SHAPES = [(4096, 4096, 512), (4096, 11008, 512), (8192, 4096, 512), (4096, 4096, 128), (2048, 4096, 512), (4096, 14336, 512), (4096, 4096, 1024)]


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from ggml_amd import native
    L = native.lib()
    tag = os.environ.get("AB_TAG", "")
    shapes = SHAPES if len(sys.argv) < 2 else [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
    for (m, k, b) in shapes:
        if (m, k, b) == bench.HEAD:
            w, x, how = bench.prescribed(bench.Q4_K, m, k, 0, m, b)
        else:
            w, how = bench.synth_blocks(bench.Q4_K, m, k, 1234), "random-valid-blocks"
            x = np.random.default_rng(4321).uniform(-1, 1, (b, k)).astype(np.float32)
        h = bench.Hot(dev, bench.Q4_K, w, m, k, x)
        abl = os.environ.pop("CDNA4_FQ_ABL", None)                      # (timing-only ablations multiply the image a complete call left in the workspace)
        h.step(); torch.cuda.synchronize()
        if abl is not None:
            os.environ["CDNA4_FQ_ABL"] = abl
        ev = min(bench.events_us(h.step, 300, 20) for _ in range(3))
        def call():                                                      # (graph capture needs the launch on the capturing stream)
            h.stream = torch.cuda.current_stream(dev).cuda_stream
            h.step()
        gr = min(bench.graph_us(dev, call, 40) for _ in range(2))
        h.stream = torch.cuda.current_stream(dev).cuda_stream
        print(json.dumps({"tag": tag, "shape": "%dx%dx%d" % (m, k, b), "route": L.ggml_cdna4_mul_mat_route(bench.Q4_K, m, k, b), "step_us_events": round(ev, 2), "step_us_graph": round(gr, 2),
                          "tflops_graph": round(2.0 * m * k * b / gr / 1e6, 1), "data": how}), flush=True)
        del h


if __name__ == "__main__":
    main()
