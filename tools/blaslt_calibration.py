"""Calibration, not product: how long does the vendor GEMM (hipBLASLt through torch.mm) take for the step's big products at the
headline shape?  Tells whether ta3n::gemm_tiles' per-launch times are near what this GPU does for such small GEMMs at all.
Usage (GPU box): python tools/blaslt_calibration.py [configs3] > gpurun_out/blaslt.txt
(configs3: the products of BASELINE configs[3] - 512+512 videos x 9 segments = 9 216 frame rows - the only shape large enough for
kernel quality, not launch structure, to decide; round 4)"""
import sys
import torch

SHAPES = [  # (name, M, N, K, transA, transB)  C[M,N] = op(A) op(B)
    ("F1      X[1010,2048] . Wsh^T[2048,512]", 1010, 512, 2048, False, True),
    ("dWsh    gZ1^T[512,1010] . X[1010,2048]", 512, 2048, 1010, True, False),
    ("gF1-ish gZ[1010,256] . W[256,2560]", 1010, 2560, 256, False, False),
    ("Z_t     cat[202,2560] . W^T[2560,256]", 202, 256, 2560, False, True),
    ("Z_t x10 batched cat[10,202,1536] . W^T", 2020, 256, 1536, False, True),
    ("Hr      R[202,256] . W1^T[256,256]", 202, 256, 256, False, True),
    ("dWtrn   gZ^T[256,202] . cat[202,2560]", 256, 2560, 202, True, False),
]


SHAPES_CONFIGS3 = [
    ("F1      X[9216,2048] . Wsh^T[2048,512]", 9216, 512, 2048, False, True),
    ("dWsh    gZ1^T[512,9216] . X[9216,2048]", 512, 2048, 9216, True, False),
    ("gF1-ish gZ[9216,256] . W[256,4608]", 9216, 4608, 256, False, False),
    ("Z_t     cat[1024,4608] . W^T[4608,256]", 1024, 256, 4608, False, True),
    ("Z_t x22 batched cat[22,1024,2560] . W^T", 22528, 256, 2560, False, True),
    ("dWtrn   gZ^T[256,1024] . cat[1024,4608]", 256, 4608, 1024, True, False),
]
if len(sys.argv) > 1 and sys.argv[1] == "configs3":
    SHAPES = SHAPES_CONFIGS3


def graphed(fn, inner):
    """us per fn() with `inner` calls captured in one graph (the host is out of the loop: torch's per-call overhead is ~18 us)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * inner)


def time_mm(a, b, ta, tb):
    out = torch.empty(a.shape[1] if ta else a.shape[0], b.shape[0] if tb else b.shape[1], device="cuda", dtype=a.dtype)
    return graphed(lambda: torch.mm(a.t() if ta else a, b.t() if tb else b, out=out), 40)


def main():
    print(torch.cuda.get_device_name(0))
    for dt in (torch.bfloat16, torch.float32):
        print(f"--- {dt}")
        for name, M, N, K, ta, tb in SHAPES:
            a = torch.randn((K, M) if ta else (M, K), device="cuda", dtype=dt)
            b = torch.randn((N, K) if tb else (K, N), device="cuda", dtype=dt)
            us = time_mm(a, b, ta, tb)
            print(f"{name:44s} {us:8.2f} us  {2.0 * M * N * K / us * 1e-6:8.1f} TFLOP/s")
    # dependent chain of the forward levels (each output feeds the next): what eight dependent vendor launches cost
    x = torch.randn(1010, 2048, device="cuda", dtype=torch.bfloat16)
    w1 = torch.randn(2048, 512, device="cuda", dtype=torch.bfloat16) * 0.02
    w2 = torch.randn(512, 512, device="cuda", dtype=torch.bfloat16) * 0.02
    def chain():
        h = x @ w1
        for _ in range(7):
            h = h @ w2
        return h
    print(f"8 dependent bf16 GEMMs (1010x2048x512 then 7 x 1010x512x512), graphed: {graphed(chain, 5) :.1f} us per chain")


if __name__ == "__main__":
    main()
