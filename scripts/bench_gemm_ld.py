"""GPU micro-benchmark: does a power-of-two row stride of the operands (lda = K = 512 doubles = 4 KB, as in
the packed panels) cost bandwidth?  C -= A B^T with padded vs unpadded leading dimensions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sif_xco2_cokriging_amd import native

M, N, K = 16384, 8192, 512
dev = torch.device("cuda:0")
torch.manual_seed(0)
h = native.Handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
h.set_option("gemm_variant", int(sys.argv[1]) if len(sys.argv) > 1 else 5)
res = {}
for pad_ab, pad_c in ((0, 0), (16, 0), (0, 16), (16, 16), (32, 32)):
    A = torch.randn(M, K + pad_ab, dtype=torch.float64, device=dev)
    B = torch.randn(N, K + pad_ab, dtype=torch.float64, device=dev)
    C = torch.randn(M, N + pad_c, dtype=torch.float64, device=dev)
    ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            h.dev_gemm_nt(C.data_ptr(), N + pad_c, A.data_ptr(), K + pad_ab, B.data_ptr(), K + pad_ab, M, N, K)
        e1.record()
        torch.cuda.synchronize()
        ts.append(2.0 * M * N * K * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    ts = sorted(ts[1:])
    print(f"pad A/B {pad_ab:2d}  pad C {pad_c:2d}: median {ts[len(ts)//2]:.1f} TF")
