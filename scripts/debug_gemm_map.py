import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sif_xco2_cokriging_amd import native
dev = torch.device("cuda:0")
h = native.Handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h.set_option("gemm_variant", v)
for (M, N, K) in ((256, 128, 64), (512, 128, 64), (256, 256, 64), (512, 256, 64), (1024, 512, 512)):
    A = (torch.arange(M, device=dev, dtype=torch.float64)[:, None] * 1000 + torch.arange(K, device=dev, dtype=torch.float64)[None, :])
    B = torch.zeros(N, K, dtype=torch.float64, device=dev)
    for j in range(N):
        B[j, j % K] = 1.0
    C = torch.zeros(M, N, dtype=torch.float64, device=dev)
    h.dev_gemm_nt(C.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
    torch.cuda.synchronize()
    ref = -(A @ B.T)
    bad = (C != ref).nonzero()
    print(f"M={M} N={N} K={K}: mismatches {len(bad)} of {M*N}")
    if len(bad):
        for (i, j) in bad[:12].tolist():
            print("  C[%d][%d] = %g expected %g" % (i, j, -C[i, j].item(), -ref[i, j].item()))
