"""GPU micro-benchmark of the MFMA GEMM tile (C -= A B^T, K = 512 as in the Cholesky trailing
update), A/B variants interleaved in one process (cdna_hip_programming.md rule 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sif_xco2_cokriging_amd import native

M, N, K = (int(x) for x in os.environ.get("CK_GEMM_MNK", "16384,8192,512").split(","))
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.randn(M, K, dtype=torch.float64, device=dev)
B = torch.randn(N, K, dtype=torch.float64, device=dev)
C = torch.randn(M, N, dtype=torch.float64, device=dev)
h = native.Handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
variants = [int(v) for v in (sys.argv[1:] or ["0", "4", "5"])]
ref = C - A @ B.T
for v in variants:
    h.set_option("gemm_variant", v)
    Cv = C.clone()
    h.dev_gemm_nt(Cv.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
    torch.cuda.synchronize()
    print(f"variant {v}: max abs err vs torch {float((Cv - ref).abs().max()):.3e}")
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        h.set_option("gemm_variant", v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            h.dev_gemm_nt(C.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(2.0 * M * N * K * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
for v in variants:
    r = sorted(res[v][1:])
    print(f"variant {v}: median {r[len(r)//2]:.1f} TF  min {r[0]:.1f}  max {r[-1]:.1f}")
