"""GPU diagnostic: which clock does the chip hold under the Cholesky trailing updates on the benchmark's data?

MI355X lowers its clock under load by an amount that depends on the operands (MI355X_MICROARCH.md, "DVFS give-back"):
the register-only MFMA microbenchmark (ck_debug_mfma_peak: operands ~1.0) holds ~2.37 GHz and defines the 77.8 TFLOP/s
issue-rate ceiling, but the GEMM kernel moves real data through L2 / LDS.  This script stamps every workgroup of
k_syrk_group_d (s_memtime / s_memrealtime, a separate instantiation: option gemm_stamps) during ordinary
factorisations and prints the clock, next to the factorisation time with and without the stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sif_xco2_cokriging_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pb = synth.conus_problem(n, seed=20003)
pv = pb["params"]
h = native.Handle(0)
h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
h.set_metric(pb["metric"])
for k in range(2):
    h.set_data(k, pb["coords"][k], pb["values"][k])
h.set_option("time_gemm", 1)
for w in (1, 4):
    print(f"mfma_peak waves/SIMD={w}:", h.mfma_peak(w, 40000), flush=True)
for stamps in (0, 1, 0, 1):
    h.assemble_joint()
    h.set_option("gemm_stamps", stamps)
    for it in range(reps):
        h.assemble_joint()
        assert h.factor() == 0
        t = h.timings()
        line = f"N={2*n} stamps={stamps} factor_ms {t['factor_ms']:.1f} syrk_ms {t['syrk_ms']:.1f}"
        if stamps:
            c = h.gemm_clock()
            peak = 256 * 128 * c["mhz_median"] * 1e6 / 1e12
            line += (f" | clock MHz median {c['mhz_median']:.0f} (5% {c['mhz_p05']:.0f}, 95% {c['mhz_p95']:.0f}) over "
                     f"{c['workgroups']} workgroups, lifetime {c['wg_us_median']:.0f} us = {c['wg_cycles_median']:.0f} cycles"
                     f" | MFMA peak at that clock {peak:.1f} TF")
        print(line, flush=True)
print("mfma_peak waves/SIMD=4 (after):", h.mfma_peak(4, 40000))
