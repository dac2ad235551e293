// ck_model.h -- host-side preparation of Matern block constants (see ck_model.cpp)
#pragma once
#include "ck_math.h"

// internal to the shared object (not part of include/cokrige.h): hidden visibility, so that the library exports exactly
// what the header declares (tests/test_abi_exports.py)
#define CK_MODEL_HIDDEN __attribute__((visibility("hidden")))
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(hidden)
void ck_matern_prepare(double nu, double len_scale, double amp, double nugget, CkMatern* m);
void ck_model_prepare(int n_procs, const double* sigma, const double* nu, const double* len_scale,
                      const double* nugget, double rho12, CkMatern* out3);
// tabulated correlation (ck_math.h): interval plan, Chebyshev fit
double ck_table_edge(int64_t interval_index);
int ck_table_plan(const CkMatern* m, int metric, double qbox_euclid, int64_t* base_out, double* q_nodes);
void ck_table_fit(const double* node_values, int n_int, int64_t base, double* coef_kmajor);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
