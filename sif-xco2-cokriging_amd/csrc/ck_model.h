// ck_model.h -- host-side preparation of Matern block constants (see ck_model.cpp)
#pragma once
#include "ck_math.h"

#ifdef __cplusplus
extern "C" {
#endif
void ck_matern_prepare(double nu, double len_scale, double amp, double nugget, CkMatern* m);
void ck_model_prepare(int n_procs, const double* sigma, const double* nu, const double* len_scale,
                      const double* nugget, double rho12, CkMatern* out3);
#ifdef __cplusplus
}
#endif
