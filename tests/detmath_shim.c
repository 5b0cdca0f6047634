/* test shim: exposes include/suma_detmath.h (the shared math specification) to Python */
#include "../include/suma_detmath.h"
#define V1(name, fn) void name(const float* x, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = fn(x[i]); }
V1(t_atan, sdm_atan) V1(t_asin, sdm_asin) V1(t_acos, sdm_acos) V1(t_sin, sdm_sin) V1(t_cos, sdm_cos)
V1(t_exp, sdm_exp) V1(t_log, sdm_log) V1(t_floor, sdm_floor) V1(t_round, sdm_round) V1(t_sqrt, sdm_sqrt)
void t_atan2(const float* y, const float* x, float* r, int n) { for (int i = 0; i < n; ++i) r[i] = sdm_atan2(y[i], x[i]); }
void t_sin_d(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = sdm_sin_d(x[i]); }
void t_cos_d(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = sdm_cos_d(x[i]); }
