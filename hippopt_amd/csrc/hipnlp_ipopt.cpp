// hipnlp_ipopt.cpp — IPOPT's C callbacks (include/hipnlp_ipopt.h) on top of the public C-ABI (include/hipnlp.h) and nothing else:
// what nlpsol's IPOPT plugin does for the reference behind opti.solver("ipopt", ...) / self._solver.solve()
// (src/hippopt/base/opti_solver.py:123-125, 479), with the engine in CasADi's place.
#include "../../include/hipnlp_ipopt.h"

#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace {
struct Sizes { int n, m, nnz; };
inline bool sizes_of(hipnlp_handle* h, Sizes& s) {
    hipnlp_dims d;
    if (!h || hipnlp_get_dims(h, &d) != HIPNLP_OK) return false;
    s.n = d.n; s.m = d.m; s.nnz = d.nnz;
    return true;
}
// one evaluation per iterate: IPOPT's new_x is the engine's new_x; a NaN / Inf (HIPNLP_E_NUMERIC) or any failure is FALSE
inline Bool served(int rc) { return rc == HIPNLP_OK ? TRUE : FALSE; }
}  // namespace

extern "C" {

Bool hipnlp_ipopt_eval_f(Index n, Number* x, Bool new_x, Number* obj_value, UserDataPtr user_data) {
    hipnlp_handle* h = static_cast<hipnlp_handle*>(user_data);
    Sizes s;
    if (!sizes_of(h, s) || n != s.n || !x || !obj_value) return FALSE;
    return served(hipnlp_eval(h, x, new_x ? 1 : 0, obj_value, nullptr, nullptr, nullptr));
}

Bool hipnlp_ipopt_eval_grad_f(Index n, Number* x, Bool new_x, Number* grad_f, UserDataPtr user_data) {
    hipnlp_handle* h = static_cast<hipnlp_handle*>(user_data);
    Sizes s;
    if (!sizes_of(h, s) || n != s.n || !x || !grad_f) return FALSE;
    return served(hipnlp_eval(h, x, new_x ? 1 : 0, nullptr, grad_f, nullptr, nullptr));
}

Bool hipnlp_ipopt_eval_g(Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr user_data) {
    hipnlp_handle* h = static_cast<hipnlp_handle*>(user_data);
    Sizes s;
    if (!sizes_of(h, s) || n != s.n || m != s.m || !x || !g) return FALSE;
    return served(hipnlp_eval(h, x, new_x ? 1 : 0, nullptr, nullptr, g, nullptr));
}

Bool hipnlp_ipopt_eval_jac_g(Index n, Number* x, Bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol, Number* values,
                             UserDataPtr user_data) {
    hipnlp_handle* h = static_cast<hipnlp_handle*>(user_data);
    Sizes s;
    if (!sizes_of(h, s) || n != s.n || m != s.m || nele_jac != s.nnz) return FALSE;
    if (!values) {   // the structure call (x is NULL): the handle's own order — CCS, or varying-first inside every knot block
        if (!iRow || !jCol) return FALSE;
        static_assert(sizeof(Index) == sizeof(int32_t), "IPOPT built with 32-bit indices");
        static_assert(sizeof(Bool) == sizeof(int), "IpStdCInterface.h's Bool is an int in the builds these callbacks were written against (see INTEGRATION.md for IPOPT builds whose Bool is bool)");
        return served(hipnlp_sparsity(h, reinterpret_cast<int32_t*>(iRow), reinterpret_cast<int32_t*>(jCol)));
    }
    if (!x) return FALSE;
    return served(hipnlp_eval(h, x, new_x ? 1 : 0, nullptr, nullptr, nullptr, values));
}

Bool hipnlp_ipopt_eval_h(Index n, Number* x, Bool new_x, Number obj_factor, Index m, Number* lambda, Bool new_lambda, Index nele_hess,
                         Index* iRow, Index* jCol, Number* values, UserDataPtr user_data) {
    (void)new_lambda;   // (the Hessian is one launch of its own; what it can reuse of the callbacks before it is the staged copy of x: new_x)
    hipnlp_handle* h = static_cast<hipnlp_handle*>(user_data);
    Sizes s;
    int64_t nh = 0;
    if (!sizes_of(h, s) || n != s.n || m != s.m || hipnlp_hess_nnz(h, &nh) != HIPNLP_OK || int64_t(nele_hess) != nh) return FALSE;
    if (!values) {
        if (!iRow || !jCol) return FALSE;
        return served(hipnlp_hess_sparsity(h, reinterpret_cast<int32_t*>(iRow), reinterpret_cast<int32_t*>(jCol)));
    }
    if (!x || !lambda) return FALSE;
    return served(hipnlp_eval_hess_at(h, x, new_x ? 1 : 0, &obj_factor, lambda, values));
}

int hipnlp_ipopt_sizes(hipnlp_handle* h, Index* n, Index* m, Index* nele_jac, Index* nele_hess) {
    hipnlp_dims d;
    if (!h) return HIPNLP_E_INVALID;
    const int rc = hipnlp_get_dims(h, &d);
    if (rc != HIPNLP_OK) return rc;
    if (n) *n = d.n;
    if (m) *m = d.m;
    if (nele_jac) *nele_jac = d.nnz;
    if (nele_hess) {
        int64_t nh = 0;
        const int rh = hipnlp_hess_nnz(h, &nh);
        if (rh != HIPNLP_OK) return rh;
        if (nh > std::numeric_limits<Index>::max()) return HIPNLP_E_INVALID;
        *nele_hess = Index(nh);
    }
    return HIPNLP_OK;
}

int hipnlp_ipopt_bounds(hipnlp_handle* h, Number* x_L, Number* x_U, Number* g_L, Number* g_U) {
    hipnlp_dims d;
    if (!h) return HIPNLP_E_INVALID;
    int rc = hipnlp_get_dims(h, &d);
    if (rc != HIPNLP_OK) return rc;
    rc = hipnlp_bounds(h, x_L, x_U, g_L, g_U);
    if (rc != HIPNLP_OK) return rc;
    const double big = 2e19;   // beyond IPOPT's nlp_lower_bound_inf / nlp_upper_bound_inf (-/+ 1e19): "no bound"
    auto clip = [big](Number* v, int count) { if (v) for (int i = 0; i < count; ++i) { if (v[i] < -big) v[i] = -big; if (v[i] > big) v[i] = big; } };
    clip(x_L, d.n); clip(x_U, d.n); clip(g_L, d.m); clip(g_U, d.m);
    return HIPNLP_OK;
}

int hipnlp_ipopt_attach(hipnlp_handle* h) {
    if (!h) return HIPNLP_E_INVALID;
    // auto-registration of IPOPT's arrays and nothing that writes into them before IPOPT asks: every callback fills exactly the array
    // it was handed (the TNLPAdapter keeps full_g_ / jac_g_ as caches keyed by the tag of x — see hipnlp_ipopt_set_early_outputs)
    int rc = hipnlp_set_auto_register(h, 1);
    if (rc == HIPNLP_OK) rc = hipnlp_set_early_outputs(h, 0);
    return rc;
}

int hipnlp_ipopt_set_early_outputs(hipnlp_handle* h, int on) {
    if (!h) return HIPNLP_E_INVALID;
    return hipnlp_set_early_outputs(h, on ? 1 : 0);   // (g and jac g only; never grad f: its destination is IPOPT's own gradient vector)
}

int hipnlp_ipopt_detach(hipnlp_handle* h) {
    if (!h) return HIPNLP_E_INVALID;
    int rc = hipnlp_set_early_outputs(h, 0);
    const int r2 = hipnlp_set_auto_register(h, 0);   // releases the registrations of IPOPT's arrays
    const int r3 = hipnlp_set_auto_register(h, 1);
    if (rc == HIPNLP_OK) rc = r2;
    if (rc == HIPNLP_OK) rc = r3;
    return rc;
}

}  // extern "C"
