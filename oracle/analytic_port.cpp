// analytic_port.cpp — CPU BASELINE (test infrastructure, never in the product; only bench.py's cpu_baseline leg and tests/ load it).
//
// SURVEY §8d / BASELINE.md §3, baselines B1 and B2: "the build's own C++ implementation of the identical per-knot maths
// (-O3 -march=native), single thread and OpenMP over knots".  This is the engine's ANALYTIC knot program (hand-derived Jacobians,
// hippopt_amd/csrc/knot_body.h — the same source the gfx950 kernel is compiled from) run on the host: one knot at a time per
// thread, every lane task of every phase in a loop, the same copy-out tables.  It is the honest CPU comparator for the callback
// throughput: what an expanded, CSE'd CasADi SX graph of the reference's NLP (base/opti_solver.py:479 -> nlp_f / nlp_grad_f /
// nlp_g / nlp_jac_g) costs per call is of this order, not of the order of the forward-AD oracle (kinodyn_oracle.cpp), which exists
// to CHECK values, not to be fast.  Values are checked against that oracle in tests/test_analytic_port.py.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../hippopt_amd/csrc/layout.h"

using namespace hipnlp;

namespace {
struct ValueEm {
    static constexpr int kTerrain = -1;   // run-time terrain switch (the device kernels are instantiated per terrain)
    double* g;
    double* jac;
    void G(int slot, int, double v) { g[slot] = v; }
    void J(int slot, int, int, double v) { jac[slot] = v; }
};
}  // namespace

struct port_handle {
    hipnlp_desc d;
    KinTables kt;
    KSettings ks;
    Layout L;
    std::vector<double> pk;
    GParams gp;
    bool params_set = false;
    std::vector<KnotScratch*> scratch;   // one per thread
    std::vector<double> cost_knot;       // [N][NCT]
    std::string err;
};

extern "C" {

port_handle* port_create(const hipnlp_desc* desc, char* err, int errlen) {
    port_handle* h = new port_handle();
    h->d = *desc;
    std::string e;
    if (Layout::make_kin_tables(desc->model, h->kt, e)) Layout::fill_terrain_tops(h->kt, desc->settings.terrain, desc->settings.n_terrain_steps, desc->settings.terrain_steps);
    if (!e.empty() || !h->L.build(desc->settings, h->kt)) {
        if (e.empty()) e = h->L.error;
        std::strncpy(err, e.c_str(), size_t(errlen - 1));
        delete h;
        return nullptr;
    }
    h->ks = Layout::make_ksettings(desc->settings);
    h->cost_knot.assign(size_t(h->L.N) * NCT, 0.0);
    return h;
}
void port_destroy(port_handle* h) {
    if (!h) return;
    for (KnotScratch* s : h->scratch) delete s;
    delete h;
}
void port_dims(const port_handle* h, int* n, int* m, int* nnz) { *n = h->L.n; *m = h->L.m; *nnz = h->L.nnz; }
int port_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void port_set_params(port_handle* h, const double* p) {
    h->pk.assign(size_t(h->L.N) * PK_STRIDE, 0.0);
    pack_params(p, h->L.N, h->pk.data(), h->gp);
    h->params_set = true;
}

// one callback set {f, grad f, g, jac g} at x; `threads` knots in flight (1 = the single-thread baseline)
int port_eval(port_handle* h, const double* x, double* f, double* grad, double* g, double* jac, int threads) {
    if (!h->params_set) return -4;
    const Layout& L = h->L;
    const int N = L.N;
    if (threads < 1) threads = 1;
    while (int(h->scratch.size()) < threads) {
        KnotScratch* s = new KnotScratch();
        std::memset(s, 0, sizeof(KnotScratch));
        h->scratch.push_back(s);
    }
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int k = 0; k < N; ++k) {
#ifdef _OPENMP
        KnotScratch* s = h->scratch[size_t(omp_get_thread_num())];
#else
        KnotScratch* s = h->scratch[0];
#endif
        for (int i = 0; i < NXK; ++i) { s->x[i] = x[NXK * k + i]; s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0; }
        for (int i = NXK; i < XPAD; ++i) { s->x[i] = 0.0; s->xm[i] = 0.0; }
        if (k == 0 || k == N - 1)
            for (int i = 0; i < NPER; ++i) s->xo[i] = x[NXK * (k == 0 ? N - 1 : 0) + periodicity_row_var(i)];
        for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
        std::memcpy(s->pk, h->pk.data() + size_t(k) * PK_STRIDE, PK_STRIDE * sizeof(double));
        KnotInfo ki{k, N, k == 0, k == N - 1};
        ValueEm em{s->g, s->jac};
        Ctx<ValueEm> cx(*s, h->kt, h->ks, h->gp, ki, em);
#define PORT_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
        HIPNLP_KNOT_PROGRAM(PORT_R, )
#undef PORT_R
        const int v = L.variant_of(k);
        const long jb = L.jac_base(k);
        const int32_t* jp = L.jperm[v].data();
        for (int i = 0; i < L.nnz_v[v]; ++i) jac[jb + i] = s->jac[jp[i]];
        if (k == N - 1) for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac[L.jac_glob_base + long(i)] = s->jac[L.jperm_glob[i]];
        const int32_t* ga = L.g_a[v].data();
        const int32_t* gb = L.g_b.data();
        for (int slot = 0; slot < gs::COUNT; ++slot) if (ga[slot] >= 0) g[ga[slot] + gb[slot] * k] = s->g[slot];
        std::memcpy(grad + size_t(NXK) * k, s->grad, NXK * sizeof(double));
        std::memcpy(h->cost_knot.data() + size_t(k) * NCT, s->cost, NCT * sizeof(double));
    }
    for (int i = 0; i < NXG; ++i) grad[size_t(NXK) * N + i] = 0.0;
    double ft = 0.0;
    for (int t = 0; t < NCT; ++t) {
        double term = 0.0;
        for (int k = 0; k < N; ++k) term += h->cost_knot[size_t(k) * NCT + t];
        ft += term;
    }
    *f = ft;
    return 0;
}

}  // extern "C"
