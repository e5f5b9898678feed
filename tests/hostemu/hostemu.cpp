// TEST-ONLY host emulation of the HIP kernel's knot program.
// It runs the *same* knot_body.h phases the gfx950 kernel runs (one "lane task" at a time, a loop
// instead of a wavefront) and the same layout tables for the copy-out, so the kernel mathematics and
// the CCS bookkeeping can be debugged against the oracle on a machine without a GPU.
// It is NOT part of the product: libhipnlp.so does not contain it and hipnlp_eval never falls back to it.
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "../../hippopt_amd/csrc/knot_hess_layout.h"
#include "../../hippopt_amd/csrc/pose_layout.h"

using namespace hipnlp;

struct ValueEm {
    static constexpr int kTerrain = -1;
    double* g;
    double* jac;
    double* hess = nullptr;
    void G(int slot, int, double v) { g[slot] = v; }
    void J(int slot, int, int, double v) { jac[slot] = v; }
    void H(int slot, int, int, double v) { hess[slot] = v; }
};

// the pose finder's device emitter (hipnlp_pose.hip PoseEm): a Jacobian staging cut to the pieces of the native numbering the pose program
// touches (pose_body.h pjs).  J indexes a pointer moved back by a constant on the device — a slot below js::V0 through J would land in front of
// the staging; JC / JD map any kept slot.  Violations are counted (hostemu_pose_map_violations), values go where the device puts them.
static long g_pose_map_violations = 0;
struct ValueEmPoseStaged {
    static constexpr int kTerrain = -1;
    double* g;
    double* stg;
    double* hess = nullptr;
    void G(int slot, int, double v) { g[slot] = v; }
    void J(int slot, int, int, double v) { if (slot < js::V0) { g_pose_map_violations++; return; } stg[pjs::index(slot)] = v; }
    void JC(int slot, int, int, double v) { if (!pjs::kept(slot)) { g_pose_map_violations++; return; } stg[pjs::index(slot)] = v; }
    void JD(int slot, int, int, double v) { JC(slot, 0, 0, v); }
    void H(int slot, int, int, double v) { hess[slot] = v; }
};

// an emitter with the compile-time traits of a device instantiation (hipnlp_knot_kernel<TERRAIN, WAVES>): task groups whose placement
// depends on them (hdyn_entries_early, knot_body.h) then run where the device runs them — used by the wave-order emulation
template <int WAVES, int TERRAIN> struct ValueEmV {
    static constexpr int kTerrain = TERRAIN;
    static constexpr int kWaves = WAVES;
    double* g;
    double* jac;
    void G(int slot, int, double v) { g[slot] = v; }
    void J(int slot, int, int, double v) { jac[slot] = v; }
};

// the COMPACT device layout of the planar callback kernel (KnotScratchT<LAYOUT_COMPACT>: shared storage for arrays with disjoint lifetimes,
// joint frames / link inertials parked in comp[], horizon-end g rows in ends.c): emulated phase by phase in program order, so a
// lifetime overlap shows up as a wrong value here, without a GPU
template <int LAYOUT> struct ValueEmC {
    static constexpr int kTerrain = -1;
    using Scratch = KnotScratchT<LAYOUT>;
    double* g;
    double* jac;
    void G(int slot, int, double v) { g[slot] = v; }
    void J(int slot, int, int, double v) { jac[slot] = v; }
};

struct hostemu_handle {
    hipnlp_desc d;
    KinTables kt;
    KSettings ks;
    Layout L;
    HessLayout HL;
    bool has_hess = false;
};

// wave-order emulation (tests only): the groups of a phase collected in table order, run wave by wave in a permuted order of the waves
struct WaveOrder {
    int waves, order;
    std::vector<std::pair<int, std::function<void()>>> groups;
    void add(int w, std::function<void()> fn) { if (w >= 0) groups.push_back({w, std::move(fn)}); }
    void flush() {
        for (int q = 0; q < waves; ++q) {
            const int w = order == 0 ? q : (order == 1 ? waves - 1 - q : (q + waves / 2) % waves);
            for (auto& gq : groups) if (gq.first == w) gq.second();
        }
        groups.clear();
    }
};
static int g_wave_order = -1;   // >= 0: the pose / Hessian emulations below run their programs through WaveOrder (hostemu_set_wave_order)

// the same evaluation on the compact scratch layout of the device's four-wave kernels
template <int LAYOUT> static int eval_compact(const hostemu_handle* h, const double* x, const double* p, double* f, double* grad, double* g, double* jac, double* cost_terms) {
    const Layout& L = h->L;
    const int N = L.N;
    std::vector<double> pk(size_t(N) * PK_STRIDE);
    GParams gp;
    pack_params(p, N, pk.data(), gp);
    for (int i = 0; i < NCT; ++i) cost_terms[i] = 0.0;
    for (int i = 0; i < L.n; ++i) grad[i] = 0.0;
    using S = KnotScratchT<LAYOUT>;
    S* s = new S();
    for (int k = 0; k < N; ++k) {
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(S) / sizeof(double), std::nan(""));
        for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
        for (int i = 0; i < NXK; ++i) { s->x[i] = x[NXK * k + i]; s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = k == 0 ? x[NXK * (N - 1) + periodicity_row_var(i)] : (k == N - 1 ? x[periodicity_row_var(i)] : 0.0);
        for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
        for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[size_t(k) * PK_STRIDE + i];
        KnotInfo ki{k, N, k == 0, k == N - 1};
        ValueEmC<LAYOUT> em{s->g, s->jac};
        Ctx<ValueEmC<LAYOUT>> cx(*s, static_cast<const KinLite&>(h->kt), h->ks, static_cast<const GParamsLite&>(gp), ki, em, &h->kt, &gp);
        if (g_wave_order < 0) {
#define HOST_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
            HIPNLP_KNOT_PROGRAM(HOST_R, )
#undef HOST_R
        } else {   // the four-wave column of the table, every phase wave by wave in a permuted order (arrays that share storage in this layout!)
            const bool planar_rt = h->ks.terrain == HIPNLP_TERRAIN_PLANAR;
            WaveOrder wo{4, g_wave_order, {}};
#define HIPNLP_W4(a, b) (planar_rt ? (a) : (b))
#define HIPNLP_W8(a, b) (planar_rt ? (a) : (b))
#define HOST_R(w4, w8, fn, nt) wo.add((w4), [&cx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_); });
#define HOST_BARRIER wo.flush();
            HIPNLP_KNOT_PROGRAM(HOST_R, HOST_BARRIER)
#undef HOST_R
#undef HOST_BARRIER
#undef HIPNLP_W4
#undef HIPNLP_W8
            wo.flush();
        }
        const int v = L.variant_of(k);
        const long jb = L.jac_base(k);
        for (int i = 0; i < L.nnz_v[v]; ++i) jac[jb + i] = s->jac[L.jperm[v][size_t(i)]];
        if (k == N - 1) for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac[L.jac_glob_base + long(i)] = s->jac[L.jperm_glob[i]];
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            const int a = L.g_a[v][size_t(slot)];
            if (a != G_NONE) g[a + L.g_b[size_t(slot)] * k] = s->g_at(slot);
        }
        for (int i = 0; i < NXK; ++i) grad[NXK * k + i] = s->grad[i];
        for (int i = 0; i < NCT; ++i) cost_terms[i] += s->cost[i];
    }
    delete s;
    double ft = 0.0;
    for (int i = 0; i < NCT; ++i) ft += cost_terms[i];
    *f = ft;
    return 0;
}

// entries of jac g that the eight-wave kernel of a launch into host memory stores behind the SECOND barrier (Layout::jslot_phase <= 1)
// and whose staged value at that point is not the final one, counted over the knots of the last wave-order evaluation
static long g_early_violations = 0;
template <int WAVES, int TERRAIN> static int eval_wave_order_t(const hostemu_handle* h, const double* x, const double* p, int order, double* f, double* grad, double* g,
                                                               double* jac, double* cost_terms) {
    constexpr int waves = WAVES;
    const Layout& L = h->L;
    const int N = L.N;
    constexpr bool planar_rt = TERRAIN == HIPNLP_TERRAIN_PLANAR;
    std::vector<double> pk(size_t(N) * PK_STRIDE);
    GParams gp;
    pack_params(p, N, pk.data(), gp);
    for (int i = 0; i < NCT; ++i) cost_terms[i] = 0.0;
    for (int i = 0; i < L.n; ++i) grad[i] = 0.0;
    KnotScratch* s = new KnotScratch();
    for (int k = 0; k < N; ++k) {
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
        for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
        for (int i = 0; i < NXK; ++i) { s->x[i] = x[NXK * k + i]; s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = k == 0 ? x[NXK * (N - 1) + periodicity_row_var(i)] : (k == N - 1 ? x[periodicity_row_var(i)] : 0.0);
        for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
        for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[size_t(k) * PK_STRIDE + i];
        KnotInfo ki{k, N, k == 0, k == N - 1};
        ValueEmV<WAVES, TERRAIN> em{s->g, s->jac};
        Ctx<ValueEmV<WAVES, TERRAIN>> cx(*s, h->kt, h->ks, gp, ki, em);
        std::vector<std::pair<int, std::function<void()>>> groups;   // (wave, group) of the current phase, in table order
        auto flush = [&]() {
            for (int q = 0; q < waves; ++q) {
                const int w = order == 0 ? q : (order == 1 ? waves - 1 - q : (q + waves / 2) % waves);
                for (auto& gq : groups) if (gq.first == w) gq.second();
            }
            groups.clear();
        };
#define HIPNLP_W4(a, b) (planar_rt ? (a) : (b))
#define HIPNLP_W8(a, b) (planar_rt ? (a) : (b))
#define HOST_R(w4, w8, fn, nt) { const int w_ = waves == 4 ? (w4) : (w8); if (w_ >= 0) groups.push_back({w_, [&cx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_); }}); }
        int barriers = 0;
        std::vector<double> snap(js::COUNT);
#define HOST_BARRIER flush(); if (++barriers == 2) std::memcpy(snap.data(), s->jac, sizeof(double) * js::COUNT);
        HIPNLP_KNOT_PROGRAM(HOST_R, HOST_BARRIER)
#undef HOST_R
#undef HOST_BARRIER
#undef HIPNLP_W4
#undef HIPNLP_W8
        flush();
        const int v = L.variant_of(k);
        const long jb = L.jac_base(k);
        if (k == 0) g_early_violations = 0;
        for (int i = 0; i < L.nnz_v[v]; ++i) {
            const int slot = L.jperm[v][size_t(i)];
            if (L.jslot_phase[size_t(slot)] <= 1 && std::memcmp(&snap[size_t(slot)], &s->jac[slot], sizeof(double)) != 0) g_early_violations++;
        }
        for (int i = 0; i < L.nnz_v[v]; ++i) jac[jb + i] = s->jac[L.jperm[v][size_t(i)]];
        if (k == N - 1) for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac[L.jac_glob_base + long(i)] = s->jac[L.jperm_glob[i]];
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            const int a = L.g_a[v][size_t(slot)];
            if (a != G_NONE) g[a + L.g_b[size_t(slot)] * k] = s->g[slot];
        }
        for (int i = 0; i < NXK; ++i) grad[NXK * k + i] = s->grad[i];
        for (int i = 0; i < NCT; ++i) cost_terms[i] += s->cost[i];
    }
    delete s;
    double ft = 0.0;
    for (int i = 0; i < NCT; ++i) ft += cost_terms[i];
    *f = ft;
    return 0;
}


extern "C" {
void hostemu_set_wave_order(int order) { g_wave_order = order; }
long hostemu_early_violations(void) { return g_early_violations; }

hostemu_handle* hostemu_create(const hipnlp_desc* desc, char* err, int errlen) {
    hostemu_handle* h = new hostemu_handle();
    h->d = *desc;
    std::string e;
    if (Layout::make_kin_tables(desc->model, h->kt, e)) Layout::fill_terrain_tops(h->kt, desc->settings.terrain, desc->settings.n_terrain_steps, desc->settings.terrain_steps);
    if (!e.empty() || !h->L.build(desc->settings, h->kt, (desc->flags & HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS) != 0, (desc->flags & HIPNLP_FLAG_JAC_VARYING_FIRST) != 0)) {
        if (e.empty()) e = h->L.error;
        std::strncpy(err, e.c_str(), size_t(errlen - 1));
        delete h;
        return nullptr;
    }
    h->ks = Layout::make_ksettings(desc->settings);
    h->has_hess = h->HL.build(desc->settings, h->kt);
    return h;
}
void hostemu_destroy(hostemu_handle* h) { delete h; }
long hostemu_hess_nnz(const hostemu_handle* h) { return h->has_hess ? h->HL.nnz : -1; }
const char* hostemu_hess_error(const hostemu_handle* h) { return h->HL.error.c_str(); }
void hostemu_hess_sparsity(const hostemu_handle* h, int* irow, int* jcol) { h->HL.pattern(irow, jcol); }
// phase of the Hessian program in which the entry at every position of a knot block is emitted; returns the length of the early run
int hostemu_hess_phases(const hostemu_handle* h, unsigned char* pos_phase /*[nnz_knot]*/, int* early_phase) {
    for (size_t i = 0; i < h->HL.pos_phase.size(); ++i) pos_phase[i] = h->HL.pos_phase[i];
    if (early_phase) *early_phase = h->HL.early_phase;
    return h->HL.early_run;
}
// the Hessian tasks of knot_hess_body.h behind the knot program, multiplier gather and copy-out as in hipnlp_knot_hess_kernel
void hostemu_hess(const hostemu_handle* h, const double* x, const double* p, double sigma, const double* lambda, double* hess) {
    const Layout& L = h->L;
    const HessLayout& HL = h->HL;
    const int N = L.N;
    std::vector<double> pk(size_t(N) * PK_STRIDE);
    GParams gp;
    pack_params(p, N, pk.data(), gp);
    KnotScratch* s = new KnotScratch();
    KHessScratch* hx = new KHessScratch();
    for (int k = 0; k < N; ++k) {
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
        std::fill(reinterpret_cast<double*>(hx), reinterpret_cast<double*>(hx) + sizeof(KHessScratch) / sizeof(double), std::nan(""));
        for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
        for (int i = 0; i < NXK; ++i) { s->x[i] = x[NXK * k + i]; s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = k == 0 ? x[NXK * (N - 1) + periodicity_row_var(i)] : (k == N - 1 ? x[periodicity_row_var(i)] : 0.0);
        for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
        for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[size_t(k) * PK_STRIDE + i];
        const int v = L.variant_of(k);
        for (int i = 0; i < 3; ++i) {
            const int slot = gs::HDYN + 3 + i, vn = L.variant_of(k + 1);
            const int a = k + 1 < N ? L.g_a[vn][size_t(slot)] : G_NONE;
            hx->lam_next[i] = a != G_NONE ? lambda[a + L.g_b[size_t(slot)] * (k + 1)] : 0.0;
        }
        hx->sigma = sigma;
        kh_fill_far_lists(h->kt, hx->far);
        KnotInfo ki{k, N, k == 0, k == N - 1};
        ValueEm em{s->g, s->jac, hx->H};
        Ctx<ValueEm> cx(*s, h->kt, h->ks, gp, ki, em);
        // multipliers by native slot, in the g staging area of the scratch (the Hessian program emits no g)
        for (int slot = 0; slot < gs::COUNT; ++slot) { const int a = L.g_a[v][size_t(slot)]; s->g[slot] = a != G_NONE ? lambda[a + L.g_b[size_t(slot)] * k] : 0.0; }
        KHCtx<ValueEm> hcx{cx, *hx, s->g};
        if (g_wave_order < 0) {
#define HOST_KIN(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
#define HOST_RH(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(hcx, t_);
            HIPNLP_KNOT_HESS_PROGRAM(HOST_KIN, HOST_RH, )
#undef HOST_KIN
#undef HOST_RH
        } else {   // every phase wave by wave in a permuted order of the four waves
            WaveOrder wo{4, g_wave_order, {}};
#define HOST_KIN(w, fn, nt) wo.add((w), [&cx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_); });
#define HOST_RH(w, fn, nt) wo.add((w), [&hcx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(hcx, t_); });
#define HOST_BARRIER wo.flush();
            HIPNLP_KNOT_HESS_PROGRAM(HOST_KIN, HOST_RH, HOST_BARRIER)
#undef HOST_KIN
#undef HOST_RH
#undef HOST_BARRIER
            wo.flush();
        }
#undef HOST_KIN
#undef HOST_RH
        for (int i = 0; i < HL.nnz_knot; ++i) hess[HL.knot_base(k) + i] = hx->H[HL.perm[size_t(i)]];
        if (k == N - 1) for (int i = 0; i < HL.n_couple; ++i) hess[HL.couple_base() + i] = hx->H[HL.perm_couple[size_t(i)]];
    }
    delete hx;
    delete s;
}
void hostemu_dims(const hostemu_handle* h, int* n, int* m, int* nnz) { *n = h->L.n; *m = h->L.m; *nnz = h->L.nnz; }
void hostemu_sparsity(const hostemu_handle* h, int* irow, int* jcol) {
    for (int i = 0; i < h->L.nnz; ++i) { irow[i] = h->L.irow[size_t(i)]; jcol[i] = h->L.jcol[size_t(i)]; }
}
// constant entries of jac g: the mask in pattern order and — what hipnlp_set_params + the host path do — an array [nnz] holding the
// constants under the parameters p at their positions, a poison value everywhere else
void hostemu_constant_mask(const hostemu_handle* h, unsigned char* mask) { h->L.constant_mask(mask); }
void hostemu_constant_fill(const hostemu_handle* h, const double* p, double poison, double* jac) {
    const Layout& L = h->L;
    std::vector<double> pk(size_t(L.N) * PK_STRIDE), cval(js::COUNT);
    GParams gp;
    pack_params(p, L.N, pk.data(), gp);
    Layout::constant_values(h->d.settings, h->kt, gp, cval.data());
    for (int i = 0; i < L.nnz; ++i) jac[i] = poison;
    for (int k = 0; k < L.N; ++k) {
        const int v = L.variant_of(k);
        for (int i = 0; i < L.nnz_v[v]; ++i) if (L.jconst_pos[v][size_t(i)]) jac[L.jac_base(k) + i] = cval[size_t(L.jperm[v][size_t(i)])];
    }
    for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac[L.jac_glob_base + long(i)] = cval[size_t(L.jperm_glob[i])];
}
void hostemu_vary_counts(const hostemu_handle* h, int* nvary /*[3]*/, int* nnz_v /*[3]*/, int* nconst_total) {
    for (int v = 0; v < 3; ++v) { nvary[v] = h->L.nvary_v[v]; nnz_v[v] = h->L.nnz_v[v]; }
    *nconst_total = h->L.nconst_total;
}
// the window of native slots the trimmed scratch of the four-wave VARY kernels stages, and what the recorder saw (Layout::vary_partition_ok)
void hostemu_vary_partition(const hostemu_handle* h, int* out /*[5]: ok, smallest varying slot, largest, window begin, window end*/) {
    out[0] = h->L.vary_partition_ok ? 1 : 0; out[1] = h->L.vary_slot_min; out[2] = h->L.vary_slot_max;
    out[3] = js::V0; out[4] = js::V0 + js::vary_slots(h->d.settings.terrain == HIPNLP_TERRAIN_PLANAR);
}
// phase of the knot program (barriers passed) in which every entry of jac g (pattern order) and every row of g gets its value
void hostemu_output_phases(const hostemu_handle* h, unsigned char* jac_phase /*[nnz]*/, unsigned char* g_phase /*[m]*/) {
    const Layout& L = h->L;
    for (int i = 0; i < L.nnz; ++i) jac_phase[i] = 255;
    for (int i = 0; i < L.m; ++i) g_phase[i] = 255;
    for (int k = 0; k < L.N; ++k) {
        const int v = L.variant_of(k);
        for (int i = 0; i < L.nnz_v[v]; ++i) jac_phase[L.jac_base(k) + i] = L.jslot_phase[size_t(L.jperm[v][size_t(i)])];
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            const int a = L.g_a[v][size_t(slot)];
            if (a != G_NONE) g_phase[a + L.g_b[size_t(slot)] * k] = L.gslot_phase[size_t(slot)];
        }
    }
    for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac_phase[L.jac_glob_base + long(i)] = L.jslot_phase[size_t(L.jperm_glob[i])];
}
void hostemu_bounds(const hostemu_handle* h, const double* p, double* lbg, double* ubg) { h->L.bounds(p, nullptr, nullptr, lbg, ubg); }
void hostemu_bounds_x(const hostemu_handle* h, const double* p, double* lbx, double* ubx) { h->L.bounds(p, lbx, ubx, nullptr, nullptr); }
void hostemu_lift_map(const hostemu_handle* h, int* kept_row) { h->L.kept_rows(kept_row); }
int hostemu_m_full(const hostemu_handle* h) { return h->L.m_full; }
void hostemu_simple_rows(const hostemu_handle* h, int* is_simple, int* var) { h->L.simple_rows(is_simple, var); }
int hostemu_num_row_blocks(const hostemu_handle* h) { return int(h->L.blocks.size()); }
void hostemu_row_block(const hostemu_handle* h, int i, const char** name, int* first, int* rows, int* k0, int* nk) {
    const RowBlock& b = h->L.blocks[size_t(i)];
    *name = b.name.c_str(); *first = b.first_row; *rows = b.rows; *k0 = b.k0; *nk = b.nk;
}

void hostemu_stage_rows(const hostemu_handle* h, int k, int* rows) {
    const int v = h->L.variant_of(k);
    for (int s = 0; s < gs::COUNT; ++s) { const int a = h->L.g_a[v][size_t(s)]; rows[s] = a != G_NONE ? a + h->L.g_b[size_t(s)] * k : -1; }
}

// The knots [kb, ke) of the horizon — what ONE shard handle of a multi-device handle evaluates (hipnlp_multi_create): its kernel reads
// x at the places the staging below reads it and writes the entries of grad f / g / jac g its knots own into the arrays of the WHOLE
// problem; cost_knot [N][NCT] receives the knots' cost partials.  Nothing else of the arrays is touched.
void hostemu_eval_range(const hostemu_handle* h, const double* x, const double* p, int kb, int ke, double* grad, double* g, double* jac, double* cost_knot) {
    const Layout& L = h->L;
    const int N = L.N;
    std::vector<double> pk(size_t(N) * PK_STRIDE);
    GParams gp;
    pack_params(p, N, pk.data(), gp);
    KnotScratch* s = new KnotScratch();
    for (int k = kb; k < ke; ++k) {
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
        for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
        for (int i = 0; i < NXK; ++i) {
            s->x[i] = x[NXK * k + i];
            s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0;
        }
        for (int i = 0; i < NPER; ++i) s->xo[i] = k == 0 ? x[NXK * (N - 1) + periodicity_row_var(i)] : (k == N - 1 ? x[periodicity_row_var(i)] : 0.0);
        for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
        for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[size_t(k) * PK_STRIDE + i];
        KnotInfo ki{k, N, k == 0, k == N - 1};
        ValueEm em{s->g, s->jac};
        Ctx<ValueEm> cx(*s, h->kt, h->ks, gp, ki, em);
#define HOST_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
        HIPNLP_KNOT_PROGRAM(HOST_R, )
#undef HOST_R
        const int v = L.variant_of(k);
        const long jb = L.jac_base(k);
        for (int i = 0; i < L.nnz_v[v]; ++i) jac[jb + i] = s->jac[L.jperm[v][size_t(i)]];
        if (k == N - 1) for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac[L.jac_glob_base + long(i)] = s->jac[L.jperm_glob[i]];
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            const int a = L.g_a[v][size_t(slot)];
            if (a != G_NONE) g[a + L.g_b[size_t(slot)] * k] = s->g[slot];
        }
        for (int i = 0; i < NXK; ++i) grad[NXK * k + i] = s->grad[i];
        if (k == N - 1) for (int i = 0; i < NXG; ++i) grad[NXK * N + i] = 0.0;   // (the global variables carry no cost: the last knot writes the zeros)
        for (int i = 0; i < NCT; ++i) cost_knot[size_t(k) * NCT + i] = s->cost[i];
    }
    delete s;
}

// The split program (knot_body.h, "Workgroup specialisation"): every knot evaluated by TWO passes on two scratches poisoned with NaN — one
// runs the kinematic half of the task groups, the other the model-free half — and every output taken from the pass that owns it (slots of g
// and jac g by the recorded owner of the slot, gradient entries and cost terms by the ownership rules).  A group of one half that read
// scratch written by the other half, or an output both halves write, shows up as a NaN or a wrong bit against hostemu_eval.
// Returns the number of outputs whose owner did not write them (still poison).
int hostemu_eval_split(const hostemu_handle* h, const double* x, const double* p, double* f, double* grad, double* g, double* jac, double* cost_terms) {
    const Layout& L = h->L;
    const int N = L.N;
    std::vector<double> pk(size_t(N) * PK_STRIDE);
    GParams gp;
    pack_params(p, N, pk.data(), gp);
    for (int i = 0; i < NCT; ++i) cost_terms[i] = 0.0;
    int unwritten = 0;
    KnotScratch* half[2] = {new KnotScratch(), new KnotScratch()};
    for (int k = 0; k < N; ++k) {
        for (int q = 0; q < 2; ++q) {
            KnotScratch* s = half[q];
            std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
            for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
            for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
            for (int i = 0; i < NXK; ++i) { s->x[i] = x[NXK * k + i]; s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0; }
            for (int i = 0; i < NPER; ++i) s->xo[i] = k == 0 ? x[NXK * (N - 1) + periodicity_row_var(i)] : (k == N - 1 ? x[periodicity_row_var(i)] : 0.0);
            for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
            for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[size_t(k) * PK_STRIDE + i];
            KnotInfo ki{k, N, k == 0, k == N - 1};
            ValueEm em{s->g, s->jac};
            Ctx<ValueEm> cx(*s, h->kt, h->ks, gp, ki, em);
            const char* both = std::getenv("HOSTEMU_SPLIT_BOTH");   // debugging aid: a group run by BOTH halves (which group does the other half miss?)
#define HOST_R(w4, w8, fn, nt) if (int(split_task_is_model_free(#fn)) == q || (both && split_same(#fn, both))) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
            HIPNLP_KNOT_PROGRAM(HOST_R, )
#undef HOST_R
        }
        const int v = L.variant_of(k);
        const long jb = L.jac_base(k);
        for (int i = 0; i < L.nnz_v[v]; ++i) { const int slot = L.jperm[v][size_t(i)]; jac[jb + i] = half[L.jslot_owner[size_t(slot)]]->jac[slot]; unwritten += std::isnan(jac[jb + i]); }
        if (k == N - 1) for (size_t i = 0; i < L.jperm_glob.size(); ++i) { const int slot = L.jperm_glob[i]; jac[L.jac_glob_base + long(i)] = half[L.jslot_owner[size_t(slot)]]->jac[slot]; }
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            const int a = L.g_a[v][size_t(slot)];
            if (a != G_NONE) { g[a + L.g_b[size_t(slot)] * k] = half[L.gslot_owner[size_t(slot)]]->g[slot]; unwritten += std::isnan(half[L.gslot_owner[size_t(slot)]]->g[slot]); }
        }
        for (int i = 0; i < NXK; ++i) { grad[NXK * k + i] = half[split_grad_is_model_free(i) ? 1 : 0]->grad[i]; unwritten += std::isnan(grad[NXK * k + i]); }
        for (int i = 0; i < NCT; ++i) { const double c = half[split_cost_is_model_free(i) ? 1 : 0]->cost[i]; cost_terms[i] += c; unwritten += std::isnan(c); }
    }
    for (int i = 0; i < NXG; ++i) grad[NXK * N + i] = 0.0;
    delete half[0];
    delete half[1];
    double ft = 0.0;
    for (int i = 0; i < NCT; ++i) ft += cost_terms[i];
    *f = ft;
    return unwritten;
}

void hostemu_eval(const hostemu_handle* h, const double* x, const double* p, double* f, double* grad, double* g, double* jac, double* cost_terms) {
    const Layout& L = h->L;
    const int N = L.N;
    std::vector<double> pk(size_t(N) * PK_STRIDE);
    GParams gp;
    pack_params(p, N, pk.data(), gp);
    for (int i = 0; i < NCT; ++i) cost_terms[i] = 0.0;
    for (int i = 0; i < L.n; ++i) grad[i] = 0.0;
    KnotScratch* s = new KnotScratch();
    for (int k = 0; k < N; ++k) {
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
        for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
        for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
        for (int i = 0; i < NXK; ++i) {
            s->x[i] = x[NXK * k + i];
            s->xm[i] = k > 0 ? x[NXK * (k - 1) + i] : 0.0;
        }
        for (int i = 0; i < NPER; ++i) s->xo[i] = k == 0 ? x[NXK * (N - 1) + periodicity_row_var(i)] : (k == N - 1 ? x[periodicity_row_var(i)] : 0.0);
        for (int i = 0; i < NXG; ++i) s->xg[i] = x[NXK * N + i];
        for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[size_t(k) * PK_STRIDE + i];
        KnotInfo ki{k, N, k == 0, k == N - 1};
        ValueEm em{s->g, s->jac};
        Ctx<ValueEm> cx(*s, h->kt, h->ks, gp, ki, em);
#define HOST_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
        HIPNLP_KNOT_PROGRAM(HOST_R, )
#undef HOST_R
        // copy-out exactly as the kernel does
        const int v = L.variant_of(k);
        const long jb = L.jac_base(k);
        for (int i = 0; i < L.nnz_v[v]; ++i) jac[jb + i] = s->jac[L.jperm[v][size_t(i)]];
        if (k == N - 1) for (size_t i = 0; i < L.jperm_glob.size(); ++i) jac[L.jac_glob_base + long(i)] = s->jac[L.jperm_glob[i]];
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            const int a = L.g_a[v][size_t(slot)];
            if (a != G_NONE) g[a + L.g_b[size_t(slot)] * k] = s->g[slot];
        }
        for (int i = 0; i < NXK; ++i) grad[NXK * k + i] = s->grad[i];
        for (int i = 0; i < NCT; ++i) cost_terms[i] += s->cost[i];
    }
    delete s;
    double ft = 0.0;
    for (int i = 0; i < NCT; ++i) ft += cost_terms[i];
    *f = ft;
}

// The knot program with the task groups of every phase run WAVE BY WAVE in a chosen order of the waves (order 0: ascending, 1: descending,
// 2: rotated by waves / 2), the groups of one wave in table order — what the device guarantees and nothing more.  Groups on different
// waves of one phase must not depend on each other (they run concurrently on the GPU): any order must give the values of the plain
// table-order run, bit for bit.  waves = 4 / 8 picks the column of the program table; the terrain picks HIPNLP_W4 / W8's alternative.
int hostemu_eval_wave_order(const hostemu_handle* h, const double* x, const double* p, int waves, int order, double* f, double* grad, double* g,
                            double* jac, double* cost_terms) {
    if ((waves != 4 && waves != 8) || order < 0 || order > 2) return -1;
    const bool planar = h->ks.terrain == HIPNLP_TERRAIN_PLANAR;
    if (waves == 4) return planar ? eval_wave_order_t<4, HIPNLP_TERRAIN_PLANAR>(h, x, p, order, f, grad, g, jac, cost_terms)
                                  : eval_wave_order_t<4, HIPNLP_TERRAIN_SMOOTH_STEPS>(h, x, p, order, f, grad, g, jac, cost_terms);
    return planar ? eval_wave_order_t<8, HIPNLP_TERRAIN_PLANAR>(h, x, p, order, f, grad, g, jac, cost_terms)
                  : eval_wave_order_t<8, HIPNLP_TERRAIN_SMOOTH_STEPS>(h, x, p, order, f, grad, g, jac, cost_terms);
}

int hostemu_eval_compact(const hostemu_handle* h, const double* x, const double* p, double* f, double* grad, double* g, double* jac, double* cost_terms) {
    return eval_compact<LAYOUT_COMPACT>(h, x, p, f, grad, g, jac, cost_terms);
}

// ---- static pose finder: the pose program of pose_body.h + the copy-out of hipnlp_pose_kernel -------------------------------
struct hostemu_pose_handle {
    hipnlp_pose_desc d;
    KinTables kt;
    KSettings ks;
    PoseHands hands;
    PoseLayout L;
};

hostemu_pose_handle* hostemu_pose_create(const hipnlp_pose_desc* desc, char* err, int errlen) {
    hostemu_pose_handle* h = new hostemu_pose_handle();
    h->d = *desc;
    std::string e;
    if (Layout::make_kin_tables(desc->model, h->kt, e)) Layout::fill_terrain_tops(h->kt, desc->settings.terrain, desc->settings.n_terrain_steps, desc->settings.terrain_steps);
    if (!e.empty() || !h->L.build(desc->settings, h->kt)) {
        if (e.empty()) e = h->L.error;
        std::strncpy(err, e.c_str(), size_t(errlen - 1));
        delete h;
        return nullptr;
    }
    h->ks = PoseLayout::make_ksettings(desc->settings);
    h->hands = PoseLayout::make_hands(desc->settings);
    return h;
}
void hostemu_pose_destroy(hostemu_pose_handle* h) { delete h; }
void hostemu_pose_dims(const hostemu_pose_handle* h, int* n, int* m, int* nnz) { *n = h->L.n; *m = h->L.m; *nnz = h->L.nnz; }
void hostemu_pose_sparsity(const hostemu_pose_handle* h, int* irow, int* jcol) {
    for (int i = 0; i < h->L.nnz; ++i) { irow[i] = h->L.irow[size_t(i)]; jcol[i] = h->L.jcol[size_t(i)]; }
}
void hostemu_pose_bounds(const hostemu_pose_handle* h, const double* p, double* lbg, double* ubg) { h->L.bounds(p, lbg, ubg); }
int hostemu_pose_num_row_blocks(const hostemu_pose_handle* h) { return int(h->L.blocks.size()); }
void hostemu_pose_row_block(const hostemu_pose_handle* h, int i, const char** name, int* first, int* rows) {
    const PoseRowBlock& b = h->L.blocks[size_t(i)];
    *name = b.name.c_str(); *first = b.first_row; *rows = b.rows;
}
void hostemu_pose_eval(const hostemu_pose_handle* h, const double* x, const double* p, double* f, double* grad, double* g, double* jac, double* cost_terms) {
    const PoseLayout& L = h->L;
    double pk[PK_STRIDE], xr[64];
    GParams gp;
    pack_pose_params(p, pk, xr, gp);
    KnotScratch* s = new KnotScratch();
    std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
    for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
    for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
    for (int i = 0; i < POSE_NX; ++i) s->x[pose_to_knot_col(i)] = x[i];
    for (int i = 0; i < 64; ++i) s->xm[i] = xr[i];
    for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[i];
    KnotInfo ki{1, 3, 0, 0};
    std::vector<double> stg(size_t(pjs::slots(false)), std::nan(""));   // (the larger of the two terrains' stagings)
    ValueEmPoseStaged em{s->g, stg.data()};
    Ctx<ValueEmPoseStaged> cx(*s, h->kt, h->ks, gp, ki, em);
    cx.hands = &h->hands;
    if (g_wave_order < 0) {
#define HOST_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
        HIPNLP_POSE_PROGRAM(HOST_R, )
#undef HOST_R
    } else {   // every phase wave by wave in a permuted order of the four waves (see hostemu_eval_wave_order)
        WaveOrder wo{4, g_wave_order, {}};
#define HOST_R(w4, w8, fn, nt) wo.add((w4), [&cx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_); });
#define HOST_BARRIER wo.flush();
        HIPNLP_POSE_PROGRAM(HOST_R, HOST_BARRIER)
#undef HOST_R
#undef HOST_BARRIER
        wo.flush();
    }
    for (int e = 0; e < L.nnz; ++e) {   // (the device table holds pjs::index of the pattern's slots; hipnlp_pose_create refuses a slot that is not kept)
        const int slot = L.jperm[size_t(e)];
        if (!pjs::kept(slot)) { g_pose_map_violations++; jac[e] = std::nan(""); continue; }
        jac[e] = stg[size_t(pjs::index(slot))];
    }
    for (int slot = 0; slot < gs::COUNT; ++slot) if (L.g_row[size_t(slot)] >= 0) g[L.g_row[size_t(slot)]] = s->g[slot];
    for (int i = 0; i < POSE_NX; ++i) grad[i] = s->grad[pose_to_knot_col(i)];
    double ft = 0.0;
    for (int t = 0; t < POSE_NCT; ++t) { cost_terms[t] = pose_cost_term(*s, t); ft += cost_terms[t]; }
    (void)ft;
    *f = s->cost[CT_POSE_TOTAL];   // (t_pose_cost_total: the kernel's f — NaN if the task ran before a term it adds was written)
    delete s;
}
long hostemu_pose_map_violations(void) { return g_pose_map_violations; }
void hostemu_pose_hess_dims(const hostemu_pose_handle* h, int* hnnz) { *hnnz = h->L.hnnz; }
void hostemu_pose_hess_sparsity(const hostemu_pose_handle* h, int* irow, int* jcol) {
    for (int i = 0; i < h->L.hnnz; ++i) { irow[i] = h->L.hrow[size_t(i)]; jcol[i] = h->L.hcol[size_t(i)]; }
}
// the Hessian tasks of pose_hess_body.h behind the pose program, copy-out of hipnlp_pose_hess_kernel
void hostemu_pose_hess(const hostemu_pose_handle* h, const double* x, const double* p, double sigma, const double* lambda, double* hess) {
    const PoseLayout& L = h->L;
    double pk[PK_STRIDE], xr[64];
    GParams gp;
    pack_pose_params(p, pk, xr, gp);
    KnotScratch* s = new KnotScratch();
    HessScratch* hx = new HessScratch();
    std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), std::nan(""));
    std::fill(reinterpret_cast<double*>(hx), reinterpret_cast<double*>(hx) + sizeof(HessScratch) / sizeof(double), std::nan(""));
    for (int i = 0; i < XPAD; ++i) { s->x[i] = 0; s->xm[i] = 0; }
    for (int i = 0; i < NPER; ++i) s->xo[i] = 0;
    for (int i = 0; i < POSE_NX; ++i) s->x[pose_to_knot_col(i)] = x[i];
    for (int i = 0; i < 64; ++i) s->xm[i] = xr[i];
    for (int i = 0; i < PK_STRIDE; ++i) s->pk[i] = pk[i];
    // the multipliers as the device kernel stages them (hipnlp_pose.hip): zeros, then thread i scatters the multiplier of ROW i to the slot of row i
    // (the inverse of g_row: every row of the pattern lives in exactly one slot — a row without a slot would lose its multiplier here)
    for (int slot = 0; slot < gs::COUNT; ++slot) hx->lam[slot] = 0.0;
    {
        std::vector<int> row_slot(size_t(L.m), -1);
        for (int slot = 0; slot < gs::COUNT; ++slot) if (L.g_row[size_t(slot)] >= 0) row_slot[size_t(L.g_row[size_t(slot)])] = slot;
        for (int r = 0; r < L.m; ++r) if (row_slot[size_t(r)] >= 0) hx->lam[row_slot[size_t(r)]] = lambda[r]; else g_pose_map_violations++;
    }
    hx->sigma = sigma;
    KnotInfo ki{1, 3, 0, 0};
    ValueEm em{s->g, s->jac, hx->H};
    Ctx<ValueEm> cx(*s, h->kt, h->ks, gp, ki, em);
    cx.hands = &h->hands;
    HCtx<ValueEm> hcx{cx, *hx};
    if (g_wave_order < 0) {
#define HOST_KIN(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
#define HOST_RH(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(hcx, t_);
        HIPNLP_POSE_HESS_PROGRAM(HOST_KIN, HOST_RH, )
#undef HOST_KIN
#undef HOST_RH
    } else {
        WaveOrder wo{4, g_wave_order, {}};
#define HOST_KIN(w, fn, nt) wo.add((w), [&cx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_); });
#define HOST_RH(w, fn, nt) wo.add((w), [&hcx]() { for (int t_ = 0; t_ < (nt); ++t_) fn(hcx, t_); });
#define HOST_BARRIER wo.flush();
        HIPNLP_POSE_HESS_PROGRAM(HOST_KIN, HOST_RH, HOST_BARRIER)
#undef HOST_KIN
#undef HOST_RH
#undef HOST_BARRIER
        wo.flush();
    }
    for (int e = 0; e < L.hnnz; ++e) hess[e] = hx->H[L.hperm[size_t(e)]];
    delete hx;
    delete s;
}
}  // extern "C"
