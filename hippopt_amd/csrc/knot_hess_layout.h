// knot_hess_layout.h — sparsity pattern of the Hessian of the Lagrangian of the kinodynamic NLP, recorded from the tasks of
// knot_hess_body.h (the same single-source pattern as the Jacobian: what a task emits IS the structure).
// Lower triangle, triplets.  Order: knot 0 block | knot 1 block | ... | knot N-1 block | periodicity coupling (last x first),
// every knot block sorted by (column, row).  All knots share one block pattern (rows that do not exist at a knot have a zero
// multiplier there; costs that start at k = 1 are switched off numerically at k = 0).
#pragma once
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "knot_hess_body.h"
#include "layout.h"

namespace hipnlp {

struct HessLayout {
    int N = 0;
    int nnz_knot = 0;                 // entries of one knot block
    int n_couple = 0;                 // periodicity-coupling entries (0 or 84)
    long nnz = 0;
    std::vector<int32_t> perm;        // [nnz_knot] position in the knot block -> native slot
    std::vector<int32_t> lrow, lcol;  // [nnz_knot] variable offsets inside the knot record
    std::vector<int32_t> perm_couple, crow, ccol;   // coupling: slot, variable of the last knot (row), of the first knot (column)
    // phase of the Hessian program (barriers passed) in which the entry at position i of a knot block is emitted — every entry is emitted
    // exactly once (a second emission is refused above), so that is when its value is final; early_run: the entries [0, early_run) of a
    // block — the point columns come first in (column, row) order — that are ALL final behind barrier number EARLY_PHASE
    std::vector<unsigned char> pos_phase;
    int early_run = 0, early_phase = 0;   // early_phase: the barrier (1 = the first) behind which the run is final — the earliest of the first three that gives the longest run
    std::string error;

    bool build(const hipnlp_settings& st, const KinTables& kt) {
        N = st.horizon;
        std::vector<int> grow(gs::COUNT, -1), jrid(js::COUNT, -1), jc(js::COUNT, -1), hrow(hk::COUNT, -1), hcol(hk::COUNT, -1);
        bool dup = false;
        KnotScratch* s = new KnotScratch();
        KHessScratch* hx = new KHessScratch();
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), 0.0);
        std::fill(reinterpret_cast<double*>(hx), reinterpret_cast<double*>(hx) + sizeof(KHessScratch) / sizeof(double), 0.0);
        s->x[QB_ + 3] = 1.0;
        kh_fill_far_lists(kt, hx->far);
        KSettings ks = Layout::make_ksettings(st);
        GParams gp{};
        gp.dt = 0.1; gp.mass = 1.0;
        KnotInfo ki{0, N, 1, 1};
        RecordEm em{grow.data(), jrid.data(), jc.data(), &dup, hrow.data(), hcol.data()};
        std::vector<unsigned char> hphase(hk::COUNT, 255);
        em.hphase = hphase.data();
        Ctx<RecordEm> cx(*s, kt, ks, gp, ki, em);
        KHCtx<RecordEm> hcx{cx, *hx, s->g};
#define HOST_KIN(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
#define HOST_RH(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(hcx, t_);
#define HOST_BAR cx.em.phase++;
        HIPNLP_KNOT_HESS_PROGRAM(HOST_KIN, HOST_RH, HOST_BAR)
#undef HOST_BAR
#undef HOST_KIN
#undef HOST_RH
        delete hx;
        delete s;
        if (dup) { error = "internal: native Hessian slot emitted twice"; return false; }
        std::vector<std::pair<std::pair<int, int>, int>> ent, cpl;   // ((col, row), slot)
        for (int slot = 0; slot < hk::COUNT; ++slot) {
            if (hrow[size_t(slot)] < 0) continue;
            if (hcol[size_t(slot)] >= COL_FIRST) cpl.push_back({{hcol[size_t(slot)] - COL_FIRST, hrow[size_t(slot)]}, slot});
            else {
                if (hrow[size_t(slot)] < hcol[size_t(slot)]) { error = "internal: Hessian entry above the diagonal"; return false; }
                ent.push_back({{hcol[size_t(slot)], hrow[size_t(slot)]}, slot});
            }
        }
        std::sort(ent.begin(), ent.end());
        std::sort(cpl.begin(), cpl.end());
        for (size_t i = 1; i < ent.size(); ++i)
            if (ent[i].first == ent[i - 1].first) { error = "internal: duplicate Hessian entry"; return false; }
        perm.clear(); lrow.clear(); lcol.clear(); perm_couple.clear(); crow.clear(); ccol.clear();
        for (auto& e : ent) { perm.push_back(e.second); lcol.push_back(e.first.first); lrow.push_back(e.first.second); }
        for (auto& e : cpl) { perm_couple.push_back(e.second); ccol.push_back(e.first.first); crow.push_back(e.first.second); }
        pos_phase.clear();
        for (int32_t slot : perm) pos_phase.push_back(hphase[size_t(slot)]);
        early_run = 0; early_phase = 0;
        for (int P = 1; P <= 3; ++P) {   // (at least the last three phases of the six still to run)
            int run = 0;
            while (run < int(pos_phase.size()) && pos_phase[size_t(run)] < P) ++run;
            if (run > early_run) { early_run = run; early_phase = P; }
        }
        nnz_knot = int(perm.size());
        n_couple = int(perm_couple.size());
        nnz = long(nnz_knot) * N + n_couple;
        return true;
    }
    long knot_base(int k) const { return long(nnz_knot) * k; }
    long couple_base() const { return long(nnz_knot) * N; }
    void pattern(int* irow, int* jcol) const {
        long e = 0;
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < nnz_knot; ++i, ++e) { irow[e] = NXK * k + lrow[size_t(i)]; jcol[e] = NXK * k + lcol[size_t(i)]; }
        for (int i = 0; i < n_couple; ++i, ++e) { irow[e] = NXK * (N - 1) + crow[size_t(i)]; jcol[e] = ccol[size_t(i)]; }
    }
};

}  // namespace hipnlp
