// layout.h — host-side structure of the NLP: the row-block directory in the reference's subject_to
// order, the map from the kernel's native slots to global rows / CCS positions, bounds.
// Header-only; used by the C-ABI implementation (hipnlp.hip) and by the test-only host emulation.
//
// Reference: call order of turnkey_planners/humanoid_kinodynamic/planner.py:124-176, row naming of
// base/multiple_shooting_solver.py:682-742,807-824 (name[0] initial-condition rows, name[k] per knot),
// Opti canonical forms.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "knot_body.h"

namespace hipnlp {

// variable behind final-state row i (0..104), or -1 for the descriptor rows; *slot = index among the 81 variable
// rows; *desc = 3c+comp of the descriptor entry.  Row order = HumanoidState.to_list() (sorted flat keys).
inline int final_row_var(int i, int* slot, int* desc) {
    *desc = -1; *slot = -1;
    if (i < 3) { *slot = i; return COM_ + i; }
    int r = i - 3;
    if (r < 72) {
        const int c = r / 9, q = r % 9;
        if (q < 3) { *desc = 3 * c + q; return -1; }
        if (q < 6) { *slot = 3 + 6 * c + (q - 3); return PT_ * c + F_ + (q - 3); }
        *slot = 3 + 6 * c + 3 + (q - 6); return PT_ * c + P_ + (q - 6);
    }
    r -= 72;
    if (r < 3) { *slot = 51 + r; return PB_ + r; }
    r -= 3;
    if (r < 4) { *slot = 54 + r; return QB_ + r; }
    r -= 4;
    *slot = 58 + r;
    return S_ + r;
}
struct RowBlock {
    std::string name;
    int first_row, rows, k0, nk;      // the reference's named constraint: rows in FULL numbering (every subject_to row)
    // detect_simple_bounds (Layout::lifted): the rows of the block that stay in g — `kept` of its `rows` per knot, row i at position
    // kidx[i] (-1: lifted into a bound on its variable), the first of them at first_kept of the reduced numbering
    int first_kept = 0, kept = 0;
    std::vector<int16_t> kidx;
};

enum Variant : int { VAR_FIRST = 0, VAR_INTERIOR = 1, VAR_LAST = 2 };

// recording emitter: remembers which row / column every native slot belongs to
struct RecordEm {
    static constexpr int kTerrain = -1;  // terrain kind read from KSettings at run time
    int* grow;        // [gs::COUNT] row id or -1
    int* jrowid;      // [js::COUNT]
    int* jcol;        // [js::COUNT]
    bool* dup;
    int* hrow = nullptr;   // [hs::COUNT] Hessian slots (pose_hess_body.h): row / column variable or -1
    int* hcol = nullptr;
    unsigned char* hphase = nullptr;   // [hk::COUNT] or null: phase of the Hessian program (barriers passed) in which the slot is emitted — every slot once
    void H(int slot, int row, int col, double) { if (hrow[slot] != -1) *dup = true; hrow[slot] = row; hcol[slot] = col; if (hphase) hphase[slot] = (unsigned char)phase; }
    void G(int slot, int rid, double) { if (grow[slot] != -1) *dup = true; grow[slot] = rid; if (gphase) gphase[slot] = (unsigned char)phase; if (gowner) gowner[slot] = owner; }
    void J(int slot, int rid, int col, double) { if (jrowid[slot] != -1) *dup = true; jrowid[slot] = rid; jcol[slot] = col; if (jphase) jphase[slot] = (unsigned char)phase; if (jowner) jowner[slot] = owner; }
    // which half of the split program (knot_body.h, split_task_is_model_free) emits a slot: 1 = the model-free half, 0 = the kinematic one
    unsigned char owner = 0;
    unsigned char* gowner = nullptr;   // [gs::COUNT] or null
    unsigned char* jowner = nullptr;   // [js::COUNT] or null
    // the phase of the knot program (barriers passed) in which a slot gets its value: what is final after the second phase can leave early
    int phase = 0;
    unsigned char* gphase = nullptr;   // [gs::COUNT] or null
    unsigned char* jphase = nullptr;   // [js::COUNT] or null
    // an entry that does not depend on x (knot_body.h, emit_jc): the slot is marked constant and its value kept (a pass over the
    // program with the handle's own parameters, Layout::constant_values, reads it back)
    unsigned char* jconst = nullptr;   // [js::COUNT] or null
    double* jcval = nullptr;           // [js::COUNT] or null
    void JC(int slot, int rid, int col, double v) { J(slot, rid, col, v); if (jconst) jconst[slot] = 1; if (jcval) jcval[slot] = v; }
};

// parameter offsets in reference creation order (tests/golden/kinodyn_structure.json)
struct ParamOffsets {
    int N;
    explicit ParamOffsets(int n) : N(n) {}
    int desc(int k, int c) const { return 24 * k + 3 * c; }
    int mass() const { return 24 * N; }
    int init() const { return 24 * N + 3; }
    int fin() const { return init() + 105; }
    int sc() const { return fin() + 105; }
    int dt() const { return sc(); }
    int gravity() const { return sc() + 1; }
    int kt() const { return sc() + 7; }
    int kbs() const { return sc() + 8; }
    int eps() const { return sc() + 9; }
    int mu() const { return sc() + 10; }
    int umax() const { return sc() + 11; }
    int fdmax() const { return sc() + 14; }
    int lmax() const { return sc() + 17; }
    int hmin() const { return sc() + 18; }
    int dmin() const { return sc() + 19; }
    int hmax() const { return sc() + 20; }
    int jpmax() const { return sc() + 21; }
    int jpmin() const { return sc() + 44; }
    int jvmax() const { return sc() + 67; }
    int jvmin() const { return sc() + 90; }
    int ref(int k) const { return sc() + 113 + 55 * k; }
    int np() const { return 79 * N + 326; }
};

struct Layout {
    int N = 0, n = 0, m = 0, nnz = 0;
    // detect_simple_bounds (main_periodic_step.py:109-110: Opti runs nlpsol with {"detect_simple_bounds": True}): rows of g that are
    // exactly ONE decision variable leave g and jac g and become bounds on that variable.  lifted: this layout IS the reduced NLP
    // (m = rows kept, the CCS pattern without the lifted rows); m_full = rows of the reference's full subject_to list either way.
    int m_full = 0, n_lifted = 0;
    bool lifted = false;
    bool has_final = false, has_per = false, has_hx0 = false;
    std::vector<RowBlock> blocks;
    int blk[RK_COUNT][NC];                // directory index of (kind, point) or -1
    // kernel tables
    std::vector<int32_t> g_a[3];          // [gs::COUNT] row of slot at knot k is g_a + g_b*k ; G_NONE = not written
    std::vector<int32_t> g_b;
    std::vector<int32_t> jperm[3];        // position within the knot's column block -> native slot (CCS order, or — vary_first — the
                                          // entries that depend on x in CCS order, then the constant ones in CCS order)
    std::vector<int32_t> jperm_glob;      // entries in the horizon-global columns (written by knot 0)
    int nnz_v[3] = {0, 0, 0};
    int jac_glob_base = 0;
    // Entries of jac g that do not depend on x (emit_jc in knot_body.h: +-1, -dt/2, the mass, +-1/4 of the linear rows): 43 % of the
    // pattern at N = 100.  jconst_slot marks the native slots, jconst_pos[v][i] position i of the column block of a variant-v knot,
    // nvary_v[v] counts the entries of a block that DO depend on x; the horizon-global entries (jperm_glob) are all constant.
    // vary_first (HIPNLP_FLAG_JAC_VARYING_FIRST): a knot's block lists its varying entries first — the kernel's stores into a host
    // array whose constants were filled once are then ONE contiguous run per knot.  IPOPT takes triplets in any order.
    bool vary_first = false;
    std::vector<unsigned char> gslot_phase, jslot_phase;   // [gs::COUNT], [js::COUNT]: phase of the knot program in which the slot gets its value (255: never)
    std::vector<unsigned char> gslot_owner, jslot_owner;   // ... and the half of the split program that emits it (1: model-free, 0: kinematic; recorded)
    std::vector<unsigned char> jconst_slot;
    std::vector<unsigned char> jconst_pos[3];
    int nvary_v[3] = {0, 0, 0};
    int nconst_total = 0;                 // constant entries of the whole pattern
    // The trimmed scratch of the four-wave VARY kernels stages the slots [js::V0, js::V0 + js::vary_slots(terrain)) only, addressed through
    // a base moved back by js::V0: every recorded slot that is NOT constant must lie in that window (a slot below it would be a store in
    // front of the staging, one behind it a store into grad[]).  Checked from what the recorder saw, not from the numbering's intent;
    // hipnlp_create launches no VARY kernel for a layout that fails it (tests/test_constant_jacobian.py asserts it holds).
    bool vary_partition_ok = false;
    int vary_slot_min = -1, vary_slot_max = -1;   // smallest / largest recorded slot that depends on x
    std::vector<int32_t> irow, jcol;      // full pattern, CCS order
    std::string error;

    static std::string point_name(int c) {
        return std::string("system.contact_points.") + (c < 4 ? "left[" : "right[") + std::to_string(c % 4) + "]";
    }
    int variant_of(int k) const { return k == 0 ? VAR_FIRST : (k == N - 1 ? VAR_LAST : VAR_INTERIOR); }
    long jac_base(int k) const { return k == 0 ? 0 : long(nnz_v[VAR_FIRST]) + long(k - 1) * nnz_v[VAR_INTERIOR]; }

    int add(const std::string& name, int rows, int k0, int nk) {
        RowBlock b{name, m_full, rows, k0, nk};
        m_full += rows * nk;
        blocks.push_back(b);
        return int(blocks.size()) - 1;
    }
    // row i of block b at its kk-th knot, in the numbering of THIS layout (-1: lifted)
    int row_of(const RowBlock& b, int kk, int i) const {
        if (!lifted) return b.first_row + b.rows * kk + i;
        return b.kidx[size_t(i)] < 0 ? -1 : b.first_kept + b.kept * kk + b.kidx[size_t(i)];
    }
    // decision variable (0..188 within the knot) behind row i of a (kind, point) block when that row is exactly one variable, else -1:
    // the x_0 == initial_state rows, the u_v and joint boxes of planner.py:386-405,699-719, the final-state rows that hold a variable
    static int simple_var(int kind, int c, int i) {
        switch (kind) {
            case RK_FDYN_X0: return PT_ * c + F_ + i;
            case RK_PDYN_X0: return PT_ * c + P_ + i;
            case RK_UB: return PT_ * c + U_ + i;
            case RK_PBDYN_X0: return PB_ + i;
            case RK_QBDYN_X0: return QB_ + i;
            case RK_SDYN_X0: return S_ + i;
            case RK_COMDYN_X0: return COM_ + i;
            case RK_JPB: return S_ + i;
            case RK_JVB: return SD_ + i;
            case RK_FIN: { int slot, desc; return final_row_var(i, &slot, &desc); }
            default: return -1;
        }
    }
    void add_dyn(const std::string& name, int L, int kin, int c, bool x0) {
        if (x0) blk[kin + 2][c] = add(name + "[0]", L, 0, 1);
        blk[kin][c] = blk[kin + 1][c] = add(name, L, 1, N - 1);
    }

    // global row of (row id) as seen from knot k, or -1
    int resolve(int rid, int k) const {
        const int kind = rid_kind(rid), c = rid_point(rid), i = rid_index(rid);
        const int bi = blk[kind][c];
        if (bi < 0) return -1;
        const RowBlock& b = blocks[size_t(bi)];
        switch (kind) {
            case RK_FDYN_OUT: case RK_PDYN_OUT: case RK_PBDYN_OUT: case RK_QBDYN_OUT: case RK_SDYN_OUT: case RK_COMDYN_OUT: case RK_HDYN_OUT:
                return (k + 1 <= N - 1) ? row_of(b, k, i) : -1;   // row of knot k+1 (k0 = 1)
            case RK_FDYN_X0: case RK_PDYN_X0: case RK_PBDYN_X0: case RK_QBDYN_X0: case RK_SDYN_X0: case RK_COMDYN_X0: case RK_HDYN_X0: case RK_PER0:
                return k == 0 ? row_of(b, 0, i) : -1;
            case RK_FIN: case RK_PERN:
                return k == N - 1 ? row_of(b, 0, i) : -1;
            default:
                return (k >= b.k0 && k < b.k0 + b.nk) ? row_of(b, k - b.k0, i) : -1;
        }
    }

    bool build(const hipnlp_settings& st, const KinTables& kt, bool lift_simple_bounds = false, bool varying_first = false) {
        N = st.horizon;
        vary_first = varying_first;
        if (N < 2) { error = "horizon must be >= 2"; return false; }
        n = NXK * N + NXG;
        m = m_full = n_lifted = 0;
        lifted = lift_simple_bounds;
        blocks.clear();
        for (int a = 0; a < RK_COUNT; ++a) for (int c = 0; c < NC; ++c) blk[a][c] = -1;
        has_final = st.final_state_type == HIPNLP_EXPR_SUBJECT_TO;
        has_per = st.periodicity_type == HIPNLP_EXPR_SUBJECT_TO;
        has_hx0 = st.periodicity_type == HIPNLP_EXPR_SKIP;  // planner.py:580-584
        // ---- directory in the reference's call order --------------------------------------------
        for (int c = 0; c < NC; ++c) {  // planner.py:124-147
            const std::string pn = point_name(c);
            add_dyn(pn + ".f_dynamics", 3, RK_FDYN_IN, c, true);
            add_dyn(pn + ".p_dynamics", 3, RK_PDYN_IN, c, true);
            blk[RK_PLANAR][c] = add(pn + ".p_planar_complementarity", 3, 0, N);
            blk[RK_DCC][c] = add(pn + ".p_dcc", 1, 0, N);
            blk[RK_HEIGHT][c] = add(pn + ".p_height", 1, 1, N - 1);
            blk[RK_NORMAL][c] = add(pn + ".f_normal", 1, 1, N - 1);
            blk[RK_FRICTION][c] = add(pn + ".f_friction", 1, 1, N - 1);
            blk[RK_UB][c] = add(pn + ".u_v_bounds", 3, 0, N);
            blk[RK_FDB][c] = add(pn + ".f_dot_bounds", 3, 0, N);
            blk[RK_KINC][c] = add(pn + ".p_kinematics_consistency", 3, 1, N - 1);
        }
        add_dyn("base_position_dynamics", 3, RK_PBDYN_IN, 0, true);       // planner.py:522-588
        add_dyn("base_quaternion_dynamics", 4, RK_QBDYN_IN, 0, true);
        add_dyn("joint_position_dynamics", NJ, RK_SDYN_IN, 0, true);
        add_dyn("com_dynamics", 3, RK_COMDYN_IN, 0, true);
        add_dyn("centroidal_momentum_dynamics", 6, RK_HDYN_IN, 0, has_hx0);
        blk[RK_UNITQ][0] = add("unitary_quaternion", 1, 1, N - 1);      // planner.py:266-425
        blk[RK_COMC][0] = add("com_kinematics_consistency", 3, 1, N - 1);
        blk[RK_CMMC][0] = add("centroidal_momentum_kinematics_consistency", 3, 0, N);
        blk[RK_AMB][0] = add("angular_momentum_bounds", 3, 0, N);
        blk[RK_COMH][0] = add("minimum_com_height", 1, 1, N - 1);
        blk[RK_FEETD][0] = add("minimum_feet_distance", 1, 1, N - 1);
        blk[RK_JPB][0] = add("joint_position_bounds", NJ, 1, N - 1);
        blk[RK_JVB][0] = add("joint_velocity_bounds", NJ, 0, N);
        if (has_final) blk[RK_FIN][0] = add("final_state_expression", 105, N - 1, 1);
        blk[RK_FEETH][0] = add("maximum_feet_relative_height", 1, 1, N - 1);  // planner.py:215-247
        if (has_per) blk[RK_PER0][0] = blk[RK_PERN][0] = add("periodicity_expression", 84, N - 1, 1);  // planner.py:897-930

        // ---- rows kept by detect_simple_bounds: per block, the position of every row among the kept ones ---------------
        for (RowBlock& b : blocks) { b.kidx.assign(size_t(b.rows), 0); for (int i = 0; i < b.rows; ++i) b.kidx[size_t(i)] = int16_t(i); b.kept = b.rows; }
        for (int kind = 0; kind < RK_COUNT; ++kind)
            for (int c = 0; c < NC; ++c) {
                if (blk[kind][c] < 0) continue;
                RowBlock& b = blocks[size_t(blk[kind][c])];
                bool any = false;
                for (int i = 0; i < b.rows; ++i) any |= simple_var(kind, c, i) >= 0;
                if (!any) continue;
                int q = 0;
                for (int i = 0; i < b.rows; ++i) b.kidx[size_t(i)] = simple_var(kind, c, i) >= 0 ? int16_t(-1) : int16_t(q++);
                b.kept = q;
            }
        {
            int at = 0;
            for (RowBlock& b : blocks) { b.first_kept = at; at += b.kept * b.nk; }
            n_lifted = m_full - at;
            m = lifted ? at : m_full;
        }

        // ---- record the kernel body's native slots --------------------------------------------------
        std::vector<int> grow(gs::COUNT, -1), jrid(js::COUNT, -1), jc(js::COUNT, -1);
        jconst_slot.assign(js::COUNT, 0);
        bool dup = false;
        {
            KnotScratch* s = new KnotScratch();
            std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), 0.0);
            s->x[QB_ + 3] = 1.0;
            KSettings ks = make_ksettings(st);
            GParams gp{};
            gp.dt = 0.1; gp.mass = 1.0;
            KnotInfo ki{0, N, 1, 1};
            RecordEm em{grow.data(), jrid.data(), jc.data(), &dup};
            em.jconst = jconst_slot.data();
            gslot_phase.assign(gs::COUNT, 255);
            jslot_phase.assign(js::COUNT, 255);
            em.gphase = gslot_phase.data();
            em.jphase = jslot_phase.data();
            gslot_owner.assign(gs::COUNT, 0);
            jslot_owner.assign(js::COUNT, 0);
            em.gowner = gslot_owner.data();
            em.jowner = jslot_owner.data();
            Ctx<RecordEm> cx(*s, kt, ks, gp, ki, em);
#define HOST_R(w4, w8, fn, nt) cx.em.owner = split_task_is_model_free(#fn) ? 1 : 0; for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
#define HOST_BAR cx.em.phase++;
            HIPNLP_KNOT_PROGRAM(HOST_R, HOST_BAR)
#undef HOST_BAR
#undef HOST_R
            delete s;
        }
        if (dup) { error = "internal: native slot emitted twice"; return false; }
        {
            const int v_end = js::V0 + js::vary_slots(st.terrain == HIPNLP_TERRAIN_PLANAR);
            vary_partition_ok = true;
            vary_slot_min = vary_slot_max = -1;
            for (int slot = 0; slot < js::COUNT; ++slot) {
                if (jrid[size_t(slot)] < 0 || jconst_slot[size_t(slot)]) continue;
                if (vary_slot_min < 0) vary_slot_min = slot;
                vary_slot_max = slot;
                if (slot < js::V0 || slot >= v_end) vary_partition_ok = false;
            }
        }

        // ---- g tables ------------------------------------------------------------------------------------
        g_b.assign(gs::COUNT, 0);
        for (int v = 0; v < 3; ++v) g_a[v].assign(gs::COUNT, G_NONE);
        const int krep[3] = {0, 1, N - 1};
        for (int slot = 0; slot < gs::COUNT; ++slot) {
            if (grow[size_t(slot)] < 0) continue;
            const int rid = grow[size_t(slot)];
            const int bi = blk[rid_kind(rid)][rid_point(rid)];
            if (bi < 0) continue;
            const int stride = blocks[size_t(bi)].nk > 1 ? (lifted ? blocks[size_t(bi)].kept : blocks[size_t(bi)].rows) : 0;
            g_b[size_t(slot)] = stride;
            for (int v = 0; v < 3; ++v) {
                if (v == VAR_INTERIOR && N < 3) continue;
                const int r = resolve(rid, krep[v]);
                if (r >= 0) g_a[v][size_t(slot)] = r - stride * krep[v];
            }
        }
        // ---- jac tables: CCS order inside the column block of a knot ------------------------------------
        jperm_glob.clear();
        for (int v = 0; v < 3; ++v) {
            jperm[v].clear();
            nnz_v[v] = 0;
            if (v == VAR_INTERIOR && N < 3) continue;
            std::vector<std::pair<std::pair<int, int>, int>> ent, glob;  // ((col,row),slot)
            for (int slot = 0; slot < js::COUNT; ++slot) {
                if (jrid[size_t(slot)] < 0) continue;
                const int r = resolve(jrid[size_t(slot)], krep[v]);
                if (r < 0) continue;
                if (jc[size_t(slot)] >= COL_GLOBAL) glob.push_back({{jc[size_t(slot)], r}, slot});
                else ent.push_back({{jc[size_t(slot)], r}, slot});
            }
            std::sort(ent.begin(), ent.end());
            for (size_t i = 1; i < ent.size(); ++i)
                if (ent[i].first == ent[i - 1].first) { error = "internal: duplicate jacobian entry"; return false; }
            if (vary_first)   // (stable: CCS order within the varying entries and within the constant ones)
                std::stable_partition(ent.begin(), ent.end(), [&](const std::pair<std::pair<int, int>, int>& e) { return !jconst_slot[size_t(e.second)]; });
            jconst_pos[v].clear();
            nvary_v[v] = 0;
            for (auto& e : ent) {
                jperm[v].push_back(e.second);
                jconst_pos[v].push_back(jconst_slot[size_t(e.second)]);
                nvary_v[v] += !jconst_slot[size_t(e.second)];
            }
            nnz_v[v] = int(ent.size());
            if (v == VAR_FIRST) {
                std::sort(glob.begin(), glob.end());
                for (auto& e : glob) {
                    if (!jconst_slot[size_t(e.second)]) { error = "internal: an entry in a horizon-global column depends on x"; return false; }
                    jperm_glob.push_back(e.second);
                }
            }
        }
        jac_glob_base = int(jac_base(N - 1)) + nnz_v[VAR_LAST];
        nnz = jac_glob_base + int(jperm_glob.size());
        // ---- full pattern ---------------------------------------------------------------------------------
        irow.clear(); jcol.clear();
        irow.reserve(size_t(nnz)); jcol.reserve(size_t(nnz));
        for (int k = 0; k < N; ++k) {
            const int v = variant_of(k);
            for (int slot : jperm[v]) { irow.push_back(resolve(jrid[size_t(slot)], k)); jcol.push_back(NXK * k + jc[size_t(slot)]); }
        }
        for (int slot : jperm_glob) { irow.push_back(resolve(jrid[size_t(slot)], 0)); jcol.push_back(NXK * N + (jc[size_t(slot)] - COL_GLOBAL)); }
        if (!vary_first)
            for (size_t i = 1; i < irow.size(); ++i)
                if (!(jcol[i - 1] < jcol[i] || (jcol[i - 1] == jcol[i] && irow[i - 1] < irow[i]))) { error = "internal: pattern not in CCS order"; return false; }
        if (int(irow.size()) != nnz) { error = "internal: nnz mismatch"; return false; }
        nconst_total = int(jperm_glob.size());
        for (int k = 0; k < N; ++k) nconst_total += nnz_v[variant_of(k)] - nvary_v[variant_of(k)];
        return true;
    }

    // is entry i of the pattern (0 .. nnz) constant?  (mask of the whole trajectory, in the order of irow / jcol)
    void constant_mask(unsigned char* mask) const {
        size_t at = 0;
        for (int k = 0; k < N; ++k) for (unsigned char c : jconst_pos[variant_of(k)]) mask[at++] = c;
        for (size_t i = 0; i < jperm_glob.size(); ++i) mask[at++] = 1;
    }
    // Values of the constant native slots under the parameters gp of ONE trajectory: the knot program run once on the host with an
    // emitter that keeps what leaves through emit_jc (values of slots that are not constant: 0).  They do not depend on the knot.
    static void constant_values(const hipnlp_settings& st, const KinTables& kt, const GParams& gp, double* cval /*[js::COUNT]*/) {
        std::vector<int> grow(gs::COUNT, -1), jrid(js::COUNT, -1), jc(js::COUNT, -1);
        std::fill(cval, cval + js::COUNT, 0.0);
        bool dup = false;
        KnotScratch* s = new KnotScratch();
        std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), 0.0);
        s->x[QB_ + 3] = 1.0;
        KSettings ks = make_ksettings(st);
        KnotInfo ki{0, st.horizon, 1, 1};
        RecordEm em{grow.data(), jrid.data(), jc.data(), &dup};
        em.jcval = cval;
        Ctx<RecordEm> cx(*s, kt, ks, gp, ki, em);
#define HOST_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
        HIPNLP_KNOT_PROGRAM(HOST_R, )
#undef HOST_R
        delete s;
    }

    // terrain description of the C-ABI -> pre-digested form the terrain jets read
    static void fill_terrain(KSettings& k, int terrain, int n_steps, const hipnlp_terrain_step* steps) {
        k.terrain = terrain;
        k.n_steps = terrain == HIPNLP_TERRAIN_SMOOTH_STEPS ? n_steps : 0;
        for (int i = 0; i < k.n_steps; ++i) {
            const hipnlp_terrain_step& t = steps[i];
            const double c = std::cos(t.orientation), sn = std::sin(t.orientation);
            TerrainStepK& o = k.steps[i];
            o.ox = t.position[0]; o.oy = t.position[1]; o.oz = t.position[2]; o.height = t.height;
            o.ax = 2.0 / t.length * c; o.ay = 2.0 / t.length * sn;       // a = (2/L) q_x,  q = Rz^T (p - o)
            o.bx = -2.0 / t.width * sn; o.by = 2.0 / t.width * c;        // b = (2/W) q_y
            o.m = 2 * t.edge_sharpness; o.r = 2 * t.side_sharpness;
            if (step_has_sloped_top(t)) o.m |= STEP_SLOPED;
        }
    }
    static bool step_has_sloped_top(const hipnlp_terrain_step& t) { return t.top_normal[0] != 0.0 || t.top_normal[1] != 0.0; }
    // the sloped tops of the steps, beside the tables the kernels read once per knot (nlp_defs.h TerrainTops): the top plane of step s is
    // pi(q) = height - (n_x q_x + n_y q_y) / n_z in the step's frame q = Rz(orientation)^T (p - position), n the normalised top_normal
    // (smooth_terrain.py:238-264); in world coordinates pi = height + px (p_x - o_x) + py (p_y - o_y)
    static void fill_terrain_tops(KinTables& kt, int terrain, int n_steps, const hipnlp_terrain_step* steps) {
        for (int i = 0; i < HIPNLP_MAX_TERRAIN_STEPS; ++i) kt.tops.px[i] = kt.tops.py[i] = 0.0;
        if (terrain != HIPNLP_TERRAIN_SMOOTH_STEPS) return;
        for (int i = 0; i < n_steps && i < HIPNLP_MAX_TERRAIN_STEPS; ++i) {
            const hipnlp_terrain_step& t = steps[i];
            if (!step_has_sloped_top(t)) continue;
            const double norm = std::sqrt(t.top_normal[0] * t.top_normal[0] + t.top_normal[1] * t.top_normal[1] + t.top_normal[2] * t.top_normal[2]);
            const double nx = t.top_normal[0] / norm, ny = t.top_normal[1] / norm, nz = t.top_normal[2] / norm;
            const double sx = -nx / nz, sy = -ny / nz;                       // d pi / d q_x, d pi / d q_y
            const double c = std::cos(t.orientation), sn = std::sin(t.orientation);
            kt.tops.px[i] = sx * c - sy * sn;                                // q_x = c dx + sn dy,  q_y = -sn dx + c dy
            kt.tops.py[i] = sx * sn + sy * c;
        }
    }
    // validation shared by hipnlp_create / hipnlp_pose_create; returns an error message or nullptr
    static const char* check_terrain(int terrain, int n_steps, const hipnlp_terrain_step* steps) {
        if (terrain != HIPNLP_TERRAIN_PLANAR && terrain != HIPNLP_TERRAIN_SMOOTH_STEPS) return "unknown terrain kind";
        if (terrain == HIPNLP_TERRAIN_SMOOTH_STEPS) {
            if (n_steps < 1 || n_steps > HIPNLP_MAX_TERRAIN_STEPS) return "n_terrain_steps must be in 1..HIPNLP_MAX_TERRAIN_STEPS";
            for (int i = 0; i < n_steps; ++i)
                if (!(steps[i].length > 0) || !(steps[i].width > 0) || steps[i].edge_sharpness < 2 || steps[i].side_sharpness < 2)
                    return "terrain step: length, width must be positive and the sharpness exponents >= 2";
            for (int i = 0; i < n_steps; ++i) {   // a zero vector = the flat top (top_normal_direction=None); else smooth_terrain.py:247-256
                const double* n = steps[i].top_normal;
                const double norm = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                if (norm == 0.0) continue;
                if (!(norm >= 1e-6)) return "terrain step: the top normal direction must be non-zero";
                if (!(std::fabs(n[2] / norm) >= 1e-6)) return "terrain step: the top normal direction must not be parallel to the xy-plane";
            }
        }
        return nullptr;
    }
    static KSettings make_ksettings(const hipnlp_settings& st) {
        KSettings k{};
        k.horizon = st.horizon;
        k.final_type = st.final_state_type;
        k.periodicity_type = st.periodicity_type;
        k.joint_reg_as_coded = st.joint_reg_as_coded;
        k.hdyn_x0 = st.periodicity_type == HIPNLP_EXPR_SKIP;
        fill_terrain(k, st.terrain, st.n_terrain_steps, st.terrain_steps);
        for (int f = 0; f < 2; ++f) for (int i = 0; i < 3; ++i) k.yaw_corner[f][i] = st.yaw_corner[f][i];
        k.final_weight = st.final_state_weight;
        k.periodicity_weight = st.periodicity_weight;
        k.m_centroid = st.contacts_centroid_cost_multiplier;
        for (int i = 0; i < 3; ++i) k.w_comvel[i] = st.com_linear_velocity_cost_weights[i];
        k.m_comvel = st.com_linear_velocity_cost_multiplier;
        k.m_frameq = st.desired_frame_quaternion_cost_multiplier;
        k.m_baseq = st.base_quaternion_cost_multiplier;
        k.m_baseqv = st.base_quaternion_velocity_cost_multiplier;
        for (int i = 0; i < NJ; ++i) k.w_jreg[i] = st.joint_regularization_cost_weights[i];
        k.m_jreg = st.joint_regularization_cost_multiplier;
        k.m_freg = st.force_regularization_cost_multiplier;
        k.m_yaw = st.foot_yaw_regularization_cost_multiplier;
        k.m_swing = st.swing_foot_height_cost_multiplier;
        k.m_ureg = st.contact_velocity_control_cost_multiplier;
        k.m_fdreg = st.contact_force_control_cost_multiplier;
        return k;
    }

    // Kinematic tables from the C descriptor; validates the topology the kernel is specialised for.
    static bool make_kin_tables(const hipnlp_robot_model& md, KinTables& kt, std::string& err) {
        kt = KinTables{};
        // Any tree rooted at link 0: the child link of joint j is link j + 1, parent[j] is ANY other link.  The joint order is the
        // reference's joints_name_list order (it fixes the x layout, variables.py:182-217) and need not be topological — ergoCub's
        // list names torso_pitch before torso_roll whatever the URDF chains first.  Everything below walks parent links, never
        // index order.
        for (int j = 0; j < NJ; ++j) {
            const int par = md.parent[j];
            if (par < 0 || par >= NL || par == j + 1) { err = "robot model: parent[j] must be a link in 0..23 other than the joint's own child link"; return false; }
            int steps = 0, l = j + 1;
            while (l > 0 && steps <= NL) { l = md.parent[l - 1]; ++steps; if (l < 0 || l >= NL) break; }
            if (l != 0) { err = "robot model: the parent links do not form a tree rooted at link 0"; return false; }
            double an = 0;
            for (int i = 0; i < 3; ++i) { kt.jf.j[j].axis[i] = md.axis[j][i]; an += md.axis[j][i] * md.axis[j][i]; }
            if (!(an > 0.999999 && an < 1.000001)) { err = "robot model: joint axes must be unit vectors"; return false; }
            for (int i = 0; i < 9; ++i) kt.jf.j[j].R_fix[i] = md.R_fix[j][i];
        }
        double total_mass = 0;
        for (int l = 0; l < NL; ++l) {
            kt.li.l[l].mass = md.mass[l];
            total_mass += md.mass[l];
            for (int i = 0; i < 3; ++i) kt.li.l[l].com[i] = md.com[l][i];
            for (int i = 0; i < 9; ++i) kt.li.l[l].inertia[i] = md.inertia[l][i];
        }
        if (!(total_mass > 0)) { err = "robot model: total mass must be positive"; return false; }
        kt.inv_total_mass = 1.0 / total_mass;
        for (int j = 0; j < NJ; ++j) { kt.leg_pos[0][j] = kt.leg_pos[1][j] = kt.chest_pos[j] = -1; }
        for (int f = 0; f < 3; ++f) {
            kt.frame_link[f] = md.frame_link[f];
            if (md.frame_link[f] < 0 || md.frame_link[f] >= NL) { err = "robot model: bad frame link"; return false; }
            for (int i = 0; i < 9; ++i) kt.frame_R[f][i] = md.frame_R[f][i];
            for (int i = 0; i < 3; ++i) kt.frame_o[f][i] = md.frame_o[f][i];
            std::vector<int> path;   // joints root -> frame link, in path order
            for (int l = md.frame_link[f]; l > 0; l = md.parent[l - 1]) path.push_back(l - 1);
            std::reverse(path.begin(), path.end());
            const int want = f == HIPNLP_FRAME_CHEST ? CHEST_PATH : LEG_PATH;
            if (int(path.size()) != want) { err = "robot model: the kernel is specialised for 6-joint leg chains and a 3-joint chest chain"; return false; }
            for (int q = 0; q < want; ++q) {
                if (f == HIPNLP_FRAME_CHEST) kt.chest_pos[path[size_t(q)]] = q;
                else { kt.leg_pos[f][path[size_t(q)]] = q; kt.leg_joint[f][q] = path[size_t(q)]; }
            }
        }
        for (int j = 0; j < NJ; ++j)
            if (kt.leg_pos[0][j] >= 0 && kt.leg_pos[1][j] >= 0) { err = "robot model: the two leg chains must be disjoint"; return false; }
        // ancestor lists (path root -> j) and descendant lists (subtree of link i)
        for (int j = 0; j < NJ; ++j) {
            kt.par_link[j] = md.parent[j];
            for (int i = 0; i < 3; ++i) kt.jf.j[j].o_fix[i] = md.o_fix[j][i];
            std::vector<int> path;
            for (int q = j + 1; q > 0; q = md.parent[q - 1]) path.push_back(q - 1);
            if (path.size() > 8) { err = "robot model: a chain is deeper than 8 joints"; return false; }
            std::reverse(path.begin(), path.end());   // root -> j, in path order
            // FRONT padded: the last element is always joint j itself, so the state before the last product is the parent's
            const int npad = 8 - int(path.size());
            for (int q = 0; q < 8; ++q) kt.anc[j][q] = int8_t(q < npad ? NJ : path[size_t(q - npad)]);
            int& ff = kt.fk_first[j < FK_SPLIT ? 0 : 1];
            if (j == 0 || j == FK_SPLIT || npad < ff) ff = npad;
        }
        for (int i = 0; i < NL; ++i) {
            int n = 0;
            for (int l = 0; l < NL; ++l) {
                bool in = false;
                for (int q = l; ; q = md.parent[q - 1]) { if (q == i) { in = true; break; } if (q == 0) break; }   // (a tree: checked above)
                if (in) kt.desc[i][n++] = int8_t(l);
            }
            kt.ndesc[i] = int16_t(n);
            for (; n < NL; ++n) kt.desc[i][n] = int8_t(NL);
        }
        {
            std::vector<int> order(NL);
            for (int i = 0; i < NL; ++i) order[size_t(i)] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kt.ndesc[a] > kt.ndesc[b]; });
            for (int i = 0; i < NL; ++i) kt.comp_order[i] = int16_t(order[size_t(i)]);
            for (int g = 0; g < NL / 4; ++g) kt.comp_cnt[g] = kt.ndesc[order[size_t(4 * g)]];   // sorted: the first of the group is the largest
        }
        for (int i = 0; i < 105; ++i) {
            int slot, desc;
            const int var = final_row_var(i, &slot, &desc);
            kt.en.fin_var[i] = int16_t(var); kt.en.fin_slot[i] = int16_t(slot); kt.en.fin_desc[i] = int16_t(desc);
        }
        for (int i = 0; i < 84; ++i) kt.en.per_var[i] = int16_t(periodicity_row_var(i));
        return true;
    }

    // Bounds of THIS layout's NLP from the parameter vector p (one trajectory); any pointer may be null.  Full layout: lbx / ubx
    // infinite (the reference adds no explicit variable bounds), lbg / ubg [m_full] the canonical row bounds.  Lifted layout: the
    // bounds of the lifted rows folded into lbx / ubx (the tightest of the rows that bound a variable), lbg / ubg [m] of the kept rows.
    void bounds(const double* p, double* lbx, double* ubx, double* lbg, double* ubg) const {
        const double inf = std::numeric_limits<double>::infinity();
        const size_t mf = static_cast<size_t>(m_full);
        if (lbx) for (int i = 0; i < n; ++i) lbx[i] = -inf;
        if (ubx) for (int i = 0; i < n; ++i) ubx[i] = inf;
        if (!lifted && !lbg && !ubg) return;
        std::vector<double> lo(mf, 0.0), hi(mf, 0.0);
        bounds_full(p, lo.data(), hi.data());
        if (!lifted) {
            if (lbg) std::copy(lo.begin(), lo.end(), lbg);
            if (ubg) std::copy(hi.begin(), hi.end(), ubg);
            return;
        }
        std::vector<int32_t> simple(mf, 0), var(mf, -1), kept(mf, -1);
        simple_rows(simple.data(), var.data());
        kept_rows(kept.data());
        for (size_t r = 0; r < mf; ++r) {
            if (kept[r] >= 0) {
                if (lbg) lbg[kept[r]] = lo[r];
                if (ubg) ubg[kept[r]] = hi[r];
            } else {
                if (lbx) lbx[var[r]] = std::max(lbx[var[r]], lo[r]);
                if (ubx) ubx[var[r]] = std::min(ubx[var[r]], hi[r]);
            }
        }
    }
    // row of THIS layout behind every row of the full list (-1: lifted); the identity for a full layout
    void kept_rows(int32_t* kept) const {
        for (const RowBlock& b : blocks)
            for (int kk = 0; kk < b.nk; ++kk)
                for (int i = 0; i < b.rows; ++i) kept[b.first_row + b.rows * kk + i] = row_of(b, kk, i);
    }
    // Canonical bounds of every row of the reference's subject_to list (CasADi Opti canon form), FULL numbering
    void bounds_full(const double* p, double* lbg, double* ubg) const {
        const double inf = std::numeric_limits<double>::infinity();
        const ParamOffsets po(N);
        auto fill = [&](int kind, int c, auto fn) {
            const int bi = blk[kind][c];
            if (bi < 0) return;
            const RowBlock& b = blocks[size_t(bi)];
            for (int kk = 0; kk < b.nk; ++kk)
                for (int i = 0; i < b.rows; ++i) { double lo, hi; fn(b.k0 + kk, i, lo, hi); lbg[b.first_row + b.rows * kk + i] = lo; ubg[b.first_row + b.rows * kk + i] = hi; }
        };
        auto eq0 = [](int, int, double& lo, double& hi) { lo = hi = 0.0; };
        auto ge0 = [inf](int, int, double& lo, double& hi) { lo = 0.0; hi = inf; };
        for (int c = 0; c < NC; ++c) {
            fill(RK_FDYN_X0, c, [&](int, int i, double& lo, double& hi) { lo = hi = p[po.init() + 9 * c + 3 + i]; });
            fill(RK_FDYN_IN, c, eq0);
            fill(RK_PDYN_X0, c, [&](int, int i, double& lo, double& hi) { lo = hi = p[po.init() + 9 * c + i]; });
            fill(RK_PDYN_IN, c, eq0);
            fill(RK_PLANAR, c, eq0);
            fill(RK_DCC, c, ge0); fill(RK_HEIGHT, c, ge0); fill(RK_NORMAL, c, ge0); fill(RK_FRICTION, c, ge0);
            fill(RK_UB, c, [&](int, int i, double& lo, double& hi) { hi = p[po.umax() + i]; lo = -hi; });
            fill(RK_FDB, c, [&](int, int i, double& lo, double& hi) { hi = p[po.fdmax() + i]; lo = -hi; });
            fill(RK_KINC, c, eq0);
        }
        fill(RK_PBDYN_X0, 0, [&](int, int i, double& lo, double& hi) { lo = hi = p[po.init() + 72 + i]; });
        fill(RK_QBDYN_X0, 0, [&](int, int i, double& lo, double& hi) { lo = hi = p[po.init() + 75 + i]; });
        fill(RK_SDYN_X0, 0, [&](int, int i, double& lo, double& hi) { lo = hi = p[po.init() + 79 + i]; });
        fill(RK_COMDYN_X0, 0, [&](int, int i, double& lo, double& hi) { lo = hi = p[po.init() + 102 + i]; });
        fill(RK_HDYN_X0, 0, eq0);
        fill(RK_PBDYN_IN, 0, eq0); fill(RK_QBDYN_IN, 0, eq0); fill(RK_SDYN_IN, 0, eq0); fill(RK_COMDYN_IN, 0, eq0); fill(RK_HDYN_IN, 0, eq0);
        fill(RK_UNITQ, 0, [](int, int, double& lo, double& hi) { lo = hi = 1.0; });
        fill(RK_COMC, 0, eq0); fill(RK_CMMC, 0, eq0);
        fill(RK_AMB, 0, [&](int, int, double& lo, double& hi) { hi = p[po.lmax()]; lo = -hi; });
        fill(RK_COMH, 0, [&](int, int, double& lo, double& hi) { lo = p[po.hmin()]; hi = inf; });
        fill(RK_FEETD, 0, [&](int, int, double& lo, double& hi) { lo = p[po.dmin()]; hi = inf; });
        fill(RK_JPB, 0, [&](int, int i, double& lo, double& hi) { lo = p[po.jpmin() + i]; hi = p[po.jpmax() + i]; });
        fill(RK_JVB, 0, [&](int, int i, double& lo, double& hi) { lo = p[po.jvmin() + i]; hi = p[po.jvmax() + i]; });
        fill(RK_FEETH, 0, [&](int, int, double& lo, double& hi) { hi = p[po.hmax()]; lo = -hi; });
        fill(RK_FIN, 0, [&](int, int i, double& lo, double& hi) { lo = hi = final_rhs(p, po, i); });
        fill(RK_PERN, 0, eq0);
    }
    // right-hand side of final-state row i: the final_state parameter matching the sorted to_list() order
    static double final_rhs(const double* p, const ParamOffsets& po, int i) {
        const int fo = po.fin();
        if (i < 3) return p[fo + 102 + i];
        int r = i - 3;
        if (r < 72) { const int c = r / 9, q = r % 9; return q < 3 ? p[fo + 9 * c + 6 + q] : (q < 6 ? p[fo + 9 * c + 3 + (q - 3)] : p[fo + 9 * c + (q - 6)]); }
        r -= 72;
        if (r < 3) return p[fo + 72 + r];
        r -= 3;
        if (r < 4) return p[fo + 75 + r];
        return p[fo + 79 + (r - 4)];
    }
    // rows that are exactly one decision variable (candidates for nlpsol detect_simple_bounds)
    void simple_rows(int32_t* is_simple, int32_t* var) const {   // (FULL numbering, whether or not this layout is lifted)
        for (int i = 0; i < m_full; ++i) { is_simple[i] = 0; var[i] = -1; }
        for (int kind = 0; kind < RK_COUNT; ++kind)
            for (int c = 0; c < NC; ++c) {
                const int bi = blk[kind][c];
                if (bi < 0) continue;
                const RowBlock& b = blocks[size_t(bi)];
                for (int kk = 0; kk < b.nk; ++kk)
                    for (int i = 0; i < b.rows; ++i) {
                        const int v = simple_var(kind, c, i);
                        if (v < 0) continue;
                        is_simple[b.first_row + b.rows * kk + i] = 1;
                        var[b.first_row + b.rows * kk + i] = NXK * (b.k0 + kk) + v;
                    }
            }
    }
};

// pack the device-side parameter records of one trajectory from the reference-order vector p
inline void pack_params(const double* p, int N, double* pk /*[N][PK_STRIDE]*/, GParams& gp) {
    const ParamOffsets po(N);
    for (int k = 0; k < N; ++k) {
        double* r = pk + size_t(k) * PK_STRIDE;
        for (int i = 0; i < 24; ++i) r[PK_DESC + i] = p[po.desc(k, 0) + i];
        for (int i = 0; i < 55; ++i) r[PK_REF + i] = p[po.ref(k) + i];
        for (int foot = 0; foot < 2; ++foot) {  // E9 needs sin/cos of yaw and of yaw + pi/2 (planner.py:831-841): parameters only
            const double yaw = r[PK_REF + (foot == 0 ? R_YAW_L : R_YAW_R)];
            r[PK_YAWSC + 4 * foot + 0] = std::sin(yaw); r[PK_YAWSC + 4 * foot + 1] = std::cos(yaw);
            r[PK_YAWSC + 4 * foot + 2] = std::sin(yaw + M_PI / 2); r[PK_YAWSC + 4 * foot + 3] = std::cos(yaw + M_PI / 2);
        }
        r[87] = 0.0;
    }
    gp.dt = p[po.dt()]; gp.kt = p[po.kt()]; gp.kbs = p[po.kbs()]; gp.eps = p[po.eps()]; gp.mu = p[po.mu()]; gp.mass = p[po.mass()];
    for (int i = 0; i < 6; ++i) gp.gravity[i] = p[po.gravity() + i];
    for (int i = 0; i < 105; ++i) gp.final_rhs[i] = Layout::final_rhs(p, po, i);
}

}  // namespace hipnlp
