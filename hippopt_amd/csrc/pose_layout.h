// pose_layout.h — host-side structure of the static pose finder NLP: row directory in the reference's subject_to order
// (turnkey_planners/humanoid_pose_finder/planner.py:360-399), native slot -> row / CCS position tables, bounds, parameters.
#pragma once
#include "layout.h"
#include "pose_hess_body.h"

namespace hipnlp {

struct PoseRowBlock { std::string name; int first_row, rows; };

// parameter offsets in the reference's creation order (tests/golden/pose_*.npz "pnames")
namespace pp {
constexpr int DESC = 0, MASS = 24, GRAV = 27, REF = 33 /* per point: p 9c, f 9c+3, descriptor 9c+6 */, REF_PB = 105, REF_QB = 108,
              REF_S = 112, REF_COM = 135, REF_FQ = 138, REF_LH = 142 /* right: + 3 */, EPS = 148, MU = 149, SMAX = 150, SMIN = 173, LH_IN = 196 /* right: + 3 */;
}

struct PoseLayout {
    int n = POSE_NX, m = 0, nnz = 0;
    std::vector<PoseRowBlock> blocks;
    int blk[RK_COUNT][NC];
    std::vector<int32_t> g_row;      // [gs::COUNT] row of the native slot or -1
    std::vector<int32_t> jperm;      // CCS position -> native jac slot
    std::vector<int32_t> irow, jcol;
    // Hessian of the Lagrangian (pose_hess_body.h): lower triangle, CCS order
    int hnnz = 0;
    std::vector<int32_t> hperm;      // CCS position -> native Hessian slot
    std::vector<int32_t> hrow, hcol;
    std::string error;

    static std::string point_name(int c) { return std::string("state.contact_points.") + (c < 4 ? "left[" : "right[") + std::to_string(c % 4) + "]"; }
    static int pose_col(int knot_col) {   // inverse of pose_to_knot_col, -1 for columns the pose has no variable for
        for (int i = 0; i < POSE_NX; ++i) if (pose_to_knot_col(i) == knot_col) return i;
        return -1;
    }
    int add(int kind, int c, const std::string& name, int rows) {
        blk[kind][c] = int(blocks.size());
        blocks.push_back({name, m, rows});
        m += rows;
        return blk[kind][c];
    }
    int resolve(int rid) const {
        const int bi = blk[rid_kind(rid)][rid_point(rid)];
        return bi < 0 ? -1 : blocks[size_t(bi)].first_row + rid_index(rid);
    }

    static KSettings make_ksettings(const hipnlp_pose_settings& st) {
        KSettings k{};
        k.horizon = 3;
        Layout::fill_terrain(k, st.terrain, st.n_terrain_steps, st.terrain_steps);
        k.m_frameq = st.desired_frame_quaternion_cost_multiplier;
        k.m_baseq = st.base_quaternion_cost_multiplier;
        for (int i = 0; i < NJ; ++i) k.w_jreg[i] = st.joint_regularization_cost_weights[i];
        k.m_jreg = st.joint_regularization_cost_multiplier;
        k.m_freg = st.force_regularization_cost_multiplier;
        k.pose_com_type = st.com_position_type;
        k.pose_left_type = st.left_point_position_type;
        k.pose_right_type = st.right_point_position_type;
        k.m_pcom = st.com_regularization_cost_multiplier;
        k.m_favg = st.average_force_regularization_cost_multiplier;
        k.m_preg = st.point_position_regularization_cost_multiplier;
        return k;
    }

    static PoseHands make_hands(const hipnlp_pose_settings& st) {
        PoseHands hd{};
        for (int h = 0; h < 2; ++h) {
            hd.type[h] = st.hand_type[h];
            hd.link[h] = st.hand_type[h] == HIPNLP_EXPR_SKIP ? 1 : st.hand_frame_link[h];
            for (int i = 0; i < 9; ++i) hd.R[h][i] = st.hand_frame_R[h][i];
            for (int i = 0; i < 3; ++i) hd.o[h][i] = st.hand_frame_o[h][i];
            hd.mult[h] = st.hand_regularization_cost_multiplier[h];
        }
        return hd;
    }

    bool build(const hipnlp_pose_settings& st, const KinTables& kt) {
        for (int h = 0; h < 2; ++h) {
            if (st.hand_type[h] != HIPNLP_EXPR_SKIP && st.hand_type[h] != HIPNLP_EXPR_SUBJECT_TO && st.hand_type[h] != HIPNLP_EXPR_MINIMIZE) { error = "hand_type: not an expression type"; return false; }
            if (st.hand_type[h] != HIPNLP_EXPR_SKIP && (st.hand_frame_link[h] < 1 || st.hand_frame_link[h] >= NL)) { error = "hand_frame_link: a link below a joint (1 .. links - 1)"; return false; }
        }
        m = 0;
        blocks.clear();
        for (int a = 0; a < RK_COUNT; ++a) for (int c = 0; c < NC; ++c) blk[a][c] = -1;
        for (int c = 0; c < NC; ++c) {   // planner.py:360-375
            const std::string pn = point_name(c);
            add(RK_PCOMPL, c, pn + ".p_complementarity", 1);
            add(RK_HEIGHT, c, pn + ".p_height", 1);
            add(RK_NORMAL, c, pn + ".f_normal", 1);
            add(RK_FRICTION, c, pn + ".f_friction", 1);
            add(RK_KINC, c, pn + ".p_kinematics_consistency", 3);
        }
        add(RK_UNITQ, 0, "unitary_quaternion", 1);               // :455-461
        add(RK_COMC, 0, "com_kinematics_consistency", 3);        // :463-485
        add(RK_PBAL, 0, "centroidal_momentum_dynamics", 6);      // :487-509
        add(RK_JPB, 0, "joint_position_bounds", NJ);             // :511-519
        if (st.com_position_type == HIPNLP_EXPR_SUBJECT_TO) add(RK_PCOMERR, 0, "com_position_error", 1);   // :566-573
        for (int h = 0; h < 2; ++h)                              // :596-660 (still inside _add_kinematics_regularization: BEFORE the feet)
            if (st.hand_type[h] == HIPNLP_EXPR_SUBJECT_TO) add(RK_PHAND, h, h == 0 ? "left_hand_position_error" : "right_hand_position_error", 3);
        for (int c = 0; c < NC; ++c) {                           // :385-396, :752-759
            const int mode = c < 4 ? st.left_point_position_type : st.right_point_position_type;
            if (mode == HIPNLP_EXPR_SUBJECT_TO) add(RK_PPREG, c, point_name(c) + ".p_regularization", 1);
        }
        const PoseHands hands = make_hands(st);
        // ---- record the native slots of the pose program ------------------------------------------------------
        std::vector<int> grow(gs::COUNT, -1), jrid(js::COUNT, -1), jc(js::COUNT, -1), hr(hs::COUNT, -1), hc(hs::COUNT, -1);
        bool dup = false;
        {
            KnotScratch* s = new KnotScratch();
            std::fill(reinterpret_cast<double*>(s), reinterpret_cast<double*>(s) + sizeof(KnotScratch) / sizeof(double), 0.0);
            s->x[QB_ + 3] = 1.0;
            KSettings ks = make_ksettings(st);
            GParams gp{};
            gp.mass = 1.0;
            KnotInfo ki{1, 3, 0, 0};
            RecordEm em{grow.data(), jrid.data(), jc.data(), &dup, hr.data(), hc.data()};
            Ctx<RecordEm> cx(*s, kt, ks, gp, ki, em);
            cx.hands = &hands;
#define HOST_R(w4, w8, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
            HIPNLP_POSE_PROGRAM(HOST_R, )
#undef HOST_R
            HessScratch* hx = new HessScratch();
            std::fill(reinterpret_cast<double*>(hx), reinterpret_cast<double*>(hx) + sizeof(HessScratch) / sizeof(double), 1.0);
            HCtx<RecordEm> hcx{cx, *hx};
#define HOST_KIN(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(cx, t_);
#define HOST_RH(w, fn, nt) for (int t_ = 0; t_ < (nt); ++t_) fn(hcx, t_);
            HIPNLP_POSE_HESS_PROGRAM(HOST_KIN, HOST_RH, )
#undef HOST_KIN
#undef HOST_RH
            delete hx;
            delete s;
        }
        if (dup) { error = "internal: native slot emitted twice"; return false; }
        g_row.assign(gs::COUNT, -1);
        for (int slot = 0; slot < gs::COUNT; ++slot) if (grow[size_t(slot)] >= 0) g_row[size_t(slot)] = resolve(grow[size_t(slot)]);
        std::vector<std::pair<std::pair<int, int>, int>> ent;   // ((col,row),slot)
        for (int slot = 0; slot < js::COUNT; ++slot) {
            if (jrid[size_t(slot)] < 0) continue;
            const int r = resolve(jrid[size_t(slot)]);
            if (r < 0) continue;
            const int col = pose_col(jc[size_t(slot)]);
            if (col < 0) { error = "internal: a pose row depends on a column the pose has no variable for"; return false; }
            ent.push_back({{col, r}, slot});
        }
        std::sort(ent.begin(), ent.end());
        for (size_t i = 1; i < ent.size(); ++i) if (ent[i].first == ent[i - 1].first) { error = "internal: duplicate jacobian entry"; return false; }
        jperm.clear(); irow.clear(); jcol.clear();
        for (auto& e : ent) { jperm.push_back(e.second); jcol.push_back(e.first.first); irow.push_back(e.first.second); }
        nnz = int(ent.size());
        // Hessian pattern: lower triangle, column major, every entry produced by exactly one slot
        std::vector<std::pair<std::pair<int, int>, int>> hent;
        for (int slot = 0; slot < hs::COUNT; ++slot) {
            if (hr[size_t(slot)] < 0) continue;
            if (hr[size_t(slot)] < hc[size_t(slot)] || hr[size_t(slot)] >= POSE_NX) { error = "internal: Hessian entry outside the lower triangle"; return false; }
            hent.push_back({{hc[size_t(slot)], hr[size_t(slot)]}, slot});
        }
        std::sort(hent.begin(), hent.end());
        for (size_t i = 1; i < hent.size(); ++i) if (hent[i].first == hent[i - 1].first) { error = "internal: duplicate Hessian entry"; return false; }
        hperm.clear(); hrow.clear(); hcol.clear();
        for (auto& e : hent) { hperm.push_back(e.second); hcol.push_back(e.first.first); hrow.push_back(e.first.second); }
        hnnz = int(hent.size());
        // every row of the directory must be produced by exactly one slot
        std::vector<int> seen(size_t(m), 0);
        for (int slot = 0; slot < gs::COUNT; ++slot) if (g_row[size_t(slot)] >= 0) seen[size_t(g_row[size_t(slot)])]++;
        for (int r = 0; r < m; ++r) if (seen[size_t(r)] != 1) { error = "internal: row " + std::to_string(r) + " not produced exactly once"; return false; }
        return true;
    }

    // canonical bounds (CasADi Opti canon form) of one pose from its parameter vector
    void bounds(const double* p, double* lbg, double* ubg) const {
        const double inf = std::numeric_limits<double>::infinity();
        auto fill = [&](int kind, int c, auto fn) {
            const int bi = blk[kind][c];
            if (bi < 0) return;
            const PoseRowBlock& b = blocks[size_t(bi)];
            for (int i = 0; i < b.rows; ++i) { double lo, hi; fn(i, lo, hi); lbg[b.first_row + i] = lo; ubg[b.first_row + i] = hi; }
        };
        auto eq0 = [](int, double& lo, double& hi) { lo = hi = 0.0; };
        auto ge0 = [inf](int, double& lo, double& hi) { lo = 0.0; hi = inf; };
        for (int c = 0; c < NC; ++c) {
            fill(RK_PCOMPL, c, ge0); fill(RK_HEIGHT, c, ge0); fill(RK_NORMAL, c, ge0); fill(RK_FRICTION, c, ge0);
            fill(RK_KINC, c, eq0); fill(RK_PPREG, c, eq0);
        }
        fill(RK_UNITQ, 0, [](int, double& lo, double& hi) { lo = hi = 1.0; });
        fill(RK_COMC, 0, eq0); fill(RK_PBAL, 0, eq0); fill(RK_PCOMERR, 0, eq0);
        for (int h = 0; h < 2; ++h) fill(RK_PHAND, h, [&](int i, double& lo, double& hi) { lo = hi = p[pp::REF_LH + 3 * h + i]; });
        fill(RK_JPB, 0, [&](int i, double& lo, double& hi) { lo = p[pp::SMIN + i]; hi = p[pp::SMAX + i]; });
    }
};

// device-side parameter records of one pose from the reference-order vector p [202]:
//   pk [PK_STRIDE] knot-style record (descriptors, chest / base quaternion and joint references),
//   xr [64] point / com references (loaded into KnotScratch::xm), gp (mass, eps, mu, gravity)
inline void pack_pose_params(const double* p, double* pk, double* xr, GParams& gp) {
    for (int i = 0; i < PK_STRIDE; ++i) pk[i] = 0.0;
    for (int i = 0; i < 64; ++i) xr[i] = 0.0;
    for (int i = 0; i < 24; ++i) pk[PK_DESC + i] = p[pp::DESC + i];
    for (int i = 0; i < 4; ++i) { pk[PK_REF + R_FQ + i] = p[pp::REF_FQ + i]; pk[PK_REF + R_BQ + i] = p[pp::REF_QB + i]; }
    for (int j = 0; j < NJ; ++j) pk[PK_REF + R_JREG + j] = p[pp::REF_S + j];
    for (int c = 0; c < NC; ++c)
        for (int i = 0; i < 3; ++i) { xr[XR_P + 3 * c + i] = p[pp::REF + 9 * c + i]; xr[XR_F + 3 * c + i] = p[pp::REF + 9 * c + 3 + i]; }
    for (int i = 0; i < 3; ++i) xr[XR_COM + i] = p[pp::REF_COM + i];
    for (int i = 0; i < 6; ++i) { xr[XR_HREF + i] = p[pp::REF_LH + i]; xr[XR_HIN + i] = p[pp::LH_IN + i]; }
    gp = GParams{};
    gp.mass = p[pp::MASS]; gp.eps = p[pp::EPS]; gp.mu = p[pp::MU];
    for (int i = 0; i < 6; ++i) gp.gravity[i] = p[pp::GRAV + i];
}

}  // namespace hipnlp
