// pose_hess_body.h — exact Hessian of the Lagrangian  sigma f(x) + lambda^T g(x)  of the static pose finder NLP (IPOPT's eval_h;
// SURVEY §8f rank 1).  The reference pose finder runs IPOPT with its default exact-Hessian option
// (turnkey_planners/humanoid_pose_finder/main.py:101 `casadi_solver_options = {}`; planner.py:334-339), CasADi derives
// nlp_hess_l by AD of the graph.  Here every second derivative is written out by hand and evaluated by tasks that run behind the
// KINEMATIC tasks of the pose program in the same workgroup (they read the kinematic quantities those left in LDS).
//
// Lower triangle in pose variable order (x [81]: per point p, f | p_b | q_b | s | com).  Second derivatives that exist:
//   point block (p_c, f_c)   relaxed complementarity / height / normal force / friction rows on the terrain (second-order jets of
//                            the terrain frame), the bilinear (p - com) x f of the static balance, the quadratic regularisations
//   (com, f_c), com          static balance, com regularisation
//   (q_b, q_b)               <M, R(q/|q|)> of the kinematic rows and of the chest-frame cost through the normalisation; unit norm row
//   (s_j, q_b), (s_j, s_i)   a base rotation acts like an outermost ancestor joint:  with  Y_j = d/d theta [ dL/ds_j ]
//                            d2L/ds_i ds_j = a_i . Y_j  (i ancestor-or-self of j),   d2L/dq_l ds_j = G_l . Y_j / |q|
// where for a point P rigidly attached below joint j:  d2P/ds_i ds_j = a_i x (a_j x (P - o_j))  (Jacobi identity).
#pragma once
#include "pose_body.h"

namespace hipnlp {

namespace pv {  // pose variable indices (reference creation order, include/hipnlp.h)
HD constexpr int P(int c, int i) { return 6 * c + i; }
HD constexpr int F(int c, int i) { return 6 * c + 3 + i; }
constexpr int PB = 48, QB = 51, S = 55, COM = 78;
}  // namespace pv

namespace hs {  // native slots of the Hessian values
constexpr int PT_PP = 0;    // [6]  lower 3x3 of (p, p):  r (r + 1) / 2 + c
constexpr int PT_FP = 6;    // [3][3]  row f_j, column p_i
constexpr int PT_FF = 15;   // [6]  lower 3x3 of (f, f)
constexpr int PT_STRIDE = 21;
constexpr int FAVG = NC * PT_STRIDE;   // [foot 2][pair 6][component 3]   (f_c', f_c), c' > c on one foot
constexpr int COMF = FAVG + 36;        // [c][6]   row com_r, column f_q  (r != q)
constexpr int COMD = COMF + 6 * NC;    // [3]      com diagonal
constexpr int QQ = COMD + 3;           // [10]     lower 4x4 of (q_b, q_b)
constexpr int QS = QQ + 10;            // [NJ][4]  row s_j, column q_l
constexpr int SS = QS + 4 * NJ;        // [NJ (NJ + 1) / 2]  row s_j, column s_i (i <= j): j (j + 1) / 2 + i; ancestor pairs only
// hand position expressions in minimize mode (2 sigma m J^T J couples p_b with itself, q_b and the joints of the hand's path)
constexpr int PBD = SS + NJ * (NJ + 1) / 2;   // [3]      (p_b, p_b) diagonal
constexpr int QPB = PBD + 3;           // [4][3]   row q_l, column p_b,i
constexpr int SPB = QPB + 12;          // [NJ][3]  row s_j, column p_b,i
constexpr int COUNT = SPB + 3 * NJ;
}  // namespace hs

struct HessScratch {
    double lam[gs::COUNT];   // multiplier of the row a native g slot belongs to (0 for slots without a row)
    double sigma, pad_;
    double Y[NJ][3];         // Y_j, the share of the com / hand / chest terms (t_hess_Y_a)
    double Y2[NJ][3];        //      the share of the contact points of the leg the joint belongs to (t_hess_Y_b); zero for the other joints
    double qq[32];           // (q_b, q_b) in three steps: [0..8] Mw, [9..17] M = Mw R_b; [18..20] ax(E), [21] tr E - 3, [22..30] E, [31] tr E (t_hess_chest)
    double H[hs::COUNT];
};

template <class Em> struct HCtx {
    Ctx<Em>& cx;
    HessScratch& hx;
};

// ---- second-order Taylor coefficients in two variables (p_x, p_y) ------------------------------------------------------
struct T2 {
    double v, x, y, xx, xy, yy;
    HD T2() : v(0.0), x(0.0), y(0.0), xx(0.0), xy(0.0), yy(0.0) {}
    HD T2(double c) : v(c), x(0.0), y(0.0), xx(0.0), xy(0.0), yy(0.0) {}
    HD T2(double v_, double x_, double y_, double xx_, double xy_, double yy_) : v(v_), x(x_), y(y_), xx(xx_), xy(xy_), yy(yy_) {}
};
HD T2 operator+(const T2& a, const T2& b) { return T2(a.v + b.v, a.x + b.x, a.y + b.y, a.xx + b.xx, a.xy + b.xy, a.yy + b.yy); }
HD T2 operator-(const T2& a, const T2& b) { return T2(a.v - b.v, a.x - b.x, a.y - b.y, a.xx - b.xx, a.xy - b.xy, a.yy - b.yy); }
HD T2 operator-(const T2& a) { return T2(-a.v, -a.x, -a.y, -a.xx, -a.xy, -a.yy); }
HD T2 operator*(const T2& a, double c) { return T2(a.v * c, a.x * c, a.y * c, a.xx * c, a.xy * c, a.yy * c); }
HD T2 operator*(double c, const T2& a) { return a * c; }
HD T2 operator*(const T2& a, const T2& b) {
    return T2(a.v * b.v, a.x * b.v + a.v * b.x, a.y * b.v + a.v * b.y,
              a.xx * b.v + 2.0 * (a.x * b.x) + a.v * b.xx,
              a.xy * b.v + a.x * b.y + a.y * b.x + a.v * b.xy,
              a.yy * b.v + 2.0 * (a.y * b.y) + a.v * b.yy);
}
HD T2 t2inv(const T2& b) {
    const double i = 1.0 / b.v, i2 = i * i, i3 = 2.0 * i2 * i;
    return T2(i, -i2 * b.x, -i2 * b.y, i3 * b.x * b.x - i2 * b.xx, i3 * b.x * b.y - i2 * b.xy, i3 * b.y * b.y - i2 * b.yy);
}
HD T2 t2sqrt(const T2& a) {
    const double r = sqrt(a.v), h = 0.5 / r, q = 0.25 / (r * a.v);
    return T2(r, a.x * h, a.y * h, a.xx * h - q * a.x * a.x, a.xy * h - q * a.x * a.y, a.yy * h - q * a.y * a.y);
}

// terrain frame (terrain_descriptor.py:45-80) with second-order dependence on (p_x, p_y); Z = third-order jet of the bump sum
struct TerrainFrame2 { T2 h, n[3], xv[3], yv[3]; };
HD void terrain_frame2(const double* Z, double pz, TerrainFrame2& t) {
    const T2 u1(-Z[1], -Z[3], -Z[4], -Z[6], -Z[7], -Z[8]);   // grad h = (u1, u2, 1)
    const T2 u2(-Z[2], -Z[4], -Z[5], -Z[7], -Z[8], -Z[9]);
    t.h = T2(pz - Z[0], -Z[1], -Z[2], -Z[3], -Z[4], -Z[5]);
    const T2 inn = t2inv(t2sqrt(T2(1.0) + u1 * u1 + u2 * u2));
    t.n[0] = u1 * inn; t.n[1] = u2 * inn; t.n[2] = inn;
    const T2 q = t.n[1] * t.n[1] + t.n[2] * t.n[2];          // same closed form as terrain_frame (knot_body.h)
    const T2 iq = t2inv(t2sqrt(q));
    t.xv[0] = q * iq; t.xv[1] = -(t.n[1] * t.n[0]) * iq; t.xv[2] = -(t.n[2] * t.n[0]) * iq;
    t.yv[0] = T2(0.0); t.yv[1] = t.n[2] * iq; t.yv[2] = -(t.n[1] * iq);
}

HD constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }   // r >= c

// --- point block (p_c, f_c): lane c (8) -------------------------------------------------------------------------------------
//   rows: complementarity  eps - h(p) (n(p).f) mass   (planner.py:680-689),  height h(p),  normal force n.f,  friction
//         -(x.f)^2 - (y.f)^2 + mu^2 (n.f)^2   (:691-722);  static balance  lam_ang . ((p - com) x f)   (:487-509)
//   costs: point position / force / average force regularisations  (:724-768)
template <class Em> HD void t_hess_point(HCtx<Em>& h, int c) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const double* lam = h.hx.lam;
    const double sigma = h.hx.sigma;
    const int gb = gs::PT_STRIDE * c, hb = hs::PT_STRIDE * c, cb = PT_ * c;
    const double l_c = lam[gb + gs::DCC], l_h = lam[gb + gs::HEIGHT], l_n = lam[gb + gs::NORMAL], l_f = lam[gb + gs::FRICTION];
    const double* la = lam + gs::HDYN + 3;   // angular rows of the static balance
    const double mass = cx.gp.mass, mu2 = cx.gp.mu * cx.gp.mu;
    const int mode = (c >> 2) == 0 ? cx.st.pose_left_type : cx.st.pose_right_type;
    const double dpp = mode == HIPNLP_EXPR_MINIMIZE ? 2.0 * sigma * cx.st.m_preg : (mode == HIPNLP_EXPR_SUBJECT_TO ? 2.0 * lam[gb + gs::UB] : 0.0);
    const double dff = sigma * (2.0 * cx.st.m_freg + 1.5 * cx.st.m_favg);   // d2/df_c^2 of m sum_c |f_c - mean|^2 = m (2 - 1/2)
    Em& em = cx.em;
    if (terrain_is_planar(cx)) {
        if (mode != HIPNLP_EXPR_SKIP) for (int i = 0; i < 3; ++i) em.H(hb + hs::PT_PP + tri(i, i), pv::P(c, i), pv::P(c, i), dpp);
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                if (i == j && i < 2) continue;   // structurally zero on the planar terrain
                em.H(hb + hs::PT_FP + 3 * j + i, pv::F(c, j), pv::P(c, i), (i == j ? -(l_c * mass) : 0.0) + skew_rc(la, j, i));
            }
        em.H(hb + hs::PT_FF + tri(0, 0), pv::F(c, 0), pv::F(c, 0), dff - 2.0 * l_f);
        em.H(hb + hs::PT_FF + tri(1, 1), pv::F(c, 1), pv::F(c, 1), dff - 2.0 * l_f);
        em.H(hb + hs::PT_FF + tri(2, 2), pv::F(c, 2), pv::F(c, 2), dff + 2.0 * mu2 * l_f);
        return;
    }
    const double* p = s.x + cb + P_;
    const double* f = s.x + cb + F_;
    double Z[10];
    terrain_Z_jet(cx.st, p[0], p[1], 3, Z, &cx.gkt->tops);
    TerrainFrame2 tf;
    terrain_frame2(Z, p[2], tf);
    const T2 nf = tf.n[0] * f[0] + tf.n[1] * f[1] + tf.n[2] * f[2];
    const T2 fcx = tf.xv[0] * f[0] + tf.xv[1] * f[1] + tf.xv[2] * f[2], fcy = tf.yv[0] * f[0] + tf.yv[1] * f[1] + tf.yv[2] * f[2];
    const T2 fric = (nf * nf) * mu2 - fcx * fcx - fcy * fcy;
    const T2 Lp = (tf.h * nf) * (-(l_c * mass)) + tf.h * l_h + nf * l_n + fric * l_f;
    em.H(hb + hs::PT_PP + tri(0, 0), pv::P(c, 0), pv::P(c, 0), Lp.xx + dpp);
    em.H(hb + hs::PT_PP + tri(1, 0), pv::P(c, 1), pv::P(c, 0), Lp.xy);
    em.H(hb + hs::PT_PP + tri(1, 1), pv::P(c, 1), pv::P(c, 1), Lp.yy + dpp);
    em.H(hb + hs::PT_PP + tri(2, 0), pv::P(c, 2), pv::P(c, 0), -(l_c * mass) * nf.x);   // d/dp_z acts on h only (dh/dp_z = 1)
    em.H(hb + hs::PT_PP + tri(2, 1), pv::P(c, 2), pv::P(c, 1), -(l_c * mass) * nf.y);
    if (mode != HIPNLP_EXPR_SKIP) em.H(hb + hs::PT_PP + tri(2, 2), pv::P(c, 2), pv::P(c, 2), dpp);
    for (int j = 0; j < 3; ++j) {
        // dL/df_j as a function of p
        const T2 q = (tf.h * tf.n[j]) * (-(l_c * mass)) + tf.n[j] * l_n + ((nf * tf.n[j]) * mu2 - fcx * tf.xv[j] - fcy * tf.yv[j]) * (2.0 * l_f);
        em.H(hb + hs::PT_FP + 3 * j + 0, pv::F(c, j), pv::P(c, 0), q.x + skew_rc(la, j, 0));
        em.H(hb + hs::PT_FP + 3 * j + 1, pv::F(c, j), pv::P(c, 1), q.y + skew_rc(la, j, 1));
        em.H(hb + hs::PT_FP + 3 * j + 2, pv::F(c, j), pv::P(c, 2), -(l_c * mass) * tf.n[j].v + skew_rc(la, j, 2));
        for (int i = 0; i <= j; ++i)
            em.H(hb + hs::PT_FF + tri(j, i), pv::F(c, j), pv::F(c, i),
                 2.0 * l_f * (mu2 * tf.n[j].v * tf.n[i].v - tf.xv[j].v * tf.xv[i].v - tf.yv[j].v * tf.yv[i].v) + (i == j ? dff : 0.0));
    }
}

// --- average force regularisation across the points of one foot: lanes (foot, pair, i) 36; static balance (com, f): lanes 36..83;
//     com diagonal: lanes 84..86 ------------------------------------------------------------------------------------------------
constexpr int HESS_MISC_TASKS = 36 + 6 * NC + 3;
template <class Em> HD void t_hess_misc(HCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    Em& em = cx.em;
    const double sigma = h.hx.sigma;
    if (t < 36) {
        const int foot = t / 18, r = t - 18 * foot, pair = r / 3, i = r - 3 * pair;
        // pairs (c' > c) of {0,1,2,3}: (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
        const int hi = pair < 1 ? 1 : (pair < 3 ? 2 : 3), lo = pair - (hi == 1 ? 0 : (hi == 2 ? 1 : 3));
        em.H(hs::FAVG + t, pv::F(4 * foot + hi, i), pv::F(4 * foot + lo, i), -0.5 * sigma * cx.st.m_favg);
    } else if (t < 36 + 6 * NC) {
        const int q = t - 36, c = q / 6, e = q - 6 * c, r = cross_row(e), col = cross_col(e);
        // lam_ang . ((p - com) x f) = f . (lam_ang x (p - com)):  d2 / d f_col d com_r = -[lam_ang]x (col, r)
        em.H(hs::COMF + q, pv::COM + r, pv::F(c, col), -skew_rc(h.hx.lam + gs::HDYN + 3, col, r));
    } else {
        const int i = t - 36 - 6 * NC;
        const int mode = cx.st.pose_com_type;
        if (mode == HIPNLP_EXPR_SKIP) return;
        em.H(hs::COMD + i, pv::COM + i, pv::COM + i, mode == HIPNLP_EXPR_MINIMIZE ? 2.0 * sigma * cx.st.m_pcom : 2.0 * h.hx.lam[gs::COMH]);
    }
}

// chest-frame rotation error  E = R_chest R(q_d)^T  (K5), once per pose (round 6: t_hess_qq_mw's nine lanes and the torso lanes of t_hess_Y_a
// each recomputed it from the frame — a quaternion, a transpose and a product at the head of the two longest chains of the fifth phase):
// 1 lane, third phase, beside t_frames (R_chest from the link's world rotation, the product t_frames forms for fr_R: it waits for nobody).
template <class Em> HD void t_hess_chest(HCtx<Em>& h, int) {
    const auto& s = h.cx.s;
    const int f = HIPNLP_FRAME_CHEST;
    double Rc[9], Rd[9], Rdt[9], E[9];
    matmul3(s.Rw[h.cx.kt.frame_link[f]], h.cx.kt.frame_R[f], Rc);
    rot_from_quat(s.pk + PK_REF + R_FQ, Rd);
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) Rdt[3 * r + q] = Rd[3 * q + r];
    matmul3(Rc, Rdt, E);
    double* o = h.hx.qq;
    for (int i = 0; i < 9; ++i) o[22 + i] = E[i];
    const double trE = E[0] + E[4] + E[8];
    o[31] = trE; o[21] = trE - 3.0;
    o[18] = E[7] - E[5]; o[19] = E[2] - E[6]; o[20] = E[3] - E[1];
}
// (the same from the frame t_frames left in the scratch: the kinodynamic Hessian's tasks, knot_hess_body.h)
template <class S> HD void chest_error(const S& s, double* E) {
    double Rd[9], Rdt[9];
    rot_from_quat(s.pk + PK_REF + R_FQ, Rd);
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) Rdt[3 * r + q] = Rd[3 * q + r];
    matmul3(s.fr_R[HIPNLP_FRAME_CHEST], Rdt, E);
}

// ---- hand position expressions (pose_body.h: t_pose_hand_pts has left r_h and e_h = P_h - ref in the scratch) ---------------------
//   subject_to: lambda_h . P_h;   minimize: sigma m |P_h - ref|^2 = second-derivative part with the weight 2 sigma m e_h  +  2 sigma m J^T J,
//   J = [ I | D_q | D_s ],  D_q(:, l) = -(r_h x G_l) / |q|,  D_s(:, j) = a_j x (r_h - o_j) for the joints j of the hand's path.
template <class Em> HD int hand_mode(const HCtx<Em>& h, int hnd) { return h.cx.hands ? h.cx.hands->type[hnd] : HIPNLP_EXPR_SKIP; }
template <class Em> HD bool hand_on_path(const HCtx<Em>& h, int hnd, int j) {
    const int jl = h.cx.hands->link[hnd] - 1;
    bool on = false;
    for (int q = 0; q < 8; ++q) on = on || (int(h.cx.kt.anc[jl][q]) == j);
    return on;
}
// weight of r_h in the Lagrangian: the multipliers of its rows / 2 sigma m e_h
template <class Em> HD void hand_weight(const HCtx<Em>& h, int hnd, double* w) {
    const int mode = hand_mode(h, hnd);
    const double* hb = pose_hand_buf(h.cx.s);
    for (int i = 0; i < 3; ++i)
        w[i] = mode == HIPNLP_EXPR_SUBJECT_TO ? h.hx.lam[gs::PT_STRIDE * hnd + gs::FDYN + i]
                                              : (mode == HIPNLP_EXPR_MINIMIZE ? 2.0 * h.hx.sigma * h.cx.hands->mult[hnd] * hb[6 + 3 * hnd + i] : 0.0);
}
template <class Em> HD double hand_jtj_scale(const HCtx<Em>& h, int hnd) {
    return hand_mode(h, hnd) == HIPNLP_EXPR_MINIMIZE ? 2.0 * h.hx.sigma * h.cx.hands->mult[hnd] : 0.0;
}
template <class Em> HD void hand_dq(const HCtx<Em>& h, int hnd, int l, double* d) {   // D_q(:, l)
    const auto& s = h.cx.s;
    const double* r = pose_hand_buf(s) + 3 * hnd;
    const double Gl[3] = {s.G[l], s.G[4 + l], s.G[8 + l]};
    double c[3];
    cross3(r, Gl, c);
    for (int i = 0; i < 3; ++i) d[i] = -c[i] * s.inv_qnorm;
}
template <class Em> HD void hand_ds(const HCtx<Em>& h, int hnd, int j, double* x) {   // D_s(:, j), j on the path
    const auto& s = h.cx.s;
    const double* r = pose_hand_buf(s) + 3 * hnd;
    double d[3];
    for (int n = 0; n < 3; ++n) d[n] = r[n] - s.ow[j + 1][n];
    cross3(s.aw[j], d, x);
}
// (p_b, p_b), (q_b, p_b), (s_j, p_b) of 2 sigma m J^T J: lanes 0..2 | 3..14 (l, i) | 15.. (j, i).  Entries exist only for hands in minimize mode.
constexpr int HESS_HAND_TASKS = 3 + 12 + 3 * NJ;
template <class Em> HD void t_hess_hand(HCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const double k0 = hand_jtj_scale(h, 0), k1 = hand_jtj_scale(h, 1);
    if (k0 == 0.0 && k1 == 0.0 && hand_mode(h, 0) != HIPNLP_EXPR_MINIMIZE && hand_mode(h, 1) != HIPNLP_EXPR_MINIMIZE) return;
    const bool m0 = hand_mode(h, 0) == HIPNLP_EXPR_MINIMIZE, m1 = hand_mode(h, 1) == HIPNLP_EXPR_MINIMIZE;
    if (t < 3) {
        cx.em.H(hs::PBD + t, pv::PB + t, pv::PB + t, k0 + k1);
    } else if (t < 15) {
        const int l = (t - 3) / 3, i = (t - 3) - 3 * l;
        double v = 0.0, d[3];
        if (m0) { hand_dq(h, 0, l, d); v += k0 * d[i]; }
        if (m1) { hand_dq(h, 1, l, d); v += k1 * d[i]; }
        cx.em.H(hs::QPB + (t - 3), pv::QB + l, pv::PB + i, v);
    } else {
        const int j = (t - 15) / 3, i = (t - 15) - 3 * j;
        const bool p0 = m0 && hand_on_path(h, 0, j), p1 = m1 && hand_on_path(h, 1, j);
        if (!p0 && !p1) return;
        double v = 0.0, x[3];
        if (p0) { hand_ds(h, 0, j, x); v += k0 * x[i]; }
        if (p1) { hand_ds(h, 1, j, x); v += k1 * x[i]; }
        cx.em.H(hs::SPB + (t - 15), pv::S + j, pv::PB + i, v);
    }
}

// --- Y_j = d/d theta [ dL/ds_j ]  (theta: world-frame rotation of the base), lane j (23) -------------------------------------------
//   L = sum_c w_c . pkin_c + w_com . com_kin + sigma m (tr E - 3)^2,   w_c = -lambda(kinematics consistency), w_com = -lambda(com consistency)
//   In two groups on two waves (one group until round 6: 2.9 k cycles of the 3.6 k cycle phase it shares with (q_b, q_b)); the consumers add the two.
template <class Em> HD void t_hess_Y_a(HCtx<Em>& h, int j) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const double* lam = h.hx.lam;
    const double* a = s.aw[j];
    const double* o = s.ow[j + 1];
    double Y[3] = {0.0, 0.0, 0.0}, t1[3], t2[3];
    {   // com: d com / d s_j = a_j x (h_sub - m_sub o_j) / M
        const double* cp = s.comp[j + 1];
        const double inv_M = cx.kt.inv_total_mass;
        for (int r = 0; r < 3; ++r) t1[r] = (cp[CH + r] - cp[CM] * o[r]) * inv_M;
        cross3(a, t1, t2);
        const double w[3] = {-lam[gs::COMC], -lam[gs::COMC + 1], -lam[gs::COMC + 2]};
        cross3(t2, w, t1);
        for (int r = 0; r < 3; ++r) Y[r] += t1[r];
    }
    for (int hnd = 0; hnd < 2; ++hnd) {   // hand points: weight w_h on r_h
        if (hand_mode(h, hnd) == HIPNLP_EXPR_SKIP || !hand_on_path(h, hnd, j)) continue;
        double w[3];
        hand_weight(h, hnd, w);
        const double* rh = pose_hand_buf(s) + 3 * hnd;
        for (int r = 0; r < 3; ++r) t1[r] = rh[r] - o[r];
        cross3(a, t1, t2);
        cross3(t2, w, t1);
        for (int r = 0; r < 3; ++r) Y[r] += t1[r];
    }
    if (cx.kt.chest_pos[j] >= 0) {
        const double* E = h.hx.qq + 22;   // (t_hess_chest)
        const double* ax = h.hx.qq + 18;
        const double trE = h.hx.qq[31], e = h.hx.qq[21];
        double Ea[3];
        matvec3(E, a, Ea);
        const double de = -dot3(ax, a);                     // d e / d s_j
        const double m2 = 2.0 * h.hx.sigma * cx.st.m_frameq;
        for (int r = 0; r < 3; ++r) Y[r] += m2 * (e * (Ea[r] - trE * a[r]) + de * (-ax[r]));
    }
    for (int r = 0; r < 3; ++r) h.hx.Y[j][r] = Y[r];
}
template <class Em> HD void t_hess_Y_b(HCtx<Em>& h, int j) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const double* lam = h.hx.lam;
    const double* a = s.aw[j];
    const double* o = s.ow[j + 1];
    double Y[3] = {0.0, 0.0, 0.0}, t1[3], t2[3];
    for (int foot = 0; foot < 2; ++foot) {
        if (cx.kt.leg_pos[foot][j] < 0) continue;
        for (int c = 4 * foot; c < 4 * foot + 4; ++c) {
            for (int r = 0; r < 3; ++r) t1[r] = s.pkin[c][r] - o[r];
            cross3(a, t1, t2);
            const double* lk = lam + gs::PT_STRIDE * c + gs::KINC;
            const double w[3] = {-lk[0], -lk[1], -lk[2]};
            cross3(t2, w, t1);
            for (int r = 0; r < 3; ++r) Y[r] += t1[r];
        }
    }
    for (int r = 0; r < 3; ++r) h.hx.Y2[j][r] = Y[r];
}
// Y_j as its consumers see it
template <class Em> HD void hess_Y(const HCtx<Em>& h, int j, double* Y) { for (int r = 0; r < 3; ++r) Y[r] = h.hx.Y[j][r] + h.hx.Y2[j][r]; }

// --- (s_j, s_i), related pairs only: lane (d, q) walks the ancestor list of joint d — k = anc[d][q] lies on the path root -> d (inclusive)
//     and carries the axis (the joint numbering need not be topological).  (Until round 4 a lane per entry of the lower triangle searched
//     its row with a loop, compared both ancestor lists and went idle for the unrelated pairs: five wave iterations for ~110 entries.)
constexpr int HESS_SS_TASKS = NJ * 8;
template <class Em> HD void t_hess_ss(HCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const int d = t >> 3, k = int(cx.kt.anc[d][t & 7]);
    if (k >= NJ) return;   // (front padding of the list)
    double Yd[3];
    hess_Y(h, d, Yd);
    double v = dot3(cx.s.aw[k], Yd);
    if (k == d) v += 2.0 * h.hx.sigma * cx.st.m_jreg * cx.st.w_jreg[d];
    for (int hnd = 0; hnd < 2; ++hnd) {   // 2 sigma m D_s^T D_s of a hand in minimize mode (both joints on its path)
        const double sc = hand_jtj_scale(h, hnd);
        if (sc == 0.0 || !hand_on_path(h, hnd, k) || !hand_on_path(h, hnd, d)) continue;
        double xk[3], xd[3];
        hand_ds(h, hnd, k, xk);
        hand_ds(h, hnd, d, xd);
        v += sc * dot3(xk, xd);
    }
    const int hi = k > d ? k : d, lo = k > d ? d : k;
    cx.em.H(hs::SS + hi * (hi + 1) / 2 + lo, pv::S + hi, pv::S + lo, v);
}

// --- (s_j, q_l): lanes (j, l) ---------------------------------------------------------------------------------------------------
template <class Em> HD void t_hess_qs(HCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const int j = t >> 2, l = t & 3;
    double Y[3];
    hess_Y(h, j, Y);
    double v = (s.G[l] * Y[0] + s.G[4 + l] * Y[1] + s.G[8 + l] * Y[2]) * s.inv_qnorm;
    for (int hnd = 0; hnd < 2; ++hnd) {   // 2 sigma m D_s^T D_q
        const double k = hand_jtj_scale(h, hnd);
        if (k == 0.0 || !hand_on_path(h, hnd, j)) continue;
        double x[3], d[3];
        hand_ds(h, hnd, j, x);
        hand_dq(h, hnd, l, d);
        v += k * dot3(x, d);
    }
    cx.em.H(hs::QS + t, pv::S + j, pv::QB + l, v);
}

// --- (q_b, q_b): lanes over the lower triangle (10); every lane forms the small matrices itself ------------------------------------
//   F(q) = <M, R(q / |q|)>,  R(qh) = I + 2 w [v]x + 2 [v]x^2:  Phi(qh) = tr M + 2 w v.ax(M) + 2 v^T M v - 2 (v.v) tr M
//   Hess_q F = J B J + ( -(g qh^T + qh g^T + (g.qh) I) + 3 (g.qh) qh qh^T ) / |q|^2,   J = (I - qh qh^T) / |q|,  g, B = gradient, Hessian of Phi
//   Three groups (one group of ten lanes until round 6, every lane forming Mw, M, B and g for itself: 3.5 k cycles, the longest chain of the
//   Hessian program): Mw entry by entry (nine lanes), M = Mw R_b entry by entry (nine lanes, behind it on its wave), the ten entries from M
//   (next phase, on the wave the joint blocks leave idle).
template <class Em> HD void t_hess_qq_mw(HCtx<Em>& h, int t) {   // lane (a, b): Mw = sum_c w_c pkin_c^T + w_com com^T + sigma 2 m e E^T  (world frame)
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const double* lam = h.hx.lam;
    const int a = t / 3, b = t - 3 * a;
    const double* E = h.hx.qq + 22;   // (t_hess_chest)
    const double e = h.hx.qq[21];
    const double m2 = 2.0 * h.hx.sigma * cx.st.m_frameq;
    double acc = -lam[gs::COMC + a] * (s.comp[0][CH + b] * cx.kt.inv_total_mass) + m2 * e * E[3 * b + a];   // com = first moment of the whole tree / mass
    for (int p = 0; p < NC; ++p) acc += -lam[gs::PT_STRIDE * p + gs::KINC + a] * s.pkin[p][b];
    for (int hnd = 0; hnd < 2; ++hnd) {   // hand points: w_h r_h^T
        if (hand_mode(h, hnd) == HIPNLP_EXPR_SKIP) continue;
        double w[3];
        hand_weight(h, hnd, w);
        acc += w[a] * (pose_hand_buf(s) + 3 * hnd)[b];
    }
    h.hx.qq[t] = acc;
}
template <class Em> HD void t_hess_qq_m(HCtx<Em>& h, int t) {    // lane (a, b): M = Mw R_b  (behind t_hess_qq_mw on its wave)
    HIPNLP_WAVE_SYNC();
    const int a = t / 3, b = t - 3 * a;
    const double* Mw = h.hx.qq;
    const double* Rb = h.cx.s.Rb;
    h.hx.qq[9 + t] = Mw[3 * a] * Rb[b] + Mw[3 * a + 1] * Rb[3 + b] + Mw[3 * a + 2] * Rb[6 + b];
}
template <class Em> HD void t_hess_qq(HCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const double* lam = h.hx.lam;
    const double sigma = h.hx.sigma;
    const int r = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3));   // row of entry t of the packed lower 4 x 4 triangle
    const int c = t - r * (r + 1) / 2;
    const double* M = h.hx.qq + 9;
    const double* axE = h.hx.qq + 18;
    const double m2 = 2.0 * sigma * cx.st.m_frameq;
    const double trM = M[0] + M[4] + M[8];
    const double al[3] = {M[7] - M[5], M[2] - M[6], M[3] - M[1]};
    const double* qh = s.qn;
    const double inv_n = s.inv_qnorm;
    double B[16], g[4];
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) B[4 * a + b] = 2.0 * (M[3 * a + b] + M[3 * b + a]) - (a == b ? 4.0 * trM : 0.0);
        B[4 * a + 3] = B[12 + a] = 2.0 * al[a];
    }
    B[15] = 0.0;
    for (int a = 0; a < 3; ++a) g[a] = 2.0 * qh[3] * al[a] + B[4 * a] * qh[0] + B[4 * a + 1] * qh[1] + B[4 * a + 2] * qh[2];
    g[3] = 2.0 * (qh[0] * al[0] + qh[1] * al[1] + qh[2] * al[2]);
    const double gq = g[0] * qh[0] + g[1] * qh[1] + g[2] * qh[2] + g[3] * qh[3];
    // (J B J)(r, c) with J = (I - qh qh^T)/|q|:  rows of J
    double Jr[4], Jc[4], BJc[4];
    for (int a = 0; a < 4; ++a) { Jr[a] = ((a == r ? 1.0 : 0.0) - qh[r] * qh[a]) * inv_n; Jc[a] = ((a == c ? 1.0 : 0.0) - qh[c] * qh[a]) * inv_n; }
    for (int a = 0; a < 4; ++a) BJc[a] = B[4 * a] * Jc[0] + B[4 * a + 1] * Jc[1] + B[4 * a + 2] * Jc[2] + B[4 * a + 3] * Jc[3];
    double v = Jr[0] * BJc[0] + Jr[1] * BJc[1] + Jr[2] * BJc[2] + Jr[3] * BJc[3];
    v += (-(g[r] * qh[c] + qh[r] * g[c] + (r == c ? gq : 0.0)) + 3.0 * gq * qh[r] * qh[c]) * (inv_n * inv_n);
    // chest cost, outer product part: d e / d q_l = -(ax(E) . G_l) / |q|
    const double ger = -(axE[0] * s.G[r] + axE[1] * s.G[4 + r] + axE[2] * s.G[8 + r]) * inv_n;
    const double gec = -(axE[0] * s.G[c] + axE[1] * s.G[4 + c] + axE[2] * s.G[8 + c]) * inv_n;
    v += m2 * ger * gec;
    for (int hnd = 0; hnd < 2; ++hnd) {   // 2 sigma m D_q^T D_q
        const double k = hand_jtj_scale(h, hnd);
        if (k == 0.0) continue;
        double dr[3], dc[3];
        hand_dq(h, hnd, r, dr);
        hand_dq(h, hnd, c, dc);
        v += k * dot3(dr, dc);
    }
    if (r == c) {
        const double* qd = s.pk + PK_REF + R_BQ;   // base quaternion error is linear in q: Hessian 2 m |q_d|^2 I  (E13)
        v += 2.0 * lam[gs::UNITQ] + 2.0 * sigma * cx.st.m_baseq * (qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
    }
    cx.em.H(hs::QQ + t, pv::QB + r, pv::QB + c, v);
}

// The Hessian program of the pose finder.  KIN(w, fn, n) runs a task of knot_body.h, RH(w, fn, n) a Hessian task, on wave w of four.
// Only the KINEMATIC part of the pose program runs (joint transforms, forward kinematics, link quantities, composites, contact-point
// kinematics: none of the rows / Jacobian columns); the Hessian tasks that need no kinematics sit on the waves the forward kinematics
// of the second phase leaves idle (round 6: the first phase is the joint transforms and the base alone), the hand block beside Y / (q, q).
// the related joint-pair lanes in two ranges on two waves
constexpr int HESS_SS_SPLIT = 2 * 64;
template <class Em> HD void t_hess_ss_a(HCtx<Em>& h, int t) { t_hess_ss(h, t); }
template <class Em> HD void t_hess_ss_b(HCtx<Em>& h, int t) { t_hess_ss(h, t + HESS_SS_SPLIT); }
#define HIPNLP_POSE_HESS_PROGRAM(KIN, RH, BARRIER)                                                                \
    KIN(0, t_joints, NJ) KIN(1, t_base, 3) KIN(1, t_kin_padding, 16)                                              \
    BARRIER                                                                                                       \
    KIN(0, t_fk_rot_a, FK_TASKS_A) KIN(0, t_link_u_a, FK_SPLIT) RH(1, t_hess_misc, HESS_MISC_TASKS) RH(2, t_hess_point, NC) \
    KIN(3, t_fk_rot_b, FK_TASKS_B) KIN(3, t_link_u_b, NJ - FK_SPLIT)                                              \
    BARRIER                                                                                                       \
    KIN(0, t_links, NL) KIN(1, t_frames, 3) KIN(2, t_link_inertia, NL) KIN(3, t_pose_hand_pts, 2) RH(3, t_hess_chest, 1) \
    BARRIER                                                                                                       \
    KIN(0, t_composite_g0, 64) KIN(1, t_composite_g1, 64) KIN(1, t_composite_g2, 64)                              \
    KIN(2, t_composite_g3, 64) KIN(2, t_composite_g4, 64) KIN(3, t_composite_g5, 64) KIN(3, t_pkin, NC)           \
    BARRIER                                                                                                       \
    RH(0, t_hess_hand, HESS_HAND_TASKS) RH(1, t_hess_Y_a, NJ) RH(2, t_hess_Y_b, NJ) RH(3, t_hess_qq_mw, 9) RH(3, t_hess_qq_m, 9) \
    BARRIER                                                                                                       \
    RH(0, t_hess_ss_a, HESS_SS_SPLIT) RH(3, t_hess_ss_b, HESS_SS_TASKS - HESS_SS_SPLIT) RH(1, t_hess_qs, 4 * NJ) RH(2, t_hess_qq, 10) \
    BARRIER

}  // namespace hipnlp
