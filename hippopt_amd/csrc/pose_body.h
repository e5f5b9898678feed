// pose_body.h — the static pose finder NLP (BASELINE config 2) on the knot machinery: one workgroup of four wavefronts
// evaluates ONE pose.  The pose's 81 variables are loaded into a knot record (velocities zero), so the kinematic tasks of
// knot_body.h (base orientation, forward kinematics as ancestor sums, composites as descendant sums, CoM / contact-point
// consistency rows with their analytic Jacobians, chest-frame and base-quaternion costs) run unchanged; the tasks below add
// what only the pose finder has.
//
// Reference (relative to /root/reference/src/hippopt/turnkey_planners/humanoid_pose_finder/planner.py):
//   :670-722 contact point feasibility (relaxed complementarity E5, height, normal force, friction cone)
//   :487-509 static balance  0 = g + sum_i [ f_i ; (p_i - com) x f_i ]   (E1, unit mass)
//   :511-519 joint position bounds (Opti_bounded -> rows of g: the pose finder runs Opti without detect_simple_bounds)
//   :521-629 regularisations (base quaternion, chest frame, com, joints)   :724-768 foot regularisations
#pragma once
#include "knot_body.h"

namespace hipnlp {

constexpr int POSE_NX = HIPNLP_POSE_NX, POSE_NP = HIPNLP_POSE_NP, POSE_NCT = HIPNLP_POSE_NCOST_TERMS;
// pose references live in the (otherwise unused) previous-knot record of the scratch: s.xm[XR_*]
enum : int { XR_P = 0, XR_F = 24, XR_COM = 48, XR_HREF = 51 /* references.left / right_hand_position */, XR_HIN = 57 /* left / right_hand_position_in_frame */, XR_COUNT = 63 };
// pose variable i (reference creation order) -> column of the knot record
HD constexpr int pose_to_knot_col(int i) {
    return i < 48 ? PT_ * (i / 6) + ((i % 6) < 3 ? P_ + (i % 6) : F_ + (i % 6) - 3)
                  : (i < 51 ? PB_ + (i - 48) : (i < 55 ? QB_ + (i - 51) : (i < 78 ? S_ + (i - 55) : COM_ + (i - 78))));
}

// The Jacobian staging of the pose finder's device kernels (round 6): of the constant region D of the native slots (nlp_defs.h, [0, js::V0))
// the pose program touches the per-point part [0, js::CG0) and, of the global part, [js::HDYN_LIN_F_OUT, js::JVB) alone — balance force
// entries, com consistency constants, the com height entry, the joint bound rows.  The staging keeps those two pieces and the varying
// regions: 428 + 809 slots on the planar terrain instead of 1 730 (3.9 KB of LDS: with the static scratch, five workgroups per CU).
// hipnlp_pose_create checks that every slot of the recorded pattern lies in a kept piece.
namespace pjs {
constexpr int D0 = js::CG0, G0 = js::HDYN_LIN_F_OUT, G1 = js::JVB, DSLOTS = D0 + (G1 - G0);
static_assert(G0 <= js::COMC_COM && js::COMC_PB + 3 <= G1 && G0 <= js::COMH_Z && js::COMH_Z < G1 && G0 <= js::JPB && js::JPB + NJ <= G1 && G1 <= js::V0, "the kept piece of the global constants");
HD constexpr bool kept(int slot) { return slot < D0 || (slot >= G0 && slot < G1) || slot >= js::V0; }
HD constexpr int index(int slot) { return slot < D0 ? slot : (slot < js::V0 ? slot - (G0 - D0) : slot - (js::V0 - DSLOTS)); }   // staging index of a kept native slot
HD constexpr int slots(bool planar) { return DSLOTS + js::vary_slots(planar); }
}  // namespace pjs

// --- contact point c: relaxed complementarity, height, normal force, friction rows; point regularisations.  lane c (8) ------
template <class Em> HD void t_pose_points(Ctx<Em>& cx, int c) {
    auto& s = cx.s;
    Em& em = cx.em;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    const double* p = s.x + cb + P_;
    const double* f = s.x + cb + F_;
    const double mass = cx.gp.mass;
    if (terrain_is_planar(cx)) {   // eps - h(p) (n . f mass)  with  h = p_z, n = e_z
        em.G(gb + gs::DCC, row_id(RK_PCOMPL, c, 0), cx.gp.eps - p[2] * (f[2] * mass));
        em.J(jb + js::PL_DCC_P, row_id(RK_PCOMPL, c, 0), cb + P_ + 2, -(f[2] * mass));
        em.J(jb + js::PL_DCC_F, row_id(RK_PCOMPL, c, 0), cb + F_ + 2, -(p[2] * mass));
        point_hnf_planar(cx, c);
    } else {
        double Z[10];
        terrain_Z_jet(cx.st, p[0], p[1], 2, Z, &cx.gkt->tops);
        TerrainFrame tf;
        terrain_frame(Z, p[2], tf);
        const D2 nf = tf.n[0] * f[0] + tf.n[1] * f[1] + tf.n[2] * f[2];
        const D2 margin = D2(cx.gp.eps) - tf.h * nf * mass;
        em.G(gb + gs::DCC, row_id(RK_PCOMPL, c, 0), margin.v);
        em.J(jb + js::DCC_P + 0, row_id(RK_PCOMPL, c, 0), cb + P_ + 0, margin.x);
        em.J(jb + js::DCC_P + 1, row_id(RK_PCOMPL, c, 0), cb + P_ + 1, margin.y);
        em.J(jb + js::DCC_P + 2, row_id(RK_PCOMPL, c, 0), cb + P_ + 2, -(nf.v * mass));   // only h depends on p_z
        for (int j = 0; j < 3; ++j) em.J(jb + js::DCC_F + j, row_id(RK_PCOMPL, c, 0), cb + F_ + j, -(tf.h.v * tf.n[j].v * mass));
        point_hnf_smooth(cx, c, tf);
    }
    // regularisations of the foot the point belongs to  (planner.py:724-768)
    const int foot = c >> 2;
    const int mode = foot == 0 ? cx.st.pose_left_type : cx.st.pose_right_type;
    double mean[3] = {0.0, 0.0, 0.0};
    for (int q = 4 * foot; q < 4 * foot + 4; ++q) for (int i = 0; i < 3; ++i) mean[i] += s.x[PT_ * q + F_ + i];
    double cp = 0.0, cf = 0.0, ca = 0.0, ep[3];
    for (int i = 0; i < 3; ++i) {
        ep[i] = p[i] - s.xm[XR_P + 3 * c + i];
        const double ef = f[i] - s.xm[XR_F + 3 * c + i];
        const double ea = f[i] - 0.25 * mean[i];
        cp += ep[i] * ep[i]; cf += ef * ef; ca += ea * ea;
        s.grad[cb + P_ + i] = mode == HIPNLP_EXPR_MINIMIZE ? 2.0 * cx.st.m_preg * ep[i] : 0.0;
        s.grad[cb + F_ + i] = 2.0 * cx.st.m_freg * ef + 2.0 * cx.st.m_favg * ea;   // sum_c (f_c - mean) = 0: no cross terms
    }
    s.c_pt[c][0] = mode == HIPNLP_EXPR_MINIMIZE ? cx.st.m_preg * cp : 0.0;
    s.c_pt[c][1] = cx.st.m_freg * cf;
    s.c_pt[c][2] = cx.st.m_favg * ca;
    if (mode == HIPNLP_EXPR_SUBJECT_TO) {   // sumsqr(p - p_ref) == 0   (base/problem.py:146-151)
        em.G(gb + gs::UB, row_id(RK_PPREG, c, 0), cp);
        for (int i = 0; i < 3; ++i) emit_jd(em, jc + js::UB + i, row_id(RK_PPREG, c, 0), cb + P_ + i, 2.0 * ep[i]);
    }
}

// where the point group runs: on the planar terrain (3.2 k cycles) beside the forward kinematics of the second phase, so that the first
// phase is the joint transforms alone; on the smooth steps (7.4 k: the jets of the bumps) from the first instruction on
template <class Em> HD void t_pose_points_first(Ctx<Em>& cx, int c) { if (!terrain_is_planar(cx)) t_pose_points(cx, c); }
template <class Em> HD void t_pose_points_second(Ctx<Em>& cx, int c) { if (terrain_is_planar(cx)) t_pose_points(cx, c); }

// --- static balance (planner.py:487-509) as three task groups (one group until round 6: its three branches ran one after the other on
//     one wave and set the length of the first phase — 6.2 k cycles of a 21.7 k cycle pose at batch, tools/diag/pose_stamps.py).  None of
//     them needs the robot model: they sit where the kinematic chain leaves a wave idle.
//     entries: lanes (c, e) 48, the Jacobian entries in p_c and f_c -----------------------------------------------------------------
constexpr int POSE_BALANCE_ENTRIES = 48, POSE_BALANCE_COM = 6;
template <class Em> HD void t_pose_balance_entries(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double* com = s.x + COM_;
    const int c = t / 6, e = t - 6 * c, r = cross_row(e), q = cross_col(e), cb = PT_ * c;
    const double* f = s.x + cb + F_;
    const double a[3] = {s.x[cb + P_] - com[0], s.x[cb + P_ + 1] - com[1], s.x[cb + P_ + 2] - com[2]};
    em.J(js::HDYN_ANG_P_OUT + 6 * c + e, row_id(RK_PBAL, 0, 3 + r), cb + P_ + q, -skew_entry(f, e));   // d (a x f)/d a = -[f]x
    em.J(js::HDYN_ANG_F_OUT + 6 * c + e, row_id(RK_PBAL, 0, 3 + r), cb + F_ + q, skew_entry(a, e));    // d (a x f)/d f =  [a]x
    if (e < 3) emit_jd(em, js::HDYN_LIN_F_OUT + 3 * c + e, row_id(RK_PBAL, 0, e), cb + F_ + e, 1.0);
}
//     rows: the three force rows and the three moment rows as two groups of three lanes (as one group of six the two branches ran one after
//     the other inside the loop over the points, every iteration waiting for its own LDS reads: 3.9 k cycles at batch) --------------------
template <class Em> HD void t_pose_balance_rows_lin(Ctx<Em>& cx, int i) {
    auto& s = cx.s;
    double acc = cx.gp.gravity[i];
    for (int c = 0; c < NC; ++c) acc += s.x[PT_ * c + F_ + i];
    cx.em.G(gs::HDYN + i, row_id(RK_PBAL, 0, i), acc);
}
template <class Em> HD void t_pose_balance_rows_ang(Ctx<Em>& cx, int i) {
    auto& s = cx.s;
    const double* com = s.x + COM_;
    double acc = cx.gp.gravity[3 + i];
    for (int c = 0; c < NC; ++c) {
        const int cb = PT_ * c;
        const double a[3] = {s.x[cb + P_] - com[0], s.x[cb + P_ + 1] - com[1], s.x[cb + P_ + 2] - com[2]};
        acc += cross_comp(a, s.x + cb + F_, i);
    }
    cx.em.G(gs::HDYN + 3 + i, row_id(RK_PBAL, 0, 3 + i), acc);
}
//     com entries: lane e (6) -------------------------------------------------------------------------------------------------------
template <class Em> HD void t_pose_balance_com(Ctx<Em>& cx, int e) {
    auto& s = cx.s;
    Em& em = cx.em;
    const int r = cross_row(e), q = cross_col(e);
    double fs[3] = {0.0, 0.0, 0.0};
    for (int c = 0; c < NC; ++c) for (int i = 0; i < 3; ++i) fs[i] += s.x[PT_ * c + F_ + i];
    em.J(js::HDYN_ANG_COM_OUT + e, row_id(RK_PBAL, 0, 3 + r), COM_ + q, skew_entry(fs, e));   // d/d com = +[sum f]x
}
// identity / zero padding slots of the ancestor lists (first phase: the forward kinematics of the second reads them), lanes e < 16
template <class Em> HD void t_kin_padding(Ctx<Em>& cx, int e) { scratch_padding(cx.s, e); }

// --- joints: bound rows, e^T diag(w) e regularisation (planner.py:575-589), local joint transform.  lane j (23) --------------
template <class Em> HD void t_pose_joints(Ctx<Em>& cx, int j) {
    auto& s = cx.s;
    Em& em = cx.em;
    em.G(gs::JPB + j, row_id(RK_JPB, 0, j), s.x[S_ + j]);
    emit_jd(em, js::JPB + j, row_id(RK_JPB, 0, j), S_ + j, 1.0);
    const double e = s.x[S_ + j] - s.pk[PK_REF + R_JREG + j];
    const double mw = cx.st.m_jreg * cx.st.w_jreg[j];
    s.c_joint[j] = mw * e * e;
    s.grad[S_ + j] = 2.0 * mw * e;
    joint_transform(cx, j);
}

// --- com position error (planner.py:566-573): lanes 0..2 components, lane 3 value.  4 tasks -----------------------------------
template <class Em> HD void t_pose_com(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const int mode = cx.st.pose_com_type;
    if (t < 3) {
        const double e = s.x[COM_ + t] - s.xm[XR_COM + t];
        s.grad[COM_ + t] = mode == HIPNLP_EXPR_MINIMIZE ? 2.0 * cx.st.m_pcom * e : 0.0;
        s.grad[PB_ + t] = 0.0;
        if (mode == HIPNLP_EXPR_SUBJECT_TO) { if (t < 2) em.J(js::COMH_XY + t, row_id(RK_PCOMERR, 0, 0), COM_ + t, 2.0 * e); else emit_jd(em, js::COMH_Z, row_id(RK_PCOMERR, 0, 0), COM_ + t, 2.0 * e); }
    } else {
        double c = 0.0;
        for (int i = 0; i < 3; ++i) { const double e = s.x[COM_ + i] - s.xm[XR_COM + i]; c += e * e; }
        s.cost[CT_COMVEL] = mode == HIPNLP_EXPR_MINIMIZE ? cx.st.m_pcom * c : 0.0;
        if (mode == HIPNLP_EXPR_SUBJECT_TO) em.G(gs::COMH, row_id(RK_PCOMERR, 0, 0), c);
    }
}

// --- hand position expressions (planner.py:596-660):  P_h = p_b + r_h,  r_h = o_L + R_L (o_frame + R_frame p_in)  (base-centred,
//     world-oriented link pose of the forward kinematics)  against references.<side>_hand_position.
//     subject_to: three rows ref <= P_h <= ref (Opti's canonical form of P_h == parameter);  minimize: mult * |P_h - ref|^2.   d r_h / d s_j = a_j x (r_h - o_j) for the joints j
//     on the path root -> hand link (kt.anc of the link's joint: at most eight), d r_h / d q_b = -[r_h]x G / |q|, d P_h / d p_b = I.
// Scratch: the per-point momentum partials hd[][][] of the knot program are unused by the pose program.
//   hb[0..5] r_h, hb[6..11] e_h = P_h - ref, hb[12..13] cost.   Native slots: the (unused) trapezoid-defect slots of the knot program —
//   g: gs::FDYN + i of point h;  jac: js::FDYN + {0: p_b,i | 1..4: q_b | 5..12: path joint q} of point 4 h + i.
template <class S> HD auto pose_hand_buf(S& s) -> decltype(&s.hd[0][0][0]) { return &s.hd[0][0][0]; }
static_assert(sizeof(KnotScratch::hd) >= 14 * sizeof(double), "hand buffer");
constexpr int POSE_HAND_PATH = 8;
// phase C (the links' world poses are there): lane h
template <class Em> HD void t_pose_hand_pts(Ctx<Em>& cx, int h) {
    auto& s = cx.s;
    double* hb = pose_hand_buf(s);
    if (!cx.hands || cx.hands->type[h] == HIPNLP_EXPR_SKIP) { hb[12 + h] = 0.0; return; }
    const PoseHands& hd = *cx.hands;
    const int L = hd.link[h];
    double q[3], t[3], r[3];
    matvec3(hd.R[h], s.xm + XR_HIN + 3 * h, q);
    for (int i = 0; i < 3; ++i) t[i] = hd.o[h][i] + q[i];
    matvec3(s.Rw[L], t, r);
    double c = 0.0;
    for (int i = 0; i < 3; ++i) {
        r[i] += s.ow[L][i];
        const double e = s.x[PB_ + i] + r[i] - s.xm[XR_HREF + 3 * h + i];
        hb[3 * h + i] = r[i];
        hb[6 + 3 * h + i] = e;
        c += e * e;
    }
    hb[12 + h] = hd.type[h] == HIPNLP_EXPR_MINIMIZE ? hd.mult[h] * c : 0.0;
}
// last phase, BEHIND t_feetd on its wave (which adds the chest cost to grad q_b), the left hand's group before the right hand's (their
// gradient shares meet in p_b, q_b and the torso joints: one group = one read-modify-write per address).  Lanes 0..2: row i /
// p_b,i / q_b (lane i takes component i, lane 0 also the fourth); lanes 3..10: joint q of the path root -> hand link.
constexpr int POSE_HAND_TASKS = 3 + POSE_HAND_PATH;
template <class Em> HD void pose_hand_rows(Ctx<Em>& cx, int t, int h) {
    auto& s = cx.s;
    Em& em = cx.em;
    HIPNLP_WAVE_SYNC();
    if (!cx.hands) return;
    const PoseHands& hd = *cx.hands;
    const int mode = hd.type[h];
    if (mode == HIPNLP_EXPR_SKIP) return;
    const double* hb = pose_hand_buf(s);
    const double* r = hb + 3 * h;
    const double* e = hb + 6 + 3 * h;
    const double m2 = 2.0 * hd.mult[h];
    if (t < 3) {
        const int i = t, jb = js::ptc(4 * h + i) + js::FDYN;
        if (mode == HIPNLP_EXPR_SUBJECT_TO) {
            // (Opti's canonical form of `position == parameter`: the row is the position, its bounds are the reference)
            em.G(gs::PT_STRIDE * h + gs::FDYN + i, row_id(RK_PHAND, h, i), s.x[PB_ + i] + r[i]);
            emit_jd(em, jb + 0, row_id(RK_PHAND, h, i), PB_ + i, 1.0);
            const double X0 = skew_rc(r, i, 0), X1 = skew_rc(r, i, 1), X2 = skew_rc(r, i, 2);
            for (int l = 0; l < 4; ++l)
                emit_jd(em, jb + 1 + l, row_id(RK_PHAND, h, i), QB_ + l, -(X0 * s.G[l] + X1 * s.G[4 + l] + X2 * s.G[8 + l]) * s.inv_qnorm);
        } else {   // minimize: gradient 2 m J^T e
            s.grad[PB_ + i] += m2 * e[i];
            for (int l = i; l < 4; l += 3) {
                double acc = 0.0;
                for (int n = 0; n < 3; ++n)
                    acc += e[n] * (skew_rc(r, n, 0) * s.G[l] + skew_rc(r, n, 1) * s.G[4 + l] + skew_rc(r, n, 2) * s.G[8 + l]);
                s.grad[QB_ + l] += -m2 * acc * s.inv_qnorm;
            }
        }
    } else {
        const int q = t - 3;
        const int j = cx.kt.anc[hd.link[h] - 1][q];   // joint q of the path root -> hand link (front padded with NJ)
        if (j >= NJ) return;
        double d[3], x[3];
        for (int n = 0; n < 3; ++n) d[n] = r[n] - s.ow[j + 1][n];
        cross3(s.aw[j], d, x);
        if (mode == HIPNLP_EXPR_SUBJECT_TO) {
            for (int i = 0; i < 3; ++i) emit_jd(em, js::ptc(4 * h + i) + js::FDYN + 5 + q, row_id(RK_PHAND, h, i), S_ + j, x[i]);
        } else {
            s.grad[S_ + j] += m2 * dot3(e, x);
        }
    }
}
template <class Em> HD void t_pose_hand_rows_l(Ctx<Em>& cx, int t) { pose_hand_rows(cx, t, 0); }
template <class Em> HD void t_pose_hand_rows_r(Ctx<Em>& cx, int t) { pose_hand_rows(cx, t, 1); }

// chest-frame orientation error of a STATIC emitter (t_frames leaves it out there): R_chest from the link's world rotation — the product t_frames
// forms for fr_R, once more, so that this group waits for nobody in its phase — then cost, d cost / d trace and ax(R_chest R_d^T) as t_frames.  1 lane.
template <class Em> HD void t_pose_chest(Ctx<Em>& cx, int) {
    if constexpr (em_static<Em>) {
        auto& s = cx.s;
        const int f = HIPNLP_FRAME_CHEST;
        double Rc[9], Rd[9], M[9], Rdt[9];
        matmul3(s.Rw[cx.kt.frame_link[f]], cx.kt.frame_R[f], Rc);
        rot_from_quat(s.pk + PK_REF + R_FQ, Rd);
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) Rdt[3 * r + q] = Rd[3 * q + r];
        matmul3(Rc, Rdt, M);
        const double e = (M[0] + M[4] + M[8]) - 3.0;
        const double on = cx.ki.first ? 0.0 : 1.0;
        const double m = on * cx.st.m_frameq;
        s.cost[CT_FRAMEQ] = m * (e * e);
        s.chest_dc = 2.0 * m * e;
        s.chest_w[0] = M[7] - M[5]; s.chest_w[1] = M[2] - M[6]; s.chest_w[2] = M[3] - M[1];
    } else { (void)cx; }
}
// the three sums over the contact points among the cost terms (point position | force | average force regularisation: c_pt[.][k]), lane k (3),
// in a phase behind t_pose_points on a wave the kinematic chain leaves idle — until round 6 the first thread of the workgroup added them
// up, 24 dependent additions, between the last barrier and the copy-out.  Parked in cost slots no task of the pose program writes.
static_assert(CT_SWING + 2 < CT_COMVEL && CT_UREG == CT_SWING + 1 && CT_FDREG == CT_SWING + 2, "three cost slots the pose program does not use");
template <class Em> HD void t_pose_cost_sums(Ctx<Em>& cx, int k) {
    auto& s = cx.s;
    double a = 0.0;
    for (int c = 0; c < NC; ++c) a += s.c_pt[c][k];
    s.cost[CT_SWING + k] = a;
}
// the cost of the pose: the nine terms in their order, one lane, in a phase behind the last task that writes one of them (t_frames and
// t_pose_hand_pts in the third) — the copy-out stores nine terms from nine lanes and this sum (until round 6 its first thread walked the
// terms between the last barrier and its stores: 1.2 k of the 2.1 k cycles of the copy-out)
constexpr int CT_POSE_TOTAL = CT_ENDS;
static_assert(CT_POSE_TOTAL > CT_JREG && CT_POSE_TOTAL != CT_FRAMEQ && CT_POSE_TOTAL != CT_BASEQ, "a cost slot the pose program does not use");
template <class S> HD double pose_cost_term(const S& s, int t);
// (the nine terms once more, side by side, for the copy-out: on the per-point partials c_pt[0..2][], which t_pose_cost_sums has consumed)
template <class S> HD auto pose_cost_terms_out(S& s) -> decltype(&s.c_pt[0][0]) { return &s.c_pt[0][0]; }
static_assert(POSE_NCT <= 3 * NC, "the terms fit on the point partials");
template <class Em> HD void t_pose_cost_total(Ctx<Em>& cx, int) {
    auto& s = cx.s;
    double tot = 0.0, v[POSE_NCT];
    for (int t = 0; t < POSE_NCT; ++t) { v[t] = pose_cost_term(s, t); tot += v[t]; }
    s.cost[CT_POSE_TOTAL] = tot;
    for (int t = 0; t < POSE_NCT; ++t) pose_cost_terms_out(s)[t] = v[t];
}
// cost term t of the pose (order of hipnlp_pose_cost_term_name) from the scratch, after the program has run
template <class S> HD double pose_cost_term(const S& s, int t) {
    switch (t) {
        case 0: return s.cost[CT_BASEQ];
        case 1: return s.cost[CT_FRAMEQ];
        case 2: return s.cost[CT_COMVEL];
        case 3: return s.cost[CT_JREG];
        case 7: case 8: return (&s.hd[0][0][0])[12 + (t - 7)];   // left / right hand position error (pose_hand_buf)
        default: return s.cost[CT_SWING + (t == 4 ? 2 : (t == 5 ? 0 : 1))];   // average force | point position | force (t_pose_cost_sums)
    }
}

// The pose program: same notation as HIPNLP_KNOT_PROGRAM (always four waves: both role ids are the same).
// The chain every pose waits for is joints / base -> forward kinematics -> links -> composites -> derivative columns -> consistency rows;
// the task groups that need no robot model (contact rows, static balance, com / quaternion / joint costs) run beside it on the waves it
// leaves idle (round 6: first phase 6.3 k -> 3 k cycles at batch; nothing of the values changes, only where a group runs).
#define HIPNLP_POSE_PROGRAM(R, BARRIER)                                                   \
    R(0, 0, t_pose_points_first, NC) R(0, 0, t_pose_balance_rows_ang, 3)                  \
    R(1, 1, t_pose_balance_rows_lin, 3) R(1, 1, t_pose_balance_com, POSE_BALANCE_COM)     \
    R(2, 2, t_pose_joints, NJ)                                                            \
    R(3, 3, t_base, 3) R(3, 3, t_kin_padding, 16)                                         \
    BARRIER                                                                               \
    R(0, 0, t_fk_rot_a, FK_TASKS_A) R(0, 0, t_link_u_a, FK_SPLIT)                         \
    R(1, 1, t_pose_points_second, NC) R(1, 1, t_pose_cost_sums, 3)                        \
    R(2, 2, t_joint_cost, 1) R(2, 2, t_pose_com, 4) R(2, 2, t_unitq, 1)                   \
    R(3, 3, t_fk_rot_b, FK_TASKS_B) R(3, 3, t_link_u_b, NJ - FK_SPLIT)                    \
    BARRIER                                                                               \
    R(0, 0, t_links, NL) R(1, 1, t_frames, 3) R(2, 2, t_link_inertia, NL) R(3, 3, t_pose_hand_pts, 2) R(3, 3, t_pose_chest, 1) \
    BARRIER                                                                               \
    R(0, 0, t_composite_g0, 64) R(1, 1, t_composite_g1, 64) R(1, 1, t_composite_g2, 64)   \
    R(2, 2, t_composite_g3, 64) R(2, 2, t_composite_g4, 64) R(3, 3, t_composite_g5, 64) R(3, 3, t_pkin, NC) \
    BARRIER                                                                               \
    R(0, 0, t_columns, NJ + 3) R(1, 1, t_pose_balance_entries, POSE_BALANCE_ENTRIES) R(2, 2, t_frame_columns, NJ) R(3, 3, t_pose_cost_total, 1) \
    BARRIER                                                                               \
    R(0, 0, t_kinc, 3 * NC) R(1, 1, t_comc, 15) R(2, 2, t_kinc_s, NC * LEG_PATH) R(3, 3, t_feetd, 4) R(3, 3, t_pose_hand_rows_l, POSE_HAND_TASKS) R(3, 3, t_pose_hand_rows_r, POSE_HAND_TASKS) \
    BARRIER

}  // namespace hipnlp
