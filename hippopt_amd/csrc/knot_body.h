// knot_body.h — the per-knot mathematics of the hipnlp engine, written once and compiled for the
// device (HIP kernel, gfx950), for the host layout recorder (layout.cpp) and for the test-only host
// emulation (tests/hostemu).  One workgroup evaluates ONE knot: its rows of g (own algebraic rows and
// the trapezoid defect that ends at the knot), its COLUMN block of jac g (so the CCS output of a knot is
// one contiguous run), its slice of grad f and its cost partials.
//
// Every `phase_*(cx, t)` is a lane task: the kernel runs task t on thread t (tasks of one phase are
// independent), with a workgroup barrier between phases.  Values leave through an emitter:
// em.G(slot, row_id, v) / em.J(slot, row_id, col, v); slots are compile-time native positions.
//
// Mathematics (citations relative to /root/reference/src/hippopt/):
//   contact rows   robot_planning/expressions/complementarity.py:27-32,71-87; contacts.py:22-24,54-66,158-166
//   defects        integrators/implicit_trapezoid.py:24-39 via base/multiple_shooting_solver.py:713-742
//   momentum       robot_planning/expressions/centroidal.py:62-64
//   kinematics     robot_planning/expressions/kinematics.py:47-69,163,249-265,337-367,428-448 (adam FK/CoM/CMM)
//                  as an O(n) spatial-algebra recursion with composite inertias; configuration derivative
//                  of the momentum:  dk/ds_j = S_j x* k_sub(j) - I_sub(j) (S_j x v_j)      
//   quaternions    robot_planning/expressions/quaternion.py:13,35-42,64-70
//   costs          turnkey_planners/humanoid_kinodynamic/planner.py:249-264,427-520,746-895
#pragma once
#include <math.h>

#include "nlp_defs.h"

namespace hipnlp {

// ---------------------------------------------------------------------------------------------------
// scratch (LDS on the device)
// ---------------------------------------------------------------------------------------------------
struct KnotScratch {
    double x[XPAD];    // knot k
    double xm[XPAD];   // knot k-1 (zeros at k = 0)
    double xo[XPAD];   // the other end of the horizon (only loaded at k = 0 and k = N-1)
    double xg[8];      // horizon-global variables (initial_state.centroidal_momentum)
    double pk[PK_STRIDE];
    // base orientation
    double qn[4], qnorm, Rb[9], G[12] /* 3x4: dtheta = G dqhat */, omega[3], dwq[12] /* d omega/d qb (3x4) */;
    // kinematics in base-centred coordinates (origin = base origin; h_ang and relative positions are
    // invariant to the base position / linear velocity)
    double Rloc[NJ][9];
    double Rw[NL][9], ow[NL][3], aw[NJ][3];
    double wv[NL][3], vo[NL][3];  // link angular velocity; velocity of the body point at the origin
    double cm[NL], ch[NL][3], cI[NL][6], ckl[NL][3], cka[NL][3];  // composite m, first moment, inertia@O, subtree momentum
    double com[3], klin[3], kang[3], hang[3];
    double dth_h[3][3];  // d hang / d theta_e   [e][i]
    double Aw[3][3];     // d hang / d omega_e   [e][i]
    double fr_R[3][9], fr_o[3][3];
    double pkin[NC][3];
    double chest_w[3], chest_dc;  // ax(R_c R_d^T);  d cost / d trace
    double hd[2][NC][6];          // per point contribution to hdot at knots k-1 (0) and k (1)
    double cen_g[3];              // d centroid cost / d p_c,i (same for the 8 points)
    double cpt[NC][3], cjt[NJ], cft[2][2];  // cost partials: per point (swing,u,fdot), per joint, per foot (freg,yaw)
    double cost[NCT];
    double grad[XPAD];
    double g[gs::COUNT];
    double jac[js::COUNT];
};

struct KnotInfo {
    int k, N;
    int first, last;  // k == 0, k == N-1   (the recorder sets both)
};

template <class Em> struct Ctx {
    KnotScratch& s;
    const KinTables& kt;
    const KSettings& st;
    const GParams& gp;
    KnotInfo ki;
    Em em;
    HD Ctx(KnotScratch& s_, const KinTables& kt_, const KSettings& st_, const GParams& gp_, KnotInfo ki_, Em em_)
        : s(s_), kt(kt_), st(st_), gp(gp_), ki(ki_), em(em_) {}
};

// ---------------------------------------------------------------------------------------------------
// tiny helpers on raw arrays
// ---------------------------------------------------------------------------------------------------
HD void cross3(const double* a, const double* b, double* r) {
    const double r0 = a[1] * b[2] - a[2] * b[1];
    const double r1 = a[2] * b[0] - a[0] * b[2];
    const double r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
HD void matvec3(const double* M, const double* v, double* r) {
    const double r0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    const double r1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    const double r2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
HD void matmul3(const double* A, const double* B, double* C) {  // C must not alias A or B
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
HD void symvec(const double* S, const double* v, double* r) {  // S = (xx,xy,xz,yy,yz,zz)
    const double r0 = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
    const double r1 = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
    const double r2 = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
HD double skew_entry(const double* a, int e) {  // entry e of [a]x, e -> (cross_row, cross_col)
    switch (e) {
        case 0: return -a[2];
        case 1: return a[1];
        case 2: return a[2];
        case 3: return -a[0];
        case 4: return -a[1];
        default: return a[0];
    }
}
// R = I + 2 w [v]x + 2 [v]x^2  (liecasadi SO3.as_matrix, xyzw)
HD void rot_from_quat(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}

// ===================================================================================================
// PHASE A tasks (all depend only on the loaded knot records; no barrier needed between them)
// ===================================================================================================

// --- A1: contact points, lane c (8).  planner.py:124-147, planar terrain ---------------------------
template <class Em> HD void phase_points(Ctx<Em>& cx, int c) {
    KnotScratch& s = cx.s;
    const double* x = s.x + PT_ * c;
    const double* xm = s.xm + PT_ * c;
    const double half = 0.5 * cx.gp.dt;
    const int gb = gs::PT_STRIDE * c, jb = js::PT_STRIDE * c, cb = PT_ * c;
    Em& em = cx.em;
    for (int i = 0; i < 3; ++i) {  // trapezoid defects of dot(f) = f_dot, dot(p) = v (T7) + x0 rows
        em.G(gb + gs::FDYN + i, row_id(RK_FDYN_IN, c, i), x[F_ + i] - (xm[F_ + i] + half * (xm[FD_ + i] + x[FD_ + i])));
        em.G(gb + gs::PDYN + i, row_id(RK_PDYN_IN, c, i), x[P_ + i] - (xm[P_ + i] + half * (xm[V_ + i] + x[V_ + i])));
        em.G(gb + gs::FDYN_X0 + i, row_id(RK_FDYN_X0, c, i), x[F_ + i]);
        em.G(gb + gs::PDYN_X0 + i, row_id(RK_PDYN_X0, c, i), x[P_ + i]);
        em.J(jb + js::FDYN + 0 + i, row_id(RK_FDYN_IN, c, i), cb + F_ + i, 1.0);
        em.J(jb + js::FDYN + 3 + i, row_id(RK_FDYN_IN, c, i), cb + FD_ + i, -half);
        em.J(jb + js::FDYN + 6 + i, row_id(RK_FDYN_OUT, c, i), cb + F_ + i, -1.0);
        em.J(jb + js::FDYN + 9 + i, row_id(RK_FDYN_OUT, c, i), cb + FD_ + i, -half);
        em.J(jb + js::FDYN + 12 + i, row_id(RK_FDYN_X0, c, i), cb + F_ + i, 1.0);
        em.J(jb + js::PDYN + 0 + i, row_id(RK_PDYN_IN, c, i), cb + P_ + i, 1.0);
        em.J(jb + js::PDYN + 3 + i, row_id(RK_PDYN_IN, c, i), cb + V_ + i, -half);
        em.J(jb + js::PDYN + 6 + i, row_id(RK_PDYN_OUT, c, i), cb + P_ + i, -1.0);
        em.J(jb + js::PDYN + 9 + i, row_id(RK_PDYN_OUT, c, i), cb + V_ + i, -half);
        em.J(jb + js::PDYN + 12 + i, row_id(RK_PDYN_X0, c, i), cb + P_ + i, 1.0);
    }
    const double pz = x[P_ + 2], fz = x[F_ + 2], vz = x[V_ + 2], fdz = x[FD_ + 2];
    // planar complementarity  v - R_t diag(tau,tau,1) u,  tau = tanh(kt h(p))   (E3; R_t = I, h = p_z)
    const double tau = tanh(cx.gp.kt * pz);
    const double dtau = cx.gp.kt * (1.0 - tau * tau);
    for (int i = 0; i < 3; ++i) {
        const double mult = i < 2 ? tau : 1.0;
        em.G(gb + gs::PLANAR + i, row_id(RK_PLANAR, c, i), x[V_ + i] - mult * x[U_ + i]);
        em.J(jb + js::PLANAR_V + i, row_id(RK_PLANAR, c, i), cb + V_ + i, 1.0);
        em.J(jb + js::PLANAR_U + i, row_id(RK_PLANAR, c, i), cb + U_ + i, -mult);
        if (i < 2) em.J(jb + js::PLANAR_PZ + i, row_id(RK_PLANAR, c, i), cb + P_ + 2, -dtau * x[U_ + i]);
    }
    // dcc margin  eps - k h (n.f) - [hdot (n.f) + h f.ndot + h (n.fdot)]   (E4; n = e_z, ndot = 0, hdot = v_z)
    em.G(gb + gs::DCC, row_id(RK_DCC, c, 0), cx.gp.eps - cx.gp.kbs * (pz * fz) - (vz * fz + pz * fdz));
    em.J(jb + js::DCC + 0, row_id(RK_DCC, c, 0), cb + P_ + 2, -cx.gp.kbs * fz - fdz);
    em.J(jb + js::DCC + 1, row_id(RK_DCC, c, 0), cb + F_ + 2, -cx.gp.kbs * pz - vz);
    em.J(jb + js::DCC + 2, row_id(RK_DCC, c, 0), cb + V_ + 2, -fz);
    em.J(jb + js::DCC + 3, row_id(RK_DCC, c, 0), cb + FD_ + 2, -pz);
    // height, normal force, friction cone  (E14, E6, E7)
    em.G(gb + gs::HEIGHT, row_id(RK_HEIGHT, c, 0), pz);
    em.J(jb + js::HEIGHT, row_id(RK_HEIGHT, c, 0), cb + P_ + 2, 1.0);
    em.G(gb + gs::NORMAL, row_id(RK_NORMAL, c, 0), fz);
    em.J(jb + js::NORMAL, row_id(RK_NORMAL, c, 0), cb + F_ + 2, 1.0);
    const double mu2 = cx.gp.mu * cx.gp.mu;
    em.G(gb + gs::FRICTION, row_id(RK_FRICTION, c, 0), -(x[F_] * x[F_]) - (x[F_ + 1] * x[F_ + 1]) + mu2 * (fz * fz));
    em.J(jb + js::FRICTION + 0, row_id(RK_FRICTION, c, 0), cb + F_ + 0, -2.0 * x[F_]);
    em.J(jb + js::FRICTION + 1, row_id(RK_FRICTION, c, 0), cb + F_ + 1, -2.0 * x[F_ + 1]);
    em.J(jb + js::FRICTION + 2, row_id(RK_FRICTION, c, 0), cb + F_ + 2, 2.0 * mu2 * fz);
    for (int i = 0; i < 3; ++i) {  // control bound rows
        em.G(gb + gs::UB + i, row_id(RK_UB, c, i), x[U_ + i]);
        em.J(jb + js::UB + i, row_id(RK_UB, c, i), cb + U_ + i, 1.0);
        em.G(gb + gs::FDB + i, row_id(RK_FDB, c, i), x[FD_ + i] * cx.gp.mass);
        em.J(jb + js::FDB + i, row_id(RK_FDB, c, i), cb + FD_ + i, cx.gp.mass);
    }
    // point-local costs (k >= 1): swing height heuristic (E10), ||u_v||^2, ||f_dot||^2
    const double on = cx.ki.first ? 0.0 : 1.0;
    const double dh = pz - s.pk[PK_REF + R_SWING];
    double* gr = s.grad + cb;
    const double msw = on * cx.st.m_swing, mur = on * cx.st.m_ureg, mfd = on * cx.st.m_fdreg;
    s.cpt[c][0] = msw * (0.5 * (dh * dh + (x[V_] * x[V_] + x[V_ + 1] * x[V_ + 1])));
    s.cpt[c][1] = mur * (x[U_] * x[U_] + x[U_ + 1] * x[U_ + 1] + x[U_ + 2] * x[U_ + 2]);
    s.cpt[c][2] = mfd * (x[FD_] * x[FD_] + x[FD_ + 1] * x[FD_ + 1] + x[FD_ + 2] * x[FD_ + 2]);
    gr[V_ + 0] = msw * x[V_]; gr[V_ + 1] = msw * x[V_ + 1]; gr[V_ + 2] = 0.0;
    gr[P_ + 0] = 0.0; gr[P_ + 1] = 0.0; gr[P_ + 2] = msw * dh;
    for (int i = 0; i < 3; ++i) { gr[U_ + i] = 2.0 * mur * x[U_ + i]; gr[FD_ + i] = 2.0 * mfd * x[FD_ + i]; gr[F_ + i] = 0.0; }
    // contribution of this point to hdot at k-1 and k   (E1)
    for (int w = 0; w < 2; ++w) {
        const double* xx = w ? s.x : s.xm;
        double r[3], t[3];
        for (int i = 0; i < 3; ++i) r[i] = xx[cb + P_ + i] - xx[COM_ + i];
        cross3(r, xx + cb + F_, t);
        for (int i = 0; i < 3; ++i) { s.hd[w][c][i] = xx[cb + F_ + i]; s.hd[w][c][3 + i] = t[i]; }
    }
}

// --- A2: trivial dynamics of base / joints / com, lane e over 33 state components.  planner.py:522-564
template <class Em> HD void phase_dyn(Ctx<Em>& cx, int e) {
    KnotScratch& s = cx.s;
    const double half = 0.5 * cx.gp.dt;
    int X, Y, L, i, kin, gslot, gx0, jslot;
    if (e < 3) { i = e; L = 3; X = PB_ + i; Y = VB_ + i; kin = RK_PBDYN_IN; gslot = gs::PBDYN; gx0 = gs::PB_X0; jslot = js::PBDYN; }
    else if (e < 7) { i = e - 3; L = 4; X = QB_ + i; Y = QD_ + i; kin = RK_QBDYN_IN; gslot = gs::QBDYN; gx0 = gs::QB_X0; jslot = js::QBDYN; }
    else if (e < 7 + NJ) { i = e - 7; L = NJ; X = S_ + i; Y = SD_ + i; kin = RK_SDYN_IN; gslot = gs::SDYN; gx0 = gs::S_X0; jslot = js::SDYN; }
    else { i = e - 7 - NJ; L = 3; X = COM_ + i; Y = H_ + i; kin = RK_COMDYN_IN; gslot = gs::COMDYN; gx0 = gs::COM_X0; jslot = js::COMDYN; }
    const int kout = kin + 1, kx0 = kin + 2;
    Em& em = cx.em;
    em.G(gslot + i, row_id(kin, 0, i), s.x[X] - (s.xm[X] + half * (s.xm[Y] + s.x[Y])));
    em.G(gx0 + i, row_id(kx0, 0, i), s.x[X]);
    em.J(jslot + 0 * L + i, row_id(kin, 0, i), X, 1.0);
    em.J(jslot + 1 * L + i, row_id(kin, 0, i), Y, -half);
    em.J(jslot + 2 * L + i, row_id(kout, 0, i), X, -1.0);
    em.J(jslot + 3 * L + i, row_id(kout, 0, i), Y, -half);
    em.J(jslot + 4 * L + i, row_id(kx0, 0, i), X, 1.0);
}

// --- A3: joint-wise rows and joint regularisation cost, lane j (23) + local joint transform -----------
template <class Em> HD void phase_joints(Ctx<Em>& cx, int j) {
    KnotScratch& s = cx.s;
    Em& em = cx.em;
    em.G(gs::JPB + j, row_id(RK_JPB, 0, j), s.x[S_ + j]);
    em.J(js::JPB + j, row_id(RK_JPB, 0, j), S_ + j, 1.0);
    em.G(gs::JVB + j, row_id(RK_JVB, 0, j), s.x[SD_ + j]);
    em.J(js::JVB + j, row_id(RK_JVB, 0, j), SD_ + j, 1.0);
    // joint_positions_error  planner.py:505-520
    const double on = cx.ki.first ? 0.0 : 1.0;
    const double m = on * cx.st.m_jreg, w = cx.st.w_jreg[j];
    const double sd = s.x[SD_ + j];
    const double t = sd + w * (s.x[S_ + j] - s.pk[PK_REF + R_JREG + j]);
    double c = t * t, gsd = 2.0 * t;
    if (cx.st.joint_reg_as_coded) { c += double(NJ - 1) * (sd * sd); gsd += 2.0 * double(NJ - 1) * sd; }
    s.cjt[j] = m * c;
    s.grad[S_ + j] = 2.0 * m * t * w;
    s.grad[SD_ + j] = m * gsd;
    // parent_R_child = R_fix * (cq (I - a a^T) + sq [a]x + a a^T)   (adam R_from_axis_angle)
    const double* a = cx.kt.axis[j];
    double sq, cq;
    sincos(s.x[S_ + j], &sq, &cq);
    double Ra[9];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) { const double aa = a[r] * a[cc]; Ra[3 * r + cc] = cq * ((r == cc ? 1.0 : 0.0) - aa) + aa; }
    Ra[1] -= sq * a[2]; Ra[2] += sq * a[1];
    Ra[3] += sq * a[2]; Ra[5] -= sq * a[0];
    Ra[6] -= sq * a[1]; Ra[7] += sq * a[0];
    matmul3(cx.kt.R_fix[j], Ra, s.Rloc[j]);
}

// --- A4: one-off global tasks --------------------------------------------------------------------------
constexpr int MISC_TASKS = 6;
template <class Em> HD void phase_misc(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    Em& em = cx.em;
    const double on = cx.ki.first ? 0.0 : 1.0;
    switch (t) {
        case 0: {  // unitary quaternion sumsqr(q) == 1 (planner.py:276-282); base quaternion error cost (E13, raw q)
            const double* q = s.x + QB_;
            em.G(gs::UNITQ, row_id(RK_UNITQ, 0, 0), q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (int l = 0; l < 4; ++l) em.J(js::UNITQ + l, row_id(RK_UNITQ, 0, 0), QB_ + l, 2.0 * q[l]);
            const double* d = s.pk + PK_REF + R_BQ;
            const double ax = -d[0], ay = -d[1], az = -d[2], aw = d[3];  // conj(q_d)
            double e[4];
            e[0] = aw * q[0] + ax * q[3] + ay * q[2] - az * q[1];
            e[1] = aw * q[1] - ax * q[2] + ay * q[3] + az * q[0];
            e[2] = aw * q[2] + ax * q[1] - ay * q[0] + az * q[3];
            e[3] = aw * q[3] - ax * q[0] - ay * q[1] - az * q[2] - 1.0;
            const double m = on * cx.st.m_baseq;
            s.cost[CT_BASEQ] = m * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
            s.grad[QB_ + 0] = 2.0 * m * (e[0] * aw + e[1] * az - e[2] * ay - e[3] * ax);
            s.grad[QB_ + 1] = 2.0 * m * (-e[0] * az + e[1] * aw + e[2] * ax - e[3] * ay);
            s.grad[QB_ + 2] = 2.0 * m * (e[0] * ay - e[1] * ax + e[2] * aw - e[3] * az);
            s.grad[QB_ + 3] = 2.0 * m * (e[0] * ax + e[1] * ay + e[2] * az + e[3] * aw);
        } break;
        case 1: {  // angular momentum bound rows h[3:]*mass; com velocity cost (k >= 0)
            double c = 0.0;
            for (int i = 0; i < 3; ++i) {
                em.G(gs::AMB + i, row_id(RK_AMB, 0, i), s.x[H_ + 3 + i] * cx.gp.mass);
                em.J(js::AMB + i, row_id(RK_AMB, 0, i), H_ + 3 + i, cx.gp.mass);
                const double e = s.x[H_ + i] - s.pk[PK_REF + R_VREF + i];
                c += e * cx.st.w_comvel[i] * e;
                s.grad[H_ + i] = 2.0 * cx.st.m_comvel * cx.st.w_comvel[i] * e;
                s.grad[H_ + 3 + i] = 0.0;
            }
            s.cost[CT_COMVEL] = cx.st.m_comvel * c;
        } break;
        case 2: {  // minimum com height: h_terrain(com) = com_z
            em.G(gs::COMH, row_id(RK_COMH, 0, 0), s.x[COM_ + 2]);
            em.J(js::COMH, row_id(RK_COMH, 0, 0), COM_ + 2, 1.0);
            for (int i = 0; i < 3; ++i) { s.grad[COM_ + i] = 0.0; s.grad[PB_ + i] = 0.0; s.grad[VB_ + i] = 0.0; }
        } break;
        case 3: {  // base quaternion velocity cost (k >= 0)
            double c = 0.0;
            for (int i = 0; i < 4; ++i) {
                const double e = s.x[QD_ + i] - s.pk[PK_REF + R_BQV + i];
                c += e * e;
                s.grad[QD_ + i] = 2.0 * cx.st.m_baseqv * e;
            }
            s.cost[CT_BASEQV] = cx.st.m_baseqv * c;
        } break;
        case 4: {  // feet centroids: relative height row + centroid cost (k >= 1)   planner.py:215-264
            double cl[3] = {0, 0, 0}, cr[3] = {0, 0, 0};
            for (int c = 0; c < 4; ++c)
                for (int i = 0; i < 3; ++i) { cl[i] += s.x[PT_ * c + P_ + i]; cr[i] += s.x[PT_ * (c + 4) + P_ + i]; }
            for (int i = 0; i < 3; ++i) { cl[i] = cl[i] / 4.0; cr[i] = cr[i] / 4.0; }
            em.G(gs::FEETH, row_id(RK_FEETH, 0, 0), cl[2] - cr[2]);
            for (int c = 0; c < NC; ++c) em.J(js::FEETH + c, row_id(RK_FEETH, 0, 0), PT_ * c + P_ + 2, c < 4 ? 0.25 : -0.25);
            double cost = 0.0;
            const double m = on * cx.st.m_centroid;
            for (int i = 0; i < 3; ++i) {
                const double e = s.pk[PK_REF + R_CREF + i] - 0.5 * (cl[i] + cr[i]);
                const double w = s.pk[PK_REF + R_CW + i];
                cost += e * w * e;
                s.cen_g[i] = -0.25 * m * w * e;  // 2 w e * d e / d p_c,i = 2 w e (-0.5/4)
            }
            s.cost[CT_CENTROID] = m * cost;
        } break;
        case 5: {  // base orientation: normalised quaternion (E11), R_b, G, omega (E12), d omega / d q_b
            const double* q = s.x + QB_;
            const double* qd = s.x + QD_;
            const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            s.qnorm = n;
            double qn[4];
            for (int i = 0; i < 4; ++i) { qn[i] = q[i] / n; s.qn[i] = qn[i]; }
            rot_from_quat(qn, s.Rb);
            const double vx = qn[0], vy = qn[1], vz = qn[2], w = qn[3];
            // G = 2 [ w I + [v]x | -v ]
            double* G = s.G;
            G[0] = 2.0 * w;   G[1] = -2.0 * vz; G[2] = 2.0 * vy;  G[3] = -2.0 * vx;
            G[4] = 2.0 * vz;  G[5] = 2.0 * w;   G[6] = -2.0 * vx; G[7] = -2.0 * vy;
            G[8] = -2.0 * vy; G[9] = 2.0 * vx;  G[10] = 2.0 * w;  G[11] = -2.0 * vz;
            for (int e = 0; e < 3; ++e) s.omega[e] = G[4 * e] * qd[0] + G[4 * e + 1] * qd[1] + G[4 * e + 2] * qd[2] + G[4 * e + 3] * qd[3];
            // omega = H(qdot) qhat,  H = 2 [ -qd_w I - [qd_v]x | qd_v ];  d omega / d q = H (I - qn qn^T) / |q|
            double H[12];
            H[0] = -2.0 * qd[3]; H[1] = 2.0 * qd[2];  H[2] = -2.0 * qd[1]; H[3] = 2.0 * qd[0];
            H[4] = -2.0 * qd[2]; H[5] = -2.0 * qd[3]; H[6] = 2.0 * qd[0];  H[7] = 2.0 * qd[1];
            H[8] = 2.0 * qd[1];  H[9] = -2.0 * qd[0]; H[10] = -2.0 * qd[3]; H[11] = 2.0 * qd[2];
            for (int e = 0; e < 3; ++e) {
                const double hq = H[4 * e] * qn[0] + H[4 * e + 1] * qn[1] + H[4 * e + 2] * qn[2] + H[4 * e + 3] * qn[3];
                for (int l = 0; l < 4; ++l) s.dwq[4 * e + l] = (H[4 * e + l] - hq * qn[l]) / n;
            }
            // root link pose and velocity in base-centred coordinates
            for (int i = 0; i < 9; ++i) s.Rw[0][i] = s.Rb[i];
            for (int i = 0; i < 3; ++i) { s.ow[0][i] = 0.0; s.wv[0][i] = s.omega[i]; s.vo[0][i] = 0.0; }
        } break;
        default: break;
    }
}

// ===================================================================================================
// PHASE B(d): forward kinematics + link velocities, one tree level per phase; lane j (joints of depth d)
// ===================================================================================================
template <class Em> HD void phase_fk_level(Ctx<Em>& cx, int j, int d) {
    if (cx.kt.depth[j] != d) return;
    KnotScratch& s = cx.s;
    const int i = j + 1, par = cx.kt.parent[j];
    matmul3(s.Rw[par], s.Rloc[j], s.Rw[i]);
    double t[3];
    matvec3(s.Rw[par], cx.kt.o_fix[j], t);
    for (int r = 0; r < 3; ++r) s.ow[i][r] = s.ow[par][r] + t[r];
    matvec3(s.Rw[i], cx.kt.axis[j], s.aw[j]);
    const double sd = s.x[SD_ + j];
    double oxa[3];
    cross3(s.ow[i], s.aw[j], oxa);
    for (int r = 0; r < 3; ++r) { s.wv[i][r] = s.wv[par][r] + s.aw[j][r] * sd; s.vo[i][r] = s.vo[par][r] + oxa[r] * sd; }
}

// ===================================================================================================
// PHASE C: per-link spatial inertia at the origin and link momentum, lane i (24 links);
//          frames (lanes 24..26)
// ===================================================================================================
template <class Em> HD void phase_links(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    if (t < NL) {
        const int i = t;
        const double m = cx.kt.mass[i];
        double c[3], RI[9], Iw[9], Rt[9];
        matvec3(s.Rw[i], cx.kt.com[i], c);
        for (int r = 0; r < 3; ++r) c[r] += s.ow[i][r];
        matmul3(s.Rw[i], cx.kt.inertia[i], RI);
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) Rt[3 * r + q] = s.Rw[i][3 * q + r];
        matmul3(RI, Rt, Iw);
        const double c2 = dot3(c, c);
        s.cm[i] = m;
        for (int r = 0; r < 3; ++r) s.ch[i][r] = m * c[r];
        s.cI[i][0] = Iw[0] + m * (c2 - c[0] * c[0]);
        s.cI[i][1] = 0.5 * (Iw[1] + Iw[3]) - m * c[0] * c[1];
        s.cI[i][2] = 0.5 * (Iw[2] + Iw[6]) - m * c[0] * c[2];
        s.cI[i][3] = Iw[4] + m * (c2 - c[1] * c[1]);
        s.cI[i][4] = 0.5 * (Iw[5] + Iw[7]) - m * c[1] * c[2];
        s.cI[i][5] = Iw[8] + m * (c2 - c[2] * c[2]);
        // link momentum about the origin:  lin = m vO + w x h ;  ang = I_O w + h x vO
        double a[3], b[3];
        cross3(s.wv[i], s.ch[i], a);
        for (int r = 0; r < 3; ++r) s.ckl[i][r] = m * s.vo[i][r] + a[r];
        symvec(s.cI[i], s.wv[i], a);
        cross3(s.ch[i], s.vo[i], b);
        for (int r = 0; r < 3; ++r) s.cka[i][r] = a[r] + b[r];
    } else if (t < NL + 3) {
        const int f = t - NL, L = cx.kt.frame_link[f];
        matmul3(s.Rw[L], cx.kt.frame_R[f], s.fr_R[f]);
        double o[3];
        matvec3(s.Rw[L], cx.kt.frame_o[f], o);
        for (int r = 0; r < 3; ++r) s.fr_o[f][r] = s.ow[L][r] + o[r];
        if (f == HIPNLP_FRAME_CHEST) {  // rotation error R_chest R(q_d)^T  (K5) -> trace and ax()
            double Rd[9], M[9], Rdt[9];
            rot_from_quat(s.pk + PK_REF + R_FQ, Rd);
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) Rdt[3 * r + q] = Rd[3 * q + r];
            matmul3(s.fr_R[f], Rdt, M);
            const double e = (M[0] + M[4] + M[8]) - 3.0;
            const double on = cx.ki.first ? 0.0 : 1.0;
            const double m = on * cx.st.m_frameq;
            s.cost[CT_FRAMEQ] = m * (e * e);
            s.chest_dc = 2.0 * m * e;
            s.chest_w[0] = M[7] - M[5]; s.chest_w[1] = M[2] - M[6]; s.chest_w[2] = M[3] - M[1];
        }
    }
}

// ===================================================================================================
// PHASE D(d): composite quantities, backward over tree levels; lane i (links of depth d gather children)
// ===================================================================================================
template <class Em> HD void phase_composite_level(Ctx<Em>& cx, int i, int d) {
    if (cx.kt.link_depth[i] != d) return;
    KnotScratch& s = cx.s;
    for (int q = 0; q < cx.kt.nchild[i]; ++q) {
        const int c = cx.kt.child[i][q];
        s.cm[i] += s.cm[c];
        for (int r = 0; r < 3; ++r) { s.ch[i][r] += s.ch[c][r]; s.ckl[i][r] += s.ckl[c][r]; s.cka[i][r] += s.cka[c][r]; }
        for (int r = 0; r < 6; ++r) s.cI[i][r] += s.cI[c][r];
    }
}

// ===================================================================================================
// PHASE E: totals (lane 0) and contact point kinematics (lanes 1..8)
// ===================================================================================================
template <class Em> HD void phase_totals(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    if (t == 0) {
        const double M = cx.kt.total_mass;
        double t3[3];
        for (int r = 0; r < 3; ++r) { s.com[r] = s.ch[0][r] / M; s.klin[r] = s.ckl[0][r]; s.kang[r] = s.cka[0][r]; }
        cross3(s.com, s.klin, t3);
        for (int r = 0; r < 3; ++r) s.hang[r] = s.kang[r] - t3[r];
    } else if (t <= NC) {
        const int c = t - 1, f = c < 4 ? 0 : 1;
        double r3[3];
        matvec3(s.fr_R[f], s.pk + PK_DESC + 3 * c, r3);
        for (int r = 0; r < 3; ++r) s.pkin[c][r] = s.fr_o[f][r] + r3[r];
    }
}

// ===================================================================================================
// PHASE F: derivative columns.  lanes 0..22: joint j ; lanes 23..25: base rotation theta_e / omega_e
// ===================================================================================================
template <class Em> HD void phase_columns(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    Em& em = cx.em;
    const double M = cx.kt.total_mass, mass = cx.gp.mass;
    double a[3], o[3] = {0.0, 0.0, 0.0};
    int i;  // link whose composite / velocity is used
    if (t < NJ) { i = t + 1; for (int r = 0; r < 3; ++r) { a[r] = s.aw[t][r]; o[r] = s.ow[i][r]; } }
    else { i = 0; for (int r = 0; r < 3; ++r) a[r] = (r == t - NJ) ? 1.0 : 0.0; }
    double oxa[3], lin[3], ang[3], t1[3], t2[3], A[3];
    cross3(o, a, oxa);
    // momentum of the subtree moving with the unit motion S = (a ; o x a):  I_sub S
    cross3(a, s.ch[i], t1);
    for (int r = 0; r < 3; ++r) lin[r] = s.cm[i] * oxa[r] + t1[r];
    symvec(s.cI[i], a, t1);
    cross3(s.ch[i], oxa, t2);
    for (int r = 0; r < 3; ++r) ang[r] = t1[r] + t2[r];
    cross3(s.com, lin, t1);
    for (int r = 0; r < 3; ++r) A[r] = ang[r] - t1[r];                // column of the centroidal momentum matrix (angular rows)
    // configuration derivative:  dk = S x* k_sub - I_sub (S x v_i)
    double xw[3], xv[3], Il[3], Ia[3], dkl[3], dka[3], dcom[3], dh[3];
    cross3(a, s.wv[i], xw);
    cross3(a, s.vo[i], t1);
    cross3(oxa, s.wv[i], t2);
    for (int r = 0; r < 3; ++r) xv[r] = t1[r] + t2[r];
    cross3(xw, s.ch[i], t1);
    for (int r = 0; r < 3; ++r) Il[r] = s.cm[i] * xv[r] + t1[r];
    symvec(s.cI[i], xw, t1);
    cross3(s.ch[i], xv, t2);
    for (int r = 0; r < 3; ++r) Ia[r] = t1[r] + t2[r];
    cross3(a, s.ckl[i], t1);
    for (int r = 0; r < 3; ++r) dkl[r] = t1[r] - Il[r];
    cross3(a, s.cka[i], t1);
    cross3(oxa, s.ckl[i], t2);
    for (int r = 0; r < 3; ++r) dka[r] = t1[r] + t2[r] - Ia[r];
    for (int r = 0; r < 3; ++r) t1[r] = s.ch[i][r] - s.cm[i] * o[r];
    cross3(a, t1, dcom);
    for (int r = 0; r < 3; ++r) dcom[r] = dcom[r] / M;
    cross3(dcom, s.klin, t1);
    cross3(s.com, dkl, t2);
    for (int r = 0; r < 3; ++r) dh[r] = dka[r] - t1[r] - t2[r];
    if (t < NJ) {
        const int j = t;
        for (int r = 0; r < 3; ++r) {
            em.J(js::COMC_S + NJ * r + j, row_id(RK_COMC, 0, r), S_ + j, -dcom[r]);
            em.J(js::CMMC_S + NJ * r + j, row_id(RK_CMMC, 0, r), S_ + j, -dh[r] / mass);
            em.J(js::CMMC_SD + NJ * r + j, row_id(RK_CMMC, 0, r), SD_ + j, -A[r] / mass);
        }
        // chest-frame orientation cost: d trace = -(ax(M) . a_j) d s_j for joints on the root->chest path
        if (cx.kt.chest_pos[j] >= 0) s.grad[S_ + j] += s.chest_dc * (-dot3(s.chest_w, a));
        // feet lateral distance  y_r . (o_l - o_r)   (K4): joints of the two leg paths
        const double* yr = s.fr_R[1];  // second column of R_rsole: entries [1],[4],[7]
        const double y[3] = {yr[1], yr[4], yr[7]};
        if (cx.kt.leg_pos[0][j] >= 0) {
            double d[3], cxd[3];
            for (int r = 0; r < 3; ++r) d[r] = s.fr_o[0][r] - o[r];
            cross3(a, d, cxd);
            em.J(js::FEETD + cx.kt.leg_pos[0][j], row_id(RK_FEETD, 0, 0), S_ + j, dot3(y, cxd));
        }
        if (cx.kt.leg_pos[1][j] >= 0) {
            double d[3], e3[3], ay[3], cxd[3];
            for (int r = 0; r < 3; ++r) { d[r] = s.fr_o[0][r] - s.fr_o[1][r]; e3[r] = s.fr_o[1][r] - o[r]; }
            cross3(a, y, ay);
            cross3(a, e3, cxd);
            em.J(js::FEETD + LEG_PATH + cx.kt.leg_pos[1][j], row_id(RK_FEETD, 0, 0), S_ + j, dot3(ay, d) - dot3(y, cxd));
        }
    } else {
        const int e = t - NJ;
        for (int r = 0; r < 3; ++r) { s.dth_h[e][r] = dh[r]; s.Aw[e][r] = A[r]; }
    }
}

// ===================================================================================================
// PHASE G: row assembly.  tasks: 0..7 contact-point kinematic consistency + hdot entries of point c ;
//          8 com rows ; 9 cmm rows ; 10 feet distance g + chest grad on q_b ; 11 hdot rows ; 12,13 foot costs
// ===================================================================================================
constexpr int ASSEMBLE_TASKS = 14;
template <class Em> HD void phase_assemble(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    Em& em = cx.em;
    const double half = 0.5 * cx.gp.dt;
    const double on = cx.ki.first ? 0.0 : 1.0;
    if (t < NC) {
        const int c = t, f = c < 4 ? 0 : 1, jb = js::PT_STRIDE * c, gb = gs::PT_STRIDE * c, cb = PT_ * c;
        const double* r = s.pkin[c];  // base-centred
        // p - (p_b + pkin)   (K1 with the normalised base quaternion, planner.py:590-632)
        for (int i = 0; i < 3; ++i) {
            em.G(gb + gs::KINC + i, row_id(RK_KINC, c, i), s.x[cb + P_ + i] - (s.x[PB_ + i] + r[i]));
            em.J(jb + js::KINC_P + i, row_id(RK_KINC, c, i), cb + P_ + i, 1.0);
            em.J(jb + js::KINC_PB + i, row_id(RK_KINC, c, i), PB_ + i, -1.0);
        }
        // d pkin / d q_b = -[r]x G / |q|   ->  row entries = +[r]x G / |q|
        const double X[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        for (int i = 0; i < 3; ++i)
            for (int l = 0; l < 4; ++l)
                em.J(jb + js::KINC_QB + 4 * i + l, row_id(RK_KINC, c, i), QB_ + l,
                     (X[3 * i] * s.G[l] + X[3 * i + 1] * s.G[4 + l] + X[3 * i + 2] * s.G[8 + l]) / s.qnorm);
        // d pkin / d s_j = a_j x (pkin - o_j) for the joints of the leg path
        for (int q = 0; q < LEG_PATH; ++q) {
            const int j = cx.kt.leg_joint[f][q];
            double d[3], cxd[3];
            for (int i = 0; i < 3; ++i) d[i] = r[i] - s.ow[j + 1][i];
            cross3(s.aw[j], d, cxd);
            for (int i = 0; i < 3; ++i) em.J(jb + js::KINC_S + LEG_PATH * i + q, row_id(RK_KINC, c, i), S_ + j, -cxd[i]);
        }
        // centroidal momentum dynamics (E1): entries of this point, identical for the IN ([k]) and OUT ([k+1]) rows
        double rc[3];
        for (int i = 0; i < 3; ++i) rc[i] = s.x[cb + P_ + i] - s.x[COM_ + i];
        for (int i = 0; i < 3; ++i) {
            em.J(js::HDYN_LIN_F_IN + 3 * c + i, row_id(RK_HDYN_IN, 0, i), cb + F_ + i, -half);
            em.J(js::HDYN_LIN_F_OUT + 3 * c + i, row_id(RK_HDYN_OUT, 0, i), cb + F_ + i, -half);
        }
        for (int e = 0; e < 6; ++e) {
            const int row = 3 + cross_row(e), col = cross_col(e);
            const double vp = half * skew_entry(s.x + cb + F_, e);   // -half * d[(p-com) x f]/dp = -half * (-[f]x)
            const double vf = -half * skew_entry(rc, e);             // -half * [p-com]x
            em.J(js::HDYN_ANG_P_IN + 6 * c + e, row_id(RK_HDYN_IN, 0, row), cb + P_ + col, vp);
            em.J(js::HDYN_ANG_P_OUT + 6 * c + e, row_id(RK_HDYN_OUT, 0, row), cb + P_ + col, vp);
            em.J(js::HDYN_ANG_F_IN + 6 * c + e, row_id(RK_HDYN_IN, 0, row), cb + F_ + col, vf);
            em.J(js::HDYN_ANG_F_OUT + 6 * c + e, row_id(RK_HDYN_OUT, 0, row), cb + F_ + col, vf);
        }
        return;
    }
    switch (t - NC) {
        case 0: {  // com == CoM(pb, qn, s)   (K2, planner.py:285-306)
            for (int i = 0; i < 3; ++i) {
                em.G(gs::COMC + i, row_id(RK_COMC, 0, i), s.x[COM_ + i] - (s.x[PB_ + i] + s.com[i]));
                em.J(js::COMC_COM + i, row_id(RK_COMC, 0, i), COM_ + i, 1.0);
                em.J(js::COMC_PB + i, row_id(RK_COMC, 0, i), PB_ + i, -1.0);
            }
            const double* r = s.com;
            const double X[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
            for (int i = 0; i < 3; ++i)
                for (int l = 0; l < 4; ++l)
                    em.J(js::COMC_QB + 4 * i + l, row_id(RK_COMC, 0, i), QB_ + l,
                         (X[3 * i] * s.G[l] + X[3 * i + 1] * s.G[4 + l] + X[3 * i + 2] * s.G[8 + l]) / s.qnorm);
        } break;
        case 1: {  // h[3:] == CMM(...)[3:] / mass   (K3, planner.py:309-339)
            const double mass = cx.gp.mass;
            for (int i = 0; i < 3; ++i) {
                em.G(gs::CMMC + i, row_id(RK_CMMC, 0, i), s.x[H_ + 3 + i] - s.hang[i] / mass);
                em.J(js::CMMC_H + i, row_id(RK_CMMC, 0, i), H_ + 3 + i, 1.0);
                for (int l = 0; l < 4; ++l) {
                    double dq = 0.0, dqd = 0.0;
                    for (int e = 0; e < 3; ++e) {
                        dq += s.dth_h[e][i] * s.G[4 * e + l] / s.qnorm + s.Aw[e][i] * s.dwq[4 * e + l];
                        dqd += s.Aw[e][i] * s.G[4 * e + l];
                    }
                    em.J(js::CMMC_QB + 4 * i + l, row_id(RK_CMMC, 0, i), QB_ + l, -dq / mass);
                    em.J(js::CMMC_QD + 4 * i + l, row_id(RK_CMMC, 0, i), QD_ + l, -dqd / mass);
                }
            }
        } break;
        case 2: {  // feet distance value (K4); chest cost gradient on q_b
            const double* yr = s.fr_R[1];
            double d[3];
            for (int i = 0; i < 3; ++i) d[i] = s.fr_o[0][i] - s.fr_o[1][i];
            em.G(gs::FEETD, row_id(RK_FEETD, 0, 0), yr[1] * d[0] + yr[4] * d[1] + yr[7] * d[2]);
            for (int l = 0; l < 4; ++l) {
                double acc = 0.0;
                for (int e = 0; e < 3; ++e) acc += -s.chest_w[e] * s.G[4 * e + l];
                s.grad[QB_ + l] += s.chest_dc * acc / s.qnorm;
            }
        } break;
        case 3: {  // centroidal momentum dynamics rows (T7 on E1) and their self / com entries
            double hdot[2][6], fs[3] = {0.0, 0.0, 0.0};
            for (int w = 0; w < 2; ++w)
                for (int i = 0; i < 6; ++i) {
                    double acc = cx.gp.gravity[i];
                    for (int c = 0; c < NC; ++c) acc += s.hd[w][c][i];
                    hdot[w][i] = acc;
                }
            for (int c = 0; c < NC; ++c) for (int i = 0; i < 3; ++i) fs[i] += s.x[PT_ * c + F_ + i];
            for (int i = 0; i < 6; ++i) {
                em.G(gs::HDYN + i, row_id(RK_HDYN_IN, 0, i), s.x[H_ + i] - (s.xm[H_ + i] + half * (hdot[0][i] + hdot[1][i])));
                em.G(gs::H_X0 + i, row_id(RK_HDYN_X0, 0, i), s.x[H_ + i] - s.xg[i]);
                em.J(js::HDYN_SELF_IN + i, row_id(RK_HDYN_IN, 0, i), H_ + i, 1.0);
                em.J(js::HDYN_SELF_OUT + i, row_id(RK_HDYN_OUT, 0, i), H_ + i, -1.0);
                em.J(js::HDYN_X0 + i, row_id(RK_HDYN_X0, 0, i), H_ + i, 1.0);
                em.J(js::HDYN_X0G + i, row_id(RK_HDYN_X0, 0, i), COL_GLOBAL + i, -1.0);
            }
            for (int e = 0; e < 6; ++e) {  // d/dcom sum (p - com) x f = [sum f]x
                const double v = -half * skew_entry(fs, e);
                em.J(js::HDYN_ANG_COM_IN + e, row_id(RK_HDYN_IN, 0, 3 + cross_row(e)), COM_ + cross_col(e), v);
                em.J(js::HDYN_ANG_COM_OUT + e, row_id(RK_HDYN_OUT, 0, 3 + cross_row(e)), COM_ + cross_col(e), v);
            }
        } break;
        case 4:
        case 5: {  // foot costs (k >= 1): force-ratio regularisation and yaw alignment   planner.py:746-853
            const int foot = t - NC - 4;
            const double mf = on * cx.st.m_freg, my = on * cx.st.m_yaw;
            const double* alpha = s.pk + PK_REF + (foot == 0 ? R_ALPHA_L : R_ALPHA_R);
            double cost = 0.0;
            for (int i = 0; i < 3; ++i) {
                double sum = 0.0, e[4], ae = 0.0;
                for (int c = 0; c < 4; ++c) sum += s.x[PT_ * (4 * foot + c) + F_ + i];
                for (int c = 0; c < 4; ++c) { e[c] = s.x[PT_ * (4 * foot + c) + F_ + i] - alpha[c] * sum; cost += e[c] * e[c]; ae += alpha[c] * e[c]; }
                for (int c = 0; c < 4; ++c) s.grad[PT_ * (4 * foot + c) + F_ + i] += 2.0 * mf * (e[c] - ae);
            }
            s.cft[foot][0] = mf * cost;
            const double yaw = s.pk[PK_REF + (foot == 0 ? R_YAW_L : R_YAW_R)];
            const int br = 4 * foot + cx.st.yaw_corner[foot][0], tr = 4 * foot + cx.st.yaw_corner[foot][1], tl = 4 * foot + cx.st.yaw_corner[foot][2];
            double s1, c1, s2, c2;
            sincos(yaw, &s1, &c1);
            sincos(yaw + M_PI / 2, &s2, &c2);
            const double* pbr = s.x + PT_ * br + P_;
            const double* ptr = s.x + PT_ * tr + P_;
            const double* ptl = s.x + PT_ * tl + P_;
            const double ef = -s1 * (ptr[0] - pbr[0]) + c1 * (ptr[1] - pbr[1]);  // E9
            const double es = -s2 * (ptl[0] - ptr[0]) + c2 * (ptl[1] - ptr[1]);
            s.cft[foot][1] = my * (0.5 * (ef * ef + es * es));
            // centroid cost gradient (same vector for every point) + yaw gradient
            for (int c = 0; c < 4; ++c)
                for (int i = 0; i < 3; ++i) s.grad[PT_ * (4 * foot + c) + P_ + i] += s.cen_g[i];
            s.grad[PT_ * br + P_ + 0] += my * ef * s1;   s.grad[PT_ * br + P_ + 1] += -my * ef * c1;
            s.grad[PT_ * tr + P_ + 0] += -my * ef * s1 + my * es * s2;
            s.grad[PT_ * tr + P_ + 1] += my * ef * c1 - my * es * c2;
            s.grad[PT_ * tl + P_ + 0] += -my * es * s2;  s.grad[PT_ * tl + P_ + 1] += my * es * c2;
        } break;
        default: break;
    }
}

// ===================================================================================================
// PHASE H: horizon-end rows (final state, periodicity), lanes over rows; only at the first / last knot
// ===================================================================================================
// variable behind final-state row i (0..104), or -1 for the descriptor rows; *slot = index among the 81 variable rows
HD int final_row_var(int i, int* slot, int* desc_point, int* desc_comp) {
    *desc_point = -1; *desc_comp = 0; *slot = -1;
    if (i < 3) { *slot = i; return COM_ + i; }
    int r = i - 3;
    if (r < 72) {
        const int c = r / 9, q = r % 9;
        if (q < 3) { *desc_point = c; *desc_comp = q; return -1; }
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
// variable behind periodicity row i (0..83)
HD int periodicity_row_var(int i) {
    if (i < 48) { const int c = i / 6, q = i % 6; return PT_ * c + (q < 3 ? U_ + q : FD_ + (q - 3)); }
    if (i < 54) return H_ + (i - 48);
    if (i < 57) return VB_ + (i - 54);
    if (i < 61) return QD_ + (i - 57);
    return SD_ + (i - 61);
}
constexpr int ENDS_TASKS = 105 + 84;
template <class Em> HD void phase_ends(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    Em& em = cx.em;
    if (t < 105) {
        if (!cx.ki.last) return;
        int slot, dp, dc;
        const int var = final_row_var(t, &slot, &dp, &dc);
        const double lhs = var >= 0 ? s.x[var] : s.pk[PK_DESC + 3 * dp + dc];
        if (cx.st.final_type == HIPNLP_EXPR_MINIMIZE) {
            const double e = lhs - cx.gp.final_rhs[t];
            s.g[gs::FIN + t] = cx.st.final_weight * e * e;        // partial, reduced in phase_reduce
            if (var >= 0) s.jac[js::FIN + slot] = 2.0 * cx.st.final_weight * e;  // grad share, applied in phase_reduce
        } else {
            em.G(gs::FIN + t, row_id(RK_FIN, 0, t), lhs);
            if (var >= 0) em.J(js::FIN + slot, row_id(RK_FIN, 0, t), var, 1.0);
        }
    } else {
        const int i = t - 105;
        const int var = periodicity_row_var(i);
        if (cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE) {
            if (!cx.ki.first && !cx.ki.last) return;
            // e = x_0 - x_{N-1};  at the last knot xo = x_0, at the first knot xo = x_{N-1}
            const double e = cx.ki.last ? (s.xo[var] - s.x[var]) : (s.x[var] - s.xo[var]);
            s.g[gs::PER + i] = cx.st.periodicity_weight * e * e;
            s.jac[js::PERN + i] = 2.0 * cx.st.periodicity_weight * e;
        } else {
            if (cx.ki.last) {
                em.G(gs::PER + i, row_id(RK_PERN, 0, i), s.xo[var] - s.x[var]);
                em.J(js::PERN + i, row_id(RK_PERN, 0, i), var, -1.0);
            }
            if (cx.ki.first) em.J(js::PER0 + i, row_id(RK_PER0, 0, i), var, 1.0);
        }
    }
}

// ===================================================================================================
// PHASE I: reductions of the cost partials (lane 0) ; cost-mode end terms applied to grad (lanes 1..)
// ===================================================================================================
template <class Em> HD void phase_reduce(Ctx<Em>& cx, int t) {
    KnotScratch& s = cx.s;
    if (t == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int p = 0; p < NC; ++p) { a += s.cpt[p][0]; b += s.cpt[p][1]; c += s.cpt[p][2]; }
        s.cost[CT_SWING] = a; s.cost[CT_UREG] = b; s.cost[CT_FDREG] = c;
        double j = 0.0;
        for (int q = 0; q < NJ; ++q) j += s.cjt[q];
        s.cost[CT_JREG] = j;
        s.cost[CT_FREG] = s.cft[0][0] + s.cft[1][0];
        s.cost[CT_YAW] = s.cft[0][1] + s.cft[1][1];
        double e = 0.0;
        if (cx.ki.last && cx.st.final_type == HIPNLP_EXPR_MINIMIZE)
            for (int i = 0; i < 105; ++i) e += s.g[gs::FIN + i];
        if (cx.ki.last && cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE)
            for (int i = 0; i < 84; ++i) e += s.g[gs::PER + i];
        s.cost[CT_ENDS] = e;
    } else if (t == 1) {
        if (cx.ki.last && cx.st.final_type == HIPNLP_EXPR_MINIMIZE)
            for (int i = 0; i < 105; ++i) {
                int slot, dp, dc;
                const int var = final_row_var(i, &slot, &dp, &dc);
                if (var >= 0) s.grad[var] += s.jac[js::FIN + slot];
            }
        if ((cx.ki.last || cx.ki.first) && cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE)
            for (int i = 0; i < 84; ++i) {
                const int var = periodicity_row_var(i);
                // d/dx_0 = +2 w e ; d/dx_{N-1} = -2 w e, with e = x_0 - x_{N-1}
                s.grad[var] += cx.ki.first && !cx.ki.last ? s.jac[js::PERN + i] : -s.jac[js::PERN + i];
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// The knot program: the ordered list of phases.  RUN(fn, ntasks) / RUNL(fn, ntasks, level) are
// supplied by the caller (device: one task per thread + barrier; host: plain loops).
// ---------------------------------------------------------------------------------------------------
#define HIPNLP_KNOT_PROGRAM(RUN, RUNL, BARRIER, MAXD)                                     \
    RUN(phase_points, NC)                                                                 \
    RUN(phase_dyn, 7 + NJ + 3)                                                            \
    RUN(phase_joints, NJ)                                                                 \
    RUN(phase_misc, MISC_TASKS)                                                           \
    BARRIER                                                                               \
    for (int d_ = 1; d_ <= (MAXD); ++d_) { RUNL(phase_fk_level, NJ, d_) BARRIER }         \
    RUN(phase_links, NL + 3)                                                              \
    BARRIER                                                                               \
    for (int d_ = (MAXD) - 1; d_ >= 0; --d_) { RUNL(phase_composite_level, NL, d_) BARRIER } \
    RUN(phase_totals, NC + 1)                                                             \
    BARRIER                                                                               \
    RUN(phase_columns, NJ + 3)                                                            \
    BARRIER                                                                               \
    RUN(phase_assemble, ASSEMBLE_TASKS)                                                   \
    RUN(phase_ends, ENDS_TASKS)                                                           \
    BARRIER                                                                               \
    RUN(phase_reduce, 2)                                                                  \
    BARRIER

}  // namespace hipnlp
