// knot_hess_body.h — exact Hessian of the Lagrangian  sigma f(x) + lambda^T g(x)  of the kinodynamic multiple-shooting NLP
// (IPOPT's eval_h; SURVEY §8f rank 1: what CasADi's nlp_hess_l computes when a script drops `hessian_approximation =
// limited-memory`).  Planar terrain and the smooth-steps terrain (its contact rows: knot_hess_terrain.h).
//
// Structure.  Every constraint row and cost term depends on ONE knot, except the trapezoid defects, which are sums of a term
// in x_{k-1} and a term in x_k (implicit_trapezoid.py:24-39), and the two horizon-end costs.  The Hessian of the Lagrangian is
// therefore block diagonal by knot (189 x 189 lower triangles), plus — periodicity in `minimize` mode only — the 84 entries
// that couple the last knot with the first.  The tasks below run BEHIND the knot program of knot_body.h in the same workgroup
// and read the kinematic quantities it left in the scratch (base-centred coordinates: origin = base origin, v_b = 0).
//
// Second derivatives that exist on the planar terrain (hand-derived; checked against forward-over-forward AD of the oracle):
//   per point      tanh complementarity (p_z, p_z), (u_v, p_z); dcc margin (f_z, p_z), (f_z, v_z), (p_z, fdot_z); friction cone
//                  (f, f); the bilinear (p - com) x f of the momentum dynamics at BOTH intervals the knot belongs to:
//                  multiplier  nu = -dt/2 (lambda_k + lambda_{k+1})  on (f, p) and (com, f); quadratic regularisations
//   across points  force-ratio regularisation (f, f') on one foot; contact centroid (p, p') over all eight points; yaw
//                  alignment (p_xy, p'_xy) between the corners of one foot
//   kinematics     contact-point and com consistency rows, chest-frame cost, feet distance (K4), centroidal momentum (K3):
//                  (q_b, s, qdot_b, sdot) block, 54 variables
// Centroidal momentum  Phi = mu . L_G(q, s, qdot, sdot),  mu = -lambda / mass.  Spatial vectors [angular; linear] about the origin:
//   S_j = [a_j; o_j x a_j] joint motion vector, I^C_j / h^C_j composite inertia / momentum of the subtree of joint j, v_j the
//   velocity of its child link, l = [mu; com x mu] (rigid rotation mu about the CoM), Phi = l . h_O.  With
//     E_j = S_j x* h^C_j - I^C_j (S_j x v_j)      (= d h_O / d s_j)         G_j = I^C_j S_j   (= d h_O / d sdot_j)
//     C_j = I^C_j (S_j x l) - S_j x* (I^C_j l)                                w_j = S_j x v_j
//   for k ancestor-or-self of j:   l . d2 h_O / ds_k ds_j = -(S_k x l) . E_j + w_k . C_j
//                                  d/ds_k [l . G_j] = -(S_k x l) . G_j ;     d/ds_j [l . G_k] = -C_j . S_k
//   plus the terms through com(s) in l:  (dcom_k x mu) . dP_j + (dcom_j x mu) . dP_k + d2com_kj . (mu x P)   (every pair).
//   A rotation of the base about world axis e acts like an outermost joint with S = [e; 0], w = S x v_0; the chain to the
//   quaternion uses  dtheta = G dq / |q|  and  omega = G qdot;  the (q_b, q_b) block is taken in the body frame,
//   Phi = u^T I~(s) w + u^T A~(s) sdot  with  u = R^T mu, w = R^T omega  (tools/diag/proto_momentum_hess.py is the numpy
//   derivation these tasks were written from).
#pragma once
#include "knot_hess_terrain.h"
#include "pose_hess_body.h"

// Loops that are deliberately NOT unrolled: the Hessian tasks are straight-line code executed once per workgroup, so the kernel's
// time follows its code size (instruction fetch); bodies that only index LDS with the loop variable stay rolled.
#if defined(__HIPCC__)
#define HIPNLP_ROLLED _Pragma("clang loop unroll(disable)")
#else
#define HIPNLP_ROLLED
#endif

namespace hipnlp {

constexpr int COL_FIRST = 2 * NXK;   // column ids >= COL_FIRST address variable (id - COL_FIRST) of the FIRST knot (periodicity coupling)

namespace hk {  // native slots of the Hessian values of one knot
// per point (stride PT): rows/cols are offsets inside the knot record
constexpr int PL_U = 0;      // [2]  (u_i, p_z), i = x, y
constexpr int DC_FP = 2, DC_FV = 3, DC_PFD = 4;   // (f_z, p_z), (f_z, v_z), (p_z, fdot_z)
constexpr int FD = 5;        // [3]  (f_i, f_i)
constexpr int HD_FP = 8;     // [6]  (f_row, p_col) off-diagonal, cross_row / cross_col order
constexpr int CF = 14;       // [6]  (com_r, f_q)
constexpr int VD = 20, FDD = 23, UD = 26;   // [3] each: diagonals of v, f_dot, u_v
constexpr int PT = 29;
constexpr int FF = NC * PT;                 // [foot 2][pair 6][i 3]   (f_c', f_c), c' > c on one foot
constexpr int PP = FF + 36;                 // [pair 36][5]  (p_c', p_c), c' >= c: same coordinate x, y, z; (y, x); (x, y)
constexpr int DG = PP + 36 * 5;             // [42] diagonals of v_b 3, p_b 3, com 3, h 6, qdot_b 4, sdot 23
constexpr int QQ = DG + 42;                 // [10] lower 4x4 (q_b, q_b)
constexpr int QQD = QQ + 10;                // [4][4] row q_r, column qdot_c
constexpr int SDQ = QQD + 16;               // [NJ][4] row sdot_j, column q_l
constexpr int SQD = SDQ + 4 * NJ;           // [NJ][4] row s_j, column qdot_l
constexpr int SQ = SQD + 4 * NJ;            // [NJ][4] row s_j, column q_l
constexpr int SSD = SQ + 4 * NJ;            // [NJ][NJ] row s_k, column sdot_l
constexpr int SS = SSD + NJ * NJ;           // lower triangle (s_j, s_i), i <= j
constexpr int PERC = SS + NJ * (NJ + 1) / 2;   // [84] periodicity cost: (x_{N-1}, x_0) coupling, written by the last knot
// smooth terrain only: dense blocks of one contact point (they replace PL_U, DC_*, FD, HD_FP, VD and the same-point part of PP)
constexpr int SP = PERC + 84;
constexpr int SP_PP = 0;      // [6]  lower 3x3 (p, p)
constexpr int SP_UP = 6;      // [3][3] row u_a, column p_j
constexpr int SP_FP = 15;     // [3][3] row f_j, column p_i
constexpr int SP_FF = 24;     // [6]  lower 3x3 (f, f)
constexpr int SP_PV = 30;     // [3][3] row p_j, column v_i
constexpr int SP_FV = 39;     // [3][3] row f_j, column v_i
constexpr int SP_VV = 48;     // [6]  lower 3x3 (v, v)
constexpr int SP_PFD = 54;    // [3][3] row p_j, column fdot_i
constexpr int SP_STRIDE = 63;
constexpr int COMXY = SP + NC * SP_STRIDE;   // (com_y, com_x) of the minimum-com-height row
constexpr int COUNT = COMXY + 1;
}  // namespace hk

struct SV6 { double a[3], l[3]; };   // spatial vector [angular; linear]

// The joint pairs that are UNRELATED (neither joint on the other's path to the root), as lists: only the terms through com(s) reach them, and
// their lanes need not be found among the related ones (until round 4 a lane per pair of the full triangle / square asked `related?` and
// went idle: two of five wave iterations of the (s, s) lanes, three of nine of the (s, sdot) ones).  A property of the model: built on the
// host (kh_fill_far_lists), staged with the tables.
struct alignas(16) KHFarLists {
    int32_t n_ss, n_ssd, n_near, pad_;
    uint16_t ss[(NJ * (NJ + 1) / 2 + 7) / 8 * 8];   // (j << 8) | i, j > i
    uint16_t ssd[(NJ * NJ + 7) / 8 * 8];            // (k << 8) | l: row s_k, column sdot_l
    uint16_t near[NJ * 8];                           // the RELATED pairs: (d << 8) | k, k on the path root -> d (inclusive) — the ancestor lists without their padding
};
static_assert(sizeof(KHFarLists) % 16 == 0, "staged in 16-byte pieces");
struct alignas(16) KHessScratch {   // (16-byte alignment: the spatial vectors and the padded triples below are read as 128-bit words)
    double lam_next[3];      // multipliers of the angular momentum-dynamics rows of the NEXT interval (owned by knot k + 1)
    double sigma;
    double Y[NJ][4], Yc[NJ][4];   // Y_j (t_kh_Y: com and contact points; t_kh_Y_chest: chest cost, zero off the chest path) — consumers add the two
    // centroidal momentum: per joint j (0..NJ-1) and per base rotation axis e (NJ + e)
    SV6 S[NJ + 3], E[NJ + 3], Gm[NJ + 3], Sxl[NJ + 3], Cv[NJ + 3], Wv[NJ + 3];
    double dcmu[NJ + 3][4];  // (d com / d (s_j | theta_e)) x mu   (the com enters through l = [mu; com x mu] only)
    double mu[3], muP[3], K[3], LG[3], IG[9], ell_l[3], com[3];
    uint32_t rel[NJ + 1];    // bit i set: joint i lies on the path root -> j (inclusive)
    double qqB[16], qqg[4], qq_axE[3], qq_m2;   // (q_b, q_b): Hessian B and gradient g of Phi(qhat) = <M, R(qhat)>, chest-error axis
    double TW[3][3];         // (theta_m, omega_m') of the centroidal momentum term (t_kh_tw -> t_kh_qqd)
    double qqMw[9];          // Mw of the (q_b, q_b) block (t_kh_qq0_mw -> t_kh_qq0)
    KHFarLists far;
    double H[hk::COUNT];     // LAST: the kernel that stores its entries straight into the destination (hipnlp.hip, DIRECT) does not allocate it
};

template <class Em> struct KHCtx {
    Ctx<Em>& cx;
    KHessScratch& hx;
    const double* lam;   // [gs::COUNT] multiplier of the row a native g slot belongs to at this knot (0 for slots without a row):
                         // gathered into the g staging area of the knot scratch once the knot program is done with it
};

HD double dot6(const SV6& x, const SV6& y) { return dot3(x.a, y.a) + dot3(x.l, y.l); }
HD void crm6(const SV6& S, const SV6& x, SV6& r) {   // S x (motion)
    double t[3];
    cross3(S.a, x.a, r.a);
    cross3(S.l, x.a, r.l);
    cross3(S.a, x.l, t);
    for (int i = 0; i < 3; ++i) r.l[i] += t[i];
}
HD void crf6(const SV6& S, const SV6& f, SV6& r) {   // S x* (force)
    double t[3];
    cross3(S.a, f.a, r.a);
    cross3(S.l, f.l, t);
    for (int i = 0; i < 3; ++i) r.a[i] += t[i];
    cross3(S.a, f.l, r.l);
}
// composite spatial inertia of link i (comp[i]: m, first moment h, rotational inertia about the origin) applied to a motion vector
HD void inertia6(const double* cp, const SV6& x, SV6& r) {
    double t[3];
    symvec(cp + CI, x.a, r.a);
    cross3(cp + CH, x.l, t);
    for (int i = 0; i < 3; ++i) r.a[i] += t[i];
    cross3(cp + CH, x.a, t);
    for (int i = 0; i < 3; ++i) r.l[i] = cp[CM] * x.l[i] - t[i];
}
HD bool is_anc(const KinTables& kt, int i, int j) {   // joint i on the path root -> j (inclusive)
    static_assert(sizeof(kt.anc[0]) == 8, "the ancestor list is read as one 8-byte word");
    const unsigned long long w = *reinterpret_cast<const unsigned long long*>(kt.anc[j]);
    bool a = false;
    for (int q = 0; q < 8; ++q) a = a || (int((w >> (8 * q)) & 0xffull) == i);
    return a;
}
// row r of the packed lower triangle that holds entry t:  r (r + 1) / 2 <= t < (r + 1)(r + 2) / 2
HD int tri_row(int t) {
    int r = int((sqrtf(8.0f * float(t) + 1.0f) - 1.0f) * 0.5f);   // (single precision: an estimate the two tests below make exact; t < 2^20)
    if ((r + 1) * (r + 2) / 2 <= t) ++r;
    if (r * (r + 1) / 2 > t) --r;
    return r;
}

// diagonal share of the horizon-end costs (final state planner.py:407-425, periodicity :897-930, `minimize` mode) for variable var
template <class Em> HD double ends_diag(const KHCtx<Em>& h, int var) {
    const Ctx<Em>& cx = h.cx;
    double v = 0.0;
    if (cx.ki.last && cx.st.final_type == HIPNLP_EXPR_MINIMIZE)
        HIPNLP_ROLLED
        for (int t = 0; t < 105; ++t) if (int(end_tables(cx).fin_var[t]) == var) v += 2.0 * h.hx.sigma * cx.st.final_weight;
    if ((cx.ki.first || cx.ki.last) && cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE)
        HIPNLP_ROLLED
        for (int t = 0; t < 84; ++t) if (int(end_tables(cx).per_var[t]) == var) v += 2.0 * h.hx.sigma * cx.st.periodicity_weight;
    return v;
}

// effective multiplier of  sum_c (p_c - com) x f_c  at this knot: the trapezoid rule puts -dt/2 hdot(x_k) into the momentum
// rows of the interval that ends here and of the one that starts here
template <class Em> HD void hdyn_multiplier(const KHCtx<Em>& h, double* nu) {
    const double half = 0.5 * h.cx.gp.dt;
    for (int i = 0; i < 3; ++i) nu[i] = -half * (h.lam[gs::HDYN + 3 + i] + h.hx.lam_next[i]);
}

// own share of the force-ratio regularisation on (f_c, f_c)
template <class Em> HD double freg_diag(const KHCtx<Em>& h, int c) {
    const Ctx<Em>& cx = h.cx;
    const int foot = c >> 2, cl = c & 3;
    const double* alpha = cx.s.pk + PK_REF + (foot == 0 ? R_ALPHA_L : R_ALPHA_R);
    const double a2 = alpha[0] * alpha[0] + alpha[1] * alpha[1] + alpha[2] * alpha[2] + alpha[3] * alpha[3];
    return 2.0 * h.hx.sigma * (cx.ki.first ? 0.0 : 1.0) * cx.st.m_freg * (1.0 - 2.0 * alpha[cl] + a2);
}

// (no multiply-add contraction inside a function that starts with this: the direct-store and the staged kernel must produce the same bits,
//  tests/test_gpu_hessian_direct.py, and where a sum has a term that is zero in most workgroups the compiler contracted it differently in the two)
#if defined(__clang__)
#define HIPNLP_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define HIPNLP_NO_CONTRACT
#endif
// the horizon-end shares of several diagonal entries at once, behind ONE (workgroup-uniform) branch: an interior knot — and every knot unless
// an end group is a cost — gets zeros without a call, and the entries that add them are then emitted in one straight line (with a call of
// ends_diag, i.e. two branches and two rolled loops, between any two emissions the slot -> position reads and the stores of the direct
// emitter went one by one: 400 cycles per entry at batch)
template <class Em> HD bool ends_are_costs(const KHCtx<Em>& h) {
    const Ctx<Em>& cx = h.cx;
    return (cx.ki.last && cx.st.final_type == HIPNLP_EXPR_MINIMIZE) || ((cx.ki.first || cx.ki.last) && cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE);
}
template <int N, class Em> HD void ends_diag_n(const KHCtx<Em>& h, int var0, double* ed) {
    for (int i = 0; i < N; ++i) ed[i] = 0.0;
    if (ends_are_costs(h)) for (int i = 0; i < N; ++i) ed[i] = ends_diag(h, var0 + i);
}

// --- point-local entries: lane c (8), in four task groups (one group until round 6: 29 entries per lane one after the other — 8.2 k cycles on
//     one wave of the first phase of a batch launch, whose other waves are done after 3.1 - 4.2 k: tools/diag/hess_stamps.py) -----------------
//     a: complementarity, dcc margin, friction cone (planar terrain; the smooth terrain's dense point blocks contain these)
template <class Em> HD void t_kh_point_a(KHCtx<Em>& h, int c) {
    HIPNLP_NO_CONTRACT
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    Em& em = cx.em;
    if (!terrain_is_planar(cx)) return;
    const double* lam = h.lam;
    const int gb = gs::PT_STRIDE * c, hb = hk::PT * c, cb = PT_ * c;
    const double* x = s.x + cb;
    // planar complementarity  v_i - tanh(kt p_z) u_i  (E3)
    const double kt = cx.gp.kt, tau = knot_tanh(kt * x[P_ + 2]), dtau = kt * (1.0 - tau * tau);
    for (int i = 0; i < 2; ++i) em.H(hb + hk::PL_U + i, cb + U_ + i, cb + P_ + 2, -lam[gb + gs::PLANAR + i] * dtau);
    // dcc margin  eps - kbs p_z f_z - (v_z f_z + p_z fdot_z)  (E4)
    const double l_d = lam[gb + gs::DCC];
    em.H(hb + hk::DC_FP, cb + F_ + 2, cb + P_ + 2, -cx.gp.kbs * l_d);
    em.H(hb + hk::DC_FV, cb + F_ + 2, cb + V_ + 2, -l_d);
    em.H(hb + hk::DC_PFD, cb + P_ + 2, cb + FD_ + 2, -l_d);
    // friction cone + own share of the force-ratio regularisation
    const double dff = freg_diag(h, c);
    const double l_f = lam[gb + gs::FRICTION], mu2 = cx.gp.mu * cx.gp.mu;
    double ed[3];
    ends_diag_n<3>(h, cb + F_, ed);
    for (int i = 0; i < 3; ++i)
        em.H(hb + hk::FD + i, cb + F_ + i, cb + F_ + i, dff + (i < 2 ? -2.0 * l_f : 2.0 * mu2 * l_f) + ed[i]);
}
//     b: nu . ((p - com) x f) in (f, p), swing height (planar terrain)
template <class Em> HD void t_kh_point_b(KHCtx<Em>& h, int c) {
    HIPNLP_NO_CONTRACT
    Ctx<Em>& cx = h.cx;
    Em& em = cx.em;
    if (!terrain_is_planar(cx)) return;
    const int hb = hk::PT * c, cb = PT_ * c;
    const double on = cx.ki.first ? 0.0 : 1.0;
    double nu[3];
    hdyn_multiplier(h, nu);
    double ed[3];
    ends_diag_n<3>(h, cb + V_, ed);
    for (int e = 0; e < 6; ++e) em.H(hb + hk::HD_FP + e, cb + F_ + cross_row(e), cb + P_ + cross_col(e), skew_rc(nu, cross_row(e), cross_col(e)));
    // swing height (E10, planar)  (k >= 1)
    for (int i = 0; i < 3; ++i) em.H(hb + hk::VD + i, cb + V_ + i, cb + V_ + i, (i < 2 ? h.hx.sigma * on * cx.st.m_swing : 0.0) + ed[i]);
}
//     c: nu . ((p - com) x f) in (com, f)
template <class Em> HD void t_kh_point_c(KHCtx<Em>& h, int c) {
    Ctx<Em>& cx = h.cx;
    const int hb = hk::PT * c, cb = PT_ * c;
    double nu[3];
    hdyn_multiplier(h, nu);
    for (int e = 0; e < 6; ++e) cx.em.H(hb + hk::CF + e, COM_ + cross_row(e), cb + F_ + cross_col(e), -skew_rc(nu, cross_col(e), cross_row(e)));
}
//     d: ||u_v||^2, ||f_dot||^2   (k >= 1)
template <class Em> HD void t_kh_point_d(KHCtx<Em>& h, int c) {
    HIPNLP_NO_CONTRACT
    Ctx<Em>& cx = h.cx;
    const int hb = hk::PT * c, cb = PT_ * c;
    const double sigma = h.hx.sigma, on = cx.ki.first ? 0.0 : 1.0;
    double edf[3], edu[3];
    ends_diag_n<3>(h, cb + FD_, edf);
    ends_diag_n<3>(h, cb + U_, edu);
    for (int i = 0; i < 3; ++i) {
        cx.em.H(hb + hk::FDD + i, cb + FD_ + i, cb + FD_ + i, 2.0 * sigma * on * cx.st.m_fdreg + edf[i]);
        cx.em.H(hb + hk::UD + i, cb + U_ + i, cb + U_ + i, 2.0 * sigma * on * cx.st.m_ureg + edu[i]);
    }
}
// Where c and d run: in the first phase behind a and b (every instantiation whose recorded entry phases matter: the full-layout kernel stores
// the leading run of a knot's block behind the second barrier, HessLayout::early_run) — or, in the COMPACT kernels of the planar terrain (batch
// launches; emitters with kBatch), in the third phase on the wave the link tasks leave idle.  Exactly one group of a pair does the work.
template <class Em, class = void> struct kh_batch_t { static constexpr bool value = false; };
template <class Em> struct kh_batch_t<Em, std::void_t<decltype(Em::kBatch)>> { static constexpr bool value = Em::kBatch && Em::kTerrain == HIPNLP_TERRAIN_PLANAR; };
template <class Em> constexpr bool kh_batch = kh_batch_t<Em>::value;
// (b in the fifth phase on the wave of t_kh_joint as well: the first phase 5.3 k -> 4.4 k cycles, x 256 unchanged, x 64 83.5 -> 85.1 us: not kept)
template <class Em> HD void t_kh_point_c_e(KHCtx<Em>& h, int c) { if constexpr (!kh_batch<Em>) t_kh_point_c(h, c); }
template <class Em> HD void t_kh_point_c_l(KHCtx<Em>& h, int c) { if constexpr (kh_batch<Em>) t_kh_point_c(h, c); }
template <class Em> HD void t_kh_point_d_e(KHCtx<Em>& h, int c) { if constexpr (!kh_batch<Em>) t_kh_point_d(h, c); }
template <class Em> HD void t_kh_point_d_l(KHCtx<Em>& h, int c) { if constexpr (kh_batch<Em>) t_kh_point_d(h, c); }

// --- force-ratio regularisation across the points of one foot (planner.py:746-771): lanes (foot, pair, i) 36 ------------------
template <class Em> HD void t_kh_ff(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const int foot = t / 18, r = t - 18 * foot, pair = r / 3, i = r - 3 * pair;
    const int hi = pair < 1 ? 1 : (pair < 3 ? 2 : 3), lo = pair - (hi == 1 ? 0 : (hi == 2 ? 1 : 3));
    const double* alpha = cx.s.pk + PK_REF + (foot == 0 ? R_ALPHA_L : R_ALPHA_R);
    const double a2 = alpha[0] * alpha[0] + alpha[1] * alpha[1] + alpha[2] * alpha[2] + alpha[3] * alpha[3];
    const double on = cx.ki.first ? 0.0 : 1.0;
    cx.em.H(hk::FF + t, PT_ * (4 * foot + hi) + F_ + i, PT_ * (4 * foot + lo) + F_ + i,
            2.0 * h.hx.sigma * on * cx.st.m_freg * (a2 - alpha[hi] - alpha[lo]));
}

// --- (p_c', p_c), c' >= c: lanes (pair, e) 180.  e = 0..2 same coordinate; e = 3: (y of c', x of c); e = 4: (x of c', y of c) ----
//   contact centroid cost (planner.py:249-264): 2 m w_i / 64 for every pair;  yaw alignment (E9, :773-853) between the
//   corners of one foot;  on the diagonal: swing height, the tanh complementarity, horizon-end costs
// contact centroid + yaw alignment share of (p_hi[a], p_lo[b]); structural tells whether a cross-coordinate entry exists at all
template <class Em> HD double pp_costs(const KHCtx<Em>& h, int hi, int lo, int a, int b, bool& yaw_struct) {
    const Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const double on = cx.ki.first ? 0.0 : 1.0;
    const int foot = hi >> 2;
    double yaw = 0.0;
    yaw_struct = false;
    if ((lo >> 2) == foot && a < 2 && b < 2) {
        const int br = cx.st.yaw_corner[foot][0], tr = cx.st.yaw_corner[foot][1], tl = cx.st.yaw_corner[foot][2];
        const int ch = hi & 3, cl = lo & 3;
        const double cfh = ch == tr ? 1.0 : (ch == br ? -1.0 : 0.0), cfl = cl == tr ? 1.0 : (cl == br ? -1.0 : 0.0);   // d ef
        const double csh = ch == tl ? 1.0 : (ch == tr ? -1.0 : 0.0), csl = cl == tl ? 1.0 : (cl == tr ? -1.0 : 0.0);   // d es
        yaw_struct = (cfh != 0.0 && cfl != 0.0) || (csh != 0.0 && csl != 0.0);
        const double* sc = s.pk + PK_YAWSC + 4 * foot;
        const double d1[2] = {-sc[0], sc[1]}, d2[2] = {-sc[2], sc[3]};
        yaw = on * cx.st.m_yaw * (cfh * cfl * d1[a] * d1[b] + csh * csl * d2[a] * d2[b]);
    }
    return h.hx.sigma * (yaw + (a == b ? 2.0 * on * cx.st.m_centroid * s.pk[PK_REF + R_CW + a] / 64.0 : 0.0));
}
template <class Em> HD void t_kh_pp(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const int pair = t / 5, e = t - 5 * pair;
    const int hi = tri_row(pair);
    const int lo = pair - hi * (hi + 1) / 2;
    const double sigma = h.hx.sigma, on = cx.ki.first ? 0.0 : 1.0;
    const int a = e < 3 ? e : (e == 3 ? 1 : 0), b = e < 3 ? e : (e == 3 ? 0 : 1);   // coordinate of c' (row), of c (column)
    if (hi == lo && (e == 4 || !terrain_is_planar(cx))) return;   // upper triangle; smooth terrain: the point's own (p, p) block is dense
    bool yaw_struct;
    double v = pp_costs(h, hi, lo, a, b, yaw_struct);
    if (e >= 3) {
        if (yaw_struct) cx.em.H(hk::PP + t, PT_ * hi + P_ + a, PT_ * lo + P_ + b, v);
        return;
    }
    if (hi == lo) {
        v += ends_diag(h, PT_ * hi + P_ + e);
        if (e == 2) {
            const double* x = s.x + PT_ * hi;
            const double kt = cx.gp.kt, tau = knot_tanh(kt * x[P_ + 2]), dtau = kt * (1.0 - tau * tau), ddtau = -2.0 * kt * tau * dtau;
            const double* lp = h.lam + gs::PT_STRIDE * hi + gs::PLANAR;
            v += sigma * on * cx.st.m_swing - (lp[0] * x[U_] + lp[1] * x[U_ + 1]) * ddtau;
        }
    }
    cx.em.H(hk::PP + t, PT_ * hi + P_ + e, PT_ * lo + P_ + e, v);
}

// Placement of the 180 (p, p) lanes (three wave iterations of ~2 k cycles, no input but the knot record and the multipliers): where the
// terrain's program leaves waves idle.  Planar: beside the forward kinematics (second phase, waves 1 and 2) and on wave 3 of the first
// phase; smooth steps: behind the forward kinematics on waves 0 and 3, whose phase lasts as long as the dense point tasks of waves
// 1 and 2.  Every lane range appears in both placements; exactly one of them does the work (as t_hdyn_entries_a / t_hdyn_entries).
template <int BEGIN, class Em> HD void t_kh_pp_planar_at(KHCtx<Em>& h, int t) { if (terrain_is_planar(h.cx)) t_kh_pp(h, t + BEGIN); }
template <int BEGIN, class Em> HD void t_kh_pp_smooth_at(KHCtx<Em>& h, int t) { if (!terrain_is_planar(h.cx)) t_kh_pp(h, t + BEGIN); }

// --- smooth terrain: the dense blocks of one contact point.  Lagrangian of the point with f, v, f_dot, u_v constant:
//       L = C_u . u + l_d (-kbs h nf - hdot nf - h (ndot . f) - h (n . fdot)) + l_h h + l_n nf + l_f (mu^2 nf^2 - (x.f)^2 - (y.f)^2)
//           + sigma m_sw/2 ((h - h_d)^2 + (x.v)^2 + (y.v)^2),      C_u = -sum_i l_pl,i (x_i tau, y_i tau, n_i),  tau = tanh(kt h)
//     with n, x, y the terrain frame, nf = n.f, hdot = grad h . v, ndot = (dn/dp) v.
// Two tasks on two waves (round 1 had ONE task per point that built the whole frame as nineteen second-order jets — 190 doubles live
// next to the truncated-Taylor arithmetic: 256 VGPRs plus ~100 spilled to scratch memory — and assembled everything from them):
//   t_kh_point_smooth_pp     lane c: the (p, p) block = Hessian of L.  Every term of L contracts the frame with a CONSTANT vector
//                            (f, v, f_dot, the multipliers): the contractions are taken while the frame still is a set of
//                            two-variable jets (J2<3>: linear combinations, and x.c = iq (q c_0 - n_0 (n_1 c_1 + n_2 c_2)),
//                            y.c = iq (n_2 c_1 - n_1 c_2): three products for a pair instead of nine), so only thirteen
//                            second-order jets ever exist.  Needs Z to fourth order (ndot . f = v . grad (n . f)).
//   t_kh_point_smooth_mixed  lane c: the mixed blocks (u, p), (f, p), (f, f), (p, f_dot), (p, v), (f, v), (v, v): GRADIENTS of
//                            dL/du, dL/df, dL/df_dot, dL/dv — first-order jets (G3) over a frame one order lower (Z to third order).
// (Measured in round 1: staging the single task over (point, bump) / point / (point, block) lanes through the dead jac area ran no
//  faster — the arithmetic of the frame was the long pole, not the assembly; what changed here is the arithmetic itself.)
constexpr int PP_STAGE = 34;   // doubles handed from t_kh_point_smooth_pp to _pp2 per contact point: the normal to SECOND order (3 x 6), the height jet (10), the Hessian of L so far (6)
// The bump jets, ONCE per (point, bump): lane (c, bump) of the first phase — the eight contact points and, as a ninth point, the com
// (minimum com height row) — evaluates  H exp(-g^r)  to FOURTH order and parks the fifteen scaled coefficients in the staging area;
// the three consumers of the next phase add the parts: t_kh_point_smooth_pp to fourth order, t_kh_point_smooth_mixed to
// third (a prefix: the coefficients are stored by total degree), the com lanes of t_kh_diag to second.  (Until round 4 each of the
// three evaluated every bump for itself, one after the other on its own lane: the powers, the exponential and the compositions of the
// bumps were 9 - 10 k of the 17 k cycles of either point task and 5 of the 6 k of the diagonal task.)  The lanes of unused bumps store
// zeros: the sums run over a fixed number of parts (adding a zero part changes no bit).
constexpr int KH_BUMP_TASKS = (NC + 1) * HIPNLP_MAX_TERRAIN_STEPS, KH_BUMP_COEF = J2<4>::NC_;
// Both staging areas are dead before the kinematic Hessian tasks start: they lie on the per-joint spatial vectors of the Hessian scratch
// (S .. Wv: written by t_kh_joint in the fifth phase), so the smooth terrain needs no LDS beyond the planar program's.
constexpr int KH_STAGE_DOUBLES = 6 * 6 * (NJ + 3);
static_assert(offsetof(KHessScratch, Wv) + sizeof(KHessScratch::Wv) - offsetof(KHessScratch, S) == sizeof(double) * KH_STAGE_DOUBLES, "S, E, Gm, Sxl, Cv, Wv are contiguous");
static_assert(PP_STAGE * NC + KH_BUMP_COEF * KH_BUMP_TASKS <= KH_STAGE_DOUBLES, "(p, p) staging and bump jets live on the spatial vectors of the later phases");
HD double* pp_stage(KHessScratch& hx, int c) { return reinterpret_cast<double*>(&hx.S[0]) + PP_STAGE * c; }
// (the six tangent-axis terms of the (p, p) block wait in the composite area of the knot scratch, which the kinematic program writes two
//  phases later)
static_assert(6 * 6 * NC <= NL * LSTR, "tangent-axis terms of the (p, p) blocks fit in comp[]");
template <class S> HD double* pp_term_stage(S& s, int c, int term) { return &s.comp[0][0] + 6 * (6 * c + term); }
HD double* kh_bump_part(KHessScratch& hx, int c, int sidx) { return reinterpret_cast<double*>(&hx.S[0]) + PP_STAGE * NC + KH_BUMP_COEF * (HIPNLP_MAX_TERRAIN_STEPS * c + sidx); }
template <class Em> HD void t_kh_bump(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    const int c = t / HIPNLP_MAX_TERRAIN_STEPS, sidx = t - HIPNLP_MAX_TERRAIN_STEPS * c;
    const double* p = c < NC ? s.x + PT_ * c + P_ : s.x + COM_;
    J2<4> bump;
    if (sidx < cx.st.n_steps) {
        terrain_bump_j<4>(cx.st.steps[sidx], p[0], p[1], bump, &cx.gkt->tops.px[sidx], &cx.gkt->tops.py[sidx]);   // (false: the bump has vanished, its jet stays zero)
        bump.c[0] += cx.st.steps[sidx].oz;
    }
    double* out = kh_bump_part(h.hx, c, sidx);
    for (int i = 0; i < KH_BUMP_COEF; ++i) out[i] = bump.c[i];
}
// Z(p_x, p_y) of the terrain at point c (NC: the com) to order K from the parts t_kh_bump left (same order of the bumps as terrain_Z_jet)
template <int K> HD J2<K> kh_terrain_Z(KHessScratch& hx, int c, int n_steps) {
    J2<K> Z;
    HIPNLP_ROLLED for (int sidx = 0; sidx < n_steps; ++sidx) {
        const double* part = kh_bump_part(hx, c, sidx);
        for (int i = 0; i < J2<K>::NC_; ++i) Z.c[i] += part[i];
    }
    return Z;
}
template <class Em> HD void t_kh_point_smooth_pp(KHCtx<Em>& h, int c) {
    Ctx<Em>& cx = h.cx;
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    const double* lam = h.lam;
    const int gb = gs::PT_STRIDE * c, cb = PT_ * c;
    const double* x = s.x + cb;
    const double* p = x + P_;
    const double* f = x + F_;
    const double* v = x + V_;
    const double* fd = x + FD_;
    const double* u = x + U_;
    const double on = cx.ki.first ? 0.0 : 1.0, msw = h.hx.sigma * on * cx.st.m_swing;
    const double l_d = lam[gb + gs::DCC], l_h = lam[gb + gs::HEIGHT], l_n = lam[gb + gs::NORMAL], l_f = lam[gb + gs::FRICTION];
    const double* lp = lam + gb + gs::PLANAR;
    const double kbs = cx.gp.kbs, mu2 = cx.gp.mu * cx.gp.mu;
    // L is ACCUMULATED term by term: every contraction leaves the two-variable jets as a second-order jet (six numbers instead of ten;
    // only n . f is needed to third order, for its gradient), goes into L at once and is dropped.  This task: the terms in h, hdot
    // and the normal; t_kh_point_smooth_pp2: the terms in the tangent axes.
    T3 L;
    {
        const J2<4> Z = kh_terrain_Z<4>(h.hx, c, cx.st.n_steps);
        const J2<3> u1 = -j2_dx(Z), u2 = -j2_dy(Z);   // grad h = (u1, u2, 1)
        T3 hT = t3_from(-j2_trunc<4, 2>(Z));
        hT.v += p[2]; hT.g[2] = 1.0;
        T3 hdot = t3_from(u1 * v[0] + u2 * v[1]);
        hdot.v += v[2];
        T3 dh = hT;
        dh.v -= s.pk[PK_REF + R_SWING];
        L = hT * l_h + (dh * dh) * (0.5 * msw);
        const J2<3> n2 = j2_rsqrt(J2<3>(1.0) + u1 * u1 + u2 * u2);
        const J2<3> n0 = u1 * n2, n1 = u2 * n2;
        {
            const J2<3> nf = n0 * f[0] + n1 * f[1] + n2 * f[2];
            const T3 nfT = t3_from(nf);
            const T3 fnd = t3_from(j2_dx(nf) * v[0] + j2_dy(nf) * v[1]);   // (dn/dp v) . f = v . grad (n . f)
            const T3 nfdT = t3_from(n0 * fd[0] + n1 * fd[1] + n2 * fd[2]);
            L = L + (hT * (nfT * kbs + fnd + nfdT) + hdot * nfT) * (-l_d) + nfT * l_n + (nfT * nfT) * (mu2 * l_f);
        }
        L = L + t3_from(n0 * lp[0] + n1 * lp[1] + n2 * lp[2]) * (-u[2]);
        // handed to t_kh_point_smooth_pp2 (next phase, another wave): the normal to second order, the height jet and the Hessian of L so
        // far (pp_stage)
        double* st = pp_stage(h.hx, c);
        for (int i = 0; i < 6; ++i) { st[i] = n0.c[i]; st[6 + i] = n1.c[i]; st[12 + i] = n2.c[i]; }
        st[18] = hT.v;
        for (int i = 0; i < 3; ++i) st[19 + i] = hT.g[i];
        for (int i = 0; i < 6; ++i) { st[22 + i] = hT.H[i]; st[28 + i] = L.H[i]; }
    }
}
// second half of the (p, p) block: the six terms that contract the TANGENT axes x, y of the terrain frame with f, v and the planar
// multipliers.  A task of its own so that the jets of the first half (Z to fourth order, grad h, hdot, the n . f family) and the
// tangent-axis jets (q, iq, the contractions) are never live together: as one task the kernel spilled 34 VGPRs at the 256 cap.
template <class Em> HD void t_kh_point_smooth_pp2(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    const double* lam = h.lam;
    HIPNLP_WAVE_SYNC();   // the first half (t_kh_point_smooth_pp) ran on this wave
    const int c = t / 6, term = t - 6 * c;     // term: x.f, y.f, x.v, y.v, x.lp, y.lp
    const int gb = gs::PT_STRIDE * c, cb = PT_ * c;
    const double* x = s.x + cb;
    const double on = cx.ki.first ? 0.0 : 1.0, msw = h.hx.sigma * on * cx.st.m_swing;
    const double* cv = term < 2 ? x + F_ : (term < 4 ? x + V_ : lam + gb + gs::PLANAR);   // the constant vector the axis is contracted with
    const double c0 = cv[0], c1 = cv[1], c2 = cv[2];
    const double scale = term < 2 ? -lam[gb + gs::FRICTION] : (term < 4 ? 0.5 * msw : -x[U_ + (term - 4)]);
    const double kt = cx.gp.kt;
    const double* st = pp_stage(h.hx, c);
    // every term ends in t3_from, which reads a jet up to second order: the whole task runs on second-order jets (a product of two of
    // them is 15 multiply-adds where the third-order one of the first half takes 35)
    J2<2> n0, n1, n2;
    T3 hT;
    for (int i = 0; i < 6; ++i) { n0.c[i] = st[i]; n1.c[i] = st[6 + i]; n2.c[i] = st[12 + i]; }
    hT.v = st[18];
    for (int i = 0; i < 3; ++i) hT.g[i] = st[19 + i];
    for (int i = 0; i < 6; ++i) hT.H[i] = st[22 + i];
    const J2<2> q = n1 * n1 + n2 * n2;            // same closed form as terrain_frame (knot_body.h)
    const J2<2> iq = j2_rsqrt(q);
    // x . c = iq (q c_0 - n_0 (n_1 c_1 + n_2 c_2)),   y . c = iq (n_2 c_1 - n_1 c_2)
    J2<2> w;
    if ((term & 1) == 0) w = q * c0 - n0 * (n1 * c1 + n2 * c2);
    else w = n2 * c1 - n1 * c2;
    const T3 axis = t3_from(iq * w);
    T3 other = axis;                               // friction cone and swing-height cost: the square; planar rows: times tanh(kt h)
    if (term >= 4) {
        const double tv = knot_tanh(kt * hT.v), t1 = kt * (1.0 - tv * tv), t2 = -2.0 * kt * tv * t1;
        other = t3_chain(hT, tv, t1, t2);
    }
    const T3 L = (axis * other) * scale;
    double* out = pp_term_stage(s, c, term);
    for (int i = 0; i < 6; ++i) out[i] = L.H[i];
}
// the (p, p) block of point c leaves: lane (c, entry) adds the two halves' shares, the cost terms and the horizon-end diagonal
template <class Em> HD void t_kh_point_smooth_pp_emit(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    if (terrain_is_planar(cx)) return;
    const int c = t / 6, e = t - 6 * c;
    const int a = e < 1 ? 0 : (e < 3 ? 1 : 2), b = e - a * (a + 1) / 2;   // e = tri(a, b), b <= a
    const int hb = hk::SP + hk::SP_STRIDE * c, cb = PT_ * c;
    const int hidx = t3h(a, b);   // xx, xy, xz, yy, yz, zz
    double v = pp_stage(h.hx, c)[28 + hidx];
    for (int term = 0; term < 6; ++term) v += pp_term_stage(cx.s, c, term)[hidx];
    bool ys;
    v += pp_costs(h, c, c, a, b, ys);
    if (a == b) v += ends_diag(h, cb + P_ + a);
    cx.em.H(hb + hk::SP_PP + e, cb + P_ + a, cb + P_ + b, v);
}
// Lane (c, r), 6 x NC lanes: every lane builds the frame of its point (first-order jets: the common prefix, 3.2 k cycles) and emits ONE
// slice of the blocks — r < 3: row j = r of the (u, p), (f, p), (f, f), (p, f_dot) blocks; r >= 3: column i = r - 3 of the (p, v), (f, v),
// (v, v) blocks.  (As one lane per point the three rows and the three columns followed one another: 3.2 k more cycles on the longest
// chain of the phase.)
// (selects of NUMBERS passed by value: `c ? a.v : b.v` on two lvalues selects between their addresses and loads once, which sends the
//  whole frame to scratch memory)
HD double sel_d(bool c, double a, double b) { return c ? a : b; }
HD G3 g3_pick2(const G3& a, const G3& b, bool first) {
    G3 r;
    r.v = sel_d(first, a.v, b.v);
    for (int k = 0; k < 3; ++k) r.g[k] = sel_d(first, a.g[k], b.g[k]);
    return r;
}
HD G3 g3_pick(const G3* a, int j) { return g3_pick2(a[0], g3_pick2(a[1], a[2], j == 1), j == 0); }
template <class Em> HD void t_kh_point_smooth_mixed(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    if (terrain_is_planar(cx)) return;
    const int c = t / 6, r = t - 6 * c;
    auto& s = cx.s;
    Em& em = cx.em;
    const double* lam = h.lam;
    const int gb = gs::PT_STRIDE * c, hb = hk::SP + hk::SP_STRIDE * c, cb = PT_ * c;
    const double* x = s.x + cb;
    const double* p = x + P_;
    const double* f = x + F_;
    const double* v = x + V_;
    const double* lp = lam + gb + gs::PLANAR;
    G3 hG, gh[3], n[3], xv[3], yv[3], dn0[3], dn1[3];   // dn0[j] = d n_j / d p_x, dn1[j] = d n_j / d p_y
    {
        const J2<3> Z = kh_terrain_Z<3>(h.hx, c, cx.st.n_steps);
        const J2<2> u1 = -j2_dx(Z), u2 = -j2_dy(Z);
        hG.v = p[2] - Z.c[0]; hG.g[0] = u1.c[0]; hG.g[1] = u2.c[0]; hG.g[2] = 1.0;
        gh[0] = g3_from(u1); gh[1] = g3_from(u2); gh[2] = G3(1.0);
        const J2<2> n2 = j2_rsqrt(J2<2>(1.0) + u1 * u1 + u2 * u2);
        const J2<2> n0 = u1 * n2, n1 = u2 * n2;
        n[0] = g3_from(n0); n[1] = g3_from(n1); n[2] = g3_from(n2);
        dn0[0] = g3_from(j2_dx(n0)); dn1[0] = g3_from(j2_dy(n0));
        dn0[1] = g3_from(j2_dx(n1)); dn1[1] = g3_from(j2_dy(n1));
        dn0[2] = g3_from(j2_dx(n2)); dn1[2] = g3_from(j2_dy(n2));
        // the tangent axes are needed to first order only
        const J2<1> m0 = j2_trunc<2, 1>(n0), m1 = j2_trunc<2, 1>(n1), m2 = j2_trunc<2, 1>(n2);
        const J2<1> q = m1 * m1 + m2 * m2;
        const J2<1> iq = j2_rsqrt(q);
        xv[0] = g3_from(q * iq); xv[1] = g3_from(-(m1 * m0) * iq); xv[2] = g3_from(-(m2 * m0) * iq);
        yv[0] = G3(0.0); yv[1] = g3_from(m2 * iq); yv[2] = g3_from(-(m1 * iq));
    }
    // (multipliers and gains are read behind the jets: the frame is what fills the registers)
    const double on = cx.ki.first ? 0.0 : 1.0, msw = h.hx.sigma * on * cx.st.m_swing;
    const double l_d = lam[gb + gs::DCC], l_n = lam[gb + gs::NORMAL], l_f = lam[gb + gs::FRICTION];
    const double kbs = cx.gp.kbs, kt = cx.gp.kt, mu2 = cx.gp.mu * cx.gp.mu;
    const double tv = knot_tanh(kt * hG.v), t1 = kt * (1.0 - tv * tv);
    G3 tau;
    tau.v = tv;
    for (int i = 0; i < 3; ++i) tau.g[i] = t1 * hG.g[i];
    const G3 nf = n[0] * f[0] + n[1] * f[1] + n[2] * f[2];
    const G3 xf = xv[0] * f[0] + xv[1] * f[1] + xv[2] * f[2], yf = yv[1] * f[1] + yv[2] * f[2];
    const G3 xvv = xv[0] * v[0] + xv[1] * v[1] + xv[2] * v[2], yvv = yv[1] * v[1] + yv[2] * v[2];
    const G3 hdot = gh[0] * v[0] + gh[1] * v[1] + gh[2] * v[2];
    double nu[3];
    hdyn_multiplier(h, nu);
    const double dff = freg_diag(h, c);
    if (r < 3) {
        const int j = r;
        const G3 nj = g3_pick(n, j), xvj = g3_pick(xv, j), yvj = g3_pick(yv, j);
        // coefficient of u_j in L: C_u,j = -(x . l_pl) tau, -(y . l_pl) tau, -(n . l_pl)
        G3 ax[3];
        HIPNLP_UNROLL for (int k = 0; k < 3; ++k) ax[k] = g3_pick2(xv[k], g3_pick2(yv[k], n[k], j == 1), j == 0);
        const G3 al = ax[0] * lp[0] + ax[1] * lp[1] + ax[2] * lp[2];
        const G3 Cu = g3_pick2(al * tau, al, j < 2) * -1.0;
        HIPNLP_UNROLL for (int i = 0; i < 3; ++i) em.H(hb + hk::SP_UP + 3 * j + i, cb + U_ + j, cb + P_ + i, Cu.g[i]);
        const G3 nd = g3_pick(dn0, j) * v[0] + g3_pick(dn1, j) * v[1];   // ndot_j = sum_i dn_j/dp_i v_i
        const G3 dLdf = (hG * nj * kbs + hdot * nj + hG * nd) * (-l_d) + nj * l_n + (nf * nj * mu2 - xf * xvj - yf * yvj) * (2.0 * l_f);
        HIPNLP_UNROLL for (int i = 0; i < 3; ++i) em.H(hb + hk::SP_FP + 3 * j + i, cb + F_ + j, cb + P_ + i, dLdf.g[i] + skew_rc(nu, j, i));
        HIPNLP_UNROLL for (int i = 0; i < 3; ++i)
            if (i <= j)
                em.H(hb + hk::SP_FF + tri(j, i), cb + F_ + j, cb + F_ + i,
                     2.0 * l_f * (mu2 * nj.v * n[i].v - xvj.v * xv[i].v - yvj.v * yv[i].v) + (i == j ? dff + ends_diag(h, cb + F_ + j) : 0.0));
        const G3 dLdfd = hG * nj * (-l_d);
        HIPNLP_UNROLL for (int i = 0; i < 3; ++i) em.H(hb + hk::SP_PFD + 3 * i + j, cb + P_ + i, cb + FD_ + j, dLdfd.g[i]);
    } else {
        const int i = r - 3;
        // dL/dv_i = l_d (-gh_i nf - h sum_j dn_j/dp_i f_j) + m_sw ((x.v) x_i + (y.v) y_i)
        G3 dnf = G3(0.0);
        if (i < 2) dnf = g3_pick2(dn0[0], dn1[0], i == 0) * f[0] + g3_pick2(dn0[1], dn1[1], i == 0) * f[1] + g3_pick2(dn0[2], dn1[2], i == 0) * f[2];
        const G3 ghi = g3_pick(gh, i), xvi = g3_pick(xv, i), yvi = g3_pick(yv, i);
        const G3 dLdv = (ghi * nf + hG * dnf) * (-l_d) + (xvv * xvi + yvv * yvi) * msw;
        HIPNLP_UNROLL for (int j = 0; j < 3; ++j) em.H(hb + hk::SP_PV + 3 * j + i, cb + P_ + j, cb + V_ + i, dLdv.g[j]);
        HIPNLP_UNROLL for (int j = 0; j < 3; ++j)
            em.H(hb + hk::SP_FV + 3 * j + i, cb + F_ + j, cb + V_ + i, -l_d * (ghi.v * n[j].v + hG.v * (i < 2 ? sel_d(i == 0, dn0[j].v, dn1[j].v) : 0.0)));
        HIPNLP_UNROLL for (int j = 0; j < 3; ++j)
            if (j <= i)
                em.H(hb + hk::SP_VV + tri(i, j), cb + V_ + i, cb + V_ + j, msw * (xvi.v * xv[j].v + yvi.v * yv[j].v) + (i == j ? ends_diag(h, cb + V_ + i) : 0.0));
    }
}

// --- remaining diagonals: lanes 42: v_b 3, p_b 3, com 3, h 6, qdot_b 4, sdot 23 ----------------------------------------------------
template <class Em> HD void t_kh_diag(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const double sigma = h.hx.sigma, on = cx.ki.first ? 0.0 : 1.0;
    int var;
    double v = 0.0;
    if (t < 3) var = VB_ + t;
    else if (t < 6) var = PB_ + (t - 3);
    else if (t < 9) {
        var = COM_ + (t - 6);
        if (!terrain_is_planar(cx) && t < 8) {   // minimum com height  com_z - Z(com_x, com_y)  (planner.py:353-358)
            const J2<2> Z = kh_terrain_Z<2>(h.hx, NC, cx.st.n_steps);   // (scaled coefficients: Z_xx = 2 c_20, Z_xy = c_11, Z_yy = 2 c_02)
            v = -h.lam[gs::COMH] * 2.0 * (t == 6 ? Z.c[J2<2>::idx(2, 0)] : Z.c[J2<2>::idx(0, 2)]);
            if (t == 7) cx.em.H(hk::COMXY, COM_ + 1, COM_, -h.lam[gs::COMH] * Z.c[J2<2>::idx(1, 1)]);
        }
    }
    else if (t < 15) { var = H_ + (t - 9); if (t < 12) v = 2.0 * sigma * cx.st.m_comvel * cx.st.w_comvel[t - 9]; }   // com velocity cost (k >= 0)
    else if (t < 19) { var = QD_ + (t - 15); v = 2.0 * sigma * cx.st.m_baseqv; }                                      // base quaternion velocity cost
    else { var = SD_ + (t - 19); v = 2.0 * sigma * on * cx.st.m_jreg * (cx.st.joint_reg_as_coded ? double(NJ) : 1.0); }   // J6
    cx.em.H(hk::DG + t, var, var, v + ends_diag(h, var));
}
// the diagonal / com-height lanes: 0.9 k cycles on the planar terrain, 4.7 k on the smooth one (the com's bump jets) — there they
// run behind the forward kinematics of the second phase, like the (p, p) lanes
template <class Em> HD void t_kh_diag_planar(KHCtx<Em>& h, int t) { if (terrain_is_planar(h.cx)) t_kh_diag(h, t); }
template <class Em> HD void t_kh_diag_smooth(KHCtx<Em>& h, int t) { if (!terrain_is_planar(h.cx)) t_kh_diag(h, t); }

// --- periodicity cost, coupling of the last knot with the first: lanes 84 (last knot only) -------------------------------------
template <class Em> HD void t_kh_percouple(KHCtx<Em>& h, int i) {
    Ctx<Em>& cx = h.cx;
    if (cx.st.periodicity_type != HIPNLP_EXPR_MINIMIZE || !cx.ki.last) return;
    cx.em.H(hk::PERC + i, int(end_tables(cx).per_var[i]), COL_FIRST + int(end_tables(cx).per_var[i]), -2.0 * h.hx.sigma * cx.st.periodicity_weight);
}

// --- centroidal momentum, shared quantities: lane 0 ---------------------------------------------------------------------------------
template <class Em> HD void t_kh_mom0(KHCtx<Em>& h, int) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    KHessScratch& hx = h.hx;
    const double* c0 = s.comp[0];
    const double M = 1.0 / cx.kt.inv_total_mass;
    for (int i = 0; i < 3; ++i) hx.mu[i] = -h.lam[gs::CMMC + i] / cx.gp.mass;
    double com[3];
    for (int i = 0; i < 3; ++i) { com[i] = c0[CH + i] * cx.kt.inv_total_mass; hx.com[i] = com[i]; }
    cross3(com, hx.mu, hx.ell_l);
    cross3(hx.mu, c0 + CKL, hx.muP);
    // L_G = L_O - com x P ;  I_G = I_O - M (|c|^2 1 - c c^T) ;  K = I_G mu
    double t[3];
    cross3(com, c0 + CKL, t);
    for (int i = 0; i < 3; ++i) hx.LG[i] = c0[CKA + i] - t[i];
    const double c2 = dot3(com, com);
    const double* I6 = c0 + CI;
    const double IO[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) hx.IG[3 * r + q] = IO[3 * r + q] - M * ((r == q ? c2 : 0.0) - com[r] * com[q]);
    matvec3(hx.IG, hx.mu, hx.K);
}

// --- centroidal momentum, per joint / base axis: lanes NJ + 3, two task groups on two waves (one group formed all six vectors in
//     round 2: 3.4 k cycles of 26 lanes, next to an idle wave): t_kh_joint S, W = S x v, E;  t_kh_joint_b G = I^C S, S x l, C, dcom x mu.
template <class Em> HD void kh_joint_axis(KHCtx<Em>& h, int t, SV6& S) {   // (selects, no branch: S stays in registers)
    auto& s = h.cx.s;
    const bool joint = t < NJ;
    const int j = joint ? t : 0;
    double l[3];
    cross3(s.ow[j + 1], s.aw[j], l);
    for (int i = 0; i < 3; ++i) { S.a[i] = joint ? s.aw[j][i] : ((i == t - NJ) ? 1.0 : 0.0); S.l[i] = joint ? l[i] : 0.0; }
}
template <class Em> HD void t_kh_joint(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    KHessScratch& hx = h.hx;
    SV6 S, v, hC, t1, t2;
    const int link = t < NJ ? t + 1 : 0;
    if (t < NJ) {
        uint32_t mask = 0u;
        for (int q = 0; q < 8; ++q) { const int a = int(cx.kt.anc[t][q]); if (a < NJ) mask |= 1u << unsigned(a); }
        hx.rel[t] = mask;
    }
    kh_joint_axis(h, t, S);
    const double* cp = s.comp[link];
    for (int i = 0; i < 3; ++i) { v.a[i] = s.wv[link][i]; v.l[i] = s.vo[link][i]; hC.a[i] = cp[CKA + i]; hC.l[i] = cp[CKL + i]; }
    hx.S[t] = S;
    crm6(S, v, hx.Wv[t]);                 // w = S x v
    crf6(S, hC, t1);
    inertia6(cp, hx.Wv[t], t2);
    for (int i = 0; i < 3; ++i) { hx.E[t].a[i] = t1.a[i] - t2.a[i]; hx.E[t].l[i] = t1.l[i] - t2.l[i]; }
}
template <class Em> HD void t_kh_joint_b(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    KHessScratch& hx = h.hx;
    SV6 S, ell, t1, t2;
    const int link = t < NJ ? t + 1 : 0;
    kh_joint_axis(h, t, S);
    const double* cp = s.comp[link];
    // l = [mu; com x mu], formed by every lane itself (t_kh_mom0 runs beside this task, not before it)
    double mu[3], comv[3];
    for (int i = 0; i < 3; ++i) { mu[i] = -h.lam[gs::CMMC + i] / cx.gp.mass; comv[i] = s.comp[0][CH + i] * cx.kt.inv_total_mass; }
    cross3(comv, mu, ell.l);
    for (int i = 0; i < 3; ++i) ell.a[i] = mu[i];
    inertia6(cp, S, hx.Gm[t]);
    crm6(S, ell, hx.Sxl[t]);
    inertia6(cp, hx.Sxl[t], t1);          // B = I^C (S x l)
    inertia6(cp, ell, t2);
    SV6 t3;
    crf6(S, t2, t3);
    for (int i = 0; i < 3; ++i) { hx.Cv[t].a[i] = t1.a[i] - t3.a[i]; hx.Cv[t].l[i] = t1.l[i] - t3.l[i]; }
    double dcv[3];
    for (int i = 0; i < 3; ++i) dcv[i] = hx.Gm[t].l[i] * cx.kt.inv_total_mass;
    cross3(dcv, mu, hx.dcmu[t]);
}

// --- Y_j = d/d theta [ dL/ds_j ]  (theta: world-frame rotation of the base) for the part of the Lagrangian that is LINEAR in
//     points rigidly attached to links: contact points, com (incl. the com inside the momentum term: weight mu x P), chest cost.
//     Same construction as t_hess_Y_a / _b of the pose finder.  lane j (23) -----------------------------------------------------------------
template <class Em> HD void t_kh_Y(KHCtx<Em>& h, int j) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const double* lam = h.lam;
    const double* a = s.aw[j];
    const double* o = s.ow[j + 1];
    double Y[3] = {0.0, 0.0, 0.0}, t1[3], t2[3];
    {
        const double* cp = s.comp[j + 1];
        const double inv_M = cx.kt.inv_total_mass;
        for (int r = 0; r < 3; ++r) t1[r] = (cp[CH + r] - cp[CM] * o[r]) * inv_M;
        cross3(a, t1, t2);
        double mu[3], muP[3];   // mu x P, formed here (t_kh_mom0 runs beside this task)
        for (int r = 0; r < 3; ++r) mu[r] = -lam[gs::CMMC + r] / cx.gp.mass;
        cross3(mu, s.comp[0] + CKL, muP);
        const double w[3] = {-lam[gs::COMC] + muP[0], -lam[gs::COMC + 1] + muP[1], -lam[gs::COMC + 2] + muP[2]};
        cross3(t2, w, t1);
        for (int r = 0; r < 3; ++r) Y[r] += t1[r];
    }
    for (int foot = 0; foot < 2; ++foot) {
        if (cx.kt.leg_pos[foot][j] < 0) continue;
        HIPNLP_ROLLED
        for (int c = 4 * foot; c < 4 * foot + 4; ++c) {
            for (int r = 0; r < 3; ++r) t1[r] = s.pkin[c][r] - o[r];
            cross3(a, t1, t2);
            const double* lk = lam + gs::PT_STRIDE * c + gs::KINC;
            const double w[3] = {-lk[0], -lk[1], -lk[2]};
            cross3(t2, w, t1);
            for (int r = 0; r < 3; ++r) Y[r] += t1[r];
        }
    }
    for (int r = 0; r < 3; ++r) h.hx.Y[j][r] = Y[r];
}
// the chest-cost part of Y_j (joints on the root -> chest path; zero elsewhere): lane j (23), on another wave than t_kh_Y
template <class Em> HD void t_kh_Y_chest(KHCtx<Em>& h, int j) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const double* a = s.aw[j];
    const double on = cx.ki.first ? 0.0 : 1.0;
    double Y[3] = {0.0, 0.0, 0.0};
    if (cx.kt.chest_pos[j] >= 0) {
        double E[9], Ea[3];
        chest_error(s, E);
        const double trE = E[0] + E[4] + E[8], e = trE - 3.0;
        const double ax[3] = {E[7] - E[5], E[2] - E[6], E[3] - E[1]};
        matvec3(E, a, Ea);
        const double de = -dot3(ax, a);
        const double m2 = 2.0 * h.hx.sigma * on * cx.st.m_frameq;
        for (int r = 0; r < 3; ++r) Y[r] += m2 * (e * (Ea[r] - trE * a[r]) + de * (-ax[r]));
    }
    for (int r = 0; r < 3; ++r) h.hx.Yc[j][r] = Y[r];
}

// --- (s_j, s_i) and (s_k, sdot_l).  Most joint pairs are UNRELATED (neither on the other's path to the root): only the dense terms
//     through com(s) reach them — a few multiply-adds.  The related pairs (each joint with the <= 8 joints of its own path) carry the
//     second derivatives proper.  Two task groups each, so that a wave iteration is either all light or all heavy:
//       t_kh_ss_far   lanes over the LIST of unrelated pairs (+ the feet-distance term of one joint on each leg)
//       t_kh_ss_near  lanes over the list of related pairs (k, d), k on the path of d
//       t_kh_ssd_far  lanes over the list of unrelated (k, l)
//       t_kh_ssd_near the same list twice: (k, l) = (a, d) and (d, a)
inline void kh_fill_far_lists(const KinLite& kt, KHFarLists& f) {   // host only
    uint32_t rel[NJ];
    for (int j = 0; j < NJ; ++j) {
        rel[j] = 0u;
        for (int q = 0; q < 8; ++q) { const int a = int(kt.anc[j][q]); if (a < NJ) rel[j] |= 1u << unsigned(a); }
    }
    auto related = [&](int i, int j) { return (((rel[j] >> unsigned(i)) | (rel[i] >> unsigned(j))) & 1u) != 0u; };
    f.n_ss = f.n_ssd = f.n_near = 0; f.pad_ = 0;
    for (auto& v : f.ss) v = 0;
    for (auto& v : f.ssd) v = 0;
    for (auto& v : f.near) v = 0;
    for (int d = 0; d < NJ; ++d) for (int q = 0; q < 8; ++q) { const int k = int(kt.anc[d][q]); if (k < NJ) f.near[f.n_near++] = uint16_t((d << 8) | k); }
    for (int j = 0; j < NJ; ++j) for (int i = 0; i < j; ++i) if (!related(i, j)) f.ss[f.n_ss++] = uint16_t((j << 8) | i);
    for (int k = 0; k < NJ; ++k) for (int l = 0; l < NJ; ++l) if (!related(k, l)) f.ssd[f.n_ssd++] = uint16_t((k << 8) | l);
}
constexpr int KH_SS_TASKS = NJ * (NJ + 1) / 2;
template <class Em> HD void t_kh_ss_far(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const KHessScratch& hx = h.hx;
    if (t >= hx.far.n_ss) return;
    const int pr = int(hx.far.ss[t]), j = pr >> 8, i = pr & 0xff;
    double v = dot3(hx.dcmu[i], hx.E[j].l) + dot3(hx.dcmu[j], hx.E[i].l);
    // feet lateral distance  D = y_r . (o_l - o_r)  (K4; base fixed), one joint on each leg:  (a_R x y) . (a_L x (o_l - o_L))
    const int Li = cx.kt.leg_pos[0][i], Lj = cx.kt.leg_pos[0][j], Ri = cx.kt.leg_pos[1][i], Rj = cx.kt.leg_pos[1][j];
    if ((Li >= 0 && Rj >= 0) || (Ri >= 0 && Lj >= 0)) {
        const double* yr = s.fr_R[1];
        const double y[3] = {yr[1], yr[4], yr[7]};
        const double* ol = s.fr_o[0];
        const int jl = Li >= 0 ? i : j, jr = Li >= 0 ? j : i;
        double u1[3], u2[3], u3[3];
        cross3(s.aw[jr], y, u1);
        for (int r = 0; r < 3; ++r) u3[r] = ol[r] - s.ow[jl + 1][r];
        cross3(s.aw[jl], u3, u2);
        v += h.lam[gs::FEETD] * dot3(u1, u2);
    }
    cx.em.H(hk::SS + j * (j + 1) / 2 + i, S_ + j, S_ + i, v);
}
constexpr int KH_NEAR_TASKS = NJ * 8;
template <class Em> HD void t_kh_ss_near(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    auto& s = cx.s;
    const KHessScratch& hx = h.hx;
    if (t >= hx.far.n_near) return;
    const int pr = int(hx.far.near[t]), d = pr >> 8, k = pr & 0xff;   // k on the path root -> d (inclusive)
    const double on = cx.ki.first ? 0.0 : 1.0;
    // centroidal momentum; points, com, chest
    double v = dot3(hx.dcmu[k], hx.E[d].l) + dot3(hx.dcmu[d], hx.E[k].l) - dot6(hx.Sxl[k], hx.E[d]) + dot6(hx.Wv[k], hx.Cv[d]) + dot3(s.aw[k], hx.Y[d]) + dot3(s.aw[k], hx.Yc[d]);
    // feet lateral distance, both joints on one leg.  For j on the left leg dD/ds_j = y . (a_j x (o_l - o_j)), on the right leg
    // dD/ds_j = (a_j x y) . (o_l - o_j)
    const int Lk = cx.kt.leg_pos[0][k], Ld = cx.kt.leg_pos[0][d], Rk = cx.kt.leg_pos[1][k], Rd = cx.kt.leg_pos[1][d];
    if ((Lk >= 0 && Ld >= 0) || (Rk >= 0 && Rd >= 0)) {
        const double* yr = s.fr_R[1];
        const double y[3] = {yr[1], yr[4], yr[7]};
        const double* ol = s.fr_o[0];
        double fd, u1[3], u2[3], u3[3];
        if (Lk >= 0) {            // y . (a_k x (a_d x (o_l - o_d)))
            for (int r = 0; r < 3; ++r) u1[r] = ol[r] - s.ow[d + 1][r];
            cross3(s.aw[d], u1, u2);
            cross3(s.aw[k], u2, u3);
            fd = dot3(y, u3);
        } else {
            cross3(s.aw[d], y, u1);                         // a_d x y
            cross3(s.aw[k], u1, u2);
            for (int r = 0; r < 3; ++r) u3[r] = ol[r] - s.ow[d + 1][r];
            fd = dot3(u2, u3);
            for (int r = 0; r < 3; ++r) u3[r] = s.ow[d + 1][r] - s.ow[k + 1][r];
            cross3(s.aw[k], u3, u2);
            fd -= dot3(u1, u2);
        }
        v += h.lam[gs::FEETD] * fd;
    }
    if (k == d) v += 2.0 * hx.sigma * on * cx.st.m_jreg * cx.st.w_jreg[d] * cx.st.w_jreg[d] + ends_diag(h, S_ + d);
    const int hi = k > d ? k : d, lo = k > d ? d : k;
    cx.em.H(hk::SS + hi * (hi + 1) / 2 + lo, S_ + hi, S_ + lo, v);
}
template <class Em> HD void t_kh_ssd_far(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const KHessScratch& hx = h.hx;
    if (t >= hx.far.n_ssd) return;
    const int pr = int(hx.far.ssd[t]), k = pr >> 8, l = pr & 0xff;
    cx.em.H(hk::SSD + k * NJ + l, S_ + k, SD_ + l, dot3(hx.dcmu[k], hx.Gm[l].l));
}
template <class Em> HD void t_kh_ssd_near(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const KHessScratch& hx = h.hx;
    const int nn = hx.far.n_near;
    if (t >= 2 * nn) return;
    const int dir = t >= nn ? 1 : 0, pr = int(hx.far.near[t - dir * nn]), d = pr >> 8, a = pr & 0xff;   // a on the path root -> d (inclusive)
    if (dir == 1 && a == d) return;
    const int k = dir == 0 ? a : d, l = dir == 0 ? d : a;   // row s_k, column sdot_l
    const double on = cx.ki.first ? 0.0 : 1.0;
    double v = dot3(hx.dcmu[k], hx.Gm[l].l);
    if (dir == 0) v += -dot6(hx.Sxl[k], hx.Gm[l]);   // k ancestor-or-self of l
    else v += -dot6(hx.Cv[k], hx.S[l]);              // l strict ancestor of k
    if (k == l) v += 2.0 * hx.sigma * on * cx.st.m_jreg * cx.st.w_jreg[k];   // J6: d2 (sd + w (s - ref))^2 / ds dsd
    cx.em.H(hk::SSD + k * NJ + l, S_ + k, SD_ + l, v);
}

// --- the joint rows of the theta-level blocks and their chain to the quaternion: lane j (NJ).  (theta_m, s_j), (theta_m, sdot_j),
//     (omega_m, s_j) of the centroidal momentum for the three base axes m stay in the lane — row j against the columns NJ + m — and go
//     straight into (s_j, q_l), (s_j, qdot_l), (sdot_j, q_l):  dtheta = G dq / |q|,  d omega / dq = dwq,  d omega / d qdot = G.
//     (Until round 4 three arrays in LDS, 69 lanes to fill them and 3 x 92 lanes behind a wave fence to contract them.)
template <class Em> HD void t_kh_theta_rows(KHCtx<Em>& h, int j) {
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const KHessScratch& hx = h.hx;
    const SV6 Ej = hx.E[j], Cj = hx.Cv[j], Gj = hx.Gm[j];
    double dj[3], TS[3], TSD[3], WS[3];
    for (int i = 0; i < 3; ++i) dj[i] = hx.dcmu[j][i];
    for (int m = 0; m < 3; ++m) {
        const int b = NJ + m;
        TS[m] = -dot6(hx.Sxl[b], Ej) + dot6(hx.Wv[b], Cj) + dot3(hx.dcmu[b], Ej.l) + dot3(dj, hx.E[b].l);   // (theta_m, s_j) without the d2com term (in Y)
        TSD[m] = dot3(hx.dcmu[b], Gj.l) - dot6(hx.Sxl[b], Gj);                                                // (theta_m, sdot_j)
        WS[m] = dot3(dj, hx.Gm[b].l) - dot6(Cj, hx.S[b]);                                                     // (omega_m, s_j)
    }
    const double Y[3] = {hx.Y[j][0] + hx.Yc[j][0], hx.Y[j][1] + hx.Yc[j][1], hx.Y[j][2] + hx.Yc[j][2]};
    for (int l = 0; l < 4; ++l) {
        double v = (s.G[l] * Y[0] + s.G[4 + l] * Y[1] + s.G[8 + l] * Y[2]) * s.inv_qnorm;
        for (int m = 0; m < 3; ++m) v += s.G[4 * m + l] * s.inv_qnorm * TS[m] + s.dwq[4 * m + l] * WS[m];
        cx.em.H(hk::SQ + 4 * j + l, S_ + j, QB_ + l, v);
        double vq = 0.0, vd = 0.0;
        for (int m = 0; m < 3; ++m) { vq += s.G[4 * m + l] * WS[m]; vd += s.G[4 * m + l] * s.inv_qnorm * TSD[m]; }
        cx.em.H(hk::SQD + 4 * j + l, S_ + j, QD_ + l, vq);
        cx.em.H(hk::SDQ + 4 * j + l, SD_ + j, QB_ + l, vd);
    }
}

// second derivative of  g . (q / |q|)  with respect to q, entry (r, c)
HD double norm2_entry(const double* g4, const double* qh, double inv_n, int r, int c) {
    const double gq = g4[0] * qh[0] + g4[1] * qh[1] + g4[2] * qh[2] + g4[3] * qh[3];
    return (-(g4[r] * qh[c] + qh[r] * g4[c] + (r == c ? gq : 0.0)) + 3.0 * gq * qh[r] * qh[c]) * (inv_n * inv_n);
}
// column l of  W(t) = 2 [ t_w 1 + [t_v]x | -t_v ]   (dtheta = W(qhat) dqhat, omega = W(qhat) qdot)
HD void What_col(const double* t, int l, double* out) {
    if (l == 3) { out[0] = -2.0 * t[0]; out[1] = -2.0 * t[1]; out[2] = -2.0 * t[2]; return; }
    for (int r = 0; r < 3; ++r) out[r] = 2.0 * ((r == l ? t[3] : 0.0) + skew_rc(t, r, l));
}

// --- (theta_m, omega_m') = (dcom_m x mu) . G_m'.l - (S_m x l) . G_m' - (I^C_0 l) . (S_m x S_m'): lanes (m, m') 9.  (Round 2 formed the
//     nine numbers inside every one of the sixteen (q_r, qdot_c) lanes, one after the other: 4.7 k cycles, the longest chain of the
//     last phase.) ------------------------------------------------------------------------------------------------------------------------------
template <class Em> HD void t_kh_tw(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    KHessScratch& hx = h.hx;
    const int m = t / 3, m2 = t - 3 * m;
    SV6 ell, Il, sx;
    for (int i = 0; i < 3; ++i) { ell.a[i] = hx.mu[i]; ell.l[i] = hx.ell_l[i]; }
    inertia6(s.comp[0], ell, Il);
    crm6(hx.S[NJ + m], hx.S[NJ + m2], sx);
    hx.TW[m][m2] = dot3(hx.dcmu[NJ + m], hx.Gm[NJ + m2].l) - dot6(hx.Sxl[NJ + m], hx.Gm[NJ + m2]) - dot6(Il, sx);
}
// --- (q_r, qdot_c): lanes 16, behind t_kh_tw on the same wave ------------------------------------------------------------------------------------
template <class Em> HD void t_kh_qqd(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const KHessScratch& hx = h.hx;
    HIPNLP_WAVE_SYNC();
    const int r = t >> 2, c = t & 3;
    double v = 0.0;
    for (int m = 0; m < 3; ++m) {
        const double gm = s.G[4 * m + r] * s.inv_qnorm;
        for (int m2 = 0; m2 < 3; ++m2) v += gm * hx.TW[m][m2] * s.G[4 * m2 + c];
    }
    // K . d2 omega / dq_r dqdot_c = K . W(J[:, r])[:, c],  J = (1 - qh qh^T) / |q|
    double Jr[4], col[3];
    for (int a = 0; a < 4; ++a) Jr[a] = ((a == r ? 1.0 : 0.0) - s.qn[r] * s.qn[a]) * s.inv_qnorm;
    What_col(Jr, c, col);
    v += dot3(hx.K, col);
    cx.em.H(hk::QQD + t, QB_ + r, QD_ + c, v);
}

// --- (q_b, q_b), shared part: lane 0 forms  M = Mw R_b  and from it the Hessian B and the gradient g of  Phi(qhat) = <M, R(qhat)>
//     (R(qh) = I + 2 w [v]x + 2 [v]x^2:  Phi = tr M + 2 w v.ax(M) + 2 v^T M v - 2 (v.v) tr M), behind t_kh_mom0 ---------------------
//     Mw = sum_c w_c pkin_c^T + w_com com^T + sigma 2 m e E^T + mu L_G^T + omega K^T  (world frame), one entry per lane (9), behind
//     t_kh_mom0 on its wave; lane 0 then goes on alone (t_kh_qq0): M = Mw R_b, B, g.
template <class Em> HD void t_kh_qq0_mw(KHCtx<Em>& h, int t) {
    HIPNLP_WAVE_SYNC();   // behind t_kh_mom0 on its wave
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    KHessScratch& hx = h.hx;
    const double* lam = h.lam;
    const double on = cx.ki.first ? 0.0 : 1.0;
    double E[9];
    chest_error(s, E);
    const double trE = E[0] + E[4] + E[8], e = trE - 3.0;
    const double m2 = 2.0 * hx.sigma * on * cx.st.m_frameq;
    if (t == 0) {
        hx.qq_axE[0] = E[7] - E[5]; hx.qq_axE[1] = E[2] - E[6]; hx.qq_axE[2] = E[3] - E[1];
        hx.qq_m2 = m2;
    }
    const int a = t / 3, b = t - 3 * a;
    double acc = -lam[gs::COMC + a] * hx.com[b] + m2 * e * E[3 * b + a] + hx.mu[a] * hx.LG[b] + s.omega[a] * hx.K[b];
    for (int p = 0; p < NC; ++p) acc += -lam[gs::PT_STRIDE * p + gs::KINC + a] * s.pkin[p][b];
    hx.qqMw[t] = acc;
}
template <class Em> HD void t_kh_qq0(KHCtx<Em>& h, int) {
    HIPNLP_WAVE_SYNC();   // behind t_kh_qq0_mw on its wave
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    KHessScratch& hx = h.hx;
    double M[9];
    matmul3(hx.qqMw, s.Rb, M);
    const double trM = M[0] + M[4] + M[8];
    const double al[3] = {M[7] - M[5], M[2] - M[6], M[3] - M[1]};
    const double* qh = s.qn;
    double* B = hx.qqB;
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) B[4 * a + b] = 2.0 * (M[3 * a + b] + M[3 * b + a]) - (a == b ? 4.0 * trM : 0.0);
        B[4 * a + 3] = B[12 + a] = 2.0 * al[a];
    }
    B[15] = 0.0;
    for (int a = 0; a < 3; ++a) hx.qqg[a] = 2.0 * qh[3] * al[a] + B[4 * a] * qh[0] + B[4 * a + 1] * qh[1] + B[4 * a + 2] * qh[2];
    hx.qqg[3] = 2.0 * (qh[0] * al[0] + qh[1] * al[1] + qh[2] * al[2]);
}

// --- (q_b, q_b): lanes over the lower triangle (10), behind t_kh_qq0 -----------------------------------------------------------------
//   Hess_q Phi(q / |q|) = J B J + ( -(g qh^T + qh g^T + (g.qh) I) + 3 (g.qh) qh qh^T ) / |q|^2,   J = (I - qh qh^T) / |q|
template <class Em> HD void t_kh_qq(KHCtx<Em>& h, int t) {
    Ctx<Em>& cx = h.cx;
    const auto& s = cx.s;
    const KHessScratch& hx = h.hx;
    const double* lam = h.lam;
    const double sigma = hx.sigma, on = cx.ki.first ? 0.0 : 1.0;
    const int r = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3));   // row of entry t of the packed lower 4 x 4 triangle
    const int c = t - r * (r + 1) / 2;
    const double* qh = s.qn;
    const double inv_n = s.inv_qnorm;
    const double* B = hx.qqB;
    double Jr[4], Jc[4], BJc[4];
    for (int a = 0; a < 4; ++a) { Jr[a] = ((a == r ? 1.0 : 0.0) - qh[r] * qh[a]) * inv_n; Jc[a] = ((a == c ? 1.0 : 0.0) - qh[c] * qh[a]) * inv_n; }
    for (int a = 0; a < 4; ++a) BJc[a] = B[4 * a] * Jc[0] + B[4 * a + 1] * Jc[1] + B[4 * a + 2] * Jc[2] + B[4 * a + 3] * Jc[3];
    double v = Jr[0] * BJc[0] + Jr[1] * BJc[1] + Jr[2] * BJc[2] + Jr[3] * BJc[3];
    v += norm2_entry(hx.qqg, qh, inv_n, r, c);
    // chest cost, outer product part: d e / d q_l = -(ax(E) . G_l) / |q|
    const double* axE = hx.qq_axE;
    const double ger = -(axE[0] * s.G[r] + axE[1] * s.G[4 + r] + axE[2] * s.G[8 + r]) * inv_n;
    const double gec = -(axE[0] * s.G[c] + axE[1] * s.G[4 + c] + axE[2] * s.G[8 + c]) * inv_n;
    v += hx.qq_m2 * ger * gec;
    // centroidal momentum, remaining terms: omega = Wq(qdot) qhat with Wq = 2 [ -qd_w 1 - [qd_v]x | qd_v ]
    {
        const double* qd = s.x + QD_;
        double g4[4];
        for (int l = 0; l < 3; ++l) g4[l] = -2.0 * qd[3] * hx.K[l] - 2.0 * (skew_rc(qd, 0, l) * hx.K[0] + skew_rc(qd, 1, l) * hx.K[1] + skew_rc(qd, 2, l) * hx.K[2]);
        g4[3] = 2.0 * dot3(qd, hx.K);
        v += norm2_entry(g4, qh, inv_n, r, c);
        double gr[3], gc[3], wr[3], wc[3], t1[3], t2[3], t3[3];
        for (int m = 0; m < 3; ++m) { gr[m] = s.G[4 * m + r] * inv_n; gc[m] = s.G[4 * m + c] * inv_n; wr[m] = s.dwq[4 * m + r]; wc[m] = s.dwq[4 * m + c]; }
        cross3(gc, hx.K, t1); v += dot3(wr, t1);
        cross3(gr, hx.K, t1); v += dot3(wc, t1);
        // (mu x g_r) . I_G (omega x g_c + dw_c) + (mu x g_c) . I_G (omega x g_r + dw_r)
        cross3(hx.mu, gr, t1);
        cross3(s.omega, gc, t2);
        for (int m = 0; m < 3; ++m) t2[m] += wc[m];
        matvec3(hx.IG, t2, t3);
        v += dot3(t1, t3);
        cross3(hx.mu, gc, t1);
        cross3(s.omega, gr, t2);
        for (int m = 0; m < 3; ++m) t2[m] += wr[m];
        matvec3(hx.IG, t2, t3);
        v += dot3(t1, t3);
    }
    if (r == c) {
        const double* qd = s.pk + PK_REF + R_BQ;   // base quaternion error is linear in q: Hessian 2 m |q_d|^2 1  (E13)
        v += 2.0 * lam[gs::UNITQ] + 2.0 * sigma * on * cx.st.m_baseq * (qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]) + ends_diag(h, QB_ + r);
    }
    cx.em.H(hk::QQ + t, QB_ + r, QB_ + c, v);
}

// the light (s, sdot) group — its list of unrelated pairs — in three lane ranges on three waves (two, three and the remaining wave iterations)
constexpr int KH_SSD_A = 2 * 64, KH_SSD_B = 3 * 64, KH_SSD_C = NJ * NJ - KH_SSD_A - KH_SSD_B;
static_assert(KH_SSD_C > 0, "(s, sdot) lane ranges");
template <class Em> HD void t_kh_ssd_far_a(KHCtx<Em>& h, int t) { t_kh_ssd_far(h, t); }
template <class Em> HD void t_kh_ssd_far_b(KHCtx<Em>& h, int t) { t_kh_ssd_far(h, t + KH_SSD_A); }
template <class Em> HD void t_kh_ssd_far_c(KHCtx<Em>& h, int t) { t_kh_ssd_far(h, t + KH_SSD_A + KH_SSD_B); }

// The Hessian program.  KIN(fn, n) runs a task of knot_body.h, RH(w, fn, n) a Hessian task, on wave w of four.  Only the KINEMATIC part
// of the knot program runs (joint transforms, forward kinematics, link momenta, composites, contact-point kinematics: none of the
// rows / Jacobian columns), and the Hessian tasks that need no kinematics fill the waves it leaves idle.
// Wave assignment from the per-wave timeline of tools/diag/hess_stamps.py (profiles/r03_hess_stamps_*.txt: cycles per task group, planar,
// N = 100): round 2's table left the (p, p) lanes (5.9 k cycles) alone on the critical wave of the first phase and 10.2 k cycles on
// wave 3 of the last phase against 5.2 - 7.6 k on the others.
#define HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER)                                                               \
    KIN(0, t_joints, NJ) RH(0, t_kh_ff, 36) KIN(1, t_base, 3) KIN(1, t_kin_padding, 16)                          \
    RH(1, t_kh_diag_planar, 42) RH(1, t_kh_percouple, 84) RH(2, t_kh_point_a, NC) RH(2, t_kh_point_b, NC) RH(2, t_kh_point_c_e, NC) RH(2, t_kh_point_d_e, NC) RH(3, t_kh_pp_planar_at<128>, 52) \
    RH(3, t_kh_bump, KH_BUMP_TASKS) RH(0, t_kh_pp_smooth_at<0>, 64) RH(1, t_kh_pp_smooth_at<64>, 64) RH(2, t_kh_pp_smooth_at<128>, 52) \
    BARRIER
#define HIPNLP_KNOT_HESS_PHASE1B(KIN, RH, BARRIER)                                                               \
    KIN(0, t_fk_rot_a, FK_TASKS_A) KIN(0, t_link_u_a, FK_SPLIT) KIN(3, t_fk_rot_b, FK_TASKS_B) KIN(3, t_link_u_b, NJ - FK_SPLIT) \
    RH(2, t_kh_point_smooth_pp, NC) RH(2, t_kh_point_smooth_pp2, 6 * NC)                                         \
    RH(1, t_kh_point_smooth_mixed, 6 * NC)                                                                       \
    RH(1, t_kh_pp_planar_at<0>, 64) RH(2, t_kh_pp_planar_at<64>, 64)                                             \
    RH(0, t_kh_diag_smooth, 42)                                                                                  \
    BARRIER
#define HIPNLP_KNOT_HESS_PHASE1C(KIN, RH, BARRIER)                                                               \
    KIN(0, t_links, NL) KIN(1, t_frames, 3) KIN(2, t_link_inertia, NL) RH(3, t_kh_point_smooth_pp_emit, 6 * NC) RH(3, t_kh_point_c_l, NC) RH(3, t_kh_point_d_l, NC) \
    BARRIER
#define HIPNLP_KNOT_HESS_PHASE1D(KIN, RH, BARRIER)                                                               \
    KIN(0, t_composite_g0, 64) KIN(1, t_composite_g1, 64) KIN(1, t_composite_g2, 64)                             \
    KIN(2, t_composite_g3, 64) KIN(2, t_composite_g4, 64) KIN(3, t_composite_g5, 64) KIN(3, t_pkin, NC)          \
    BARRIER
#define HIPNLP_KNOT_HESS_PHASE2(KIN, RH, BARRIER)                                                                \
    RH(0, t_kh_joint, NJ + 3) RH(1, t_kh_Y, NJ) RH(2, t_kh_mom0, 1) RH(2, t_kh_qq0_mw, 9) RH(2, t_kh_qq0, 1)     \
    RH(3, t_kh_joint_b, NJ + 3) RH(3, t_kh_Y_chest, NJ)                                                          \
    BARRIER
#define HIPNLP_KNOT_HESS_PHASE3(KIN, RH, BARRIER)                                                                \
    RH(0, t_kh_ss_near, KH_NEAR_TASKS) RH(0, t_kh_ssd_far_a, KH_SSD_A) RH(0, t_kh_ssd_far_c, KH_SSD_C)           \
    RH(1, t_kh_ssd_near, 2 * KH_NEAR_TASKS) RH(1, t_kh_ssd_far_b, KH_SSD_B)                                      \
    RH(2, t_kh_ss_far, KH_SS_TASKS) RH(2, t_kh_tw, 9) RH(2, t_kh_qqd, 16)                                        \
    RH(3, t_kh_theta_rows, NJ) RH(3, t_kh_qq, 10)                                                                \
    BARRIER
// diagnostic builds only (tools/diag/hess_phases.sh): -DHIPNLP_HESS_DIAG_PHASES=n runs the first n of the six phases (the values are then wrong)
#if !defined(HIPNLP_HESS_DIAG_PHASES)
#define HIPNLP_HESS_DIAG_PHASES 6
#endif
#if HIPNLP_HESS_DIAG_PHASES >= 6
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1B(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1C(KIN, RH, BARRIER) \
    HIPNLP_KNOT_HESS_PHASE1D(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE2(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE3(KIN, RH, BARRIER)
#elif HIPNLP_HESS_DIAG_PHASES == 5
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1B(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1C(KIN, RH, BARRIER) \
    HIPNLP_KNOT_HESS_PHASE1D(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE2(KIN, RH, BARRIER)
#elif HIPNLP_HESS_DIAG_PHASES == 4
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1B(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1C(KIN, RH, BARRIER) \
    HIPNLP_KNOT_HESS_PHASE1D(KIN, RH, BARRIER)
#elif HIPNLP_HESS_DIAG_PHASES == 3
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1B(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1C(KIN, RH, BARRIER)
#elif HIPNLP_HESS_DIAG_PHASES == 2
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1B(KIN, RH, BARRIER)
#elif HIPNLP_HESS_DIAG_PHASES == 1
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER) HIPNLP_KNOT_HESS_PHASE1A(KIN, RH, BARRIER)
#else
#define HIPNLP_KNOT_HESS_PROGRAM(KIN, RH, BARRIER)
#endif

}  // namespace hipnlp
