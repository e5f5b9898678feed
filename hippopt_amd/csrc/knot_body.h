// knot_body.h — the per-knot mathematics of the hipnlp engine, written once and compiled for the
// device (HIP kernel, gfx950), for the host layout recorder (layout.h) and for the test-only host
// emulation (tests/hostemu).  One workgroup of FOUR (or eight) wavefronts evaluates ONE knot: its rows of g (own
// algebraic rows and the trapezoid defect that ends at the knot), its COLUMN block of jac g (so the CCS
// output of a knot is one contiguous run), its slice of grad f and its cost partials.
//
// Every `t_*(cx, t)` is a lane task.  The knot program at the bottom of this file assigns task groups
// to waves ("roles"): tasks of one group run on consecutive lanes of one wave, different groups of a
// phase run concurrently on different waves (different SIMDs of the CU), a workgroup barrier separates
// phases.  Values leave through an emitter: em.G(slot, row_id, v) / em.J(slot, row_id, col, v); slots
// are compile-time native positions in LDS.
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

#include <type_traits>
#include <utility>

#include "nlp_defs.h"
#include "knot_tanh.h"

// Lanes of ONE wavefront exchanging data through LDS (no other wave involved): the hardware executes a
// wave's LDS operations in order, so only the compiler must be kept from moving accesses across this point.
#if defined(__HIPCC__)
#define HIPNLP_UNROLL _Pragma("unroll")
#else
#define HIPNLP_UNROLL
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define HIPNLP_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier()
#define HIPNLP_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // a value the caller knows to be the same in every lane -> SGPR
// Everything above this line is issued before anything below it: used between a batch of independent LDS reads and the first
// use of their values.  Left alone, the scheduler puts every read next to its use and waits for each one in turn (eight
// dependent round trips of ~130 cycles for an eight-term sum).  Only where a sum of LDS values is long: staging the operands of
// ordinary tasks this way measured slower.
#define HIPNLP_ISSUE_FENCE() __builtin_amdgcn_sched_barrier(0)
// An integer the compiler must take as it is (no instruction): behind it, `x ^ literal` stays ONE xor — left to itself the compiler splits
// a word it knows to be (a << 7) | (b << 4) into its parts again and pays two or three operations per derived address.
#define HIPNLP_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define HIPNLP_OPAQUE(x) ((void)0)
#define HIPNLP_ISSUE_FENCE() ((void)0)
#define HIPNLP_WAVE_SYNC() ((void)0)
#define HIPNLP_UNIFORM(x) (x)
#endif

namespace hipnlp {

// ---------------------------------------------------------------------------------------------------
// scratch (LDS on the device)
// ---------------------------------------------------------------------------------------------------
// stride of the per-link records own[] / comp[] (16 used): 17 doubles, so that lanes indexed by LINK hit different LDS banks
constexpr int LSTR = 17;
struct EndTerms { double c[105 + 84], g[105 + 84]; };  // minimize-mode end rows: cost partial and gradient share of row i

// Joint record: 16 doubles [ L (9) | of (3) | c (3) | sd ] in eight 16-byte chunks, at a stride of 144 bytes.  The FK lanes read the SAME
// chunk of DIFFERENT records in one instruction (ds_read_b128: sixteen lanes share the 256-byte bank row); at the natural 128-byte stride
// those all sit in the same four banks.  At nine 16-byte slots per record, chunk q of record i lies in slot (9 i + q) mod 16: records
// whose numbers differ by less than sixteen never meet.  (Rounds 1 - 4 kept the 128-byte stride and stored chunk q at position q ^ (i & 7):
// the same freedom from conflicts, but every chunk address was an xor of a per-lane word — twelve integer instructions per ancestor step
// of the forward kinematics where a multiply and eight immediate offsets do, round 5: -170 VALU instructions per knot.)
struct alignas(16) JointRec { double d[18]; };
static_assert(sizeof(JointRec) == 144, "joint record");
enum : int { JR_L = 0, JR_OF = 9, JR_C = 12, JR_SD = 15 };
HD constexpr int jr_pos(int, int e) { return e; }   // physical position of logical double e of record i

// Two layouts of the scratch (LAYOUT_*):
//   KnotScratchT<LAYOUT_FULL>     every array has its own storage (host recorders / emulation, eight-wave and pose / Hessian kernels)
//   KnotScratchT<LAYOUT_COMPACT>  the device layout of the four-wave callback kernels (four workgroups per CU need <= 40 KB of LDS per
//                                 workgroup): arrays whose lifetimes do not overlap share storage —
//                          * own[] (written in phase C, read in phase D) lives on top of the joint records Jr[] (written in phase A,
//                            dead once the forward kinematics of phase B has read them); own[NL] (the zero slot written in phase A)
//                            lies behind the last joint record;
//                          * the g rows of the horizon ends (final state, periodicity: first / last knot only, written in phase E)
//                            live in EndTerms::c of the same union (the minimize-mode cost partials they exclude row by row);
//                          * smooth terrain: comp[] (composites: written in phase D) holds the bump jets and then, IN PLACE, the terrain
//                            frames of the eight points during phases A - B (the full layout parks the frames in own[], which here
//                            lies on the joint records);
//                          * the tables that are read once per knot — joint frames (phase A), link inertials (phase C) — and the
//                            horizon-end tables (first / last knot) are not staged into LDS at all: the lanes that need them read
//                            them from global memory (parking them in comp[] instead measured 1.4 % slower).
constexpr int LAYOUT_FULL = 0, LAYOUT_COMPACT = 1, LAYOUT_COMPACT_NOG = 2 /* compact, and no staging of g: the pose finder's Hessian kernel, whose program emits no row */;
template <int LAYOUT> struct ScratchJrOwn;
template <> struct ScratchJrOwn<LAYOUT_FULL> {
    JointRec Jr[NJ + 1];
    union {  // own[] is dead once the composites are formed; the (rare) minimize-mode end terms reuse its space
        double own[NL + 1][LSTR];  // per link, same layout as comp; slot NL = 0 (padding of the descendant lists)
        EndTerms ends;
    };
    double g[gs::COUNT];
};
template <> struct ScratchJrOwn<LAYOUT_COMPACT> {
    union {
        JointRec Jr[NJ + 1];
        double own[NL + 1][LSTR];
        EndTerms ends;
    };
    double g[gs::FIN];   // slots >= gs::FIN (horizon-end rows) are ends.c[slot - gs::FIN]
};
template <> struct ScratchJrOwn<LAYOUT_COMPACT_NOG> {
    union {
        JointRec Jr[NJ + 1];
        double own[NL + 1][LSTR];
        EndTerms ends;
    };
};
// (compact layout: own[] lies ON the joint records, its zero slot own[NL] on the last two of them — written with the rest of own[], by
//  t_links in the third phase, when the forward kinematics has read the records)

// periodicity variables of the other end of the horizon, by periodicity row (only loaded at k = 0 and k = N-1): not in a trimmed scratch
template <bool HAS> struct ScratchXo { double xo[NPER]; };
template <> struct ScratchXo<false> {};
// per point contribution to hdot at knots k-1 (0) and k (1): not in the trimmed scratch of the smooth terrain (hd_of below)
template <bool HAS> struct ScratchHd { double hd[2][NC][6]; };
template <> struct ScratchHd<false> {};
// JSLOTS: slots of the Jacobian staging `jac` (default: all of them).  A VARY kernel (hipnlp.hip) keeps the slots behind js::V0 only —
// js::vary_slots(terrain) of them — and addresses them through a pointer moved back by js::V0.
template <int LAYOUT, int JSLOTS = js::COUNT, bool STATIC = false> struct alignas(16) KnotScratchT : ScratchJrOwn<LAYOUT>, ScratchXo<JSLOTS == js::COUNT>,
                                                                                 ScratchHd<JSLOTS != js::vary_slots(false)> {
    static constexpr int layout = LAYOUT;
    static constexpr bool compact = LAYOUT != LAYOUT_FULL;
    // trimmed (the four-wave VARY kernels: 32 KB of LDS per workgroup and not a byte more on the smooth terrain): besides the
    // Jacobian staging, the periodicity variables of the other horizon end — 84 doubles only the first and the last knot read, once —
    // stay in global memory (per_other below), and the knot records carry one pad word instead of three
    static constexpr bool trimmed = JSLOTS != js::COUNT;
    // The trimmed scratch of the SMOOTH terrain (the four-wave VARY kernel: LDS is handed out in 1 280-byte granules on gfx950, five
    // workgroups per CU means 25 granules = 32 000 B each) keeps the per-point momentum shares hd[][][] — 768 B — on the world
    // rotations Rw[1..11]: in that kernel the shares are written (t_points_vec) and consumed (t_hdyn_rows_a, behind it on its wave)
    // inside the FIRST phase, the forward kinematics writes Rw[1..] in the SECOND; t_base writes Rw[0] only.  (Valid for exactly the
    // instantiation hdyn_entries_early<Em> describes; the planar kernel sums the shares in the second phase and keeps its own array.)
    static constexpr bool hd_on_rw = JSLOTS == js::vary_slots(false) && JSLOTS != js::COUNT;
    static constexpr int xpad = trimmed ? NXK + 1 : XPAD;
    // STATIC (the pose finder's device kernels, Em::kStatic: no velocity of the record is non-zero and no task group that reads one runs):
    // the scratch carries neither the velocity arrays Uj / wv / vo (zero-length below: the layout of every other instantiation is what it
    // was) nor a whole previous-knot record — the pose finder keeps its 63 reference values there (pose_body.h XR_*)
    static constexpr bool has_velocities = !STATIC;
    static constexpr int xm_len = STATIC ? 64 : xpad;
    alignas(16) double x[xpad];    // knot k   (x, xm, xg, pk: 16-byte aligned, staged by direct global -> LDS loads)
    double xm[xm_len]; // knot k-1 (zeros at k = 0)
    double xg[8];      // horizon-global variables (initial_state.centroidal_momentum)
    double pk[PK_STRIDE];
    // base orientation
    double qn[4], qnorm, inv_qnorm, Rb[9], G[12] /* 3x4: dtheta = G dqhat */, omega[3], dwq[12] /* d omega/d qb (3x4) */;
    // kinematics in base-centred coordinates (origin = base origin; h_ang and relative positions are
    // invariant to the base position / linear velocity)
    // per joint, ONE 128-byte record the FK lanes stream with 16-byte LDS reads (ScratchJrOwn::Jr): [ parent_R_child (9) | o_fix (3) |
    // c = R_fix axis = parent-frame joint axis (3) | sdot ]; slot NJ = [identity | 0 | 0 | 0] (padding of the ancestor lists)
    double Uj[STATIC ? 0 : NJ + 1][3];   // (o_j x a_j) sdot_j ; slot NJ = 0
    double Rw[NL][9], ow[NL][3], aw[NJ + 1][3];  // padding slot aw[NJ] = 0
    double wv[STATIC ? 0 : NL][3], vo[STATIC ? 0 : NL][3];  // link angular velocity; velocity of the body point at the origin
    double comp[NL][LSTR];  // composite per link: [m | first moment h (3) | inertia@O xx,xy,xz,yy,yz,zz (6) | subtree momentum lin (3) | ang (3)]
    double com[3], klin[3], kang[3], hang[3];
    double dth_h[3][3];  // d hang / d theta_e   [e][i]
    double Aw[3][3];     // d hang / d omega_e   [e][i]
    double fr_R[3][9], fr_o[3][3];
    double pkin[NC][3];
    double chest_w[3], chest_dc;  // ax(R_c R_d^T);  d cost / d trace
    double cen_g[3];              // d centroid cost / d p_c,i (same for the 8 points)
    double yaw_e[2][2];  // per foot: alignment errors (forward, sideways); (the sin / cos of the yaw references are read where they are: pk[PK_YAWSC ..])
    // cost partials, reduced by t_reduce
    double c_pt[NC][3], c_joint[NJ], c_force[2][3], c_yaw[2];
    double cost[NCT];
    double grad[xpad];
    double jac[JSLOTS];
    // value of native g slot `slot` / where the emitter stores it
    HD double& g_at(int slot) {
        if constexpr (compact) return slot >= gs::FIN ? this->ends.c[slot - gs::FIN] : this->g[slot];
        else return this->g[slot];
    }
};
using KnotScratch = KnotScratchT<LAYOUT_FULL>;

enum : int { CM = 0, CH = 1, CI = 4, CKL = 10, CKA = 13 };  // offsets inside KnotScratch::comp[i]

struct KnotInfo {
    int k, N;
    int first, last;  // k == 0, k == N-1   (the recorder sets both)
};

// emitter traits: Em::Scratch (default KnotScratch) selects the scratch layout
template <class Em, class = void> struct ScratchOf { using type = KnotScratch; };
template <class Em> struct ScratchOf<Em, std::void_t<typename Em::Scratch>> { using type = typename Em::Scratch; };

template <class Em> struct Ctx {
    using Scratch = typename ScratchOf<Em>::type;
    using Kin = std::conditional_t<Scratch::compact, KinLite, KinTables>;
    using GP = std::conditional_t<Scratch::compact, GParamsLite, GParams>;
    Scratch& s;
    const Kin& kt;
    const KSettings& st;
    const GP& gp;
    KnotInfo ki;
    Em em;
    // the FULL tables wherever they live (compact device layout: global memory; elsewhere: the same objects as kt / gp)
    const KinTables* gkt;
    const GParams* ggp;
    const PoseHands* hands = nullptr;   // pose finder only (set by its kernels / host expansions behind the constructor)
    const double* x_other = nullptr;    // trimmed scratch only: the knot record of the OTHER horizon end in global memory (first knot: the last one's, and vice versa)
    HD Ctx(Scratch& s_, const Kin& kt_, const KSettings& st_, const GP& gp_, KnotInfo ki_, Em em_, const KinTables* gkt_ = nullptr, const GParams* ggp_ = nullptr)
        : s(s_), kt(kt_), st(st_), gp(gp_), ki(ki_), em(em_), gkt(gkt_), ggp(ggp_) {
        if constexpr (!Scratch::compact) {
            if (!gkt) gkt = &kt_;
            if (!ggp) ggp = &gp_;
        }
    }
};

// ---- tables that are read once per knot, from the full tables (gkt: the LDS copy in the full layout, global memory in the compact one)
// joint frames of joint j: phase A only
template <class Em> HD const double* kin_R_fix(const Ctx<Em>& cx, int j) {
    return cx.gkt->jf.j[j].R_fix;
}
template <class Em> HD const double* kin_o_fix(const Ctx<Em>& cx, int j) {
    return cx.gkt->jf.j[j].o_fix;
}
template <class Em> HD const double* kin_axis(const Ctx<Em>& cx, int j) {
    return cx.gkt->jf.j[j].axis;
}
// link inertials of link i: phase C only
template <class Em> HD double kin_mass(const Ctx<Em>& cx, int i) {
    return cx.gkt->li.l[i].mass;
}
template <class Em> HD const double* kin_com(const Ctx<Em>& cx, int i) {
    return cx.gkt->li.l[i].com;
}
template <class Em> HD const double* kin_inertia(const Ctx<Em>& cx, int i) {
    return cx.gkt->li.l[i].inertia;
}
// g rows of the horizon ends (native slots >= gs::FIN): compact layout -> ends.c (see KnotScratchT)
template <class Em> HD void emit_g_end(Ctx<Em>& cx, int slot, int id, double v) {
    if constexpr (Ctx<Em>::Scratch::compact) { (void)id; cx.s.ends.c[slot - gs::FIN] = v; }
    else cx.em.G(slot, id, v);
}
// the per-point momentum shares [2][NC][6] (see KnotScratchT::hd_on_rw)
template <class S> HD double* hd_of(S& s) {
    if constexpr (S::hd_on_rw) { static_assert(sizeof(double) * 2 * NC * 6 <= sizeof(double) * 9 * (NL - 1), "hd on Rw[1..]"); return &s.Rw[1][0]; }
    else return &s.hd[0][0][0];
}
// periodicity variable i of the other horizon end (first / last knot only): LDS copy, or — trimmed scratch — global memory
template <class Em> HD double per_other(const Ctx<Em>& cx, int i) {
    if constexpr (Ctx<Em>::Scratch::trimmed) return cx.x_other[periodicity_row_var(i)];
    else return cx.s.xo[i];
}
// horizon-end tables and final-state values: first / last knot only (compact layout: straight from global memory)
template <class Em> HD const EndTables& end_tables(const Ctx<Em>& cx) { return cx.gkt->en; }
template <class Em> HD const double* final_rhs(const Ctx<Em>& cx) { return cx.ggp->final_rhs; }

// base of contact point c's block of VARYING jac slots (nlp_defs.h, region P: its stride depends on the terrain); js::ptc(c) is the base
// of the point's block of constant slots
template <class Em> HD int js_pt(const Ctx<Em>& cx, int c);
// The terrain kind is a COMPILE-TIME constant of the device emitters (one kernel instantiation per terrain: the planar kernel
// must not pay registers for the smooth-terrain jets) and a run-time value (Em::kTerrain < 0) for the host-side recorders.
template <class Em> HD bool terrain_is_planar(const Ctx<Em>& cx) {
    if constexpr (Em::kTerrain >= 0) return Em::kTerrain == HIPNLP_TERRAIN_PLANAR;
    else return cx.st.terrain == HIPNLP_TERRAIN_PLANAR;
}

template <class Em> HD int js_pt(const Ctx<Em>& cx, int c) { return js::PV0 + js::pt_stride(terrain_is_planar(cx)) * c; }

// Entries of jac g that do NOT depend on x — literals and parameters only: +-1, -dt/2, the mass, +-1/4 (the trapezoid defects
// integrators/implicit_trapezoid.py:24-39, the x0 rows base/multiple_shooting_solver.py:713-742, the single-variable bound rows
// planner.py:386-405,699-719) — leave through emit_jc.  An emitter with a JC member sees them as such: the layout recorder marks the
// slot constant (Layout::jconst), the parameter pass of hipnlp_set_params collects the values (a handle's host destinations are
// filled with them once and its kernels then store the varying entries only); every other emitter takes them as ordinary entries.
template <class Em, class = void> struct em_has_jc : std::false_type {};
template <class Em> struct em_has_jc<Em, std::void_t<decltype(std::declval<Em&>().JC(0, 0, 0, 0.0))>> : std::true_type {};
// an entry that DEPENDS on x but lives in a slot of the constant region D (only the pose finder has such: it reuses slots of rows it does not
// have — pose_body.h): an emitter with a JD member maps the region itself (the pose kernels stage a compacted copy of it)
template <class Em, class = void> struct em_has_jd : std::false_type {};
template <class Em> struct em_has_jd<Em, std::void_t<decltype(std::declval<Em&>().JD(0, 0, 0, 0.0))>> : std::true_type {};
template <class Em> HD void emit_jd(Em& em, int slot, int rid, int col, double v) {
    if constexpr (em_has_jd<Em>::value) em.JD(slot, rid, col, v);
    else em.J(slot, rid, col, v);
}
template <class Em> HD void emit_jc(Em& em, int slot, int rid, int col, double v) {
    if constexpr (em_has_jc<Em>::value) em.JC(slot, rid, col, v);
    else em.J(slot, rid, col, v);
}

// ---------------------------------------------------------------------------------------------------
// tiny helpers on raw arrays
// ---------------------------------------------------------------------------------------------------
HD void cross3(const double* a, const double* b, double* r) {
    const double r0 = a[1] * b[2] - a[2] * b[1];
    const double r1 = a[2] * b[0] - a[0] * b[2];
    const double r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
HD double cross_comp(const double* a, const double* b, int i) {  // component i of a x b
    const int i1 = i == 2 ? 0 : i + 1, i2 = i == 0 ? 2 : i - 1;
    return a[i1] * b[i2] - a[i2] * b[i1];
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
HD double skew_rc(const double* a, int r, int c) {  // [a]x(r, c):  (0,1)=-a2 (0,2)=a1 (1,0)=a2 (1,2)=-a0 (2,0)=-a1 (2,1)=a0
    if (r == c) return 0.0;
    const int k = 3 - r - c;
    return (c == (r + 2) % 3) ? a[k] : -a[k];
}
// R = I + 2 w [v]x + 2 [v]x^2  (liecasadi SO3.as_matrix, xyzw)
HD void rot_from_quat(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}

// 1 / sqrt(x).  Device: hardware estimate (v_rsq_f64) + two Newton steps (one IEEE sqrt and one IEEE division cost several
// hundred cycles of dependent instructions); x = 0 gives a non-finite result, as 1 / sqrt(0) does.
HD double inv_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    r = r * (1.5 - hx * r * r);
    r = r * (1.5 - hx * r * r);
    return r;
#else
    return 1.0 / sqrt(x);
#endif
}

// ---------------------------------------------------------------------------------------------------
// Smooth step terrain (TerrainSum of SmoothTerrain.step, robot_planning/utilities/smooth_terrain.py:201-227,266-336,
// terrain_sum.py:19-38):  h(p) = p_z - Z(p_x, p_y),  Z = sum_s [ H_s exp(-g_s^r) + o_s,z ],  g = a^m + b^m.
// The rows need the normal n = grad h / |grad h|, the orientation R_t = [x y n] (terrain_descriptor.py:45-80), the
// directional derivatives hdot = grad h . v and ndot = (dn/dp) v (complementarity.py:74-75) AND their derivatives with
// respect to p: partials of Z up to THIRD order, evaluated in closed form (Faa di Bruno on psi(g) = H exp(-g^r)); everything
// that is built from them is propagated with a 2-direction dual number (d/dp_x, d/dp_y; nothing but h depends on p_z).
// ---------------------------------------------------------------------------------------------------
struct D2 {
    double v, x, y;
    HD D2() : v(0.0), x(0.0), y(0.0) {}
    HD D2(double c) : v(c), x(0.0), y(0.0) {}
    HD D2(double v_, double x_, double y_) : v(v_), x(x_), y(y_) {}
};
HD D2 operator+(const D2& a, const D2& b) { return D2(a.v + b.v, a.x + b.x, a.y + b.y); }
HD D2 operator-(const D2& a, const D2& b) { return D2(a.v - b.v, a.x - b.x, a.y - b.y); }
HD D2 operator-(const D2& a) { return D2(-a.v, -a.x, -a.y); }
HD D2 operator*(const D2& a, const D2& b) { return D2(a.v * b.v, a.x * b.v + a.v * b.x, a.y * b.v + a.v * b.y); }
HD D2 operator*(const D2& a, double c) { return D2(a.v * c, a.x * c, a.y * c); }
HD D2 operator*(double c, const D2& a) { return D2(a.v * c, a.x * c, a.y * c); }
HD D2 operator/(const D2& a, const D2& b) { const double q = a.v / b.v, i = 1.0 / b.v; return D2(q, (a.x - q * b.x) * i, (a.y - q * b.y) * i); }
HD D2 d2sqrt(const D2& a) { const double r = sqrt(a.v), i = 0.5 / r; return D2(r, a.x * i, a.y * i); }
HD D2 d2tanh(const D2& a) { const double t = knot_tanh(a.v), d = 1.0 - t * t; return D2(t, a.x * d, a.y * d); }

// x^n by squaring.  The first five bits without a branch (the terrain exponents: n < 32 for a sharpness up to 16; the lanes of a bump
// task hold different bumps, so a loop over the bits of n is a divergent one), a loop for anything beyond; the products that reach
// the result are the same, in the same order, either way.
HD double ipow_d(double x, int n) {
    double r = 1.0, b = x;
    for (int i = 0; i < 5; ++i) { r = (n & 1) ? r * b : r; b = b * b; n >>= 1; }
    while (n > 0) { if (n & 1) r *= b; n >>= 1; if (n) b *= b; }
    return r;
}

// contribution of ONE bump to the jet (out is accumulated into).  slope_x / slope_y: TerrainTops of this step, read only when the step says
// its top is sloped (step_sloped): Z_s = exp(-g^r) pi,  pi = height + slope_x dx + slope_y dy  (smooth_terrain.py:211, 238-264).
HD void terrain_bump_jet(const TerrainStepK& t, double px, double py, int order, double* out, const double* slope_x = nullptr, const double* slope_y = nullptr) {
    out[0] += t.oz;
    const double dx = px - t.ox, dy = py - t.oy;
    const double a = t.ax * dx + t.ay * dy, b = t.bx * dx + t.by * dy;
    const int m = step_m(t), r = t.r;
    const double am3 = ipow_d(a, m - 3), bm3 = ipow_d(b, m - 3);
    const double A0 = am3 * a * a * a, A1 = m * am3 * a * a, A2 = double(m * (m - 1)) * am3 * a, A3 = double(m * (m - 1) * (m - 2)) * am3;
    const double B0 = bm3 * b * b * b, B1 = m * bm3 * b * b, B2 = double(m * (m - 1)) * bm3 * b, B3 = double(m * (m - 1) * (m - 2)) * bm3;
    const double g = A0 + B0;
    const double gr3 = ipow_d(g, r - 3);
    const double w = gr3 * g * g * g;
    if (!(w < 700.0)) return;  // exp(-w) underflows (also catches inf / nan of far-away points): the bump and all its derivatives vanish
    const double w1 = r * gr3 * g * g, w2 = double(r * (r - 1)) * gr3 * g, w3 = double(r * (r - 1) * (r - 2)) * gr3;
    if (step_sloped(t) && slope_x) {
        // F = exp(-g^r) and its partials (the flat-top formulas below with height 1), then the product with the LINEAR top pi
        const double e0 = exp(-w), e1 = -e0 * w1, e2 = e0 * (w1 * w1 - w2), e3 = e0 * (-w1 * w1 * w1 + 3.0 * w1 * w2 - w3);
        const double pix = *slope_x, piy = *slope_y, pi0 = t.height + pix * dx + piy * dy;
        const double gx = A1 * t.ax + B1 * t.bx, gy = A1 * t.ay + B1 * t.by;
        const double Fx = e1 * gx, Fy = e1 * gy;
        out[0] += e0 * pi0;
        out[1] += Fx * pi0 + e0 * pix;
        out[2] += Fy * pi0 + e0 * piy;
        if (order < 2) return;
        const double gxx = A2 * t.ax * t.ax + B2 * t.bx * t.bx, gxy = A2 * t.ax * t.ay + B2 * t.bx * t.by, gyy = A2 * t.ay * t.ay + B2 * t.by * t.by;
        const double Fxx = e2 * gx * gx + e1 * gxx, Fxy = e2 * gx * gy + e1 * gxy, Fyy = e2 * gy * gy + e1 * gyy;
        out[3] += Fxx * pi0 + 2.0 * Fx * pix;
        out[4] += Fxy * pi0 + Fx * piy + Fy * pix;
        out[5] += Fyy * pi0 + 2.0 * Fy * piy;
        if (order < 3) return;
        const double gxxx = A3 * t.ax * t.ax * t.ax + B3 * t.bx * t.bx * t.bx, gxxy = A3 * t.ax * t.ax * t.ay + B3 * t.bx * t.bx * t.by;
        const double gxyy = A3 * t.ax * t.ay * t.ay + B3 * t.bx * t.by * t.by, gyyy = A3 * t.ay * t.ay * t.ay + B3 * t.by * t.by * t.by;
        const double Fxxx = e3 * gx * gx * gx + e2 * (3.0 * gxx * gx) + e1 * gxxx, Fxxy = e3 * gx * gx * gy + e2 * (gxx * gy + 2.0 * gxy * gx) + e1 * gxxy;
        const double Fxyy = e3 * gx * gy * gy + e2 * (gyy * gx + 2.0 * gxy * gy) + e1 * gxyy, Fyyy = e3 * gy * gy * gy + e2 * (3.0 * gyy * gy) + e1 * gyyy;
        out[6] += Fxxx * pi0 + 3.0 * Fxx * pix;
        out[7] += Fxxy * pi0 + 2.0 * Fxy * pix + Fxx * piy;
        out[8] += Fxyy * pi0 + 2.0 * Fxy * piy + Fyy * pix;
        out[9] += Fyyy * pi0 + 3.0 * Fyy * piy;
        return;
    }
    const double psi = t.height * exp(-w);
    const double p1 = -psi * w1, p2 = psi * (w1 * w1 - w2), p3 = psi * (-w1 * w1 * w1 + 3.0 * w1 * w2 - w3);
    const double gx = A1 * t.ax + B1 * t.bx, gy = A1 * t.ay + B1 * t.by;
    out[0] += psi;
    out[1] += p1 * gx;
    out[2] += p1 * gy;
    if (order < 2) return;
    const double gxx = A2 * t.ax * t.ax + B2 * t.bx * t.bx, gxy = A2 * t.ax * t.ay + B2 * t.bx * t.by, gyy = A2 * t.ay * t.ay + B2 * t.by * t.by;
    out[3] += p2 * gx * gx + p1 * gxx;
    out[4] += p2 * gx * gy + p1 * gxy;
    out[5] += p2 * gy * gy + p1 * gyy;
    if (order < 3) return;
    const double gxxx = A3 * t.ax * t.ax * t.ax + B3 * t.bx * t.bx * t.bx, gxxy = A3 * t.ax * t.ax * t.ay + B3 * t.bx * t.bx * t.by;
    const double gxyy = A3 * t.ax * t.ay * t.ay + B3 * t.bx * t.by * t.by, gyyy = A3 * t.ay * t.ay * t.ay + B3 * t.by * t.by * t.by;
    out[6] += p3 * gx * gx * gx + p2 * (3.0 * gxx * gx) + p1 * gxxx;
    out[7] += p3 * gx * gx * gy + p2 * (gxx * gy + 2.0 * gxy * gx) + p1 * gxxy;
    out[8] += p3 * gx * gy * gy + p2 * (gyy * gx + 2.0 * gxy * gy) + p1 * gxyy;
    out[9] += p3 * gy * gy * gy + p2 * (3.0 * gyy * gy) + p1 * gyyy;
}
// Z and its partials up to third order: out = [Z, Zx, Zy, Zxx, Zxy, Zyy, Zxxx, Zxxy, Zxyy, Zyyy]
HD void terrain_Z_jet(const KSettings& st, double px, double py, int order, double* out, const TerrainTops* tops = nullptr) {
    for (int i = 0; i < 10; ++i) out[i] = 0.0;
    for (int sidx = 0; sidx < st.n_steps; ++sidx)
        terrain_bump_jet(st.steps[sidx], px, py, order, out, tops ? &tops->px[sidx] : nullptr, tops ? &tops->py[sidx] : nullptr);
}

// ===================================================================================================
// PHASE A — everything that depends only on the loaded knot records
// ===================================================================================================

// --- contact points, component-wise: lane (c, i), 24 tasks.  planner.py:721-744, 646-654, 699-719 ----
// (the trapezoid defects of the point states are a task group of their own: they share nothing with the rest but the knot record)
template <class Em> HD void t_points_dyn(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    const int c = t / 3, i = t - 3 * c;
    const double* x = s.x + PT_ * c;
    const double* xm = s.xm + PT_ * c;
    const double half = 0.5 * cx.gp.dt;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    Em& em = cx.em;
    // trapezoid defects of dot(f) = f_dot, dot(p) = v (T7) + x0 rows
    em.G(gb + gs::FDYN + i, row_id(RK_FDYN_IN, c, i), x[F_ + i] - (xm[F_ + i] + half * (xm[FD_ + i] + x[FD_ + i])));
    em.G(gb + gs::PDYN + i, row_id(RK_PDYN_IN, c, i), x[P_ + i] - (xm[P_ + i] + half * (xm[V_ + i] + x[V_ + i])));
    em.G(gb + gs::FDYN_X0 + i, row_id(RK_FDYN_X0, c, i), x[F_ + i]);
    em.G(gb + gs::PDYN_X0 + i, row_id(RK_PDYN_X0, c, i), x[P_ + i]);
    emit_jc(em, jc + js::FDYN + 0 + i, row_id(RK_FDYN_IN, c, i), cb + F_ + i, 1.0);
    emit_jc(em, jc + js::FDYN + 3 + i, row_id(RK_FDYN_IN, c, i), cb + FD_ + i, -half);
    emit_jc(em, jc + js::FDYN + 6 + i, row_id(RK_FDYN_OUT, c, i), cb + F_ + i, -1.0);
    emit_jc(em, jc + js::FDYN + 9 + i, row_id(RK_FDYN_OUT, c, i), cb + FD_ + i, -half);
    emit_jc(em, jc + js::FDYN + 12 + i, row_id(RK_FDYN_X0, c, i), cb + F_ + i, 1.0);
    emit_jc(em, jc + js::PDYN + 0 + i, row_id(RK_PDYN_IN, c, i), cb + P_ + i, 1.0);
    emit_jc(em, jc + js::PDYN + 3 + i, row_id(RK_PDYN_IN, c, i), cb + V_ + i, -half);
    emit_jc(em, jc + js::PDYN + 6 + i, row_id(RK_PDYN_OUT, c, i), cb + P_ + i, -1.0);
    emit_jc(em, jc + js::PDYN + 9 + i, row_id(RK_PDYN_OUT, c, i), cb + V_ + i, -half);
    emit_jc(em, jc + js::PDYN + 12 + i, row_id(RK_PDYN_X0, c, i), cb + P_ + i, 1.0);
}
template <class Em> HD void t_points_vec(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    const int c = t / 3, i = t - 3 * c;
    const double* x = s.x + PT_ * c;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    Em& em = cx.em;
    const double pz = x[P_ + 2];
    const bool planar = terrain_is_planar(cx);
    emit_jc(em, jc + js::PLANAR_V + i, row_id(RK_PLANAR, c, i), cb + V_ + i, 1.0);
    if (planar) {  // planar complementarity  v - R_t diag(tau,tau,1) u,  tau = tanh(kt h(p))   (E3; R_t = I, h = p_z)
        const double tau = knot_tanh(cx.gp.kt * pz);
        const double mult = i < 2 ? tau : 1.0;
        em.G(gb + gs::PLANAR + i, row_id(RK_PLANAR, c, i), x[V_ + i] - mult * x[U_ + i]);
        if (i < 2) em.J(jb + js::PL_U + i, row_id(RK_PLANAR, c, i), cb + U_ + i, -mult);
        else emit_jc(em, jb + js::PL_UZ, row_id(RK_PLANAR, c, i), cb + U_ + i, -1.0);
        if (i < 2) em.J(jb + js::PL_P + i, row_id(RK_PLANAR, c, i), cb + P_ + 2, -(cx.gp.kt * (1.0 - tau * tau)) * x[U_ + i]);
    }
    // control bound rows
    em.G(gb + gs::UB + i, row_id(RK_UB, c, i), x[U_ + i]);
    emit_jc(em, jc + js::UB + i, row_id(RK_UB, c, i), cb + U_ + i, 1.0);
    em.G(gb + gs::FDB + i, row_id(RK_FDB, c, i), x[FD_ + i] * cx.gp.mass);
    emit_jc(em, jc + js::FDB + i, row_id(RK_FDB, c, i), cb + FD_ + i, cx.gp.mass);
    // gradient of the point-local costs (k >= 1): swing height (E10), ||u_v||^2, ||f_dot||^2
    const double on = cx.ki.first ? 0.0 : 1.0;
    double* gr = s.grad + cb;
    if (planar) {  // (smooth terrain: t_terrain_hnf writes these two)
        gr[V_ + i] = i < 2 ? on * cx.st.m_swing * x[V_ + i] : 0.0;
        gr[P_ + i] = i == 2 ? on * cx.st.m_swing * (pz - s.pk[PK_REF + R_SWING]) : 0.0;
    }
    gr[U_ + i] = 2.0 * on * cx.st.m_ureg * x[U_ + i];
    gr[FD_ + i] = 2.0 * on * cx.st.m_fdreg * x[FD_ + i];
    gr[F_ + i] = 0.0;
    // contribution of this point to hdot at k-1 and k   (E1): component i of f and of (p - com) x f
    // (the two components of p - com the lane needs, read at its own indices: a three-entry register array indexed by the lane's
    //  component is eight selects per product)
    const int i1 = i == 2 ? 0 : i + 1, i2 = i == 0 ? 2 : i - 1;
    for (int w = 0; w < 2; ++w) {
        const double* xx = w ? s.x : s.xm;
        const double r1 = xx[cb + P_ + i1] - xx[COM_ + i1], r2 = xx[cb + P_ + i2] - xx[COM_ + i2];
        hd_of(s)[(w * NC + c) * 6 + i] = xx[cb + F_ + i];
        hd_of(s)[(w * NC + c) * 6 + 3 + i] = r1 * xx[cb + F_ + i2] - r2 * xx[cb + F_ + i1];   // component i of (p - com) x f
    }
}

// height, normal and orientation of the smooth terrain at one point with their d/dp_x, d/dp_y (E15: terrain_descriptor.py:45-80)
// from the jet of Z (second order suffices for the frame itself; the tangents of u1, u2 use Zxx, Zxy, Zyy)
struct TerrainFrame { D2 u1, u2, h, inn, n[3], xv[3], yv[3]; };
HD void terrain_frame(const double* Z, double pz, TerrainFrame& t) {
    t.u1 = D2(-Z[1], -Z[3], -Z[4]);           // grad h = (u1, u2, 1); the D2 tangents are d/dp_x, d/dp_y
    t.u2 = D2(-Z[2], -Z[4], -Z[5]);
    t.h = D2(pz - Z[0], -Z[1], -Z[2]);
    const D2 nn = d2sqrt(D2(1.0) + t.u1 * t.u1 + t.u2 * t.u2);
    t.inn = D2(1.0) / nn;
    t.n[0] = t.u1 * t.inn; t.n[1] = t.u2 * t.inn; t.n[2] = t.inn;
    // R_t = [xv yv n]:  y0 = n x e_x = (0, n_z, -n_y), x = y0 x n = (n_y^2 + n_z^2, -n_y n_x, -n_z n_x), xv = x / |x|, yv = n x xv
    // (terrain_descriptor.py:64-72).  n is a unit vector orthogonal to y0, so |x|^2 = (n_y^2 + n_z^2) =: q and
    // n x (y0 x n) = y0: xv = x / sqrt(q), yv = y0 / sqrt(q) — the same vectors with half the arithmetic.
    const D2 q = t.n[1] * t.n[1] + t.n[2] * t.n[2];
    const D2 iq = D2(1.0) / d2sqrt(q);
    t.xv[0] = q * iq; t.xv[1] = -(t.n[1] * t.n[0]) * iq; t.xv[2] = -(t.n[2] * t.n[0]) * iq;
    t.yv[0] = D2(0.0); t.yv[1] = t.n[2] * iq; t.yv[2] = -(t.n[1] * iq);
}

// height, normal force, friction cone rows of contact point c on the smooth terrain  (E15-E17, E6, E7); shared with pose_body.h
template <class Em> HD void point_hnf_smooth(Ctx<Em>& cx, int c, const TerrainFrame& tf) {
    Em& em = cx.em;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    const double* f = cx.s.x + cb + F_;
    const D2 *n = tf.n, *xv = tf.xv, *yv = tf.yv;
    const double gradh[3] = {tf.u1.v, tf.u2.v, 1.0};
    const D2 nf = n[0] * f[0] + n[1] * f[1] + n[2] * f[2];
    em.G(gb + gs::HEIGHT, row_id(RK_HEIGHT, c, 0), tf.h.v);
    for (int j = 0; j < 2; ++j) em.J(jb + js::HEIGHT_XY + j, row_id(RK_HEIGHT, c, 0), cb + P_ + j, gradh[j]);
    emit_jc(em, jc + js::HEIGHT_Z, row_id(RK_HEIGHT, c, 0), cb + P_ + 2, 1.0);
    em.G(gb + gs::NORMAL, row_id(RK_NORMAL, c, 0), nf.v);
    em.J(jb + js::NORMAL_P + 0, row_id(RK_NORMAL, c, 0), cb + P_ + 0, nf.x);
    em.J(jb + js::NORMAL_P + 1, row_id(RK_NORMAL, c, 0), cb + P_ + 1, nf.y);
    for (int j = 0; j < 3; ++j) em.J(jb + js::NORMAL_F + j, row_id(RK_NORMAL, c, 0), cb + F_ + j, n[j].v);
    const double mu2 = cx.gp.mu * cx.gp.mu;
    const D2 fcx = xv[0] * f[0] + xv[1] * f[1] + xv[2] * f[2], fcy = yv[0] * f[0] + yv[1] * f[1] + yv[2] * f[2], fcz = nf;
    const D2 fric = -(fcx * fcx) - (fcy * fcy) + (fcz * fcz) * mu2;
    em.G(gb + gs::FRICTION, row_id(RK_FRICTION, c, 0), fric.v);
    em.J(jb + js::FRICTION_P + 0, row_id(RK_FRICTION, c, 0), cb + P_ + 0, fric.x);
    em.J(jb + js::FRICTION_P + 1, row_id(RK_FRICTION, c, 0), cb + P_ + 1, fric.y);
    for (int j = 0; j < 3; ++j)
        em.J(jb + js::FRICTION_F + j, row_id(RK_FRICTION, c, 0), cb + F_ + j, -2.0 * fcx.v * xv[j].v - 2.0 * fcy.v * yv[j].v + 2.0 * mu2 * fcz.v * n[j].v);
}
// the same three rows on the planar terrain (E14: h = p_z, n = e_z, R_t = I)
template <class Em> HD void point_hnf_planar(Ctx<Em>& cx, int c) {
    Em& em = cx.em;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    const double* x = cx.s.x + cb;
    const double pz = x[P_ + 2], fz = x[F_ + 2];
    em.G(gb + gs::HEIGHT, row_id(RK_HEIGHT, c, 0), pz);
    emit_jc(em, jc + js::HEIGHT_Z, row_id(RK_HEIGHT, c, 0), cb + P_ + 2, 1.0);
    em.G(gb + gs::NORMAL, row_id(RK_NORMAL, c, 0), fz);
    emit_jc(em, jb + js::PL_NORMAL_FZ, row_id(RK_NORMAL, c, 0), cb + F_ + 2, 1.0);
    const double mu2 = cx.gp.mu * cx.gp.mu;
    em.G(gb + gs::FRICTION, row_id(RK_FRICTION, c, 0), -(x[F_] * x[F_]) - (x[F_ + 1] * x[F_ + 1]) + mu2 * (fz * fz));
    em.J(jb + js::FRICTION_F + 0, row_id(RK_FRICTION, c, 0), cb + F_ + 0, -2.0 * x[F_]);
    em.J(jb + js::FRICTION_F + 1, row_id(RK_FRICTION, c, 0), cb + F_ + 1, -2.0 * x[F_ + 1]);
    em.J(jb + js::FRICTION_F + 2, row_id(RK_FRICTION, c, 0), cb + F_ + 2, 2.0 * mu2 * fz);
}

// --- contact point rows on the SMOOTH terrain (E3, E4, E6, E7, E10, E14-E17) in two stages:
//   t_terrain_stage (phase A, lane c): third-order jet of the bump sum and the terrain frame with its d/dp_x, d/dp_y, parked in
//       the (not yet used) per-link area of the scratch;
//   t_terrain_planar / _dcc / _hnf / _swing (phase B, lane c each, on different waves): the rows, their Jacobian entries, the swing cost.
struct TerrainStage { TerrainFrame tf; double Zh[7]; };   // Zh = Z[3..9]: second and third derivatives (for udot)
static_assert(sizeof(TerrainStage) * NC <= sizeof(double) * NL * LSTR, "terrain staging must fit in own[0..NL) (own[NL] is the zero slot) / in comp[]");
template <class S> HD TerrainStage* terrain_stage(S& s, int c) {
    // compact layouts: own[] lies on the joint records, which are live while the terrain frames are: the frames take the place of the
    // bump jets in comp[] (written IN PLACE by t_terrain_stage: every lane has read its jets before any lane stores its frame)
    if constexpr (S::compact) return reinterpret_cast<TerrainStage*>(&s.comp[0][0]) + c;
    else return reinterpret_cast<TerrainStage*>(&s.own[0][0]) + c;
}

// the jet of ONE bump at ONE point: lane (c, bump), (NC + 1) x HIPNLP_MAX_TERRAIN_STEPS tasks — the eight contact points and, as a ninth
// point, the com (the minimum com height row needs h_terrain(com): same instruction stream, four more lanes of the same wave; as a
// task of its own on another wave — one lane per bump — it was the longest chain of the first phase of the four-wave stairs kernel
// at batch: 3.7 k cycles next to a 3.6 k cycles t_small).  Parked in the composite area (written only three phases later); the
// lanes of unused bumps store zeros, so that the sums below run over a FIXED number of parts, all read in one round trip (adding a
// zero part changes no bit).  (Splitting a bump's jet further over three lanes by derivative order measured slower: the common
// prefix — powers and the exponential — dominates and is then computed three times.)  t_terrain_stage follows on the same wave,
// adds the bumps in order and builds the frames; the com's three numbers wait at the very end of comp[] — behind the bump jets
// and the frames that take their place in the compact layouts — for t_com_height two phases later.
constexpr int TERRAIN_BUMP_TASKS = (NC + 1) * HIPNLP_MAX_TERRAIN_STEPS;
template <class S> HD double* terrain_bump_part(S& s, int c, int sidx) { return &s.comp[0][0] + 10 * (HIPNLP_MAX_TERRAIN_STEPS * c + sidx); }
template <class S> HD double* terrain_com_part(S& s, int sidx) { return &s.comp[0][0] + NL * LSTR - 3 * HIPNLP_MAX_TERRAIN_STEPS + 3 * sidx; }
static_assert(sizeof(TerrainStage) * NC + sizeof(double) * 3 * HIPNLP_MAX_TERRAIN_STEPS <= sizeof(double) * NL * LSTR &&
              sizeof(double) * (10 * NC * HIPNLP_MAX_TERRAIN_STEPS + 3 * HIPNLP_MAX_TERRAIN_STEPS) <= sizeof(double) * NL * LSTR, "bump jets / frames and the com parts behind them must fit in comp[]");
template <class Em> HD void t_terrain_bump(Ctx<Em>& cx, int t) {
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    const int c = t / HIPNLP_MAX_TERRAIN_STEPS, sidx = t - HIPNLP_MAX_TERRAIN_STEPS * c;
    const double* p = c < NC ? s.x + PT_ * c + P_ : s.x + COM_;
    double Z[10];
    for (int i = 0; i < 10; ++i) Z[i] = 0.0;
    if (sidx < cx.st.n_steps) terrain_bump_jet(cx.st.steps[sidx], p[0], p[1], 3, Z, &cx.gkt->tops.px[sidx], &cx.gkt->tops.py[sidx]);
    if (c < NC) {
        double* out = terrain_bump_part(s, c, sidx);
        for (int i = 0; i < 10; ++i) out[i] = Z[i];
    } else {
        double* out = terrain_com_part(s, sidx);
        for (int i = 0; i < 3; ++i) out[i] = Z[i];
    }
}
// minimum com height on the smooth terrain: com_z - h_terrain(com) (planner.py:352-362; the planar row is part of t_small).  One lane,
// in the third phase, on a wave with nothing else to do there.
template <class Em> HD void t_com_height(Ctx<Em>& cx, int) {
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    Em& em = cx.em;
    double Z[3] = {0.0, 0.0, 0.0};
    for (int sidx = 0; sidx < HIPNLP_MAX_TERRAIN_STEPS; ++sidx) {   // same order as terrain_Z_jet (unused bumps: zeros)
        const double* part = terrain_com_part(s, sidx);
        for (int i = 0; i < 3; ++i) Z[i] += part[i];
    }
    em.G(gs::COMH, row_id(RK_COMH, 0, 0), s.x[COM_ + 2] - Z[0]);
    em.J(js::COMH_XY + 0, row_id(RK_COMH, 0, 0), COM_ + 0, -Z[1]);
    em.J(js::COMH_XY + 1, row_id(RK_COMH, 0, 0), COM_ + 1, -Z[2]);
    emit_jc(em, js::COMH_Z, row_id(RK_COMH, 0, 0), COM_ + 2, 1.0);
}
template <class Em> HD void t_terrain_stage(Ctx<Em>& cx, int c) {
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    HIPNLP_WAVE_SYNC();   // the bump jets of this wave's t_terrain_bump
    const double* p = s.x + PT_ * c + P_;
    double Z[10];
    for (int i = 0; i < 10; ++i) Z[i] = 0.0;
    auto sum_parts = [&](int cc, double* z) {
        for (int sidx = 0; sidx < HIPNLP_MAX_TERRAIN_STEPS; ++sidx) {   // same order as terrain_Z_jet (unused bumps: zeros)
            const double* part = terrain_bump_part(s, cc, sidx);
            for (int i = 0; i < 10; ++i) z[i] += part[i];
        }
    };
#if defined(__HIP_DEVICE_COMPILE__)
    sum_parts(c, Z);
#else
    // Host (recorders, emulation): the lanes of a task run one after the other.  In the compact layouts the frame of point c
    // overwrites bump jets of points c, c + 1 that the device's lanes have all read by then (one wave, lockstep): read them for
    // every lane when the first lane runs, as the wave does.
    if constexpr (std::remove_reference_t<decltype(s)>::compact) {
        static thread_local double zall[NC][10];
        if (c == 0)
            for (int cc = 0; cc < NC; ++cc) { for (int i = 0; i < 10; ++i) zall[cc][i] = 0.0; sum_parts(cc, zall[cc]); }
        for (int i = 0; i < 10; ++i) Z[i] = zall[c][i];
    } else sum_parts(c, Z);
#endif
    HIPNLP_WAVE_SYNC();   // (compact layouts: the frames below overwrite the bump jets every lane of this wave has just read)
    TerrainStage* st = terrain_stage(s, c);
    terrain_frame(Z, p[2], st->tf);
    for (int i = 0; i < 7; ++i) st->Zh[i] = Z[3 + i];
}
// planar complementarity  v - R_t diag(tau,tau,1) u   (E3)
template <class Em> HD void t_terrain_planar(Ctx<Em>& cx, int c) {
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    Em& em = cx.em;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    const double* v = s.x + cb + V_;
    const double* u = s.x + cb + U_;
    const TerrainFrame& tf = terrain_stage(s, c)->tf;
    const D2 *n = tf.n, *xv = tf.xv, *yv = tf.yv;
    const double kt = cx.gp.kt;
    const D2 tau = d2tanh(tf.h * kt);
    const double dtau_z = kt * (1.0 - tau.v * tau.v);                       // d tau / d p_z
    for (int i = 0; i < 3; ++i) {
        const D2 r = xv[i] * tau * u[0] + yv[i] * tau * u[1] + n[i] * u[2];
        em.G(gb + gs::PLANAR + i, row_id(RK_PLANAR, c, i), v[i] - r.v);
        em.J(jb + js::PLANAR_U + 3 * i + 0, row_id(RK_PLANAR, c, i), cb + U_ + 0, -(xv[i].v * tau.v));
        em.J(jb + js::PLANAR_U + 3 * i + 1, row_id(RK_PLANAR, c, i), cb + U_ + 1, -(yv[i].v * tau.v));
        em.J(jb + js::PLANAR_U + 3 * i + 2, row_id(RK_PLANAR, c, i), cb + U_ + 2, -n[i].v);
        em.J(jb + js::PLANAR_P + 3 * i + 0, row_id(RK_PLANAR, c, i), cb + P_ + 0, -r.x);
        em.J(jb + js::PLANAR_P + 3 * i + 1, row_id(RK_PLANAR, c, i), cb + P_ + 1, -r.y);
        em.J(jb + js::PLANAR_P + 3 * i + 2, row_id(RK_PLANAR, c, i), cb + P_ + 2, -(xv[i].v * u[0] + yv[i].v * u[1]) * dtau_z);
    }
}
// dcc margin  eps - k h (n.f) - [hdot (n.f) + h f.ndot + h (n.fdot)]   (E4)
template <class Em> HD void t_terrain_dcc(Ctx<Em>& cx, int c) {
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    Em& em = cx.em;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    const double* f = s.x + cb + F_;
    const double* v = s.x + cb + V_;
    const double* fd = s.x + cb + FD_;
    const TerrainStage* st = terrain_stage(s, c);
    const TerrainFrame& tf = st->tf;
    const double* Z = st->Zh - 3;   // Z[3..9]
    const D2 &u1 = tf.u1, &u2 = tf.u2, &h = tf.h, &inn = tf.inn;
    const D2* n = tf.n;
    const D2 ud1(-(Z[3] * v[0] + Z[4] * v[1]), -(Z[6] * v[0] + Z[7] * v[1]), -(Z[7] * v[0] + Z[8] * v[1]));  // (d grad h/dp) v
    const D2 ud2(-(Z[4] * v[0] + Z[5] * v[1]), -(Z[7] * v[0] + Z[8] * v[1]), -(Z[8] * v[0] + Z[9] * v[1]));
    const D2 hdot = u1 * v[0] + u2 * v[1] + D2(v[2]);
    const D2 ndu = n[0] * ud1 + n[1] * ud2;                                  // n . udot   (udot_z = 0)
    D2 nd[3] = {(ud1 - n[0] * ndu) * inn, (ud2 - n[1] * ndu) * inn, (-(n[2] * ndu)) * inn};  // ndot = (udot - n (n.udot)) / |grad h|
    const double kbs = cx.gp.kbs;
    const D2 nf = n[0] * f[0] + n[1] * f[1] + n[2] * f[2];
    const D2 nfd = n[0] * fd[0] + n[1] * fd[1] + n[2] * fd[2];
    const D2 fnd = nd[0] * f[0] + nd[1] * f[1] + nd[2] * f[2];
    const D2 margin = D2(cx.gp.eps) - h * nf * kbs - (hdot * nf + h * fnd + h * nfd);
    em.G(gb + gs::DCC, row_id(RK_DCC, c, 0), margin.v);
    em.J(jb + js::DCC_P + 0, row_id(RK_DCC, c, 0), cb + P_ + 0, margin.x);
    em.J(jb + js::DCC_P + 1, row_id(RK_DCC, c, 0), cb + P_ + 1, margin.y);
    em.J(jb + js::DCC_P + 2, row_id(RK_DCC, c, 0), cb + P_ + 2, -kbs * nf.v - (fnd.v + nfd.v));   // only h depends on p_z (dh/dp_z = 1)
    const double gradh[3] = {u1.v, u2.v, 1.0};
    for (int j = 0; j < 3; ++j) {
        em.J(jb + js::DCC_F + j, row_id(RK_DCC, c, 0), cb + F_ + j, -kbs * h.v * n[j].v - hdot.v * n[j].v - h.v * nd[j].v);
        em.J(jb + js::DCC_FD + j, row_id(RK_DCC, c, 0), cb + FD_ + j, -h.v * n[j].v);
        // d/dv_j: hdot = grad h . v ; ndot = (dn/dp) v -> d ndot / d v_j = dn/dp_j (zero for j = z)
        const double dndvj_f = j == 0 ? (n[0].x * f[0] + n[1].x * f[1] + n[2].x * f[2]) : (j == 1 ? (n[0].y * f[0] + n[1].y * f[1] + n[2].y * f[2]) : 0.0);
        em.J(jb + js::DCC_V + j, row_id(RK_DCC, c, 0), cb + V_ + j, -gradh[j] * nf.v - h.v * dndvj_f);
    }
}
// height, normal force, friction cone rows; swing height heuristic (E10):  0.5 [ (h - hd)^2 + |(R_t^T v)_xy|^2 ]   (k >= 1)
template <class Em> HD void t_terrain_hnf(Ctx<Em>& cx, int c) {
    if (terrain_is_planar(cx)) return;
    point_hnf_smooth(cx, c, terrain_stage(cx.s, c)->tf);
}
// swing height heuristic on the smooth terrain (value and gradient; t_foot_costs adds to that gradient: same wave, behind this)
template <class Em> HD void t_terrain_swing(Ctx<Em>& cx, int c) {
    if (terrain_is_planar(cx)) return;
    auto& s = cx.s;
    const int cb = PT_ * c;
    const double* v = s.x + cb + V_;
    const TerrainFrame& tf = terrain_stage(s, c)->tf;
    const D2 *xv = tf.xv, *yv = tf.yv;
    const double on = cx.ki.first ? 0.0 : 1.0;
    const double msw = on * cx.st.m_swing;
    const D2 dh = tf.h - D2(s.pk[PK_REF + R_SWING]);
    const D2 pvx = xv[0] * v[0] + xv[1] * v[1] + xv[2] * v[2], pvy = yv[0] * v[0] + yv[1] * v[1] + yv[2] * v[2];
    const D2 sw = (dh * dh + pvx * pvx + pvy * pvy) * 0.5;
    s.c_pt[c][0] = msw * sw.v;
    double* gr = s.grad + cb;
    gr[P_ + 0] = msw * sw.x; gr[P_ + 1] = msw * sw.y; gr[P_ + 2] = msw * dh.v;
    for (int j = 0; j < 3; ++j) gr[V_ + j] = msw * (pvx.v * xv[j].v + pvy.v * yv[j].v);
}

// --- contact points, scalar rows and cost values: lane c, 8 tasks.  planner.py:656-697, 855-895 --------
template <class Em> HD void t_points_scalar(Ctx<Em>& cx, int c) {
    auto& s = cx.s;
    const double* x = s.x + PT_ * c;
    const int gb = gs::PT_STRIDE * c, jb = js_pt(cx, c), jc = js::ptc(c), cb = PT_ * c;
    Em& em = cx.em;
    const double on = cx.ki.first ? 0.0 : 1.0;
    s.c_pt[c][1] = on * cx.st.m_ureg * (x[U_] * x[U_] + x[U_ + 1] * x[U_ + 1] + x[U_ + 2] * x[U_ + 2]);
    s.c_pt[c][2] = on * cx.st.m_fdreg * (x[FD_] * x[FD_] + x[FD_ + 1] * x[FD_ + 1] + x[FD_ + 2] * x[FD_ + 2]);
    if (!terrain_is_planar(cx)) return;   // smooth terrain: t_terrain_stage / _planar / _dcc / _hnf
    const double pz = x[P_ + 2], fz = x[F_ + 2], vz = x[V_ + 2], fdz = x[FD_ + 2];
    // dcc margin  eps - k h (n.f) - [hdot (n.f) + h f.ndot + h (n.fdot)]   (E4; n = e_z, ndot = 0, hdot = v_z)
    em.G(gb + gs::DCC, row_id(RK_DCC, c, 0), cx.gp.eps - cx.gp.kbs * (pz * fz) - (vz * fz + pz * fdz));
    em.J(jb + js::PL_DCC_P, row_id(RK_DCC, c, 0), cb + P_ + 2, -cx.gp.kbs * fz - fdz);
    em.J(jb + js::PL_DCC_F, row_id(RK_DCC, c, 0), cb + F_ + 2, -cx.gp.kbs * pz - vz);
    em.J(jb + js::PL_DCC_V, row_id(RK_DCC, c, 0), cb + V_ + 2, -fz);
    em.J(jb + js::PL_DCC_FD, row_id(RK_DCC, c, 0), cb + FD_ + 2, -pz);
    point_hnf_planar(cx, c);
    // value of the swing-height cost (k >= 1)
    const double dh = pz - s.pk[PK_REF + R_SWING];
    s.c_pt[c][0] = on * cx.st.m_swing * (0.5 * (dh * dh + (x[V_] * x[V_] + x[V_ + 1] * x[V_ + 1])));
}

// sums of the point-local cost partials: lane = term (swing, u_v, f_dot), runs behind t_points_scalar on the same wave
template <class Em> HD void t_points_cost(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    HIPNLP_WAVE_SYNC();
    double acc = 0.0;
    for (int p = 0; p < NC; ++p) acc += s.c_pt[p][t];
    s.cost[CT_SWING + t] = acc;
}

// identity / zero padding slots of the ancestor and descendant lists, lanes e < 16
template <class S> HD void scratch_padding(S& s, int e) {
    // (the zero slot own[NL] of the descendant lists: t_links — in the compact layouts it lies on joint records that are live here)
    if (e < 16) s.Jr[NJ].d[jr_pos(NJ, e)] = (e < 9 && e % 4 == 0) ? 1.0 : 0.0;   // L = I, of = c = 0, sd = 0
    if (e < 3) { s.aw[NJ][e] = 0.0; if constexpr (S::has_velocities) s.Uj[NJ][e] = 0.0; }
}

// --- trivial dynamics of base / joints / com, lane e over 33 state components.  planner.py:522-564 -----
template <class Em> HD void t_dyn(Ctx<Em>& cx, int e) {
    auto& s = cx.s;
    const double half = 0.5 * cx.gp.dt;
    int L, i, kin, jslot;
    if (e < 3) { i = e; L = 3; kin = RK_PBDYN_IN; jslot = js::PBDYN; }
    else if (e < 7) { i = e - 3; L = 4; kin = RK_QBDYN_IN; jslot = js::QBDYN; }
    else if (e < 7 + NJ) { i = e - 7; L = NJ; kin = RK_SDYN_IN; jslot = js::SDYN; }
    else { i = e - 7 - NJ; L = 3; kin = RK_COMDYN_IN; jslot = js::COMDYN; }
    const int kout = kin + 1, kx0 = kin + 2;
    // The four state blocks are neighbours in the knot record and in the native g slots: the variable, its rate and the two g slots of lane
    // e are e plus a constant (two for the variable, three for the rate) — what the VARY kernels, whose constant entries are gone, have left
    // of this task is four reads, three operations and two stores, and the chain above (kept for the row ids and the constant entries' slots)
    // was a dozen selects in front of them.
    static_assert(QB_ == PB_ + 3 && COM_ == S_ + NJ && QD_ == VB_ + 3, "state blocks of t_dyn");
    static_assert(gs::QBDYN == gs::PBDYN + 3 && gs::SDYN == gs::PBDYN + 7 && gs::COMDYN == gs::PBDYN + 7 + NJ, "g slots of t_dyn");
    static_assert(gs::QB_X0 == gs::PB_X0 + 3 && gs::S_X0 == gs::PB_X0 + 7 && gs::COM_X0 == gs::PB_X0 + 7 + NJ, "x0 slots of t_dyn");
    const int X = e + (e < 7 ? PB_ : S_ - 7);
    const int Y = e + (e < 7 ? VB_ : (e < 7 + NJ ? SD_ - 7 : H_ - 7 - NJ));
    Em& em = cx.em;
    // padding slots of the ancestor / descendant lists (identity rotation, zero terms), one element per lane
    scratch_padding(s, e);
    em.G(gs::PBDYN + e, row_id(kin, 0, i), s.x[X] - (s.xm[X] + half * (s.xm[Y] + s.x[Y])));
    em.G(gs::PB_X0 + e, row_id(kx0, 0, i), s.x[X]);
    emit_jc(em, jslot + 0 * L + i, row_id(kin, 0, i), X, 1.0);
    emit_jc(em, jslot + 1 * L + i, row_id(kin, 0, i), Y, -half);
    emit_jc(em, jslot + 2 * L + i, row_id(kout, 0, i), X, -1.0);
    emit_jc(em, jslot + 3 * L + i, row_id(kout, 0, i), Y, -half);
    emit_jc(em, jslot + 4 * L + i, row_id(kx0, 0, i), X, 1.0);
}

// parent_R_child = R_fix * (cq (I - a a^T) + sq [a]x + a a^T)   (adam R_from_axis_angle) -> the joint record s.Jr[j]
// (a rotation about the joint axis leaves the axis in place: parent_R_child axis = R_fix axis, a constant)
template <class Em> HD void joint_transform(Ctx<Em>& cx, int j) {
    auto& s = cx.s;
    const double* a = kin_axis(cx, j);
    double sq, cq;
    sincos(s.x[S_ + j], &sq, &cq);
    double Ra[9];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) { const double aa = a[r] * a[cc]; Ra[3 * r + cc] = cq * ((r == cc ? 1.0 : 0.0) - aa) + aa; }
    Ra[1] -= sq * a[2]; Ra[2] += sq * a[1];
    Ra[3] += sq * a[2]; Ra[5] -= sq * a[0];
    Ra[6] -= sq * a[1]; Ra[7] += sq * a[0];
    double L[9], c[3];
    matmul3(kin_R_fix(cx, j), Ra, L);
    matvec3(kin_R_fix(cx, j), a, c);
    double* rec = s.Jr[j].d;
    for (int e = 0; e < 9; ++e) rec[JR_L + e] = L[e];
    for (int r = 0; r < 3; ++r) { rec[JR_OF + r] = kin_o_fix(cx, j)[r]; rec[JR_C + r] = c[r]; }
    rec[JR_SD] = s.x[SD_ + j];
}

// --- joint-wise rows, joint regularisation cost and the local joint transform, lane j (23) -------------
template <class Em> HD void t_joint_rows(Ctx<Em>& cx, int j) {
    auto& s = cx.s;
    Em& em = cx.em;
    em.G(gs::JPB + j, row_id(RK_JPB, 0, j), s.x[S_ + j]);
    emit_jc(em, js::JPB + j, row_id(RK_JPB, 0, j), S_ + j, 1.0);
    em.G(gs::JVB + j, row_id(RK_JVB, 0, j), s.x[SD_ + j]);
    emit_jc(em, js::JVB + j, row_id(RK_JVB, 0, j), SD_ + j, 1.0);
    // joint_positions_error  planner.py:505-520
    const double on = cx.ki.first ? 0.0 : 1.0;
    const double m = on * cx.st.m_jreg, w = cx.st.w_jreg[j];
    const double sd = s.x[SD_ + j];
    const double t = sd + w * (s.x[S_ + j] - s.pk[PK_REF + R_JREG + j]);
    double c = t * t, gsd = 2.0 * t;
    if (cx.st.joint_reg_as_coded) { c += double(NJ - 1) * (sd * sd); gsd += 2.0 * double(NJ - 1) * sd; }
    s.c_joint[j] = m * c;
    s.grad[S_ + j] = 2.0 * m * t * w;
    s.grad[SD_ + j] = m * gsd;
}
// the local joint transforms (records of the FK pass), on a wave of their own: the longest task of the first phase
template <class Em> HD void t_joints(Ctx<Em>& cx, int j) { joint_transform(cx, j); }

template <class Em> HD void t_joint_cost(Ctx<Em>& cx, int) {  // behind t_joints on the same wave
    auto& s = cx.s;
    HIPNLP_WAVE_SYNC();
    double acc = 0.0;
    for (int q = 0; q < NJ; ++q) acc += s.c_joint[q];
    s.cost[CT_JREG] = acc;
}

// --- unitary quaternion row (planner.py:276-282) + base quaternion error cost (E13, raw q), 1 task ------
template <class Em> HD void t_unitq(Ctx<Em>& cx, int) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double on = cx.ki.first ? 0.0 : 1.0;
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
}

// --- small global rows / costs: lanes 0..2 component-wise, lane 3 scalar: 4 tasks ---------------------------
template <class Em> HD void t_small(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    if (t < 3) {  // angular momentum bound rows h[3:]*mass (planner.py:342-350); gradient of the com velocity cost
        const int i = t;
        em.G(gs::AMB + i, row_id(RK_AMB, 0, i), s.x[H_ + 3 + i] * cx.gp.mass);
        emit_jc(em, js::AMB + i, row_id(RK_AMB, 0, i), H_ + 3 + i, cx.gp.mass);
        const double e = s.x[H_ + i] - s.pk[PK_REF + R_VREF + i];
        s.grad[H_ + i] = 2.0 * cx.st.m_comvel * cx.st.w_comvel[i] * e;
        s.grad[H_ + 3 + i] = 0.0;
        s.grad[COM_ + i] = 0.0; s.grad[PB_ + i] = 0.0; s.grad[VB_ + i] = 0.0;
    } else {      // minimum com height: h_terrain(com) = com_z ; com velocity and base quaternion velocity costs (k >= 0)
        if (terrain_is_planar(cx)) {   // (smooth terrain: t_com_height, behind the bump jets)
            em.G(gs::COMH, row_id(RK_COMH, 0, 0), s.x[COM_ + 2]);
            emit_jc(em, js::COMH_Z, row_id(RK_COMH, 0, 0), COM_ + 2, 1.0);
        }
        double c = 0.0;
        for (int i = 0; i < 4; ++i) {
            const double e = s.x[QD_ + i] - s.pk[PK_REF + R_BQV + i];
            c += e * e;
            s.grad[QD_ + i] = 2.0 * cx.st.m_baseqv * e;
        }
        s.cost[CT_BASEQV] = cx.st.m_baseqv * c;
        double cv = 0.0;
        for (int i = 0; i < 3; ++i) { const double e = s.x[H_ + i] - s.pk[PK_REF + R_VREF + i]; cv += e * cx.st.w_comvel[i] * e; }
        s.cost[CT_COMVEL] = cx.st.m_comvel * cv;
    }
}

// --- feet: lane 0 centroids (relative height row + centroid cost, planner.py:215-264);
//           lanes 1,2 yaw alignment errors of the left / right foot (E9, planner.py:773-853): 3 tasks -------
template <class Em> HD void t_feet(Ctx<Em>& cx, int t) {   // t = 1, 2: yaw of the left / right foot
    auto& s = cx.s;
    const double on = cx.ki.first ? 0.0 : 1.0;
    {
        const int foot = t - 1;   // (t = 0 was the centroid lane: t_feet_centroid below)
        const double yaw = s.pk[PK_REF + (foot == 0 ? R_YAW_L : R_YAW_R)];
        const int br = 4 * foot + cx.st.yaw_corner[foot][0], tr = 4 * foot + cx.st.yaw_corner[foot][1], tl = 4 * foot + cx.st.yaw_corner[foot][2];
        // sin/cos of the yaw reference and of yaw + pi/2: parameters only, precomputed by pack_params
        const double* sc = s.pk + PK_YAWSC + 4 * foot;
        const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
        (void)yaw;
        const double* pbr = s.x + PT_ * br + P_;
        const double* ptr = s.x + PT_ * tr + P_;
        const double* ptl = s.x + PT_ * tl + P_;
        const double ef = -s1 * (ptr[0] - pbr[0]) + c1 * (ptr[1] - pbr[1]);
        const double es = -s2 * (ptl[0] - ptr[0]) + c2 * (ptl[1] - ptr[1]);
        s.yaw_e[foot][0] = ef; s.yaw_e[foot][1] = es;
        s.c_yaw[foot] = on * cx.st.m_yaw * (0.5 * (ef * ef + es * es));
    }
}

// the two branches of t_feet as task groups of their own, so that they can sit on different waves of phase A
// centroids: lane i < 3 the component i of both centroids (relative height row on lane 2), its share of the centroid cost and of the
// cost's gradient; lane c < 8 the constant entry of the height row in p_c,z.  (One lane for all of it until round 6: 24 reads and three
// error terms one after the other, 1.7 - 1.9 k cycles at batch, as long as the base task it shares a wave with in the four-wave kernels.)
// The three cost shares wait in chest_w[] — written by t_frames two phases later — for t_foot_cost_sum in the second phase.
constexpr int FEET_CENTROID_TASKS = NC;
template <class S> HD double* centroid_cost_parts(S& s) { return s.chest_w; }
template <class Em> HD void t_feet_centroid(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    emit_jc(em, js::FEETH + t, row_id(RK_FEETH, 0, 0), PT_ * t + P_ + 2, t < 4 ? 0.25 : -0.25);
    if (t >= 3) return;
    const int i = t;
    const double on = cx.ki.first ? 0.0 : 1.0;
    double cl = 0.0, cr = 0.0;
    for (int c = 0; c < 4; ++c) { cl += s.x[PT_ * c + P_ + i]; cr += s.x[PT_ * (c + 4) + P_ + i]; }
    cl = cl / 4.0; cr = cr / 4.0;
    if (i == 2) em.G(gs::FEETH, row_id(RK_FEETH, 0, 0), cl - cr);
    const double m = on * cx.st.m_centroid;
    const double e = s.pk[PK_REF + R_CREF + i] - 0.5 * (cl + cr);
    const double w = s.pk[PK_REF + R_CW + i];
    centroid_cost_parts(s)[i] = e * w * e;
    s.cen_g[i] = -0.25 * m * w * e;  // 2 w e * d e / d p_c,i = 2 w e (-0.5/4)
}
template <class Em> HD void t_feet_yaw(Ctx<Em>& cx, int foot) { t_feet(cx, 1 + foot); }

// --- base orientation: normalised quaternion (E11), R_b, G, omega (E12), d omega / d q_b: lane e = row (3 tasks) ---------
template <class Em> HD void t_base(Ctx<Em>& cx, int e) {
    // lane e works on row e with the cyclic index triple (e, e1, e2): every entry of R, G, H has one closed form in
    // (a, b, c) = (q_e, q_e1, q_e2), so no lane indexes a register array with a run-time index
    auto& s = cx.s;
    const double* q = s.x + QB_;
    const double* qd = s.x + QD_;
    const int e1 = e == 2 ? 0 : e + 1, e2 = e == 0 ? 2 : e - 1;
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double inv_n = inv_sqrt(n2);
    const double a = q[e] * inv_n, b = q[e1] * inv_n, c = q[e2] * inv_n, w = q[3] * inv_n;
    if (e == 0) { s.qnorm = n2 * inv_n; s.inv_qnorm = inv_n; s.qn[3] = w; }
    s.qn[e] = a;
    // row e of R = I + 2 w [v]x + 2 [v]x^2
    const double Ree = 1.0 - 2.0 * (b * b + c * c), Re1 = 2.0 * (a * b - w * c), Re2 = 2.0 * (a * c + w * b);
    s.Rb[3 * e + e] = Ree; s.Rb[3 * e + e1] = Re1; s.Rb[3 * e + e2] = Re2;
    s.Rw[0][3 * e + e] = Ree; s.Rw[0][3 * e + e1] = Re1; s.Rw[0][3 * e + e2] = Re2;
    // G = 2 [ w I + [v]x | -v ] ; omega = G qdot ; H = 2 [ -qd_w I - [qd_v]x | qd_v ] ; d omega/d q = H (I - qn qn^T)/|q|
    const double Ge = 2.0 * w, Ge1 = -2.0 * c, Ge2 = 2.0 * b, G3 = -2.0 * a;
    const double He = -2.0 * qd[3], He1 = 2.0 * qd[e2], He2 = -2.0 * qd[e1], H3 = 2.0 * qd[e];
    const double om = Ge * qd[e] + Ge1 * qd[e1] + Ge2 * qd[e2] + G3 * qd[3];
    const double hq = He * a + He1 * b + He2 * c + H3 * w;
    double* Gr = s.G + 4 * e;
    double* Dr = s.dwq + 4 * e;
    Gr[e] = Ge; Gr[e1] = Ge1; Gr[e2] = Ge2; Gr[3] = G3;
    Dr[e] = (He - hq * a) * inv_n; Dr[e1] = (He1 - hq * b) * inv_n; Dr[e2] = (He2 - hq * c) * inv_n; Dr[3] = (H3 - hq * w) * inv_n;
    s.omega[e] = om;
    s.ow[0][e] = 0.0;   // root link in base-centred coordinates
    if constexpr (std::remove_reference_t<decltype(s)>::has_velocities) { s.wv[0][e] = om; s.vo[0][e] = 0.0; }
}

// ===================================================================================================
// PHASE B — wave 0: forward kinematics + link velocities, ONE LANE PER ROOT->LEAF CHAIN (the chain state
// stays in registers; shared prefixes are recomputed, identical values are stored twice).
// Other waves meanwhile: momentum-dynamics rows, foot costs, horizon-end rows.
// ===================================================================================================
// Forward kinematics and link velocities as ANCESTOR SUMS: lane (joint j, component r), 69 tasks per pass.
// Every lane walks its own (padded) ancestor list, so no lane waits for another one inside a pass; the passes
// run back to back on one wave (HIPNLP_WAVE_SYNC between them).
// Emitters of a STATIC problem (Em::kStatic: the pose finder's device kernels — every velocity of the knot record is zero): the kinematic
// tasks leave out what only the velocities feed — link angular velocities and origin velocities, link / subtree momenta, the
// velocity-product columns — and what reads them.  No value the pose finder emits depends on any of it (the host recorders and the
// emulation run the full tasks on the zero velocities: same outputs).
template <class Em, class = void> struct em_static_t { static constexpr bool value = false; };
template <class Em> struct em_static_t<Em, std::void_t<decltype(Em::kStatic)>> { static constexpr bool value = Em::kStatic; };
template <class Em> constexpr bool em_static = em_static_t<Em>::value;
constexpr int FK_TASKS_A = 3 * FK_SPLIT, FK_TASKS_B = 3 * (NJ - FK_SPLIT);
template <class Em> HD void t_fk_rot_at(Ctx<Em>& cx, int t, int q0) {
    // lane (joint j, component r) walks the front-padded ancestor list of j ONCE and accumulates, for its component r:
    //   row r of R_j = row r of R_b L_a1 ... L_j        o_j[r] = sum_a (row r of R_parent(a)) . o_fix_a
    //   a_j[r] = (row r of R_j) . axis_j                w_j[r] = omega[r] + sum_a a_a[r] sdot_a
    // (padding slot NJ: L = I, o_fix = c = sdot = 0)
    auto& s = cx.s;
    const int j = t / 3, r = t - 3 * j;
    double v0 = s.Rb[3 * r], v1 = s.Rb[3 * r + 1], v2 = s.Rb[3 * r + 2];
    double o = 0.0, w = s.omega[r], ar = 0.0;
    // q0: every list of this task group is padded up to there (group-uniform: the arithmetic of the leading identity steps is skipped).
    // The record of step q + 1 is read while step q is computed (two register buffers, static step numbers): a step is one LDS round
    // trip PLUS three dependent levels of fp64 arithmetic otherwise, ~360 cycles, eight times over on the longest chain of phase B.
    // (the eight ancestor indices stay packed in the 8-byte word they are stored as: two registers instead of eight)
    const unsigned long long ia_w = *reinterpret_cast<const unsigned long long*>(cx.kt.anc[j]);
    double rec[2][16];   // the eight 16-byte chunks of a record
    auto fetch = [&](int q, double* rc) {   // (one multiply for the record's address, the eight chunks at immediate offsets)
        const double* rp = s.Jr[int((ia_w >> (8 * q)) & 0xffull)].d;
        HIPNLP_UNROLL
        for (int e = 0; e < 16; ++e) rc[e] = rp[e];
    };
    // (the records of the leading identity steps are not read either: eight 16-byte reads — 32 LDS-array cycles — per skipped step, and the
    //  LDS array is the busiest unit of the batch launches beside the VALU; q0 is group-uniform: scalar branches)
    if (q0 == 0) fetch(0, rec[0]);
    HIPNLP_UNROLL
    for (int q = 0; q < 8; ++q) {
        if (q + 1 < 8 && q + 1 >= q0) fetch(q + 1, rec[(q + 1) & 1]);
        if (q < q0) continue;
        const double* rq = rec[q & 1];
        const double* L = rq + JR_L;
        const double* of = rq + JR_OF;
        const double* cc = rq + JR_C;
        o += v0 * of[0] + v1 * of[1] + v2 * of[2];
        ar = v0 * cc[0] + v1 * cc[1] + v2 * cc[2];   // (row r of R_parent) . (R_fix axis) = (row r of R_a) . axis
        if constexpr (!em_static<Em>) w += ar * rq[JR_SD];
        const double n0 = v0 * L[0] + v1 * L[3] + v2 * L[6];
        const double n1 = v0 * L[1] + v1 * L[4] + v2 * L[7];
        const double n2 = v0 * L[2] + v1 * L[5] + v2 * L[8];
        v0 = n0; v1 = n1; v2 = n2;
    }
    // (the store addresses are derived from the task number AGAIN, behind a wall the compiler cannot see through: kept from the top of the
    //  task, they are live across the whole walk — 64 registers of records in flight — and the five-per-CU kernels, at 96 registers, spilled one)
    int t2 = t;
    HIPNLP_OPAQUE(t2);
    const int j2 = t2 / 3, r2 = t2 - 3 * j2;
    s.Rw[j2 + 1][3 * r2] = v0; s.Rw[j2 + 1][3 * r2 + 1] = v1; s.Rw[j2 + 1][3 * r2 + 2] = v2;
    s.aw[j2][r2] = ar;
    s.ow[j2 + 1][r2] = o;
    if constexpr (!em_static<Em>) s.wv[j2 + 1][r2] = w;
}
template <class Em> HD void t_fk_rot_a(Ctx<Em>& cx, int t) { t_fk_rot_at(cx, t, HIPNLP_UNIFORM(cx.kt.fk_first[0])); }
template <class Em> HD void t_fk_rot_b(Ctx<Em>& cx, int t) { t_fk_rot_at(cx, t + FK_TASKS_A, HIPNLP_UNIFORM(cx.kt.fk_first[1])); }
// (Measured alternative: FK as a parallel prefix over the ancestor chains — pointer doubling with affine elements (L, of, wl),
//  three steps + the base element instead of up to seven ancestor steps, ping-pong between the joint records and comp[].  Correct
//  (all GPU tests), phase B 3.5 k -> 2.9 k cycles, +1.3 % at batch 64, but -2.6 % on the 100-knot step, whose duration is set by
//  the workgroup that finishes the cost reduction; not kept.)
// (Round 2 repeated it in place on the joint records, two lanes per joint — lane (j, 0): two columns of P and t, lane (j, 1): the
//  third column, v and u = c_j; 46 lanes of ONE wave, a third of the LDS reads and half the fp64 instructions of the ancestor sums;
//  0 scratch at 104-128 VGPRs.  Correct (65 GPU tests) and again not kept: 8.21 us against 7.88 on the 100-knot launch, -3 % at
//  batch 4, +-0.5 % at x 64 / x 1024 / stairs 200 x 16: three DEPENDENT rounds of load -> 9 fma levels -> store -> wave fence are a
//  longer chain than eight pipelined ancestor reads, and at batch the saved LDS reads were not what the launch waited for.)
// (A level-synchronous variant — one step per joint, the wave walking the tree level by level behind wave-level fences — was
//  measured at 4.7 k cycles against 3.4 k for these ancestor sums: a level costs two dependent LDS round trips, ~680 cycles.)
// velocity of the body point of link i at the (base-centred) origin: vO_i = sum over the joints a on the path root -> i of
// U_a = (o_a x a_a) sdot_a.  t_link_u (lane j) forms U_j, t_links (same wave, behind it) sums the ancestors' (padding: U_NJ = 0)
template <class Em> HD void t_link_u(Ctx<Em>& cx, int j) {
    if constexpr (em_static<Em>) { (void)cx; (void)j; return; }
    auto& s = cx.s;
    HIPNLP_WAVE_SYNC();   // o_j, a_j: components written by three lanes of this wave (t_fk_rot_a / _b)
    double u[3];
    cross3(s.ow[j + 1], s.aw[j], u);
    const double sd = s.x[SD_ + j];
    for (int r = 0; r < 3; ++r) s.Uj[j][r] = u[r] * sd;
}
template <class Em> HD void t_link_u_a(Ctx<Em>& cx, int j) { t_link_u(cx, j); }              // behind t_fk_rot_a on its wave
template <class Em> HD void t_link_u_b(Ctx<Em>& cx, int j) { t_link_u(cx, j + FK_SPLIT); }   // behind t_fk_rot_b on its wave
template <class S, class K> HD void link_origin_velocity(const S& s, const K& kt, int i, double* v) {
    v[0] = v[1] = v[2] = 0.0;
    if (i == 0) return;
    double u[8][3];
    const unsigned long long aw8 = *reinterpret_cast<const unsigned long long*>(kt.anc[i - 1]);   // (the eight indices as the one word they are stored as)
    HIPNLP_UNROLL
    for (int q = 0; q < 8; ++q) {
        const double* uq = s.Uj[int((aw8 >> (8 * q)) & 0xffull)];
        for (int r = 0; r < 3; ++r) u[q][r] = uq[r];
    }
    HIPNLP_ISSUE_FENCE();   // all 24 reads in flight before the first add
    HIPNLP_UNROLL
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 3; ++r) v[r] += u[q][r];
}

// (the branches of t_hdyn diverge inside a wave: entries and rows + com entries are task groups of their own)
template <class Em> HD void t_hdyn(Ctx<Em>& cx, int t);
// The entry tasks depend on the knot record only.  In the four-wave kernel of the SMOOTH terrain the first phase is set by one wave
// (bump jets -> terrain frames) while the others wait, so they run there (t_hdyn_entries_a) instead of in the second phase; every
// other instantiation (emitters without kWaves: host recorders / emulation; the planar and the eight-wave kernels) runs them in
// the second phase.  Exactly one of the two task groups does the work.
template <class Em, class = void> struct em_waves { static constexpr int value = 0; };
template <class Em> struct em_waves<Em, std::void_t<decltype(Em::kWaves)>> { static constexpr int value = Em::kWaves; };
template <class Em> constexpr bool hdyn_entries_early = em_waves<Em>::value == 4 && Em::kTerrain == HIPNLP_TERRAIN_SMOOTH_STEPS;
template <class Em> HD void t_hdyn_entries_a(Ctx<Em>& cx, int t) { if constexpr (hdyn_entries_early<Em>) t_hdyn(cx, t); }
template <class Em> HD void t_hdyn_entries(Ctx<Em>& cx, int t) { if constexpr (!hdyn_entries_early<Em>) t_hdyn(cx, t); }
// The six rows sum the per-point shares t_points_vec left in hd[]: behind it on ITS wave in the first phase where the entries run
// early (four-wave kernel on the smooth terrain: that phase waits for the terrain chain of another wave anyway, while the second phase
// was set by the wave that carried these rows behind its half of the forward kinematics — 6.8 k against 4.0 - 4.8 k cycles under
// load, profiles/r03_callback_stamps_stairs_B32.txt); in the second phase everywhere else.  Exactly one of the two groups works.
template <class Em> HD void t_hdyn_rows_a(Ctx<Em>& cx, int t) {
    if constexpr (hdyn_entries_early<Em>) {
        HIPNLP_WAVE_SYNC();   // hd[] of this wave's t_points_vec
        t_hdyn(cx, t + 48);
    }
}
template <class Em> HD void t_hdyn_rows(Ctx<Em>& cx, int t) { if constexpr (!hdyn_entries_early<Em>) t_hdyn(cx, t + 48); }
// --- centroidal momentum dynamics (T7 on E1): lanes (c, e) 48 entry tasks + lanes 48..59 row tasks -----------
constexpr int HDYN_TASKS = 60;
template <class Em> HD void t_hdyn(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double half = 0.5 * cx.gp.dt;
    if (t < 48) {
        const int c = t / 6, e = t - 6 * c, cb = PT_ * c;
        double rc[3];
        for (int i = 0; i < 3; ++i) rc[i] = s.x[cb + P_ + i] - s.x[COM_ + i];
        const int row = 3 + cross_row(e), col = cross_col(e);
        const double vp = half * skew_entry(s.x + cb + F_, e);   // -half * d[(p-com) x f]/dp = -half * (-[f]x)
        const double vf = -half * skew_entry(rc, e);             // -half * [p-com]x
        em.J(js::HDYN_ANG_P_IN + 6 * c + e, row_id(RK_HDYN_IN, 0, row), cb + P_ + col, vp);
        em.J(js::HDYN_ANG_P_OUT + 6 * c + e, row_id(RK_HDYN_OUT, 0, row), cb + P_ + col, vp);
        em.J(js::HDYN_ANG_F_IN + 6 * c + e, row_id(RK_HDYN_IN, 0, row), cb + F_ + col, vf);
        em.J(js::HDYN_ANG_F_OUT + 6 * c + e, row_id(RK_HDYN_OUT, 0, row), cb + F_ + col, vf);
        if (e < 3) {
            emit_jc(em, js::HDYN_LIN_F_IN + 3 * c + e, row_id(RK_HDYN_IN, 0, e), cb + F_ + e, -half);
            emit_jc(em, js::HDYN_LIN_F_OUT + 3 * c + e, row_id(RK_HDYN_OUT, 0, e), cb + F_ + e, -half);
        }
    } else if (t < 54) {  // the six rows
        const int i = t - 48;
        double h0 = cx.gp.gravity[i], h1 = cx.gp.gravity[i];
        for (int c = 0; c < NC; ++c) { h0 += hd_of(s)[c * 6 + i]; h1 += hd_of(s)[(NC + c) * 6 + i]; }
        em.G(gs::HDYN + i, row_id(RK_HDYN_IN, 0, i), s.x[H_ + i] - (s.xm[H_ + i] + half * (h0 + h1)));
        em.G(gs::H_X0 + i, row_id(RK_HDYN_X0, 0, i), s.x[H_ + i] - s.xg[i]);
        emit_jc(em, js::HDYN_SELF_IN + i, row_id(RK_HDYN_IN, 0, i), H_ + i, 1.0);
        emit_jc(em, js::HDYN_SELF_OUT + i, row_id(RK_HDYN_OUT, 0, i), H_ + i, -1.0);
        emit_jc(em, js::HDYN_X0 + i, row_id(RK_HDYN_X0, 0, i), H_ + i, 1.0);
        emit_jc(em, js::HDYN_X0G + i, row_id(RK_HDYN_X0, 0, i), COL_GLOBAL + i, -1.0);
    } else {              // d/dcom sum (p - com) x f = [sum f]x
        const int e = t - 54;
        double fs[3] = {0.0, 0.0, 0.0};
        for (int c = 0; c < NC; ++c) for (int i = 0; i < 3; ++i) fs[i] += s.x[PT_ * c + F_ + i];
        const double v = -half * skew_entry(fs, e);
        em.J(js::HDYN_ANG_COM_IN + e, row_id(RK_HDYN_IN, 0, 3 + cross_row(e)), COM_ + cross_col(e), v);
        em.J(js::HDYN_ANG_COM_OUT + e, row_id(RK_HDYN_OUT, 0, 3 + cross_row(e)), COM_ + cross_col(e), v);
    }
}

// --- foot costs (k >= 1): lanes (foot, i) 6: force-ratio regularisation; lanes 6..29 (c, i): gradient of the
//     centroid and yaw costs on p.  planner.py:746-853, 249-264 ----------------------------------------------
constexpr int FOOT_TASKS = 30;
template <class Em> HD void t_foot_costs(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    HIPNLP_WAVE_SYNC();   // smooth terrain: the swing-cost gradient of the same wave's t_terrain_hnf is accumulated into below
    const double on = cx.ki.first ? 0.0 : 1.0;
    if (t < 6) {
        const int foot = t / 3, i = t - 3 * foot;
        const double mf = on * cx.st.m_freg;
        const double* alpha = s.pk + PK_REF + (foot == 0 ? R_ALPHA_L : R_ALPHA_R);
        double sum = 0.0, e[4], ae = 0.0, cost = 0.0;
        for (int c = 0; c < 4; ++c) sum += s.x[PT_ * (4 * foot + c) + F_ + i];
        for (int c = 0; c < 4; ++c) { e[c] = s.x[PT_ * (4 * foot + c) + F_ + i] - alpha[c] * sum; cost += e[c] * e[c]; ae += alpha[c] * e[c]; }
        for (int c = 0; c < 4; ++c) s.grad[PT_ * (4 * foot + c) + F_ + i] += 2.0 * mf * (e[c] - ae);
        s.c_force[foot][i] = mf * cost;
    } else {
        const int q = t - 6, c = q / 3, i = q - 3 * c, foot = c / 4, cl = c - 4 * foot;
        double gsum = s.cen_g[i];
        if (i < 2) {
            const double my = on * cx.st.m_yaw;
            const double ef = s.yaw_e[foot][0], es = s.yaw_e[foot][1];
            const double* sc = s.pk + PK_YAWSC + 4 * foot;   // sin / cos of the yaw reference and of yaw + pi/2 (pack_params)
            const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
            // d ef: (p_tr - p_br) . (-s1, c1) ; d es: (p_tl - p_tr) . (-s2, c2)
            const double df = i == 0 ? -s1 : c1, ds = i == 0 ? -s2 : c2;
            if (cl == cx.st.yaw_corner[foot][0]) gsum += -my * ef * df;
            if (cl == cx.st.yaw_corner[foot][1]) gsum += my * ef * df - my * es * ds;
            if (cl == cx.st.yaw_corner[foot][2]) gsum += my * es * ds;
        }
        s.grad[PT_ * c + P_ + i] += gsum;
    }
}

template <class Em> HD void t_foot_cost_sum(Ctx<Em>& cx, int t) {  // behind t_foot_costs (force partials) / t_feet (yaw)
    auto& s = cx.s;
    HIPNLP_WAVE_SYNC();
    if (t == 0) s.cost[CT_FREG] = ((s.c_force[0][0] + s.c_force[0][1]) + s.c_force[0][2]) + ((s.c_force[1][0] + s.c_force[1][1]) + s.c_force[1][2]);
    else if (t == 1) s.cost[CT_YAW] = s.c_yaw[0] + s.c_yaw[1];
    else {   // the centroid cost from the three shares t_feet_centroid left (its own order of the sum: component 0, 1, 2)
        const double on = cx.ki.first ? 0.0 : 1.0;
        const double* cp = centroid_cost_parts(s);
        s.cost[CT_CENTROID] = (on * cx.st.m_centroid) * ((cp[0] + cp[1]) + cp[2]);
    }
}

// --- horizon-end rows (final state planner.py:407-425, periodicity :897-930), lanes over rows ----------------
constexpr int ENDS_TASKS = 105 + 84;
template <class Em> HD void t_ends(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    if (!cx.ki.first && !cx.ki.last) return;
    s.ends.c[t] = 0.0;
    s.ends.g[t] = 0.0;
    if (t < 105) {
        if (!cx.ki.last) return;
        const int var = end_tables(cx).fin_var[t];
        const double lhs = var >= 0 ? s.x[var] : s.pk[PK_DESC + end_tables(cx).fin_desc[t]];
        if (cx.st.final_type == HIPNLP_EXPR_MINIMIZE) {
            const double e = lhs - final_rhs(cx)[t];
            s.ends.c[t] = cx.st.final_weight * e * e;
            s.ends.g[t] = 2.0 * cx.st.final_weight * e;
        } else if (cx.st.final_type == HIPNLP_EXPR_SUBJECT_TO) {
            emit_g_end(cx, gs::FIN + t, row_id(RK_FIN, 0, t), lhs);
            if (var >= 0) emit_jc(em, js::FIN + end_tables(cx).fin_slot[t], row_id(RK_FIN, 0, t), var, 1.0);
        }
    } else {
        const int i = t - 105;
        const int var = end_tables(cx).per_var[i];
        if (cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE) {
            if (!cx.ki.first && !cx.ki.last) return;
            // e = x_0 - x_{N-1};  at the last knot xo = x_0, at the first knot xo = x_{N-1}
            const double xoi = per_other(cx, i);
            const double e = cx.ki.last ? (xoi - s.x[var]) : (s.x[var] - xoi);
            if (cx.ki.last) s.ends.c[t] = cx.st.periodicity_weight * e * e;
            s.ends.g[t] = (cx.ki.last ? -2.0 : 2.0) * cx.st.periodicity_weight * e;  // d/dx_{N-1} = -2we, d/dx_0 = +2we
        } else if (cx.st.periodicity_type == HIPNLP_EXPR_SUBJECT_TO) {
            if (cx.ki.last) {
                emit_g_end(cx, gs::PER + i, row_id(RK_PERN, 0, i), per_other(cx, i) - s.x[var]);
                emit_jc(em, js::PERN + i, row_id(RK_PERN, 0, i), var, -1.0);
            }
            if (cx.ki.first) emit_jc(em, js::PER0 + i, row_id(RK_PER0, 0, i), var, 1.0);
        }
    }
}

// ===================================================================================================
// PHASE C — per-link spatial inertia at the origin and link momentum, lane i (24 links); frames (3 lanes)
// ===================================================================================================
// t_links (lane i): first moment, origin velocity and momentum of the link;  t_link_inertia (lane i, another wave): rotational
// inertia about the origin.  The momentum takes I_O w as R (I (R^T w)) + m (c^2 w - c (c.w)), so neither waits for the other.
template <class Em> HD void t_links(Ctx<Em>& cx, int i) {
    auto& s = cx.s;
    const double m = kin_mass(cx, i);
    if (i < 16) s.own[NL][i] = 0.0;   // the zero slot of the descendant lists (scratch_padding)
    if constexpr (em_static<Em>) {    // mass and first moment only (the momenta are zero and nothing reads them)
        const double* R = s.Rw[i];
        double c[3];
        matvec3(R, kin_com(cx, i), c);
        double* cp = s.own[i];
        cp[CM] = m;
        for (int r = 0; r < 3; ++r) cp[CH + r] = m * (c[r] + s.ow[i][r]);
        return;
    }
    double vo[3];
    link_origin_velocity(s, cx.kt, i, vo);
    for (int r = 0; r < 3; ++r) s.vo[i][r] = vo[r];
    const double* R = s.Rw[i];
    const double* w = s.wv[i];
    double c[3], h[3], wl[3], Iwl[3], Iw[3];
    matvec3(R, kin_com(cx, i), c);
    for (int r = 0; r < 3; ++r) c[r] += s.ow[i][r];
    for (int r = 0; r < 3; ++r) wl[r] = R[r] * w[0] + R[3 + r] * w[1] + R[6 + r] * w[2];   // R^T w
    matvec3(kin_inertia(cx, i), wl, Iwl);
    matvec3(R, Iwl, Iw);
    const double c2 = dot3(c, c), cw = dot3(c, w);
    double* cp = s.own[i];
    cp[CM] = m;
    for (int r = 0; r < 3; ++r) { h[r] = m * c[r]; cp[CH + r] = h[r]; }
    // link momentum about the origin:  lin = m vO + w x h ;  ang = I_O w + h x vO
    double a[3], b[3];
    cross3(w, h, a);
    for (int r = 0; r < 3; ++r) cp[CKL + r] = m * vo[r] + a[r];
    cross3(h, vo, b);
    for (int r = 0; r < 3; ++r) cp[CKA + r] = Iw[r] + m * (c2 * w[r] - c[r] * cw) + b[r];
}
template <class Em> HD void t_link_inertia(Ctx<Em>& cx, int i) {
    auto& s = cx.s;
    const double m = kin_mass(cx, i);
    const double* R = s.Rw[i];
    double c[3], RI[9];
    matvec3(R, kin_com(cx, i), c);
    for (int r = 0; r < 3; ++r) c[r] += s.ow[i][r];
    matmul3(R, kin_inertia(cx, i), RI);
    // (R I R^T)(r, q) = RI row r . R row q; the six entries of the symmetric part
    const double c2 = dot3(c, c);
    double* cp = s.own[i] + CI;
    cp[0] = dot3(RI, R) + m * (c2 - c[0] * c[0]);
    cp[1] = 0.5 * (dot3(RI, R + 3) + dot3(RI + 3, R)) - m * c[0] * c[1];
    cp[2] = 0.5 * (dot3(RI, R + 6) + dot3(RI + 6, R)) - m * c[0] * c[2];
    cp[3] = dot3(RI + 3, R + 3) + m * (c2 - c[1] * c[1]);
    cp[4] = 0.5 * (dot3(RI + 3, R + 6) + dot3(RI + 6, R + 3)) - m * c[1] * c[2];
    cp[5] = dot3(RI + 6, R + 6) + m * (c2 - c[2] * c[2]);
}
template <class Em> HD void t_frames(Ctx<Em>& cx, int f) {
    auto& s = cx.s;
    const int L = cx.kt.frame_link[f];
    matmul3(s.Rw[L], cx.kt.frame_R[f], s.fr_R[f]);
    double o[3];
    matvec3(s.Rw[L], cx.kt.frame_o[f], o);
    for (int r = 0; r < 3; ++r) s.fr_o[f][r] = s.ow[L][r] + o[r];
    // (static emitters — the pose finder's device kernels — form the chest error in a task group of their own, beside this one: its lane did
    //  twice the work of the two foot frames, and this group set the length of the third phase once the link task had lost its velocities)
    if constexpr (em_static<Em>) return;
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

// ===================================================================================================
// PHASE D — composite quantities.  Stage 1: one lane per leaf walks up its single-child chain with the
// running sums in registers.  Stage 2: one lane finishes the links that have several children and their
// ancestors (reverse topological order), then the totals.  Contact-point kinematics ride along.
// ===================================================================================================
// composite (subtree) quantities as DESCENDANT SUMS: lane (link slot, component r).  The links are visited in the order of
// decreasing subtree size (kt.comp_order) and a wave iteration runs as many steps as the largest subtree among its four links
// (kt.comp_cnt), so it costs: 24 + 6 + 4 + 3 + 2 + 1 steps on the ergoCub tree instead of 6 x 24.
// An iteration has a fixed cost of several hundred cycles of index chasing on top of its steps: six iteration groups, spread
// 1-2-2-1 over four waves or one per wave over eight.
static_assert(NL * 16 == 6 * 64, "six wave iterations of composite tasks (16 used components per link)");
template <class Em> HD void t_composite(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    const int i = cx.kt.comp_order[t >> 4], r = t & 15;
    if constexpr (em_static<Em>) { if (r >= CKL) return; }   // (subtree momenta: not formed, not read)
    // trip count of the wave iteration = largest subtree among its four links (uniform: scalar loop); shorter lists are padded
    // with the zero slot own[NL]
    const int cnt = HIPNLP_UNIFORM(int(cx.kt.comp_cnt[t >> 6]));
    // the whole descendant list of the link first (24 bytes, three 8-byte reads in flight together), then eight records per
    // step, all eight reads in flight before the first add
    static_assert(NL == 24 && sizeof(cx.kt.desc[0]) == 24, "descendant lists are read as three 8-byte words");
    const unsigned long long* dw = reinterpret_cast<const unsigned long long*>(cx.kt.desc[i]);
    const unsigned long long w3[3] = {dw[0], dw[1], dw[2]};
    double acc = 0.0;
    if (cnt <= 4) {   // (wave-uniform) the small subtrees — four of the six iteration groups of the ergoCub tree: four records, not eight
        double v[4];  // (the terms left out are the zero slot's: no bit of the sum changes)
        HIPNLP_UNROLL
        for (int u = 0; u < 4; ++u) v[u] = s.own[int((w3[0] >> (8 * u)) & 0xffull)][r];
        HIPNLP_ISSUE_FENCE();
        HIPNLP_UNROLL
        for (int u = 0; u < 4; ++u) acc += v[u];
        s.comp[i][r] = acc;
        return;
    }
    HIPNLP_UNROLL
    for (int c = 0; c < 3; ++c) {
        if (8 * c >= cnt) break;   // (wave-uniform)
        double v[8];
        HIPNLP_UNROLL
        for (int u = 0; u < 8; ++u) v[u] = s.own[int((w3[c] >> (8 * u)) & 0xffull)][r];
        HIPNLP_ISSUE_FENCE();
        HIPNLP_UNROLL
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    s.comp[i][r] = acc;
}
template <class Em> HD void t_composite_g0(Ctx<Em>& cx, int t) { t_composite(cx, t); }          // the four largest subtrees
template <class Em> HD void t_composite_g1(Ctx<Em>& cx, int t) { t_composite(cx, t + 64); }
template <class Em> HD void t_composite_g2(Ctx<Em>& cx, int t) { t_composite(cx, t + 128); }
template <class Em> HD void t_composite_g3(Ctx<Em>& cx, int t) { t_composite(cx, t + 192); }
template <class Em> HD void t_composite_g4(Ctx<Em>& cx, int t) { t_composite(cx, t + 256); }
template <class Em> HD void t_composite_g5(Ctx<Em>& cx, int t) { t_composite(cx, t + 320); }   // the four smallest
template <class Em> HD void t_pkin(Ctx<Em>& cx, int c) {
    auto& s = cx.s;
    const int f = c < 4 ? 0 : 1;
    double r3[3];
    matvec3(s.fr_R[f], s.pk + PK_DESC + 3 * c, r3);
    for (int r = 0; r < 3; ++r) s.pkin[c][r] = s.fr_o[f][r] + r3[r];
}

// ===================================================================================================
// PHASE F — derivative columns.  lanes 0..22: joint j ; lanes 23..25: base rotation theta_e / omega_e
// ===================================================================================================
template <class Em> HD void t_columns(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    if constexpr (em_static<Em>) {   // the com columns alone: d com / d s_j = a_j x (h_sub - m_sub o_j) / M
        const double inv_M = cx.kt.inv_total_mass;
        if (t == 0) for (int r = 0; r < 3; ++r) s.com[r] = s.comp[0][CH + r] * inv_M;
        if (t < NJ) {
            const int i = t + 1;
            const double cm = s.comp[i][CM];
            double a[3], t1[3], dcom[3];
            for (int r = 0; r < 3; ++r) { a[r] = s.aw[t][r]; t1[r] = s.comp[i][CH + r] - cm * s.ow[i][r]; }
            cross3(a, t1, dcom);
            for (int r = 0; r < 3; ++r) em.J(js::COMC_S + NJ * r + t, row_id(RK_COMC, 0, r), S_ + t, -(dcom[r] * inv_M));
        }
        return;
    }
    const double inv_M = cx.kt.inv_total_mass, inv_mass = 1.0 / cx.gp.mass;
    // totals of the whole tree (every lane needs them; lane 0 publishes them for the row assembly)
    double com[3], klin[3], kang[3], t1[3], t2[3];
    for (int r = 0; r < 3; ++r) { com[r] = s.comp[0][CH + r] * inv_M; klin[r] = s.comp[0][CKL + r]; kang[r] = s.comp[0][CKA + r]; }
    if (t == 0) {
        cross3(com, klin, t1);
        for (int r = 0; r < 3; ++r) { s.com[r] = com[r]; s.klin[r] = klin[r]; s.kang[r] = kang[r]; s.hang[r] = kang[r] - t1[r]; }
    }
    double a[3], o[3] = {0.0, 0.0, 0.0};
    int i;  // link whose composite / velocity is used
    if (t < NJ) { i = t + 1; for (int r = 0; r < 3; ++r) { a[r] = s.aw[t][r]; o[r] = s.ow[i][r]; } }
    else { i = 0; for (int r = 0; r < 3; ++r) a[r] = (r == t - NJ) ? 1.0 : 0.0; }
    const double cm = s.comp[i][CM];
    double ch[3], cI[6], ckl[3], cka[3];
    for (int r = 0; r < 3; ++r) { ch[r] = s.comp[i][CH + r]; ckl[r] = s.comp[i][CKL + r]; cka[r] = s.comp[i][CKA + r]; }
    for (int r = 0; r < 6; ++r) cI[r] = s.comp[i][CI + r];
    double oxa[3];
    cross3(o, a, oxa);
    // configuration derivative:  dk = S x* k_sub - I_sub (S x v_i)
    double xw[3], xv[3], Il[3], Ia[3], dkl[3], dka[3], dcom[3], dh[3];
    cross3(a, s.wv[i], xw);
    cross3(a, s.vo[i], t1);
    cross3(oxa, s.wv[i], t2);
    for (int r = 0; r < 3; ++r) xv[r] = t1[r] + t2[r];
    cross3(xw, ch, t1);
    for (int r = 0; r < 3; ++r) Il[r] = cm * xv[r] + t1[r];
    symvec(cI, xw, t1);
    cross3(ch, xv, t2);
    for (int r = 0; r < 3; ++r) Ia[r] = t1[r] + t2[r];
    cross3(a, ckl, t1);
    for (int r = 0; r < 3; ++r) dkl[r] = t1[r] - Il[r];
    cross3(a, cka, t1);
    cross3(oxa, ckl, t2);
    for (int r = 0; r < 3; ++r) dka[r] = t1[r] + t2[r] - Ia[r];
    for (int r = 0; r < 3; ++r) t1[r] = ch[r] - cm * o[r];
    cross3(a, t1, dcom);
    for (int r = 0; r < 3; ++r) dcom[r] = dcom[r] * inv_M;
    cross3(dcom, klin, t1);
    cross3(com, dkl, t2);
    for (int r = 0; r < 3; ++r) dh[r] = dka[r] - t1[r] - t2[r];
    if (t < NJ) {
        const int j = t;
        for (int r = 0; r < 3; ++r) {
            em.J(js::COMC_S + NJ * r + j, row_id(RK_COMC, 0, r), S_ + j, -dcom[r]);
            em.J(js::CMMC_S + NJ * r + j, row_id(RK_CMMC, 0, r), S_ + j, -dh[r] * inv_mass);
        }
    } else {
        const int e = t - NJ;
        for (int r = 0; r < 3; ++r) s.dth_h[e][r] = dh[r];
    }
}

// joint columns of the frame-based terms, lane j (23), on a wave of its own in the column phase: chest-frame orientation cost
// gradient and the feet lateral distance row
template <class Em> HD void t_frame_columns(Ctx<Em>& cx, int j) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double* a = s.aw[j];
    const double* o = s.ow[j + 1];
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
}

// columns of the centroidal momentum matrix (angular rows): d hang / d sdot_j = I_sub S_j about the CoM; lanes as t_columns,
// on another wave of the same phase
template <class Em> HD void t_cmm_columns(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double inv_M = cx.kt.inv_total_mass, inv_mass = 1.0 / cx.gp.mass;
    double com[3], a[3], o[3] = {0.0, 0.0, 0.0}, t1[3], t2[3];
    for (int r = 0; r < 3; ++r) com[r] = s.comp[0][CH + r] * inv_M;
    int i;
    if (t < NJ) { i = t + 1; for (int r = 0; r < 3; ++r) { a[r] = s.aw[t][r]; o[r] = s.ow[i][r]; } }
    else { i = 0; for (int r = 0; r < 3; ++r) a[r] = (r == t - NJ) ? 1.0 : 0.0; }
    const double cm = s.comp[i][CM];
    double ch[3], cI[6], oxa[3], lin[3], ang[3];
    for (int r = 0; r < 3; ++r) ch[r] = s.comp[i][CH + r];
    for (int r = 0; r < 6; ++r) cI[r] = s.comp[i][CI + r];
    cross3(o, a, oxa);
    cross3(a, ch, t1);
    for (int r = 0; r < 3; ++r) lin[r] = cm * oxa[r] + t1[r];
    symvec(cI, a, t1);
    cross3(ch, oxa, t2);
    for (int r = 0; r < 3; ++r) ang[r] = t1[r] + t2[r];
    cross3(com, lin, t1);
    if (t < NJ) { for (int r = 0; r < 3; ++r) em.J(js::CMMC_SD + NJ * r + t, row_id(RK_CMMC, 0, r), SD_ + t, -(ang[r] - t1[r]) * inv_mass); }
    else { for (int r = 0; r < 3; ++r) s.Aw[t - NJ][r] = ang[r] - t1[r]; }
}

// ===================================================================================================
// PHASE G — row assembly
// ===================================================================================================
// contact-point kinematic consistency  p - (p_b + pkin)  (K1, planner.py:590-632): lane (c, i), 24 tasks
template <class Em> HD void t_kinc(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const int c = t / 3, i = t - 3 * c, jb = js_pt(cx, c), jc = js::ptc(c), gb = gs::PT_STRIDE * c, cb = PT_ * c;
    const double* r = s.pkin[c];  // base-centred
    em.G(gb + gs::KINC + i, row_id(RK_KINC, c, i), s.x[cb + P_ + i] - (s.x[PB_ + i] + r[i]));
    emit_jc(em, jc + js::KINC_P + i, row_id(RK_KINC, c, i), cb + P_ + i, 1.0);
    emit_jc(em, jc + js::KINC_PB + i, row_id(RK_KINC, c, i), PB_ + i, -1.0);
    // d pkin / d q_b = -[r]x G / |q|   ->  row entries = +[r]x G / |q|
    const double X0 = skew_rc(r, i, 0), X1 = skew_rc(r, i, 1), X2 = skew_rc(r, i, 2);
    for (int l = 0; l < 4; ++l)
        em.J(jb + js::KINC_QB + 4 * i + l, row_id(RK_KINC, c, i), QB_ + l, (X0 * s.G[l] + X1 * s.G[4 + l] + X2 * s.G[8 + l]) * s.inv_qnorm);
}
// d pkin / d s_j = a_j x (pkin - o_j) for the joints of the leg path: lane (c, q) 48 tasks, three rows each (another wave)
template <class Em> HD void t_kinc_s(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const int c = t / LEG_PATH, q = t - LEG_PATH * c, f = c < 4 ? 0 : 1, jb = js_pt(cx, c);
    const int j = cx.kt.leg_joint[f][q];
    const double* r = s.pkin[c];
    double d[3], x[3];
    for (int n = 0; n < 3; ++n) d[n] = r[n] - s.ow[j + 1][n];
    cross3(s.aw[j], d, x);
    for (int i = 0; i < 3; ++i) em.J(jb + js::KINC_S + LEG_PATH * i + q, row_id(RK_KINC, c, i), S_ + j, -x[i]);
}
// com == CoM(pb, qn, s)  (K2, planner.py:285-306): lanes (i, l) 12 for the q_b entries, lanes 12..14 rows
template <class Em> HD void t_comc(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double* r = s.com;
    if (t < 12) {
        const int i = t / 4, l = t - 4 * i;
        const double X0 = skew_rc(r, i, 0), X1 = skew_rc(r, i, 1), X2 = skew_rc(r, i, 2);
        em.J(js::COMC_QB + 4 * i + l, row_id(RK_COMC, 0, i), QB_ + l, (X0 * s.G[l] + X1 * s.G[4 + l] + X2 * s.G[8 + l]) * s.inv_qnorm);
    } else {
        const int i = t - 12;
        em.G(gs::COMC + i, row_id(RK_COMC, 0, i), s.x[COM_ + i] - (s.x[PB_ + i] + r[i]));
        emit_jc(em, js::COMC_COM + i, row_id(RK_COMC, 0, i), COM_ + i, 1.0);
        emit_jc(em, js::COMC_PB + i, row_id(RK_COMC, 0, i), PB_ + i, -1.0);
    }
}
// h[3:] == CMM(...)[3:] / mass  (K3, planner.py:309-339): lanes (i, l) 12 for q_b / qdot_b, lanes 12..14 rows
template <class Em> HD void t_cmmc(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    const double inv_mass = 1.0 / cx.gp.mass;
    if (t < 12) {
        const int i = t / 4, l = t - 4 * i;
        double dq = 0.0, dqd = 0.0;
        for (int e = 0; e < 3; ++e) {
            dq += s.dth_h[e][i] * s.G[4 * e + l] * s.inv_qnorm + s.Aw[e][i] * s.dwq[4 * e + l];
            dqd += s.Aw[e][i] * s.G[4 * e + l];
        }
        em.J(js::CMMC_QB + 4 * i + l, row_id(RK_CMMC, 0, i), QB_ + l, -dq * inv_mass);
        em.J(js::CMMC_QD + 4 * i + l, row_id(RK_CMMC, 0, i), QD_ + l, -dqd * inv_mass);
    } else {
        const int i = t - 12;
        em.G(gs::CMMC + i, row_id(RK_CMMC, 0, i), s.x[H_ + 3 + i] - s.hang[i] * inv_mass);
        emit_jc(em, js::CMMC_H + i, row_id(RK_CMMC, 0, i), H_ + 3 + i, 1.0);
    }
}
// feet distance value (K4) on lane 4; chest cost gradient on q_b, lanes 0..3: 5 tasks
template <class Em> HD void t_feetd(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    Em& em = cx.em;
    if (t == 4) {
        const double* yr = s.fr_R[1];
        double d[3];
        for (int i = 0; i < 3; ++i) d[i] = s.fr_o[0][i] - s.fr_o[1][i];
        em.G(gs::FEETD, row_id(RK_FEETD, 0, 0), yr[1] * d[0] + yr[4] * d[1] + yr[7] * d[2]);
    } else {
        const int l = t;
        double acc = 0.0;
        for (int e = 0; e < 3; ++e) acc += -s.chest_w[e] * s.G[4 * e + l];
        s.grad[QB_ + l] += s.chest_dc * acc * s.inv_qnorm;
    }
}

// minimize-mode end terms: lane 0 sums the cost partials, lanes 1.. add the gradient shares (phase G, behind t_feetd)
constexpr int ENDS_FINISH_TASKS = 1 + ENDS_TASKS;
template <class Em> HD void t_ends_finish(Ctx<Em>& cx, int t) {
    auto& s = cx.s;
    const bool active = (cx.ki.last || cx.ki.first) && (cx.st.final_type == HIPNLP_EXPR_MINIMIZE || cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE);
    // Nothing to add (interior knots; subject_to ends): no read-modify-write of grad, so no ordering behind t_feetd either.  (With the
    // fence first, every one of the three wave iterations of this group waited for the wave's outstanding LDS operations and then
    // for the settings words: 0.9 k cycles on an interior knot, 1.4 k on the first knot — the slowest workgroup of a launch — for
    // storing one zero.)
    if (!active) { if (t == 0) s.cost[CT_ENDS] = 0.0; return; }
    HIPNLP_WAVE_SYNC();
    if (t == 0) {
        // (only the rows that ARE costs: in the compact layout the c[] entries of subject_to rows hold their g values)
        double e = 0.0;
        if (active && cx.ki.last) {
            if (cx.st.final_type == HIPNLP_EXPR_MINIMIZE) for (int i = 0; i < 105; ++i) e += s.ends.c[i];
            if (cx.st.periodicity_type == HIPNLP_EXPR_MINIMIZE) for (int i = 105; i < ENDS_TASKS; ++i) e += s.ends.c[i];
        }
        s.cost[CT_ENDS] = e;
    } else if (active) {
        const int i = t - 1;
        const double gv = s.ends.g[i];
        if (gv != 0.0) {
            const int var = i < 105 ? end_tables(cx).fin_var[i] : end_tables(cx).per_var[i - 105];
            if (var >= 0) s.grad[var] += gv;  // every variable appears in at most one final row and one periodicity row,
        }                                      // and the two row sets are disjoint (states vs controls/velocities)
    }
}

// ---------------------------------------------------------------------------------------------------
// Where a task group runs in the FOUR-WAVE kernels of the planar terrain (round 6).  At batch a launch is (workgroups / resident
// workgroups) x the lifetime of ONE workgroup (tools/diag/stamps.py at x 64: 22.6 k cycles, of which the first phase 5.4 k — four waves carry
// 18 k cycles of task groups there, but the only ones the second phase waits for are the joint transforms, 2.9 k, and the base, 2.4 k).  The
// groups below need nothing but the knot record and nobody waits for them before the last phases: in that kernel they run where the
// kinematic chain leaves a wave idle (fn_l: second / third phase); everywhere else — eight-wave kernels, smooth terrain, host recorders and
// emulation (emitters without kWaves) — in the first phase as before (fn_e).  Exactly one of the two groups of a pair does the work.
template <class Em> constexpr bool sched4p = em_waves<Em>::value == 4 && Em::kTerrain == HIPNLP_TERRAIN_PLANAR;
#define HIPNLP_TASK_PAIR(fn)                                                                                        \
    template <class Em> HD void fn##_e(Ctx<Em>& cx, int t) { if constexpr (!sched4p<Em>) fn(cx, t); }              \
    template <class Em> HD void fn##_l(Ctx<Em>& cx, int t) { if constexpr (sched4p<Em>) fn(cx, t); }
// (Every group that writes a COST TERM stays inside the first three phases: the cost partials of a knot are published behind the third barrier, hipnlp.hip pub_step.)
HIPNLP_TASK_PAIR(t_joint_rows)   // second phase: grad S_ / SD_ are first added to in the fifth (t_frame_columns), c_joint is read by t_joint_cost in the third
HIPNLP_TASK_PAIR(t_unitq)        // third phase: grad QB_ is first added to in the last (t_feetd)
HIPNLP_TASK_PAIR(t_small)        // third phase: grad H_ / COM_ / PB_ / VB_ / QD_: only t_ends_finish (last phase) adds
HIPNLP_TASK_PAIR(t_points_dyn)   // fifth phase: rows and constant entries of the knot record alone, no cost term
#undef HIPNLP_TASK_PAIR

// ---------------------------------------------------------------------------------------------------
// The knot program.  R(w4, w8, fn, n): run tasks 0..n-1 of fn on the lanes of wave w4 of a four-wave workgroup, or of wave w8
// of an eight-wave workgroup (the kernel variant for launches that leave most CUs idle: one workgroup per CU anyway, so the
// phases are spread over twice the waves); host: plain loop.  Groups of one phase run concurrently on different waves;
// BARRIER separates phases.  Groups that rely on running BEHIND another group of their wave (HIPNLP_WAVE_SYNC) share both ids.
// HIPNLP_W4(p, s) / HIPNLP_W8(p, s): wave of the four- / eight-wave workgroup on the planar / on the smooth terrain (defined where the device expands the
// program; host expansions ignore the wave ids).  The eight waves sit two per SIMD (w and w + 4): a phase's longest task wants a
// partner with little to issue — on the planar terrain the terrain tasks are empty, so the pairing differs between the two.
// Same-wave chain of the last phase (do not separate): t_feetd -> t_ends_finish (both add into grad q_b: on two waves the two
// read-modify-writes of one scratch entry would race — tried for +0.3 %, caught by the wave-order test below, undone).
// Same-wave chains of the second phase (do not separate): t_terrain_swing -> t_points_cost (sums the per-point costs the swing task wrote)
// -> t_foot_costs (adds to the swing gradient) -> t_foot_cost_sum.  (The host expansions run the groups in table order and cannot see a
// group moved to another wave of the same phase: tests/test_kernel_body_hostemu.py runs every phase wave by wave in permuted wave orders
// for that, and the GPU parity tests — test_smooth_terrain_matches_oracle caught one such move.)
// A group listed twice with wave -1 in one of the places (t_joint_cost, t_pkin) runs in a different PHASE in the two kernel variants; such
// groups only write scratch (no emitter calls), so the host expansions, which run both, compute the same thing twice.
// ---------------------------------------------------------------------------------------------------
#define HIPNLP_KNOT_PROGRAM(R, BARRIER)                                                   \
    R(0, 0, t_points_vec, 3 * NC) R(HIPNLP_W4(-1, 0), -1, t_hdyn_rows_a, HDYN_TASKS - 48) R(0, HIPNLP_W8(7, 2), t_joint_rows_e, NJ) \
    R(HIPNLP_W4(1, 3), HIPNLP_W8(7, 2), t_points_scalar, NC) R(HIPNLP_W4(1, 0), 3, t_dyn, 7 + NJ + 3) R(1, 7, t_terrain_bump, TERRAIN_BUMP_TASKS) R(1, 7, t_terrain_stage, NC) R(HIPNLP_W4(1, 0), 1, t_unitq_e, 1) \
    R(2, HIPNLP_W8(2, 4), t_joints, NJ) R(2, 1, t_feet_yaw, 2) R(HIPNLP_W4(3, 2), HIPNLP_W8(4, 1), t_feet_centroid, FEET_CENTROID_TASKS) R(HIPNLP_W4(-1, 2), -1, t_hdyn_entries_a, 48) \
    R(3, 5, t_base, 3) R(3, 6, t_small_e, 4) R(3, 3, t_points_dyn_e, 3 * NC) \
    BARRIER                                                                               \
    R(0, 0, t_fk_rot_a, FK_TASKS_A) R(0, 0, t_link_u_a, FK_SPLIT)                         \
    R(3, 1, t_fk_rot_b, FK_TASKS_B) R(3, 1, t_link_u_b, NJ - FK_SPLIT)                    \
    R(1, 2, t_hdyn_entries, 48) R(HIPNLP_W4(1, 3), 3, t_hdyn_rows, HDYN_TASKS - 48)                     \
    R(2, -1, t_joint_rows_l, NJ)                                                          \
    R(2, 7, t_terrain_hnf, NC) R(2, HIPNLP_W8(6, 4), t_terrain_swing, NC) R(2, HIPNLP_W8(6, 4), t_points_cost, 3) \
    R(2, HIPNLP_W8(6, 4), t_foot_costs, FOOT_TASKS) R(2, HIPNLP_W8(6, 4), t_foot_cost_sum, 3) \
    R(1, 5, t_terrain_planar, NC) R(HIPNLP_W4(0, 1), 6, t_terrain_dcc, NC) R(-1, 7, t_joint_cost, 1)    \
    BARRIER                                                                               \
    R(0, 0, t_links, NL) R(1, 1, t_frames, 3) R(2, 2, t_link_inertia, NL) R(2, -1, t_unitq_l, 1) \
    R(3, -1, t_small_l, 4) R(3, -1, t_joint_cost, 1) R(3, 3, t_com_height, 1)             \
    BARRIER                                                                               \
    R(0, 0, t_composite_g0, 64) R(1, 1, t_composite_g1, 64) R(1, 2, t_composite_g2, 64)   \
    R(2, 3, t_composite_g3, 64) R(2, 4, t_composite_g4, 64) R(3, 5, t_composite_g5, 64) R(-1, 6, t_pkin, NC) \
    BARRIER                                                                               \
    R(0, 0, t_columns, NJ + 3) R(1, 1, t_cmm_columns, NJ + 3) R(1, -1, t_points_dyn_l, 3 * NC) R(2, 2, t_frame_columns, NJ) R(3, 3, t_ends, ENDS_TASKS) R(3, -1, t_pkin, NC) \
    BARRIER                                                                               \
    R(0, 0, t_kinc, 3 * NC) R(1, 1, t_comc, 15) R(2, 2, t_cmmc, 15) R(2, 3, t_kinc_s, NC * LEG_PATH) \
    R(3, 4, t_feetd, 5) R(3, 4, t_ends_finish, ENDS_FINISH_TASKS)                         \
    BARRIER

// ---------------------------------------------------------------------------------------------------
// Workgroup specialisation (hipnlp.hip, SPLIT instantiations): the knot program cut into two halves that run on TWO workgroups of one
// launch — the task groups that need the robot model (joints, base, forward kinematics, links, composites, derivative columns, the
// kinematic consistency rows; with them the cheap groups the kinematic ones lean on: t_joint_rows / t_joint_cost initialise grad S_ / SD_
// and t_unitq grad QB_, to which kinematic groups add; t_dyn writes the padding slots of the ancestor lists), and the MODEL-FREE ones below: contact-point rows, terrain, linear trapezoid defects,
// momentum dynamics, point costs, horizon ends.  No group of one half reads scratch a group of the other half writes, every native slot
// of g / jac g and every cost term is written by one half, and the gradient splits by variable: entries [QB_, COM_) belong to the kinematic
// half, the others (point variables, v_b, qdot_b, p_b, com, h) to the model-free one — tests/test_split_program.py runs the halves on two
// poisoned scratches and finds the merged outputs equal to the one-scratch run, bit for bit.
// (Horizon-end terms in `minimize` mode add into every gradient entry: launches of such settings keep the one-workgroup kernels.)
// ---------------------------------------------------------------------------------------------------
HD constexpr bool split_same(const char* a, const char* b) { while (*a && *a == *b) { ++a; ++b; } return *a == *b; }
HD constexpr bool split_task_is_model_free(const char* n) {
    const char* mf[] = {"t_points_vec", "t_points_scalar", "t_points_dyn", "t_terrain_bump", "t_terrain_stage", "t_terrain_planar", "t_terrain_dcc",
                        "t_terrain_hnf", "t_terrain_swing", "t_points_cost", "t_foot_costs", "t_foot_cost_sum", "t_feet_yaw", "t_feet_centroid",
                        "t_hdyn_rows_a", "t_hdyn_entries_a", "t_hdyn_entries", "t_hdyn_rows", "t_small", "t_small_e", "t_small_l", "t_points_dyn_e", "t_points_dyn_l",
                        "t_com_height", "t_ends", "t_ends_finish"};
    for (const char* k : mf) if (split_same(n, k)) return true;
    return false;
}
HD constexpr bool split_grad_is_model_free(int i) { return i < QB_ || i >= COM_; }   // (x layout: ... | q_b 130 | sdot 134 | s 157 | com 180 | h 183)
HD constexpr bool split_cost_is_model_free(int t) { return !(t == CT_FRAMEQ || t == CT_BASEQ || t == CT_JREG); }
static_assert(QB_ == 130 && SD_ == 134 && S_ == 157 && COM_ == 180 && VB_ == 120, "gradient ownership of the split program follows the x layout");

}  // namespace hipnlp
