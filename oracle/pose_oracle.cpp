// TEST INFRASTRUCTURE — CPU oracle of the static pose finder NLP (BASELINE config 2).  NOT product code: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY STATUS: the ASSEMBLY (variable / parameter / constraint order, Opti canonical forms, cost scaling) is pinned
// against the reference's own Python executed in the build container on the CasADi-API stand-in
// (tests/golden/pose_*.npz, tools/gen_pose_fixtures.py).  The third-party arithmetic (CasADi AD, adam FK/CoM on the real
// ergoCub URDF, liecasadi) is NOT available here: "parity unpinned" for those — the published algorithms are restated.
//
// Restates (citations relative to /root/reference/src/hippopt/):
//   turnkey_planners/humanoid_pose_finder/planner.py:323-399   Planner.__init__ (call order)
//     :670-722 _add_contact_point_feasibility     :631-668 _add_contact_kinematic_consistency
//     :446-519 _add_kinematics_constraints        :521-629 _add_kinematics_regularization
//     :724-768 _add_foot_regularization
//   base/problem.py:95-174 (add_cost / add_constraint / add_expression; a non-comparison expression in subject_to mode
//   becomes `expression == 0`, the scaling is dropped)
// The whole NLP is ONE generic function of x (81 inputs), instantiated with double / Dual (forward AD, 81 directions) / Dep.
#include <algorithm>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "kinodyn_formulas.hpp"

namespace oracle {
namespace pose {

enum : int { NJ = HIPNLP_NJ, NC = HIPNLP_NC, NX = HIPNLP_POSE_NX, NP = HIPNLP_POSE_NP,
             X_PB = 48, X_QB = 51, X_S = 55, X_COM = 78,
             // parameters (reference creation order; tests/golden/pose_*.npz "pnames")
             P_DESC = 0, P_MASS = 24, P_GRAV = 27, P_REF = 33 /* per point: p 9c, f 9c+3, descriptor 9c+6 */,
             P_REF_PB = 105, P_REF_QB = 108, P_REF_S = 112, P_REF_COM = 135, P_REF_FQ = 138, P_REF_LH = 142 /* right: + 3 */, P_EPS = 148, P_MU = 149,
             P_SMAX = 150, P_SMIN = 173, P_LH_IN = 196 /* right: + 3 */ };
constexpr double kInf = std::numeric_limits<double>::infinity();
enum : int { CT_BASEQ = 0, CT_FRAMEQ, CT_COM, CT_JOINT, CT_FAVG, CT_PREG, CT_FREG, CT_LHAND, CT_RHAND, NCT };
static_assert(NCT == HIPNLP_POSE_NCOST_TERMS, "cost terms");
static const char* kCostNames[NCT] = {"base_quaternion_error", "frame_rotation_error", "com_position_error", "joint_positions_error",
                                      "average_force_regularization", "point_position_regularization", "force_regularization",
                                      "left_hand_position_error", "right_hand_position_error"};

struct RowBlock { std::string name; int first, rows; };

static std::string point_name(int c) { return std::string("state.contact_points.") + (c < 4 ? "left[" : "right[") + std::to_string(c % 4) + "]"; }

// g (and the 7 cost terms) of one pose.  `rows`/`lb`/`ub` are filled when non-null (structure pass).
template <class S> void evaluate(const hipnlp_pose_desc& d, const S* x, const double* p, std::vector<S>& g, S* cost,
                                 std::vector<RowBlock>* blocks, std::vector<double>* lb, std::vector<double>* ub) {
    const hipnlp_pose_settings& st = d.settings;
    const hipnlp_robot_model& md = d.model;
    auto par = [&](int off) { return ParamMaker<S>::make(p[off]); };
    auto par3 = [&](int off) { return v3<S>(par(off), par(off + 1), par(off + 2)); };
    auto x3 = [&](int off) { return v3<S>(x[off], x[off + 1], x[off + 2]); };
    g.clear();
    auto row = [&](const S& v, double lo, double hi) { g.push_back(v); if (lb) { lb->push_back(lo); ub->push_back(hi); } };
    auto block = [&](const std::string& name, int rows) { if (blocks) blocks->push_back({name, int(g.size()), rows}); };
    const TerrainSpec terrain(st);
    Q4<S> qb; for (int i = 0; i < 4; ++i) qb[i] = x[X_QB + i];
    const Q4<S> qn = quaternion_xyzw_normalization(qb);                              // planner.py:343-349
    const M3<S> Rb = rotation_from_quaternion_xyzw(qn);
    const V3<S> pb = x3(X_PB);
    const S* s = x + X_S;
    const S mass = par(P_MASS), eps = par(P_EPS), mu = par(P_MU);
    V3<S> P[NC], F[NC];
    for (int c = 0; c < NC; ++c) { P[c] = x3(6 * c); F[c] = x3(6 * c + 3); }
    for (int c = 0; c < NC; ++c) {                                                   // planner.py:360-375
        const std::string pn = point_name(c);
        block(pn + ".p_complementarity", 1);
        row(relaxed_complementarity_margin(terrain, P[c], scale(F[c], mass), eps), 0.0, kInf);   // :680-689 (f * mass)
        block(pn + ".p_height", 1);
        row(terrain_height(terrain, P[c]), 0.0, kInf);
        block(pn + ".f_normal", 1);
        row(normal_force_component(terrain, P[c], F[c]), 0.0, kInf);
        block(pn + ".f_friction", 1);
        row(friction_cone_square_margin(terrain, P[c], F[c], mu), 0.0, kInf);
        block(pn + ".p_kinematics_consistency", 3);
        const LinkPose<S> fr = frame_pose(md, c < 4 ? HIPNLP_FRAME_LEFT_SOLE : HIPNLP_FRAME_RIGHT_SOLE, pb, Rb, s);
        const V3<S> kin = fr.o + mul(fr.R, par3(P_DESC + 3 * c));
        for (int i = 0; i < 3; ++i) row(P[c][i] - kin[i], 0.0, 0.0);
    }
    block("unitary_quaternion", 1);                                                  // :455-461
    row(qb[0] * qb[0] + qb[1] * qb[1] + qb[2] * qb[2] + qb[3] * qb[3], 1.0, 1.0);
    block("com_kinematics_consistency", 3);                                          // :463-485
    const V3<S> com = x3(X_COM);
    const V3<S> comk = center_of_mass_position(md, pb, Rb, s);
    for (int i = 0; i < 3; ++i) row(com[i] - comk[i], 0.0, 0.0);
    block("centroidal_momentum_dynamics", 6);                                        // :487-509  (unit mass, == 0)
    {
        S hdot[6];
        double grav[6];
        for (int i = 0; i < 6; ++i) grav[i] = 0.0;
        centroidal_dynamics_with_point_forces(grav, com, P, F, NC, hdot);
        for (int i = 0; i < 6; ++i) row(par(P_GRAV + i) + hdot[i], 0.0, 0.0);         // gravity is an Opti parameter here
    }
    block("joint_position_bounds", NJ);                                              // :511-519 (Opti_bounded)
    for (int j = 0; j < NJ; ++j) row(s[j], p[P_SMIN + j], p[P_SMAX + j]);

    for (int t = 0; t < NCT; ++t) cost[t] = S(0.0);
    // ---- _add_kinematics_regularization  :521-629
    {
        Q4<S> qd; for (int i = 0; i < 4; ++i) qd[i] = par(P_REF_QB + i);
        const Q4<S> e = quaternion_xyzw_error(qb, qd);                               // raw (un-normalised) quaternion, :529-537
        cost[CT_BASEQ] = S(st.base_quaternion_cost_multiplier) * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
        Q4<S> fq; for (int i = 0; i < 4; ++i) fq[i] = par(P_REF_FQ + i);
        const S tr = rotation_error_trace(md, HIPNLP_FRAME_CHEST, pb, Rb, s, fq) - S(3.0);
        cost[CT_FRAMEQ] = S(st.desired_frame_quaternion_cost_multiplier) * (tr * tr);
        const V3<S> ce = com - par3(P_REF_COM);
        const S ce2 = dot(ce, ce);
        if (st.com_position_type == HIPNLP_EXPR_MINIMIZE) cost[CT_COM] = S(st.com_regularization_cost_multiplier) * ce2;
        else if (st.com_position_type == HIPNLP_EXPR_SUBJECT_TO) { block("com_position_error", 1); row(ce2, 0.0, 0.0); }
        S acc = S(0.0);
        for (int j = 0; j < NJ; ++j) { const S ej = s[j] - par(P_REF_S + j); acc = acc + ej * S(st.joint_regularization_cost_weights[j]) * ej; }
        cost[CT_JOINT] = S(st.joint_regularization_cost_multiplier) * acc;
        // hand position expressions  :596-660  (cs.MX(position == reference): three equality rows in subject_to mode,
        // multiplier * sumsqr(position - reference) in minimize mode, base/problem.py:95-174)
        for (int hnd = 0; hnd < 2; ++hnd) {
            if (st.hand_type[hnd] == HIPNLP_EXPR_SKIP) continue;
            bool needed[HIPNLP_NL];
            chain_mask(md, st.hand_frame_link[hnd], needed);
            LinkPose<S> links[HIPNLP_NL];
            all_link_poses(md, pb, Rb, s, links, needed);
            const LinkPose<S>& lk = links[st.hand_frame_link[hnd]];
            // point_position_from_kinematics (expressions/kinematics.py): world_H_frame applied to the point in the frame
            const V3<S> pos = lk.o + mul(lk.R, v3c<S>(st.hand_frame_o[hnd]) + mul(m3c<S>(st.hand_frame_R[hnd]), par3(P_LH_IN + 3 * hnd)));
            const V3<S> e = pos - par3(P_REF_LH + 3 * hnd);
            if (st.hand_type[hnd] == HIPNLP_EXPR_MINIMIZE) cost[CT_LHAND + hnd] = S(st.hand_regularization_cost_multiplier[hnd]) * dot(e, e);
            else {   // Opti's canonical form of `expression == parameter`: g = expression, lbg = ubg = the parameter's value
                block(hnd == 0 ? "left_hand_position_error" : "right_hand_position_error", 3);
                for (int i = 0; i < 3; ++i) row(pos[i], p[P_REF_LH + 3 * hnd + i], p[P_REF_LH + 3 * hnd + i]);
            }
        }
    }
    // ---- _add_foot_regularization  :724-768 (left, then right)
    for (int foot = 0; foot < 2; ++foot) {
        const int mode = foot == 0 ? st.left_point_position_type : st.right_point_position_type;
        V3<S> sum = v3<S>(S(0.0), S(0.0), S(0.0));
        for (int c = 4 * foot; c < 4 * foot + 4; ++c) sum = sum + F[c];
        for (int c = 4 * foot; c < 4 * foot + 4; ++c) {
            const V3<S> e = F[c] - scale(sum, S(0.25));
            cost[CT_FAVG] = cost[CT_FAVG] + S(st.average_force_regularization_cost_multiplier) * dot(e, e);
        }
        for (int c = 4 * foot; c < 4 * foot + 4; ++c) {
            const V3<S> ep = P[c] - par3(P_REF + 9 * c);
            const S ep2 = dot(ep, ep);
            if (mode == HIPNLP_EXPR_MINIMIZE) cost[CT_PREG] = cost[CT_PREG] + S(st.point_position_regularization_cost_multiplier) * ep2;
            else if (mode == HIPNLP_EXPR_SUBJECT_TO) { block(point_name(c) + ".p_regularization", 1); row(ep2, 0.0, 0.0); }
            const V3<S> ef = F[c] - par3(P_REF + 9 * c + 3);
            cost[CT_FREG] = cost[CT_FREG] + S(st.force_regularization_cost_multiplier) * dot(ef, ef);
        }
    }
}

struct Handle {
    hipnlp_pose_desc d;
    int m = 0, nnz = 0;
    std::vector<RowBlock> blocks;
    std::vector<int> irow, jcol;
    double cost_terms[NCT];
};

}  // namespace pose
}  // namespace oracle

using namespace oracle;
using namespace oracle::pose;

extern "C" {

Handle* oracle_pose_create(const hipnlp_pose_desc* desc) {
    if (desc->settings.terrain != HIPNLP_TERRAIN_PLANAR && desc->settings.terrain != HIPNLP_TERRAIN_SMOOTH_STEPS) return nullptr;
    Handle* h = new Handle();
    h->d = *desc;
    // structure: trace with every parameter symbolic and x seeded
    std::vector<double> p(NP, 0.5);
    std::vector<Dep> x(NX), g;
    for (int i = 0; i < NX; ++i) x[size_t(i)] = Dep::seed(0.0, i);
    Dep cost[NCT];
    std::vector<double> lb, ub;
    evaluate<Dep>(h->d, x.data(), p.data(), g, cost, &h->blocks, &lb, &ub);
    h->m = int(g.size());
    for (int c = 0; c < NX; ++c)
        for (int r = 0; r < h->m; ++r)
            if (!g[size_t(r)].is_const && g[size_t(r)].mask.test(size_t(c))) { h->irow.push_back(r); h->jcol.push_back(c); }
    h->nnz = int(h->irow.size());
    return h;
}
void oracle_pose_destroy(Handle* h) { delete h; }
void oracle_pose_dims(const Handle* h, int* n, int* m, int* nnz, int* np) { *n = NX; *m = h->m; *nnz = h->nnz; *np = NP; }
void oracle_pose_sparsity(const Handle* h, int* irow, int* jcol) {
    std::copy(h->irow.begin(), h->irow.end(), irow);
    std::copy(h->jcol.begin(), h->jcol.end(), jcol);
}
int oracle_pose_num_row_blocks(const Handle* h) { return int(h->blocks.size()); }
void oracle_pose_row_block(const Handle* h, int i, const char** name, int* first, int* rows) {
    *name = h->blocks[size_t(i)].name.c_str(); *first = h->blocks[size_t(i)].first; *rows = h->blocks[size_t(i)].rows;
}
void oracle_pose_bounds(const Handle* h, const double* p, double* lbg, double* ubg) {
    std::vector<double> x(NX, 0.1), g, lb, ub;
    x[X_QB + 3] = 1.0;
    double cost[NCT];
    evaluate<double>(h->d, x.data(), p, g, cost, nullptr, &lb, &ub);
    std::copy(lb.begin(), lb.end(), lbg);
    std::copy(ub.begin(), ub.end(), ubg);
}
void oracle_pose_eval(Handle* h, const double* x, const double* p, double* f, double* grad, double* g, double* jac) {
    Dual::K = NX;
    std::vector<Dual> xd(NX), gd;
    for (int i = 0; i < NX; ++i) xd[size_t(i)] = Dual::seed(x[i], i);
    std::vector<Dual> cost(NCT);
    evaluate<Dual>(h->d, xd.data(), p, gd, cost.data(), nullptr, nullptr, nullptr);
    Dual tot(0.0);
    for (int t = 0; t < NCT; ++t) { h->cost_terms[t] = cost[size_t(t)].v; tot = tot + cost[size_t(t)]; }
    if (f) *f = tot.v;
    if (grad) for (int i = 0; i < NX; ++i) grad[i] = tot.d[i];
    if (g) for (int r = 0; r < h->m; ++r) g[r] = gd[size_t(r)].v;
    if (jac) for (int e = 0; e < h->nnz; ++e) jac[e] = gd[size_t(h->irow[size_t(e)])].d[h->jcol[size_t(e)]];
    Dual::K = kMaxDir;
}
// Exact Hessian of the Lagrangian  sigma f(x) + lambda^T g(x)  (what CasADi's nlp_hess_l hands IPOPT's eval_h; the reference
// pose finder runs IPOPT with its default exact-Hessian option: humanoid_pose_finder/main.py:101, casadi_solver_options = {}).
// Forward-over-forward AD: the outer D1<> carries ONE direction j, the inner Dual all 81; H [81][81] dense, row major
// (H[j][i] = d^2 L / dx_i dx_j; symmetric up to rounding).
void oracle_pose_hess(Handle* h, const double* x, const double* p, double sigma, const double* lambda, double* H) {
    Dual::K = NX;
    typedef D1<Dual> DD;
    std::vector<DD> xd(NX), gd;
    std::vector<DD> cost(NCT);
    for (int j = 0; j < NX; ++j) {
        for (int i = 0; i < NX; ++i) xd[size_t(i)] = DD(Dual::seed(x[i], i), Dual(i == j ? 1.0 : 0.0));
        evaluate<DD>(h->d, xd.data(), p, gd, cost.data(), nullptr, nullptr, nullptr);
        Dual acc(0.0);
        for (int t = 0; t < NCT; ++t) acc = acc + Dual(sigma) * cost[size_t(t)].d;
        for (int r = 0; r < h->m; ++r) if (lambda[r] != 0.0) acc = acc + Dual(lambda[r]) * gd[size_t(r)].d;
        for (int i = 0; i < NX; ++i) H[size_t(j) * NX + i] = acc.d[i];
    }
    Dual::K = kMaxDir;
}
void oracle_pose_cost_terms(const Handle* h, double* out) { for (int t = 0; t < NCT; ++t) out[t] = h->cost_terms[t]; }
const char* oracle_pose_cost_term_name(int i) { return (i >= 0 && i < NCT) ? kCostNames[i] : ""; }

}  // extern "C"
