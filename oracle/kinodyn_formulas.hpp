// TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/README.md). Not product code.
//
// Generic (scalar-type-templated) restatement of the closed-form per-knot functions of
// hippopt's robot_planning layer.  Citations are relative to /root/reference/src/hippopt/.
// For the three un-vendored libraries (casadi, adam-robotics, liecasadi; no versions pinned,
// setup.cfg:52-74) the published algorithm is restated and the reference *call site* is cited.
#pragma once
#include <cmath>

#include "../include/hipnlp.h"
#include "scalar_types.hpp"

namespace oracle {

using std::cos;
using std::exp;
using std::sin;
using std::sqrt;
using std::tanh;

template <class S> struct V3 {
    S a[3];
    S& operator[](int i) { return a[i]; }
    const S& operator[](int i) const { return a[i]; }
};
template <class S> struct M3 {
    S m[3][3];
};

template <class S> V3<S> v3(const S& x, const S& y, const S& z) { V3<S> r; r[0] = x; r[1] = y; r[2] = z; return r; }
template <class S> V3<S> v3c(const double* c) { V3<S> r; for (int i = 0; i < 3; ++i) r[i] = S(c[i]); return r; }
template <class S> V3<S> operator+(const V3<S>& a, const V3<S>& b) { V3<S> r; for (int i = 0; i < 3; ++i) r[i] = a[i] + b[i]; return r; }
template <class S> V3<S> operator-(const V3<S>& a, const V3<S>& b) { V3<S> r; for (int i = 0; i < 3; ++i) r[i] = a[i] - b[i]; return r; }
template <class S> V3<S> scale(const V3<S>& a, const S& s) { V3<S> r; for (int i = 0; i < 3; ++i) r[i] = a[i] * s; return r; }
template <class S> S dot(const V3<S>& a, const V3<S>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// casadi cross(a,b): [a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0]
template <class S> V3<S> cross(const V3<S>& a, const V3<S>& b) {
    return v3<S>(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
template <class S> M3<S> m3c(const double* c) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = S(c[3 * i + j]); return r; }
template <class S> M3<S> eye3() { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = S(i == j ? 1.0 : 0.0); return r; }
template <class S> M3<S> mul(const M3<S>& a, const M3<S>& b) {
    M3<S> r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
template <class S> V3<S> mul(const M3<S>& a, const V3<S>& b) {
    V3<S> r;
    for (int i = 0; i < 3; ++i) r[i] = a.m[i][0] * b[0] + a.m[i][1] * b[1] + a.m[i][2] * b[2];
    return r;
}
template <class S> M3<S> transpose(const M3<S>& a) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i]; return r; }
template <class S> M3<S> skew(const V3<S>& v) {
    M3<S> r = eye3<S>();
    r.m[0][0] = S(0.0); r.m[1][1] = S(0.0); r.m[2][2] = S(0.0);
    r.m[0][1] = -v[2]; r.m[0][2] = v[1];
    r.m[1][0] = v[2];  r.m[1][2] = -v[0];
    r.m[2][0] = -v[1]; r.m[2][1] = v[0];
    return r;
}
template <class S> M3<S> add(const M3<S>& a, const M3<S>& b) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
template <class S> M3<S> scale(const M3<S>& a, const S& s) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] * s; return r; }

// ---------------------------------------------------------------------------------------------
// quaternions, xyzw (robot_planning/variables/floating_base.py:23-25)
// ---------------------------------------------------------------------------------------------
template <class S> struct Q4 { S a[4]; S& operator[](int i) { return a[i]; } const S& operator[](int i) const { return a[i]; } };

// E11  expressions/quaternion.py:13  liecasadi.Quaternion(xyzw=q).normalize() = q / norm_2(q)
template <class S> Q4<S> quaternion_xyzw_normalization(const Q4<S>& q) {
    S n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    Q4<S> r;
    for (int i = 0; i < 4; ++i) r[i] = q[i] / n;
    return r;
}
// liecasadi SO3.from_quat(q).as_matrix():  I + 2 w [v]x + 2 [v]x^2   (call sites:
// expressions/kinematics.py:49-51,175-177,265-267,428-448; expressions/quaternion.py:64-70)
template <class S> M3<S> rotation_from_quaternion_xyzw(const Q4<S>& q) {
    V3<S> v = v3<S>(q[0], q[1], q[2]);
    M3<S> K = skew(v);
    M3<S> K2 = mul(K, K);
    return add(add(eye3<S>(), scale(K, S(2.0) * q[3])), scale(K2, S(2.0)));
}
// Hamilton product in xyzw (liecasadi Quaternion.__mul__, used by SO3.__mul__, quaternion.py:67)
template <class S> Q4<S> quaternion_product(const Q4<S>& a, const Q4<S>& b) {
    Q4<S> r;
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    r[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    return r;
}
// E12  expressions/quaternion.py:35-42
template <class S> V3<S> quaternion_velocity_to_right_trivialized_angular_velocity(const Q4<S>& q, const Q4<S>& qdot) {
    V3<S> qi = v3<S>(q[0], q[1], q[2]);
    V3<S> qdi = v3<S>(qdot[0], qdot[1], qdot[2]);
    V3<S> c = cross(qdi, qi);
    V3<S> r;
    for (int i = 0; i < 3; ++i) r[i] = S(2.0) * (-(qdot[3] * qi[i]) + q[3] * qdi[i] - c[i]);
    return r;
}
// E13  expressions/quaternion.py:64-70 : (SO3(qd).inverse() * SO3(q)).as_quat() - Identity.as_quat()
template <class S> Q4<S> quaternion_xyzw_error(const Q4<S>& q, const Q4<S>& qd) {
    Q4<S> qdi;  // liecasadi SO3.inverse(): conjugate
    qdi[0] = -qd[0]; qdi[1] = -qd[1]; qdi[2] = -qd[2]; qdi[3] = qd[3];
    Q4<S> e = quaternion_product(qdi, q);
    e[3] = e[3] - S(1.0);
    return e;
}

// ---------------------------------------------------------------------------------------------
// terrains.  ts.terrain == HIPNLP_TERRAIN_PLANAR : PlanarTerrain (robot_planning/utilities/planar_terrain.py:7-41),
//            h = p_z, n = e_z, R_t = I, all as CONSTANTS (so the structural pattern is the planar one).
//            HIPNLP_TERRAIN_SMOOTH_STEPS : TerrainSum (terrain_sum.py:19-38) of SmoothTerrain.step bumps
//            (smooth_terrain.py:201-227 height, :266-336 step), normal / orientation from the TerrainDescriptor defaults
//            (terrain_descriptor.py:45-80), derivatives the way the reference takes them (cs.gradient / cs.jtimes -> D1<>).
// ---------------------------------------------------------------------------------------------
struct TerrainSpec {   // the terrain part of hipnlp_settings / hipnlp_pose_settings
    int terrain, n_terrain_steps;
    const hipnlp_terrain_step* terrain_steps;
    TerrainSpec(const hipnlp_settings& s) : terrain(s.terrain), n_terrain_steps(s.n_terrain_steps), terrain_steps(s.terrain_steps) {}
    TerrainSpec(const hipnlp_pose_settings& s) : terrain(s.terrain), n_terrain_steps(s.n_terrain_steps), terrain_steps(s.terrain_steps) {}
};
template <class S> S ipow(const S& x, int n) {  // x ** n for the integer-valued float exponents 2*edge_sharpness, 2*side_sharpness
    S r = S(1.0), b = x;
    while (n > 0) { if (n & 1) r = r * b; n >>= 1; if (n) b = b * b; }
    return r;
}
template <class S> S terrain_height(const TerrainSpec& ts, const V3<S>& p) {
    if (ts.terrain == HIPNLP_TERRAIN_PLANAR) return p[2];
    S h = p[2];  // sum_s h_s - (n-1) p_z  with  h_s = p_z - (z_terrain,s + o_s,z)   (terrain_sum.py:30-34)
    for (int i = 0; i < ts.n_terrain_steps; ++i) {
        const hipnlp_terrain_step& st = ts.terrain_steps[i];
        const double c = std::cos(st.orientation), sn = std::sin(st.orientation);
        S dx = p[0] - S(st.position[0]), dy = p[1] - S(st.position[1]);
        S qx = S(c) * dx + S(sn) * dy, qy = S(-sn) * dx + S(c) * dy;           // inv(T) (p - offset), T = Rz(orientation)
        S g = ipow(S(2.0 / st.length) * qx, 2 * st.edge_sharpness) + ipow(S(2.0 / st.width) * qy, 2 * st.edge_sharpness);
        // top surface pi(q_xy): the height, or — SmoothTerrain.step(top_normal_direction=n) — the plane through (0, 0, height) with the
        // normalised normal n:  height - n_x / n_z q_x - n_y / n_z q_y   (smooth_terrain.py:238-264; a zero vector = None = flat)
        S top = S(st.height);
        const double nn = std::sqrt(st.top_normal[0] * st.top_normal[0] + st.top_normal[1] * st.top_normal[1] + st.top_normal[2] * st.top_normal[2]);
        if (nn > 0.0) {
            const double nx = st.top_normal[0] / nn, ny = st.top_normal[1] / nn, nz = st.top_normal[2] / nn;
            top = S(-nx / nz) * qx + S(-ny / nz) * qy + S(st.height);
        }
        S z_terrain = exp(-ipow(g, 2 * st.side_sharpness)) * top;               // smooth_terrain.py:211
        h = h - (z_terrain + S(st.position[2]));
    }
    return h;
}
template <class S> V3<S> terrain_gradient(const TerrainSpec& ts, const V3<S>& p) {  // cs.gradient(height(p), p)
    V3<S> g;
    for (int i = 0; i < 3; ++i) {
        V3<D1<S>> pd;
        for (int j = 0; j < 3; ++j) pd[j] = D1<S>(p[j], S(i == j ? 1.0 : 0.0));
        g[i] = terrain_height(ts, pd).d;
    }
    return g;
}
template <class S> V3<S> terrain_normal(const TerrainSpec& ts, const V3<S>& p) {
    if (ts.terrain == HIPNLP_TERRAIN_PLANAR) return v3<S>(S(0.0), S(0.0), S(1.0));
    V3<S> g = terrain_gradient(ts, p);                                   // terrain_descriptor.py:49-53
    S n = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    return v3<S>(g[0] / n, g[1] / n, g[2] / n);
}
template <class S> M3<S> terrain_orientation(const TerrainSpec& ts, const V3<S>& p) {
    if (ts.terrain == HIPNLP_TERRAIN_PLANAR) return eye3<S>();
    V3<S> n = terrain_normal(ts, p);                                     // terrain_descriptor.py:64-72
    V3<S> y = cross(n, v3<S>(S(1.0), S(0.0), S(0.0)));
    V3<S> x = cross(y, n);
    S xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    x = v3<S>(x[0] / xn, x[1] / xn, x[2] / xn);
    y = cross(n, x);
    M3<S> R;
    for (int i = 0; i < 3; ++i) { R.m[i][0] = x[i]; R.m[i][1] = y[i]; R.m[i][2] = n[i]; }
    return R;
}
// jtimes(height, p, v) and jtimes(normal, p, v) of complementarity.py:74-75
template <class S> S terrain_height_derivative(const TerrainSpec& ts, const V3<S>& p, const V3<S>& v) {
    if (ts.terrain == HIPNLP_TERRAIN_PLANAR) return v[2];
    V3<D1<S>> pd;
    for (int j = 0; j < 3; ++j) pd[j] = D1<S>(p[j], v[j]);
    return terrain_height(ts, pd).d;
}
template <class S> V3<S> terrain_normal_derivative(const TerrainSpec& ts, const V3<S>& p, const V3<S>& v) {
    if (ts.terrain == HIPNLP_TERRAIN_PLANAR) return v3<S>(S(0.0), S(0.0), S(0.0));
    V3<D1<S>> pd;
    for (int j = 0; j < 3; ++j) pd[j] = D1<S>(p[j], v[j]);
    V3<D1<S>> n = terrain_normal(ts, pd);
    return v3<S>(n[0].d, n[1].d, n[2].d);
}

// ---------------------------------------------------------------------------------------------
// closed-form expressions
// ---------------------------------------------------------------------------------------------
// E3  expressions/complementarity.py:27-32
template <class S> V3<S> dcc_planar_complementarity(const TerrainSpec& terrain, const V3<S>& p, const S& kt, const V3<S>& u) {
    S tau = tanh(kt * terrain_height(terrain, p));
    V3<S> mu = v3<S>(tau * u[0], tau * u[1], u[2]);
    return mul(terrain_orientation(terrain, p), mu);
}
// E4  expressions/complementarity.py:71-87
template <class S> S dcc_complementarity_margin(const TerrainSpec& terrain, const V3<S>& p, const V3<S>& f, const V3<S>& v, const V3<S>& fdot,
                                                const S& k_bs, const S& eps) {
    S height = terrain_height(terrain, p);
    V3<S> n = terrain_normal(terrain, p);
    S height_derivative = terrain_height_derivative(terrain, p, v);
    V3<S> normal_derivative = terrain_normal_derivative(terrain, p, v);
    S normal_force = dot(n, f);
    S normal_force_derivative = dot(n, fdot);
    S complementarity = height * normal_force;
    S csi = height_derivative * normal_force + height * dot(f, normal_derivative) + height * normal_force_derivative;
    return eps - k_bs * complementarity - csi;
}
// E5  expressions/complementarity.py:113-150
template <class S> S relaxed_complementarity_margin(const TerrainSpec& terrain, const V3<S>& p, const V3<S>& f, const S& eps) {
    return eps - terrain_height(terrain, p) * dot(terrain_normal(terrain, p), f);
}
// E6  expressions/contacts.py:22-24
template <class S> S normal_force_component(const TerrainSpec& terrain, const V3<S>& p, const V3<S>& f) { return dot(terrain_normal(terrain, p), f); }
// E7  expressions/contacts.py:54-66
template <class S> S friction_cone_square_margin(const TerrainSpec& terrain, const V3<S>& p, const V3<S>& f, const S& mu) {
    V3<S> fc = mul(transpose(terrain_orientation(terrain, p)), f);
    return S(-1.0) * (fc[0] * fc[0]) + S(-1.0) * (fc[1] * fc[1]) + (mu * mu) * (fc[2] * fc[2]);
}
// E9  expressions/contacts.py:132
template <class S> S contact_points_yaw_alignment_error(const V3<S>& p0, const V3<S>& p1, const S& yaw) {
    return -sin(yaw) * (p1[0] - p0[0]) + cos(yaw) * (p1[1] - p0[1]);
}
// E10 expressions/contacts.py:158-166
template <class S> S swing_height_heuristic(const TerrainSpec& terrain, const V3<S>& p, const V3<S>& v, const S& hd) {
    S dh = terrain_height(terrain, p) - hd;
    V3<S> pv = mul(transpose(terrain_orientation(terrain, p)), v);
    return S(0.5) * (dh * dh + (pv[0] * pv[0] + pv[1] * pv[1]));
}
// E1  expressions/centroidal.py:62-64   (assume_unitary_mass=True at planner.py:573-577)
template <class S> void centroidal_dynamics_with_point_forces(const double* gravity, const V3<S>& com, const V3<S>* p, const V3<S>* f,
                                                              int n, S* hdot /*6*/) {
    for (int i = 0; i < 6; ++i) hdot[i] = S(gravity[i]);
    for (int c = 0; c < n; ++c) {
        V3<S> t = cross(p[c] - com, f[c]);
        for (int i = 0; i < 3; ++i) { hdot[i] = hdot[i] + f[c][i]; hdot[3 + i] = hdot[3 + i] + t[i]; }
    }
}

// ---------------------------------------------------------------------------------------------
// kinematics — adam-robotics KinDynComputations restated (SURVEY Appendix A conventions).
// Call sites: expressions/kinematics.py:47 (CMM), :163 (CoM), :249 (FK), :339-340, :432.
// ---------------------------------------------------------------------------------------------
// adam spatial_math R_from_axis_angle: cq (I - a a^T) + sq [a]x + a a^T
template <class S> M3<S> rotation_from_axis_angle(const double* axis, const S& q) {
    S cq = cos(q), sq = sin(q);
    M3<S> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double aa = axis[i] * axis[j];
            r.m[i][j] = cq * S((i == j ? 1.0 : 0.0) - aa) + S(aa);
        }
    r.m[0][1] = r.m[0][1] - sq * S(axis[2]); r.m[0][2] = r.m[0][2] + sq * S(axis[1]);
    r.m[1][0] = r.m[1][0] + sq * S(axis[2]); r.m[1][2] = r.m[1][2] - sq * S(axis[0]);
    r.m[2][0] = r.m[2][0] - sq * S(axis[1]); r.m[2][1] = r.m[2][1] + sq * S(axis[0]);
    return r;
}

template <class S> struct LinkPose { M3<S> R; V3<S> o; };

// joints in an order in which every joint comes after the joint that moves its parent link (the joint numbering is the
// reference's joints_name_list order, which need not follow the tree)
inline void joint_evaluation_order(const hipnlp_robot_model& md, int* order /*NJ*/) {
    bool done[HIPNLP_NL];
    for (int i = 0; i < HIPNLP_NL; ++i) done[i] = i == 0;
    int n = 0;
    while (n < HIPNLP_NJ) {
        const int before = n;
        for (int j = 0; j < HIPNLP_NJ; ++j)
            if (!done[j + 1] && done[md.parent[j]]) { done[j + 1] = true; order[n++] = j; }
        if (n == before) break;   // not a tree rooted at link 0 (rejected by the engine's own model check)
    }
    for (int j = 0; n < HIPNLP_NJ && j < HIPNLP_NJ; ++j) if (!done[j + 1]) { done[j + 1] = true; order[n++] = j; }
}

// world_H_link for every link: H_b * prod(parent_H_child(s_j))  (adam forward_kinematics)
template <class S> void all_link_poses(const hipnlp_robot_model& md, const V3<S>& pb, const M3<S>& Rb, const S* s, LinkPose<S>* out,
                                       const bool* needed = nullptr) {
    out[0].R = Rb; out[0].o = pb;
    int order[HIPNLP_NJ];
    joint_evaluation_order(md, order);
    for (int q = 0; q < HIPNLP_NJ; ++q) {
        const int j = order[q];
        if (needed && !needed[j + 1]) continue;
        const LinkPose<S>& par = out[md.parent[j]];
        M3<S> Rloc = mul(m3c<S>(md.R_fix[j]), rotation_from_axis_angle(md.axis[j], s[j]));
        out[j + 1].R = mul(par.R, Rloc);
        out[j + 1].o = par.o + mul(par.R, v3c<S>(md.o_fix[j]));
    }
}
inline void chain_mask(const hipnlp_robot_model& md, int link, bool* needed /*NL*/) {
    for (int i = 0; i < HIPNLP_NL; ++i) needed[i] = false;
    needed[0] = true;
    while (link > 0) { needed[link] = true; link = md.parent[link - 1]; }
}
// K1 building block: forward_kinematics_fun(frame)(H_b, s)   expressions/kinematics.py:249-265
template <class S> LinkPose<S> frame_pose(const hipnlp_robot_model& md, int frame, const V3<S>& pb, const M3<S>& Rb, const S* s) {
    bool needed[HIPNLP_NL];
    chain_mask(md, md.frame_link[frame], needed);
    LinkPose<S> links[HIPNLP_NL];
    all_link_poses(md, pb, Rb, s, links, needed);
    const LinkPose<S>& L = links[md.frame_link[frame]];
    LinkPose<S> r;
    r.R = mul(L.R, m3c<S>(md.frame_R[frame]));
    r.o = L.o + mul(L.R, v3c<S>(md.frame_o[frame]));
    return r;
}
inline double total_mass(const hipnlp_robot_model& md) { double m = 0; for (int i = 0; i < HIPNLP_NL; ++i) m += md.mass[i]; return m; }
// K2  CoM_position_fun()(H_b, s)    expressions/kinematics.py:163-197
template <class S> V3<S> center_of_mass_position(const hipnlp_robot_model& md, const V3<S>& pb, const M3<S>& Rb, const S* s) {
    LinkPose<S> links[HIPNLP_NL];
    all_link_poses(md, pb, Rb, s, links);
    V3<S> acc = v3<S>(S(0.0), S(0.0), S(0.0));
    for (int i = 0; i < HIPNLP_NL; ++i) acc = acc + scale(links[i].o + mul(links[i].R, v3c<S>(md.com[i])), S(md.mass[i]));
    return scale(acc, S(1.0 / total_mass(md)));
}
// K3  centroidal_momentum_matrix_fun()(H_b, s) @ [pb_dot; omega; s_dot]   expressions/kinematics.py:47-69,108
// Restated as the physical definition the matrix encodes (mixed velocity representation):
//   h_lin = sum_i m_i cdot_i ;  h_ang = sum_i [ R_i I_i R_i^T w_i + m_i (c_i - com) x cdot_i ]
template <class S> void centroidal_momentum(const hipnlp_robot_model& md, const V3<S>& pb, const M3<S>& Rb, const S* s,
                                            const V3<S>& pb_dot, const V3<S>& omega, const S* s_dot, S* h /*6*/) {
    LinkPose<S> links[HIPNLP_NL];
    all_link_poses(md, pb, Rb, s, links);
    V3<S> w[HIPNLP_NL], vo[HIPNLP_NL];  // angular velocity and linear velocity of the link-frame origin
    w[0] = omega; vo[0] = pb_dot;
    int order[HIPNLP_NJ];
    joint_evaluation_order(md, order);
    for (int q = 0; q < HIPNLP_NJ; ++q) {
        const int j = order[q];
        int par = md.parent[j];
        V3<S> a = mul(links[j + 1].R, v3c<S>(md.axis[j]));
        w[j + 1] = w[par] + scale(a, s_dot[j]);
        vo[j + 1] = vo[par] + cross(w[par], links[j + 1].o - links[par].o);
    }
    const double M = total_mass(md);
    V3<S> c[HIPNLP_NL], cd[HIPNLP_NL];
    V3<S> com = v3<S>(S(0.0), S(0.0), S(0.0)), lin = com;
    for (int i = 0; i < HIPNLP_NL; ++i) {
        V3<S> r = mul(links[i].R, v3c<S>(md.com[i]));
        c[i] = links[i].o + r;
        cd[i] = vo[i] + cross(w[i], r);
        com = com + scale(c[i], S(md.mass[i] / M));
        lin = lin + scale(cd[i], S(md.mass[i]));
    }
    V3<S> ang = v3<S>(S(0.0), S(0.0), S(0.0));
    for (int i = 0; i < HIPNLP_NL; ++i) {
        M3<S> Iw = mul(mul(links[i].R, m3c<S>(md.inertia[i])), transpose(links[i].R));
        ang = ang + mul(Iw, w[i]) + scale(cross(c[i] - com, cd[i]), S(md.mass[i]));
    }
    for (int i = 0; i < 3; ++i) { h[i] = lin[i]; h[3 + i] = ang[i]; }
}
// K4  expressions/kinematics.py:337-367  (base pose = identity)
template <class S> V3<S> frames_relative_position(const hipnlp_robot_model& md, int reference_frame, int target_frame, const S* s) {
    V3<S> pb = v3<S>(S(0.0), S(0.0), S(0.0));
    M3<S> Rb = eye3<S>();
    LinkPose<S> ref = frame_pose(md, reference_frame, pb, Rb, s);
    LinkPose<S> tgt = frame_pose(md, target_frame, pb, Rb, s);
    M3<S> Rt = transpose(ref.R);
    V3<S> t = mul(Rt, ref.o);
    return mul(Rt, tgt.o) - t;
}
// K5  expressions/kinematics.py:428-448 : R_frame * R(qd)^T ; cost uses (trace - 3)^2, planner.py:471-477
template <class S> S rotation_error_trace(const hipnlp_robot_model& md, int frame, const V3<S>& pb, const M3<S>& Rb, const S* s, const Q4<S>& qd) {
    LinkPose<S> f = frame_pose(md, frame, pb, Rb, s);
    M3<S> E = mul(f.R, transpose(rotation_from_quaternion_xyzw(qd)));
    return E.m[0][0] + E.m[1][1] + E.m[2][2];
}

}  // namespace oracle
