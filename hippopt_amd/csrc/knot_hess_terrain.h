// knot_hess_terrain.h — second derivatives of the contact-point rows on the SMOOTH terrain (TerrainSum of SmoothTerrain.step,
// robot_planning/utilities/smooth_terrain.py:201-227,266-336; terrain_descriptor.py:45-80; complementarity.py:27-32,71-87;
// contacts.py:22-66,158-166).  The dcc margin contains  ndot = (dn/dp) v,  so its Hessian with respect to p needs the THIRD
// derivatives of the normal, i.e. the FOURTH derivatives of the bump sum Z.  Instead of more closed forms, the Hessian tasks
// carry truncated Taylor polynomials:
//   J2<K>  a function of (p_x, p_y) up to total degree K, scaled coefficients c_ij = d^(i+j) f / dx^i dy^j / (i! j!)
//          (product = truncated polynomial product; composition with an analytic g by Horner on its Taylor coefficients);
//          Z as J2<4>, the terrain frame (n, x, y) and grad h as J2<3>, dn/dp as J2<2> by differentiation;
//   T3     a function of (p_x, p_y, p_z) to second order (value, gradient, Hessian), in which the Lagrangian of the point is
//          assembled with f, v, f_dot, u_v as constants: its Hessian is the (p, p) block (t_kh_point_smooth_pp);
//   G3     value and gradient in (p_x, p_y, p_z): the mixed blocks are the gradients of dL/df, dL/dv, ... and need the terrain one
//          order lower (Z as J2<3>, the frame as J2<2>; t_kh_point_smooth_mixed).
#pragma once
#include "knot_body.h"

namespace hipnlp {

template <int K> struct J2 {
    static constexpr int NC_ = (K + 1) * (K + 2) / 2;
    double c[NC_];
    HD J2() { for (int i = 0; i < NC_; ++i) c[i] = 0.0; }
    HD explicit J2(double v) { for (int i = 0; i < NC_; ++i) c[i] = 0.0; c[0] = v; }
    HD static constexpr int idx(int i, int j) { return (i + j) * (i + j + 1) / 2 + j; }   // x^i y^j
};
template <int K> HD J2<K> operator+(const J2<K>& a, const J2<K>& b) { J2<K> r; for (int i = 0; i < J2<K>::NC_; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
template <int K> HD J2<K> operator-(const J2<K>& a, const J2<K>& b) { J2<K> r; for (int i = 0; i < J2<K>::NC_; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
template <int K> HD J2<K> operator-(const J2<K>& a) { J2<K> r; for (int i = 0; i < J2<K>::NC_; ++i) r.c[i] = -a.c[i]; return r; }
template <int K> HD J2<K> operator*(const J2<K>& a, double s) { J2<K> r; for (int i = 0; i < J2<K>::NC_; ++i) r.c[i] = a.c[i] * s; return r; }
template <int K> HD J2<K> operator*(const J2<K>& a, const J2<K>& b) {
    J2<K> r;
    for (int d1 = 0; d1 <= K; ++d1)
        for (int j1 = 0; j1 <= d1; ++j1) {
            const double av = a.c[d1 * (d1 + 1) / 2 + j1];
            for (int d2 = 0; d1 + d2 <= K; ++d2)
                for (int j2 = 0; j2 <= d2; ++j2) {
                    const int d = d1 + d2;
                    r.c[d * (d + 1) / 2 + j1 + j2] += av * b.c[d2 * (d2 + 1) / 2 + j2];
                }
        }
    return r;
}
// g(a) for an analytic g with Taylor coefficients gn[n] = g^(n)(a_0) / n!: Horner in d = a - a_0.  d has no constant term, so the
// value after j steps is only needed — and only meaningful — up to total degree j: step j multiplies a degree-(j-1) polynomial with the
// degree >= 1 part of d, truncated at degree j (2 + 9 + 25 + 55 = 91 multiply-adds for K = 4 where four full truncated products
// take 280; the terms kept are the same, in the same order, so the coefficients are the ones the full products give).
template <int K, int ORD> HD void j2_horner_step(J2<K>& r, const J2<K>& a, double g) {
    double out[(ORD + 1) * (ORD + 2) / 2];
    for (int i = 0; i < (ORD + 1) * (ORD + 2) / 2; ++i) out[i] = 0.0;
    for (int d1 = 0; d1 < ORD; ++d1)
        for (int j1 = 0; j1 <= d1; ++j1) {
            const double av = r.c[d1 * (d1 + 1) / 2 + j1];
            for (int d2 = 1; d1 + d2 <= ORD; ++d2)
                for (int j2 = 0; j2 <= d2; ++j2) {
                    const int d = d1 + d2;
                    out[d * (d + 1) / 2 + j1 + j2] += av * a.c[d2 * (d2 + 1) / 2 + j2];
                }
        }
    out[0] = g;
    for (int i = 0; i < (ORD + 1) * (ORD + 2) / 2; ++i) r.c[i] = out[i];
}
template <int K, int ORD> HD void j2_horner(J2<K>& r, const J2<K>& a, const double* gn) {
    if constexpr (ORD <= K) {
        j2_horner_step<K, ORD>(r, a, gn[K - ORD]);
        j2_horner<K, ORD + 1>(r, a, gn);
    }
}
template <int K> HD J2<K> j2_compose(const J2<K>& a, const double* gn) {
    J2<K> r(gn[K]);
    j2_horner<K, 1>(r, a, gn);
    return r;
}
// (a0 + ax X + ay Y)^m for an integer m > K, in closed form (multinomial theorem): the coefficient of X^i Y^j is
// m (m-1) ... (m-i-j+1) / (i! j!)  a0^(m-i-j) ax^i ay^j.  No polynomial products and no temporaries: the two footprint powers of every
// bump were half of the truncated products of the terrain jet and most of its register pressure.
HD constexpr double j2_inv_fact(int n) { return n <= 1 ? 1.0 : (n == 2 ? 0.5 : (n == 3 ? 1.0 / 6.0 : (n == 4 ? 1.0 / 24.0 : 1.0 / 120.0))); }   // n <= 5
template <int K> HD J2<K> j2_linpow(double a0, double ax, double ay, int m) {
    J2<K> r;
    double p[K + 1];              // a0^(m-d), d = 0..K
    p[K] = ipow_d(a0, m - K);
    for (int d = K - 1; d >= 0; --d) p[d] = p[d + 1] * a0;
    double xi[K + 1], yj[K + 1];  // ax^i / i!, ay^j / j!  (reciprocal factorials as constants: no division in the task)
    xi[0] = yj[0] = 1.0;
    double xp = 1.0, yp = 1.0;
    for (int i = 1; i <= K; ++i) { xp *= ax; yp *= ay; xi[i] = xp * j2_inv_fact(i); yj[i] = yp * j2_inv_fact(i); }
    double ff = 1.0;              // m (m-1) ... (m-d+1)
    for (int d = 0; d <= K; ++d) {
        const double fp = ff * p[d];
        for (int i = 0; i <= d; ++i) r.c[J2<K>::idx(i, d - i)] = fp * xi[i] * yj[d - i];
        ff = ff * double(m - d);
    }
    return r;
}
template <int K> HD J2<K> j2_rsqrt(const J2<K>& a) {   // a^(-1/2), a_0 > 0: Taylor coefficients g_n = g_(n-1) (-1/2 - (n-1)) / (n a_0)
    double gn[K + 1];
    const double r = sqrt(a.c[0]), inv = 1.0 / a.c[0];
    gn[0] = r * inv;
    for (int n = 1; n <= K; ++n) gn[n] = gn[n - 1] * ((-0.5 - double(n - 1)) / double(n)) * inv;
    return j2_compose(a, gn);
}
template <int K> HD J2<K - 1> j2_dx(const J2<K>& a) {
    J2<K - 1> r;
    for (int i = 0; i + 1 <= K; ++i) for (int j = 0; i + 1 + j <= K; ++j) r.c[J2<K - 1>::idx(i, j)] = double(i + 1) * a.c[J2<K>::idx(i + 1, j)];
    return r;
}
template <int K> HD J2<K - 1> j2_dy(const J2<K>& a) {
    J2<K - 1> r;
    for (int j = 0; j + 1 <= K; ++j) for (int i = 0; i + j + 1 <= K; ++i) r.c[J2<K - 1>::idx(i, j)] = double(j + 1) * a.c[J2<K>::idx(i, j + 1)];
    return r;
}
template <int K, int L> HD J2<L> j2_trunc(const J2<K>& a) { J2<L> r; for (int i = 0; i < J2<L>::NC_; ++i) r.c[i] = a.c[i]; return r; }

// One bump  H exp(-g^r)  of the terrain as a jet of order K, g = a^m + b^m with the footprint coordinates a, b LINEAR in (p_x, p_y):
// their powers in closed form, and ONE composition with the Taylor coefficients of  F(t) = H exp(-t^r)  at t = g_0
// (s_n = C(r, n) g_0^(r-n) are those of t^r;  E = exp(-s):  n e_n = -sum_k k s_k e_(n-k)) instead of a composition for the power and
// another for the exponential.  false: exp(-w) underflows (same guard as terrain_bump_jet; also catches inf / nan of far-away
// points) — the bump and all its derivatives vanish, `out` is left alone.
// slope_x / slope_y (TerrainTops; read only for a step with a sloped top): the jet of exp(-g^r) times the LINEAR top
// pi = height + slope_x dx + slope_y dy — in scaled coefficients a convolution with a three-term polynomial.
template <int K> HD bool terrain_bump_j(const TerrainStepK& t, double px, double py, J2<K>& out, const double* slope_x = nullptr, const double* slope_y = nullptr) {
    const double dx = px - t.ox, dy = py - t.oy;
    const bool sloped = step_sloped(t) && slope_x;
    const J2<K> g = j2_linpow<K>(t.ax * dx + t.ay * dy, t.ax, t.ay, step_m(t)) + j2_linpow<K>(t.bx * dx + t.by * dy, t.bx, t.by, step_m(t));
    double sn[K + 1], en[K + 1];
    sn[K] = ipow_d(g.c[0], t.r - K);                                 // g_0^(r-n) ...
    for (int n = K - 1; n >= 0; --n) sn[n] = sn[n + 1] * g.c[0];
    double ff = 1.0;                                                 // ... times C(r, n) = r (r-1) ... (r-n+1) / n!
    for (int n = 0; n <= K; ++n) { sn[n] *= ff * j2_inv_fact(n); ff *= double(t.r - n); }
    if (!(sn[0] < 700.0)) return false;
    en[0] = (sloped ? 1.0 : t.height) * exp(-sn[0]);
    for (int n = 1; n <= K; ++n) {
        double acc = 0.0;
        for (int k = 1; k <= n; ++k) acc += double(k) * sn[k] * en[n - k];
        en[n] = acc * (-1.0 / double(n));
    }
    out = j2_compose(g, en);
    if (sloped) {
        const double pix = *slope_x, piy = *slope_y, pi0 = t.height + pix * dx + piy * dy;
        J2<K> f = out;
        for (int d = 0; d <= K; ++d)
            for (int j = 0; j <= d; ++j) {
                const int i = d - j;
                double v = f.c[J2<K>::idx(i, j)] * pi0;
                if (i > 0) v += f.c[J2<K>::idx(i - 1, j)] * pix;
                if (j > 0) v += f.c[J2<K>::idx(i, j - 1)] * piy;
                out.c[J2<K>::idx(i, j)] = v;
            }
    }
    return true;
}
// Z(p_x, p_y) of the terrain as a jet of order K
template <int K> HD J2<K> terrain_Z_j(const KSettings& st, double px, double py, const TerrainTops* tops = nullptr) {
    J2<K> Z;
    for (int sidx = 0; sidx < st.n_steps; ++sidx) {
        Z.c[0] += st.steps[sidx].oz;
        J2<K> bump;
        if (terrain_bump_j<K>(st.steps[sidx], px, py, bump, tops ? &tops->px[sidx] : nullptr, tops ? &tops->py[sidx] : nullptr)) Z = Z + bump;
    }
    return Z;
}

// ---- second order in three variables (p_x, p_y, p_z) -------------------------------------------------------------------------------
struct T3 {
    double v, g[3], H[6];   // H: xx, xy, xz, yy, yz, zz
    HD T3() : v(0.0) { for (int i = 0; i < 3; ++i) g[i] = 0.0; for (int i = 0; i < 6; ++i) H[i] = 0.0; }
    HD explicit T3(double c) : v(c) { for (int i = 0; i < 3; ++i) g[i] = 0.0; for (int i = 0; i < 6; ++i) H[i] = 0.0; }
};
HD constexpr int t3h(int a, int b) { return a <= b ? (a == 0 ? b : (a == 1 ? 2 + b : 5)) : (b == 0 ? a : (b == 1 ? 2 + a : 5)); }
HD T3 operator+(const T3& a, const T3& b) { T3 r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] + b.g[i]; for (int i = 0; i < 6; ++i) r.H[i] = a.H[i] + b.H[i]; return r; }
HD T3 operator-(const T3& a, const T3& b) { T3 r; r.v = a.v - b.v; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] - b.g[i]; for (int i = 0; i < 6; ++i) r.H[i] = a.H[i] - b.H[i]; return r; }
HD T3 operator*(const T3& a, double s) { T3 r; r.v = a.v * s; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] * s; for (int i = 0; i < 6; ++i) r.H[i] = a.H[i] * s; return r; }
HD T3 operator*(const T3& a, const T3& b) {
    T3 r;
    r.v = a.v * b.v;
    for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i];
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) r.H[t3h(i, j)] = a.H[t3h(i, j)] * b.v + a.g[i] * b.g[j] + a.g[j] * b.g[i] + a.v * b.H[t3h(i, j)];
    return r;
}
HD T3 t3_chain(const T3& a, double f0, double f1, double f2) {   // f(a): f, f', f'' at a.v
    T3 r;
    r.v = f0;
    for (int i = 0; i < 3; ++i) r.g[i] = f1 * a.g[i];
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) r.H[t3h(i, j)] = f1 * a.H[t3h(i, j)] + f2 * a.g[i] * a.g[j];
    return r;
}
template <int K> HD T3 t3_from(const J2<K>& a) {   // a function of (p_x, p_y) only
    T3 r;
    r.v = a.c[0]; r.g[0] = a.c[J2<K>::idx(1, 0)]; r.g[1] = a.c[J2<K>::idx(0, 1)];
    r.H[0] = 2.0 * a.c[J2<K>::idx(2, 0)]; r.H[1] = a.c[J2<K>::idx(1, 1)]; r.H[3] = 2.0 * a.c[J2<K>::idx(0, 2)];
    return r;
}

// ---- first order in three variables (p_x, p_y, p_z): value and gradient --------------------------------------------------------------
struct G3 {
    double v, g[3];
    HD G3() : v(0.0) { g[0] = g[1] = g[2] = 0.0; }
    HD explicit G3(double c) : v(c) { g[0] = g[1] = g[2] = 0.0; }
};
HD G3 operator+(const G3& a, const G3& b) { G3 r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
HD G3 operator-(const G3& a, const G3& b) { G3 r; r.v = a.v - b.v; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
HD G3 operator*(const G3& a, double s) { G3 r; r.v = a.v * s; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] * s; return r; }
HD G3 operator*(const G3& a, const G3& b) { G3 r; r.v = a.v * b.v; for (int i = 0; i < 3; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
template <int K> HD G3 g3_from(const J2<K>& a) {   // a function of (p_x, p_y) only
    G3 r;
    r.v = a.c[0]; r.g[0] = a.c[J2<K>::idx(1, 0)]; r.g[1] = a.c[J2<K>::idx(0, 1)];
    return r;
}

}  // namespace hipnlp
