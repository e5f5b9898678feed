// TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/README.md). Not product code.
//
// Scalar types the oracle's single generic restatement of the reference formulas is
// instantiated with:
//   double   -> values
//   Dual     -> forward-mode AD (what CasADi does to obtain nlp_jac_g / nlp_grad_f)
//   Dep      -> structural dependency tracer (reproduces CasADi-SX style structural sparsity:
//               products with a literal constant 0 vanish, x+0 -> x)
#pragma once
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

constexpr int kMaxDir = 192;  // every block of the NLP has <= 192 scalar inputs (periodicity: 168)

struct Dual {
    double v;
    double d[kMaxDir];
    static thread_local int K;  // active number of directions (<= kMaxDir)
    Dual() : v(0.0) { std::memset(d, 0, sizeof(double) * K); }
    Dual(double c) : v(c) { std::memset(d, 0, sizeof(double) * K); }  // NOLINT implicit: constants
    struct NoInit {};
    explicit Dual(NoInit) {}
    Dual(const Dual& o) : v(o.v) { std::memcpy(d, o.d, sizeof(double) * K); }  // only the active directions
    Dual& operator=(const Dual& o) { v = o.v; std::memcpy(d, o.d, sizeof(double) * K); return *this; }
    static Dual seed(double value, int dir) {
        Dual r(value);
        r.d[dir] = 1.0;
        return r;
    }
};
inline thread_local int Dual::K = kMaxDir;

inline Dual operator+(const Dual& a, const Dual& b) {
    Dual r{Dual::NoInit{}}; r.v = a.v + b.v;
    for (int i = 0; i < Dual::K; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
inline Dual operator-(const Dual& a, const Dual& b) {
    Dual r{Dual::NoInit{}}; r.v = a.v - b.v;
    for (int i = 0; i < Dual::K; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
inline Dual operator-(const Dual& a) {
    Dual r{Dual::NoInit{}}; r.v = -a.v;
    for (int i = 0; i < Dual::K; ++i) r.d[i] = -a.d[i];
    return r;
}
inline Dual operator*(const Dual& a, const Dual& b) {
    Dual r{Dual::NoInit{}}; r.v = a.v * b.v;
    for (int i = 0; i < Dual::K; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
inline Dual operator/(const Dual& a, const Dual& b) {
    Dual r{Dual::NoInit{}}; r.v = a.v / b.v;
    const double inv = 1.0 / b.v;
    for (int i = 0; i < Dual::K; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
inline Dual chain(const Dual& a, double fv, double fd) {
    Dual r{Dual::NoInit{}}; r.v = fv;
    for (int i = 0; i < Dual::K; ++i) r.d[i] = fd * a.d[i];
    return r;
}
inline Dual sqrt(const Dual& a) { double s = std::sqrt(a.v); return chain(a, s, 0.5 / s); }
inline Dual sin(const Dual& a) { return chain(a, std::sin(a.v), std::cos(a.v)); }
inline Dual cos(const Dual& a) { return chain(a, std::cos(a.v), -std::sin(a.v)); }
inline Dual tanh(const Dual& a) { double t = std::tanh(a.v); return chain(a, t, 1.0 - t * t); }
inline Dual exp(const Dual& a) { double e = std::exp(a.v); return chain(a, e, e); }
inline double value_of(const Dual& a) { return a.v; }

// ---- dependency tracer -------------------------------------------------------------
struct Dep {
    std::bitset<kMaxDir> mask;  // bit i set: depends structurally on block input i
    bool is_const;
    double c;       // value when is_const
    Dep() : mask(), is_const(true), c(0.0) {}
    Dep(double cv) : mask(), is_const(true), c(cv) {}  // NOLINT
    static Dep seed(double, int dir) { Dep r; r.is_const = false; r.mask.set(size_t(dir)); return r; }
    bool zero() const { return is_const && c == 0.0; }
};
inline Dep dep_join(const Dep& a, const Dep& b) { Dep r; r.is_const = false; r.mask = a.mask | b.mask; return r; }
inline Dep operator+(const Dep& a, const Dep& b) {
    if (a.is_const && b.is_const) return Dep(a.c + b.c);
    if (a.zero()) return b;
    if (b.zero()) return a;
    return dep_join(a, b);
}
inline Dep operator-(const Dep& a) { if (a.is_const) return Dep(-a.c); return a; }
inline Dep operator-(const Dep& a, const Dep& b) {
    if (a.is_const && b.is_const) return Dep(a.c - b.c);
    if (b.zero()) return a;
    if (a.zero()) return -b;
    return dep_join(a, b);
}
inline Dep operator*(const Dep& a, const Dep& b) {
    if (a.is_const && b.is_const) return Dep(a.c * b.c);
    if (a.zero() || b.zero()) return Dep(0.0);
    return dep_join(a, b);
}
inline Dep operator/(const Dep& a, const Dep& b) {
    if (a.is_const && b.is_const) return Dep(a.c / b.c);
    if (a.zero()) return Dep(0.0);
    return dep_join(a, b);
}
inline Dep dep_unary(const Dep& a, double cv) { if (a.is_const) return Dep(cv); return a; }
inline Dep sqrt(const Dep& a) { return dep_unary(a, std::sqrt(a.c)); }
inline Dep sin(const Dep& a) { return dep_unary(a, std::sin(a.c)); }
inline Dep cos(const Dep& a) { return dep_unary(a, std::cos(a.c)); }
inline Dep tanh(const Dep& a) { return dep_unary(a, std::tanh(a.c)); }
inline Dep exp(const Dep& a) { return dep_unary(a, std::exp(a.c)); }

inline double value_of(double a) { return a; }

// ---- generic first-order forward-mode wrapper over ANY scalar type (nests: D1<D1<S>>) -----------------------
// Used where the reference itself differentiates: cs.gradient(height, p) (terrain_descriptor.py:49-52) and
// cs.jtimes(height / normal, p, v) (complementarity.py:74-75).
template <class S> struct D1 {
    S v, d;
    D1() : v(0.0), d(0.0) {}
    D1(double c) : v(c), d(0.0) {}  // NOLINT
    D1(const S& val, const S& der) : v(val), d(der) {}
};
template <class S> D1<S> operator+(const D1<S>& a, const D1<S>& b) { return D1<S>(a.v + b.v, a.d + b.d); }
template <class S> D1<S> operator-(const D1<S>& a, const D1<S>& b) { return D1<S>(a.v - b.v, a.d - b.d); }
template <class S> D1<S> operator-(const D1<S>& a) { return D1<S>(-a.v, -a.d); }
template <class S> D1<S> operator*(const D1<S>& a, const D1<S>& b) { return D1<S>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class S> D1<S> operator/(const D1<S>& a, const D1<S>& b) { S q = a.v / b.v; return D1<S>(q, (a.d - q * b.d) / b.v); }
template <class S> D1<S> sqrt(const D1<S>& a) { using std::sqrt; S r = sqrt(a.v); return D1<S>(r, a.d / (S(2.0) * r)); }
template <class S> D1<S> exp(const D1<S>& a) { using std::exp; S e = exp(a.v); return D1<S>(e, e * a.d); }
template <class S> D1<S> sin(const D1<S>& a) { using std::sin; using std::cos; return D1<S>(sin(a.v), cos(a.v) * a.d); }
template <class S> D1<S> cos(const D1<S>& a) { using std::sin; using std::cos; return D1<S>(cos(a.v), -(sin(a.v) * a.d)); }
template <class S> D1<S> tanh(const D1<S>& a) { using std::tanh; S t = tanh(a.v); return D1<S>(t, (S(1.0) - t * t) * a.d); }

// an Opti *parameter*: symbolic for the structure tracer (never a literal constant), a number otherwise
template <class S> struct ParamMaker { static S make(double v) { return S(v); } };
template <> struct ParamMaker<Dep> { static Dep make(double) { Dep r; r.is_const = false; return r; } };

// seeding helper: make an S from a value and a direction index
template <class S> struct Seeder { static S make(double v, int dir) { return S::seed(v, dir); } };
template <> struct Seeder<double> { static double make(double v, int) { return v; } };

}  // namespace oracle
