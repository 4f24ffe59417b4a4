// gp_dual.h -- forward-mode derivatives for covariance functions given as source text (gphip_create_custom).
//
// The caller's function body is a template over its arithmetic type T.  The kernel build instantiates it with T = double /
// float; the gradient reduction (custom_grad_kernel, gp_kernels.h) instantiates the SAME text with T = Dual<S, NP>: a value
// and its NP partial derivatives with respect to the function's hyper-parameters P(0) .. P(NP-1), propagated through every
// operator and math function below.  One factorisation then yields dl/dtheta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) for
// an arbitrary function, as the named kernels do with their hand-written derivatives (row f3; the reference's Laplace /
// MAP route differentiates numerically, LaplaceApproximation.wl:177-238).
//
// Text rules (this file is embedded in the library next to gp_kernels.h and compiled by hiprtc, and by g++ in
// tests/test_dual_numbers.py): no #include, no std::.  Math functions are called as ::f so that the overloads declared here
// never recurse through the implicit scalar -> Dual conversion.
#pragma once
#ifndef GP_HD
#define GP_HD __host__ __device__ __forceinline__
#endif

namespace gphip {

template <typename A> struct gp_is_num { static constexpr bool value = false; };
#define GP_NUM(A) template <> struct gp_is_num<A> { static constexpr bool value = true; };
GP_NUM(int) GP_NUM(unsigned) GP_NUM(long) GP_NUM(unsigned long) GP_NUM(long long) GP_NUM(unsigned long long) GP_NUM(float) GP_NUM(double)
#undef GP_NUM
template <bool B, typename R> struct gp_if {};
template <typename R> struct gp_if<true, R> { typedef R type; };
#define GP_NUM_ARG(A) typename A, typename gp_if<gp_is_num<A>::value, int>::type = 0

template <typename S, int NP>
struct Dual {
    S v;
    S g[NP];
    GP_HD Dual() : v((S)0) {
        for (int k = 0; k < NP; ++k) g[k] = (S)0;
    }
    template <GP_NUM_ARG(A)>
    GP_HD Dual(A a) : v((S)a) {
        for (int k = 0; k < NP; ++k) g[k] = (S)0;
    }
    // hyper-parameter k: value with a unit derivative in its own slot
    static GP_HD Dual param(S val, int k) {
        Dual r(val);
        if (k >= 0 && k < NP) r.g[k] = (S)1;
        return r;
    }
    // f(v) with derivative df: the chain rule every unary function below is
    GP_HD Dual chain(S f, S df) const {
        Dual r;
        r.v = f;
        for (int k = 0; k < NP; ++k) r.g[k] = df * g[k];
        return r;
    }
    GP_HD Dual operator-() const { return chain(-v, (S)-1); }
    GP_HD Dual operator+() const { return *this; }
    GP_HD Dual& operator+=(const Dual& o) {
        v += o.v;
        for (int k = 0; k < NP; ++k) g[k] += o.g[k];
        return *this;
    }
    GP_HD Dual& operator-=(const Dual& o) {
        v -= o.v;
        for (int k = 0; k < NP; ++k) g[k] -= o.g[k];
        return *this;
    }
    GP_HD Dual& operator*=(const Dual& o) {
        for (int k = 0; k < NP; ++k) g[k] = g[k] * o.v + v * o.g[k];
        v *= o.v;
        return *this;
    }
    GP_HD Dual& operator/=(const Dual& o) {
        const S inv = (S)1 / o.v;
        v *= inv;
        for (int k = 0; k < NP; ++k) g[k] = (g[k] - v * o.g[k]) * inv;
        return *this;
    }
};

#define GP_DUAL_BINOP(op, asg)                                                                                      \
    template <typename S, int NP> GP_HD Dual<S, NP> operator op(Dual<S, NP> a, const Dual<S, NP>& b) { a asg b; return a; }      \
    template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> operator op(Dual<S, NP> a, A b) { a asg Dual<S, NP>(b); return a; } \
    template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> operator op(A a, const Dual<S, NP>& b) { Dual<S, NP> r(a); r asg b; return r; }
GP_DUAL_BINOP(+, +=)
GP_DUAL_BINOP(-, -=)
GP_DUAL_BINOP(*, *=)
GP_DUAL_BINOP(/, /=)
#undef GP_DUAL_BINOP
#define GP_DUAL_CMP(op)                                                                                             \
    template <typename S, int NP> GP_HD bool operator op(const Dual<S, NP>& a, const Dual<S, NP>& b) { return a.v op b.v; }     \
    template <typename S, int NP, GP_NUM_ARG(A)> GP_HD bool operator op(const Dual<S, NP>& a, A b) { return a.v op (S)b; }    \
    template <typename S, int NP, GP_NUM_ARG(A)> GP_HD bool operator op(A a, const Dual<S, NP>& b) { return (S)a op b.v; }
GP_DUAL_CMP(<)
GP_DUAL_CMP(>)
GP_DUAL_CMP(<=)
GP_DUAL_CMP(>=)
GP_DUAL_CMP(==)
GP_DUAL_CMP(!=)
#undef GP_DUAL_CMP

// The scalar overloads of the same names stay visible inside namespace gphip (the caller's function lives there): without
// these declarations an unqualified exp(x) on a double would only see the Dual overloads below.
using ::exp; using ::log; using ::log1p; using ::expm1; using ::sqrt; using ::pow; using ::fabs; using ::sin; using ::cos; using ::tan;
using ::tanh; using ::sinh; using ::cosh; using ::atan; using ::erf; using ::erfc; using ::fmin; using ::fmax;

#define GP_DUAL_UNARY(name, f, df)                                                    \
    template <typename S, int NP> GP_HD Dual<S, NP> name(const Dual<S, NP>& a) {          \
        const S x = a.v;                                                              \
        const S fx = (f);                                                             \
        return a.chain(fx, (df));                                                     \
    }
GP_DUAL_UNARY(exp, ::exp(x), fx)
GP_DUAL_UNARY(expm1, ::expm1(x), fx + (S)1)
GP_DUAL_UNARY(log, ::log(x), (S)1 / x)
GP_DUAL_UNARY(log1p, ::log1p(x), (S)1 / ((S)1 + x))
GP_DUAL_UNARY(sqrt, ::sqrt(x), x > (S)0 ? (S)0.5 / fx : (S)0)                // (d sqrt at 0: the one-sided limit is infinite -- a
                                                                             //  distance's |x - y| there has derivative 0 in every P)
GP_DUAL_UNARY(fabs, ::fabs(x), x > (S)0 ? (S)1 : (x < (S)0 ? (S)-1 : (S)0))
GP_DUAL_UNARY(sin, ::sin(x), ::cos(x))
GP_DUAL_UNARY(cos, ::cos(x), -::sin(x))
GP_DUAL_UNARY(tan, ::tan(x), (S)1 + fx * fx)
GP_DUAL_UNARY(tanh, ::tanh(x), (S)1 - fx * fx)
GP_DUAL_UNARY(sinh, ::sinh(x), ::cosh(x))
GP_DUAL_UNARY(cosh, ::cosh(x), ::sinh(x))
GP_DUAL_UNARY(atan, ::atan(x), (S)1 / ((S)1 + x * x))
GP_DUAL_UNARY(erf, ::erf(x), (S)1.1283791670955126 * ::exp(-x * x))          // 2 / sqrt(pi)
GP_DUAL_UNARY(erfc, ::erfc(x), (S)-1.1283791670955126 * ::exp(-x * x))
#undef GP_DUAL_UNARY

// a^b: the exponent's derivative term v log(a) only where the exponent really depends on a parameter (a constant
// exponent on a negative base must not turn 0 * log(negative) into NaN)
template <typename S, int NP>
GP_HD Dual<S, NP> pow(const Dual<S, NP>& a, const Dual<S, NP>& b) {
    Dual<S, NP> r;
    r.v = ::pow(a.v, b.v);
    const S da = (b.v == (S)0 || (a.v == (S)0 && b.v < (S)1)) ? (S)0 : b.v * ::pow(a.v, b.v - (S)1);   // (a = 0, b < 1: as sqrt at 0)
    bool bdep = false;
    for (int k = 0; k < NP; ++k) bdep = bdep || b.g[k] != (S)0;
    const S db = bdep ? r.v * ::log(a.v) : (S)0;
    // (a.g[k] == 0 skips the base's term: Power(r2, 0.5) on the diagonal has da = inf and a.g = 0 -- inf * 0 would
    //  turn every partial derivative into NaN; sqrt() above guards the same point)
    for (int k = 0; k < NP; ++k)
        r.g[k] = (a.g[k] != (S)0 ? da * a.g[k] : (S)0) + (b.g[k] != (S)0 ? db * b.g[k] : (S)0);
    return r;
}
template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> pow(const Dual<S, NP>& a, A b) { return pow(a, Dual<S, NP>(b)); }
template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> pow(A a, const Dual<S, NP>& b) { return pow(Dual<S, NP>(a), b); }
template <typename S, int NP> GP_HD Dual<S, NP> fmin(const Dual<S, NP>& a, const Dual<S, NP>& b) { return b.v < a.v ? b : a; }
template <typename S, int NP> GP_HD Dual<S, NP> fmax(const Dual<S, NP>& a, const Dual<S, NP>& b) { return b.v > a.v ? b : a; }
template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> fmin(const Dual<S, NP>& a, A b) { return fmin(a, Dual<S, NP>(b)); }
template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> fmin(A a, const Dual<S, NP>& b) { return fmin(Dual<S, NP>(a), b); }
template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> fmax(const Dual<S, NP>& a, A b) { return fmax(a, Dual<S, NP>(b)); }
template <typename S, int NP, GP_NUM_ARG(A)> GP_HD Dual<S, NP> fmax(A a, const Dual<S, NP>& b) { return fmax(Dual<S, NP>(a), b); }

// hyper-parameter k of the function as the arithmetic type the body is instantiated with (the body's P(k))
template <typename T> struct gp_param_of {
    static GP_HD T get(const double* __restrict__ p, int k) { return (T)p[k]; }
};
template <typename S, int NP> struct gp_param_of<Dual<S, NP>> {
    static GP_HD Dual<S, NP> get(const double* __restrict__ p, int k) { return Dual<S, NP>::param((S)p[k], k); }
};
#undef GP_NUM_ARG

}  // namespace gphip
