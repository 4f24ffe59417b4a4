// Host check of csrc/gp_dual.h (tests/test_dual_numbers.py): every operator and function of the forward-mode type against
// central differences of the same expression evaluated in plain doubles.  Prints "name value_err max_grad_err" lines.
#include <cmath>
#include <cstdio>
#define GP_HD inline
#include "gp_dual.h"

using gphip::Dual;
typedef Dual<double, 3> D3;

template <typename T> struct Par { T a, b, c; };

// expressions in the style of a caller's covariance function body: constants, mixed scalar / T arithmetic, every function
template <typename T> T f_arith(Par<T> p, double x) {
    T s = 0;
    s += p.a * x;
    s -= 2 * p.b;
    s *= (p.c + 1.5f);
    s /= (1 + p.a * p.a);
    return -s + (3 - p.b) / (p.c * 2.0) + x / p.a - (T)0.25 * s;
}
template <typename T> T f_explog(Par<T> p, double x) { return exp(-p.a * x) + log(p.b + x * x) + log1p(p.c * p.c) + expm1(p.a * 0.1); }
template <typename T> T f_sqrtpow(Par<T> p, double x) { return sqrt(p.a + x * x) + pow(p.b, 2) + pow(p.c, p.a) + pow(x + 2.0, p.b) + pow(p.a * p.b, 1.5); }
template <typename T> T f_trig(Par<T> p, double x) { return sin(p.a * x) * cos(p.b) + tan(p.c * 0.3) + atan(p.a / p.b); }
template <typename T> T f_hyp(Par<T> p, double x) { return tanh(p.a) + sinh(p.b * x) - cosh(p.c) * 0.01; }
template <typename T> T f_erf(Par<T> p, double x) { return erf(p.a * x) + erfc(p.b) * p.c; }
template <typename T> T f_absminmax(Par<T> p, double x) { return fabs(p.a - 2.0) * fmin(p.b, 5.0) + fmax(p.c, p.a * 4) + fmin(2, p.c) + fabs(-p.b); }
template <typename T> T f_negpow(Par<T> p, double x) { return pow(T(-1.5) * p.a, 2) + pow(p.b - 10.0, 3); }      // negative bases, constant exponents
template <typename T> T f_cmp(Par<T> p, double x) { return (p.a < p.b && p.b >= 0.5 && 1 < p.c && p.c != p.a) ? p.a * p.b : p.c; }
template <typename T> T f_se(Par<T> p, double x) {                // SE kernel with a periodic factor
    const T u = (x - 0.3) / p.a;
    return p.b * p.b * exp((T)-0.5 * u * u) * (1 + p.c * cos(3.14159 * x));
}

template <typename F, typename G>
void check(const char* name, F fd, G fdual) {
    const double th[3] = {0.7, 1.3, 2.1};
    const double x = 0.45;
    Par<D3> pd{D3::param(th[0], 0), D3::param(th[1], 1), D3::param(th[2], 2)};
    const D3 r = fdual(pd, x);
    Par<double> p0{th[0], th[1], th[2]};
    const double v0 = fd(p0, x);
    double worst = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double h = 1e-6 * std::fabs(th[k]);
        double tp[3] = {th[0], th[1], th[2]}, tm[3] = {th[0], th[1], th[2]};
        tp[k] += h; tm[k] -= h;
        const double num = (fd(Par<double>{tp[0], tp[1], tp[2]}, x) - fd(Par<double>{tm[0], tm[1], tm[2]}, x)) / (2 * h);
        const double err = std::fabs(r.g[k] - num) / std::fmax(1.0, std::fabs(num));
        worst = std::fmax(worst, err);
    }
    std::printf("%s %.3e %.3e\n", name, std::fabs(r.v - v0) / std::fmax(1.0, std::fabs(v0)), worst);
}
#define CHECK(f) check(#f, f<double>, f<D3>)

int main() {
    CHECK(f_arith); CHECK(f_explog); CHECK(f_sqrtpow); CHECK(f_trig); CHECK(f_hyp); CHECK(f_erf); CHECK(f_absminmax);
    CHECK(f_negpow); CHECK(f_cmp); CHECK(f_se);
    // the hyper-parameter accessor the generated P(k) expands to
    const double pp[3] = {0.7, 1.3, 2.1};
    const D3 q = gphip::gp_param_of<D3>::get(pp, 1);
    const double s = gphip::gp_param_of<double>::get(pp, 2);
    std::printf("param %.3e %.3e\n", std::fabs(q.v - 1.3) + std::fabs(s - 2.1), std::fabs(q.g[0]) + std::fabs(q.g[1] - 1.0) + std::fabs(q.g[2]));
    // sqrt at 0 and fabs at 0 stay finite
    const D3 z = sqrt(D3::param(0.0, 0)) + fabs(D3::param(0.0, 1));
    std::printf("at_zero %.3e %.3e\n", std::fabs(z.v), (std::isfinite(z.g[0]) && std::isfinite(z.g[1])) ? 0.0 : 1.0);
    // Power(r2, 0.5) on the diagonal: r2 = (x - x)^2 / l^2 is a CONSTANT zero (no parameter dependence), the exponent a constant:
    // b a^(b-1) = inf must not meet the zero partials (inf * 0 = NaN in every derivative); with a parameter-dependent zero
    // base the derivative follows sqrt's convention (0)
    const D3 r2 = D3(0.0) / (D3::param(0.7, 0) * D3::param(0.7, 0));
    const D3 pz = D3::param(1.3, 1) * pow(r2, 0.5) + pow(D3::param(0.0, 2), 0.5) + pow(r2, D3(0.25));
    std::printf("pow_at_zero %.3e %.3e\n", std::fabs(pz.v), (std::isfinite(pz.g[0]) && std::isfinite(pz.g[1]) && std::isfinite(pz.g[2]) &&
                                                            pz.g[0] == 0.0 && pz.g[1] == 0.0 && pz.g[2] == 0.0) ? 0.0 : 1.0);
    return 0;
}
