// rtc_dyn.h -- hiprtc bound at run time (dlopen), for handles whose covariance function arrives as source text
// (gphip_create_custom).  The reference evaluates ANY `kernel @@ points[[{i,j}]]` (BGP:29-33); the named kernels of this
// library are compiled offline, everything else is compiled here into the SAME kernel build: the text of gp_kernels.h is
// EMBEDDED in the library at build time (.incbin below -- a libgphip.so copied anywhere can create such handles, and the text
// can never describe other structs than the ones the library was compiled with), its [rtc-begin] .. [rtc-end] region is
// prefixed with GP_CUSTOM_KERNEL and followed by the caller's function.  $GPHIP_SRC_DIR (a directory holding a gp_kernels.h)
// overrides the embedded text for development; it must carry the library's GP_RTC_ABI.  Code objects are cached per process
// by (function text, type, architecture): a second handle with the same function costs a hash lookup, not a compilation.
// libgphip.so carries no link-time dependency on hiprtc; a process that never creates such a handle never loads it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include "gp_kernels.h"

#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

// the text of gp_kernels.h, as compiled (the build passes -I <csrc>; host pass only)
#if !defined(__HIP_DEVICE_COMPILE__)
__asm__(".pushsection .rodata\n"
        ".hidden gphip_rtc_embedded_src\n"
        ".global gphip_rtc_embedded_src\n"
        "gphip_rtc_embedded_src:\n"
        ".incbin \"gp_kernels.h\"\n"
        ".byte 0\n"
        ".hidden gphip_rtc_embedded_dual\n"
        ".global gphip_rtc_embedded_dual\n"
        "gphip_rtc_embedded_dual:\n"
        ".incbin \"gp_dual.h\"\n"
        ".byte 0\n"
        ".popsection\n");
#endif
extern "C" const char gphip_rtc_embedded_src[];
extern "C" const char gphip_rtc_embedded_dual[];       // gp_dual.h: the forward-mode type the gradient instantiates the function with

namespace gphip {

struct RtcApi {
    void* so = nullptr;
    typedef struct _hiprtcProgram* prog_t;
    int (*CreateProgram)(prog_t*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*DestroyProgram)(prog_t*) = nullptr;
    int (*AddNameExpression)(prog_t, const char*) = nullptr;
    int (*CompileProgram)(prog_t, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(prog_t, size_t*) = nullptr;
    int (*GetProgramLog)(prog_t, char*) = nullptr;
    int (*GetLoweredName)(prog_t, const char*, const char**) = nullptr;
    int (*GetCodeSize)(prog_t, size_t*) = nullptr;
    int (*GetCode)(prog_t, char*) = nullptr;
    bool ok() const { return so != nullptr; }
};

inline const RtcApi& rtc() {
    static RtcApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("GPHIP_HIPRTC_PATH");
        const char* cands[] = {env, "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
        for (const char* c : cands) {
            if (!c || !*c) continue;
            void* so = dlopen(c, RTLD_NOW);
            if (!so) continue;
            RtcApi a;
            a.so = so;
#define GP_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(so, sym))
            GP_SYM(CreateProgram, "hiprtcCreateProgram");
            GP_SYM(DestroyProgram, "hiprtcDestroyProgram");
            GP_SYM(AddNameExpression, "hiprtcAddNameExpression");
            GP_SYM(CompileProgram, "hiprtcCompileProgram");
            GP_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
            GP_SYM(GetProgramLog, "hiprtcGetProgramLog");
            GP_SYM(GetLoweredName, "hiprtcGetLoweredName");
            GP_SYM(GetCodeSize, "hiprtcGetCodeSize");
            GP_SYM(GetCode, "hiprtcGetCode");
#undef GP_SYM
            if (a.CreateProgram && a.DestroyProgram && a.AddNameExpression && a.CompileProgram && a.GetProgramLogSize &&
                a.GetProgramLog && a.GetLoweredName && a.GetCodeSize && a.GetCode) {
                api = a;
                return;
            }
            dlclose(so);
        }
    });
    return api;
}

// the [rtc-begin] .. [rtc-end] region of gp_kernels.h: the embedded text, or $GPHIP_SRC_DIR/gp_kernels.h (developer override)
inline bool rtc_kernel_source(std::string& region, std::string& dual, std::string& why) {
    std::string text, where = "the text embedded in the library";
    if (const char* env = getenv("GPHIP_SRC_DIR")) {
        where = std::string(env) + "/gp_kernels.h";
        std::ifstream f(where), f2(std::string(env) + "/gp_dual.h");
        if (!f || !f2) { why = where + " / gp_dual.h (GPHIP_SRC_DIR) cannot be read"; return false; }
        std::stringstream ss, s2;
        ss << f.rdbuf();
        s2 << f2.rdbuf();
        text = ss.str();
        dual = s2.str();
    } else {
        text = gphip_rtc_embedded_src;
        dual = gphip_rtc_embedded_dual;
    }
    const size_t b = text.find("// [rtc-begin]"), e = text.find("// [rtc-end]");
    if (b == std::string::npos || e == std::string::npos || e < b) { why = where + " has no [rtc-begin] / [rtc-end] region"; return false; }
    region = text.substr(b, e - b);
    // the text must describe the same structs the library was compiled with
    const std::string tag = "#define GP_RTC_ABI ";
    const size_t t = region.find(tag);
    if (t == std::string::npos || atoi(region.c_str() + t + tag.size()) != GP_RTC_ABI) {
        why = where + " does not match this build of the library (GP_RTC_ABI differs): rebuild, or unset GPHIP_SRC_DIR";
        return false;
    }
    return true;
}

constexpr int RTC_SPEC_MAXD = 32;          // beyond: the generic-dimension program (its loops stay rolled, the points come from global memory)
struct RtcResult {
    std::vector<char> code;                      // the code object for hipModuleLoadData
    std::string build, diag, prep;               // lowered names of the three kernels of the value program ...
    std::string grad;                            // ... or of the gradient program's one kernel (custom_grad_kernel)
};

// body: the statements of   template <typename T> T k(X, Y, P, D)   -- X(k) / Y(k) coordinate k of the two points, P(k)
// hyper-parameter k (of type T), D the input dimension, T the arithmetic type; must `return` the covariance.
// grad_ncp < 0: the value program (kernel build, k(x, x) kernels).  grad_ncp >= 1: the gradient program -- the same text
// instantiated with T = Dual<S, grad_ncp> inside custom_grad_kernel (intermediates that depend on P must be of type T there,
// which they are in a body that follows the documented form).
inline bool rtc_compile_uncached(const std::string& body, int dtype, const char* arch, int grad_ncp, int spec_d, RtcResult& out, std::string& why) {
    const RtcApi& api = rtc();
    if (!api.ok()) { why = "hiprtc could not be loaded (libhiprtc.so; set GPHIP_HIPRTC_PATH)"; return false; }
    std::string region, dual;
    if (!rtc_kernel_source(region, dual, why)) return false;
    std::string src = "#define GP_CUSTOM_KERNEL 1\n#define GP_HD __device__ __forceinline__\n";
    if (grad_ncp >= 1) src += "#define GP_CUSTOM_GRAD 1\n#define GP_NCP " + std::to_string(grad_ncp) + "\n";
    // the program of ONE handle: its input dimension as a compile-time constant (up to RTC_SPEC_MAXD), so that `D` in the
    // function's text is one and its dimension loops unroll
    if (spec_d >= 1 && spec_d <= RTC_SPEC_MAXD) src += "#define GP_D " + std::to_string(spec_d) + "\n";
    src += dual + "\n" + region;
    // names a Mathematica CForm of the function uses (GPHIP.wl translates a pure-function kernel that way)
    src += "\nnamespace gphip {\n"
           "template <typename A, typename B> __device__ __forceinline__ auto Power(A a, B b) -> decltype(a * b * 1.0f) { typedef decltype(a * b * 1.0f) R; return pow((R)a, (R)b); }\n"
           // (an integer exponent -- what CForm prints for x^2, 1/l^2 -- is repeated multiplication: exact derivative, no pow call)
           "template <typename A> __device__ __forceinline__ auto Power(A a, int n) -> decltype(a * 1.0f) { typedef decltype(a * 1.0f) R; R r = (R)1, b = (R)a; "
           "unsigned m = n < 0 ? 0u - (unsigned)n : (unsigned)n; while (m) { if (m & 1u) r = r * b; m >>= 1; if (m) b = b * b; } return n < 0 ? (R)1 / r : r; }\n"
           "template <typename A> __device__ __forceinline__ A Sqrt(A a) { return sqrt(a); }\n"
           "template <typename A> __device__ __forceinline__ A Exp(A a) { return exp(a); }\n"
           "template <typename A> __device__ __forceinline__ A Log(A a) { return log(a); }\n"
           "template <typename A> __device__ __forceinline__ A Abs(A a) { return fabs(a); }\n"
           "template <typename A> __device__ __forceinline__ A Sin(A a) { return sin(a); }\n"
           "template <typename A> __device__ __forceinline__ A Cos(A a) { return cos(a); }\n"
           "template <typename A> __device__ __forceinline__ A Tanh(A a) { return tanh(a); }\n"
           "template <typename A> __device__ __forceinline__ A Sinh(A a) { return sinh(a); }\n"
           "template <typename A> __device__ __forceinline__ A Cosh(A a) { return cosh(a); }\n"
           "template <typename A> __device__ __forceinline__ A Tan(A a) { return tan(a); }\n"
           "template <typename A> __device__ __forceinline__ A ArcTan(A a) { return atan(a); }\n"
           "template <typename A> __device__ __forceinline__ A Erf(A a) { return erf(a); }\n"
           "template <typename A> __device__ __forceinline__ A Erfc(A a) { return erfc(a); }\n"
           "template <typename A, typename B> __device__ __forceinline__ auto Min(A a, B b) -> decltype(a + b) { typedef decltype(a + b) R; return a < b ? (R)a : (R)b; }\n"
           "template <typename A, typename B> __device__ __forceinline__ auto Max(A a, B b) -> decltype(a + b) { typedef decltype(a + b) R; return a > b ? (R)a : (R)b; }\n"
           "constexpr double Pi = 3.14159265358979323846, E = 2.71828182845904523536;\n"
           "template <typename T, typename S>\n"
           "__device__ __forceinline__ T gphip_custom_k(PointRef<S> X, PointRef<S> Y, const double* __restrict__ Pp, int gphip_d_) {\n"
           "#ifdef GP_D\n    constexpr int D = GP_D; (void)gphip_d_;\n#else\n    const int D = gphip_d_;\n#endif\n"
           "#define P(k) (gp_param_of<T>::get(Pp, (k)))\n";
    src += body;
    src += "\n#undef P\n}\n}  // namespace gphip\n";
    const std::string ty = dtype == 64 ? "double" : "float";
    std::vector<std::pair<std::string, std::string*>> names;
    if (grad_ncp >= 1) {
        names.push_back({"gphip::custom_grad_kernel<" + ty + ">", &out.grad});
    } else {
        names.push_back({"gphip::kbuild_kernel<" + ty + ", 0, 3>", &out.build});
        names.push_back({"gphip::custom_diag_kernel<" + ty + ">", &out.diag});
        names.push_back({"gphip::custom_prep_kernel<" + ty + ">", &out.prep});
    }
    RtcApi::prog_t prog = nullptr;
    if (api.CreateProgram(&prog, src.c_str(), "gphip_custom_kernel.hip", 0, nullptr, nullptr) != 0) {
        why = "hiprtcCreateProgram failed";
        return false;
    }
    for (auto& n : names) api.AddNameExpression(prog, n.first.c_str());
    const std::string archopt = std::string("--offload-arch=") + arch;
    // -freciprocal-math: x / p may become x * (1 / p) -- with the hyper-parameters loop invariant (see kbuild_kernel) the per-entry
    // fp64 divisions of a typical function (one per dimension, ~15 instructions each) leave the inner loop; <= 1 ulp per quotient,
    // the same form the named kernels use (inputs pre-multiplied by 1 / l_k).  No other fast-math relaxation.
    const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-freciprocal-math"};
    const int rc = api.CompileProgram(prog, 4, opts);
    if (rc != 0) {
        size_t n = 0;
        api.GetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) api.GetProgramLog(prog, &log[0]);
        why = "the covariance function does not compile:\n" + log;
        api.DestroyProgram(&prog);
        return false;
    }
    bool ok = true;
    for (auto& n : names) {
        const char* low = nullptr;
        ok = ok && api.GetLoweredName(prog, n.first.c_str(), &low) == 0 && low;
        if (ok) *n.second = low;
    }
    size_t sz = 0;
    ok = ok && api.GetCodeSize(prog, &sz) == 0 && sz > 0;
    if (ok) {
        out.code.resize(sz);
        ok = api.GetCode(prog, out.code.data()) == 0;
    }
    api.DestroyProgram(&prog);
    if (!ok) why = "hiprtc produced no code object";
    return ok;
}

// per-process cache of code objects, keyed by everything the compilation depends on (the kernel text is fixed per process)
inline std::shared_ptr<const RtcResult> rtc_compile_custom(const std::string& body, int dtype, const char* arch, std::string& why,
                                                            bool* cache_hit = nullptr, int grad_ncp = -1, int spec_d = 0) {
    if (spec_d < 1 || spec_d > RTC_SPEC_MAXD) spec_d = 0;
    static std::mutex mu;
    static std::map<std::string, std::shared_ptr<const RtcResult>> cache;
    const char* dev = getenv("GPHIP_SRC_DIR");
    const std::string key = std::string(arch) + "|" + std::to_string(dtype) + "|" + std::to_string(grad_ncp) + "|" + std::to_string(spec_d) + "|" + (dev ? dev : "") + "|" + body;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (cache_hit) *cache_hit = it != cache.end();
    if (it != cache.end()) return it->second;
    auto r = std::make_shared<RtcResult>();
    if (!rtc_compile_uncached(body, dtype, arch, grad_ncp, spec_d, *r, why)) return nullptr;
    cache.emplace(key, r);
    return r;
}

}  // namespace gphip
