"""Static consistency of the Wolfram-Language side of the boundary (no Wolfram kernel exists here, so the
package itself cannot run): every LibraryFunctionLoad in bayesianinference_amd/wl/GPHIP.wl must name a function
the LibraryLink shim exports, with the argument count the shim checks; the shim must compile (g++, against the
tests-only stand-in header) and reject a wrong argument count with LIBRARY_FUNCTION_ERROR; and the package must
keep the reference-shaped keys of "GaussianProcessData" (BayesianGaussianProcess.wl:314-321)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = os.path.join(ROOT, "bayesianinference_amd", "wl", "GPHIP.wl")
WL_SAMPLER = os.path.join(ROOT, "bayesianinference_amd", "wl", "GPHIPSampler.wl")      # nestedSamplingHIP, loaded by GPHIP.wl
SHIM = os.path.join(ROOT, "bayesianinference_amd", "csrc", "librarylink_shim.cpp")


def wl_bindings():
    """{shim function name: number of arguments} from the load["name", {argument types}, result] bindings of the two package
    files (load = LibraryFunctionLoad[$GPHIPLibrary, ..]; the type shorthands m2, v1, any, iv count as one argument each)."""
    text = open(WL).read() + open(WL_SAMPLER).read()
    text = re.sub(r"\(\*.*?\*\)", "", text, flags=re.S)
    assert "load[name_, args_, ret_] := LibraryFunctionLoad[$GPHIPLibrary, name, args, ret];" in text
    out = {}
    for m in re.finditer(r'\bload\["(\w+)",\s*\{', text):
        i, depth, args, seen = m.end(), 1, 0, False
        while depth:                                   # count top-level elements of the argument list
            ch = text[i]
            if ch in "{[":
                depth += 1
                seen = True
            elif ch in "}]":
                depth -= 1
            elif ch == "," and depth == 1:
                args += 1
            elif not ch.isspace():
                seen = True
            i += 1
        out[m.group(1)] = args + 1 if seen else 0      # ({} = no arguments)
    return out


def shim_argc():
    """{function name: N} from `if (argc != N)` at the top of every entry point of the shim."""
    text = open(SHIM).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(
        r"EXTERN_C DLLEXPORT int (gphip_wl_\w+)\([^)]*\)\s*\{\s*if \(argc != (\d+)\)", text)}


def test_every_wl_binding_matches_a_shim_entry_point():
    wl, shim = wl_bindings(), shim_argc()
    assert len(wl) == 22 and set(wl) == set(shim), (sorted(wl), sorted(shim))
    assert wl == shim, {k: (wl[k], shim[k]) for k in wl if wl[k] != shim[k]}


def test_shim_compiles_against_stub_header_and_exports_the_lifecycle():
    from bayesianinference_amd import _lib, build
    _lib.load()                                        # maps libgphip.so (and its HIP runtime) first
    path = build.build_wl_stub()
    lib = C.CDLL(path)
    for name in list(shim_argc()) + ["WolframLibrary_getVersion", "WolframLibrary_initialize",
                                      "WolframLibrary_uninitialize"]:
        assert hasattr(lib, name), name
    lib.WolframLibrary_getVersion.restype = C.c_int64
    assert lib.WolframLibrary_getVersion() >= 1
    lib.drv_libdata.restype = C.c_void_p
    data = C.c_void_p(lib.drv_libdata())
    assert lib.WolframLibrary_initialize(data) == 0
    # a wrong argument count is LIBRARY_FUNCTION_ERROR (6) before anything is touched
    for name, n in shim_argc().items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        assert fn(data, n + 1, None, None) == 6, name


def test_wl_package_stays_thin():
    """the untestable WL side stays small, all logic that needs no Wolfram kernel lives behind the C ABI."""
    assert len(open(WL).read().splitlines()) <= 300
    assert len(open(WL_SAMPLER).read().splitlines()) <= 100
    for path in (WL, WL_SAMPLER):                      # no text processing / parsing left on this side
        text = open(path).read()
        assert "RegularExpression" not in text and "StringSplit" not in text and "RandomVariate" not in text, path
    for path in (WL, WL_SAMPLER):                      # brackets balance (outside strings and comments)
        t, stack, i = open(path).read(), [], 0
        close = {")": "(", "]": "[", "}": "{"}
        while i < len(t):
            if t.startswith("(*", i):
                depth, i = 1, i + 2
                while depth:
                    if t.startswith("(*", i):
                        depth, i = depth + 1, i + 2
                    elif t.startswith("*)", i):
                        depth, i = depth - 1, i + 2
                    else:
                        i += 1
                continue
            if t[i] == '"':
                i += 1
                while t[i] != '"':
                    i += 2 if t[i] == "\\" else 1
                i += 1
                continue
            if t.startswith("<|", i):
                stack.append("<|"); i += 2; continue
            if t.startswith("|>", i):
                assert stack.pop() == "<|", (path, i); i += 2; continue
            if t[i] in "([{":
                stack.append(t[i])
            elif t[i] in ")]}":
                assert stack.pop() == close[t[i]], (path, t[max(0, i - 60):i + 1])
            i += 1
        assert not stack, (path, stack)


def test_wl_package_keeps_reference_shapes():
    text = open(WL).read()
    for key in ('"KernelFunction"', '"NuggetFunction"', '"MeanFunction"', '"CovarianceFunction"',
                '"InverseCovarianceFunction"', '"LogLikelihoodFunction"', '"Data"', '"PriorDistribution"',
                '"Parameters"'):                       # BGP:310-325
        assert key in text, key
    assert '"Inverse" ->' in text and '"LogDet" ->' in text            # BGP:137-141
    assert 'Throw[$MachineLogZero, "MatInv"]' in text                  # BGP:133
    # the HIP prediction rule must be PREPENDED to the reference's down-values (an appended rule
    # of equal specificity is never reached)
    assert re.search(r"DownValues\[predictFromGaussianProcess\]\s*=\s*Prepend\[", text)
    assert re.search(r"DownValues\[predictiveDistribution\]\s*=\s*Join\[\s*\{", text)      # new rules FIRST
    # brackets NEST properly once comments and strings are blanked (the nearest thing to a syntax check available
    # without a Wolfram kernel; it also catches a "*)" inside a comment, which ends the comment early)
    def blank(m):
        return re.sub(r"[^\n]", " ", m.group(0))
    code = re.sub(r"\(\*.*?\*\)", blank, text, flags=re.S)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', blank, code)
    stack, line = [], 1
    pairs = {")": "(", "]": "[", "}": "{"}
    for ch in code:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in ")]}":
            assert stack and stack[-1][0] == pairs[ch], (ch, line, stack[-1:] )
            stack.pop()
    assert not stack, stack[-3:]
    assert code.count("<|") == code.count("|>")
    # the reference's own argument list, the fall-through for non-native kernels, the device count and the native sampler
    assert re.search(r"defineGaussianProcessHIP\[\s*dataIn_List.*?kerf_, nugf_, meanf_,", text, flags=re.S)     # BGP:228-234
    assert "Return @ defineGaussianProcess[dataIn -> dataOut, kerf, nugf, meanf, variables, variablePrior" in text
    assert "Mod[$KernelID, Max[gpDevices[], 1]]" in text and "Mod[$KernelID, 8]" not in text
    sampler = open(WL_SAMPLER).read()
    assert "evidenceSampling[" in sampler and "gpNested[" in sampler and 'Get[FileNameJoin[{DirectoryName[$InputFileName], "GPHIPSampler.wl"}]]' in text   # BS:1158-1291
    assert "expressionToFunction[nugf, vars -> paramVector]" in text                                       # BGP:257-262
