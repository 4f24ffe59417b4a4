"""The blocked (8 x 8 super-tile) enumeration of the trailing update's tile list (gp_kernels.h: blocked_tri_decode) must be a
bijection onto the lower triangle for every size -- a tile visited twice or never would corrupt the factor silently only
for some N.  The function is pure integer / sqrt arithmetic, so its text is compiled for the HOST (g++) and enumerated."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "bayesianinference_amd", "csrc", "gp_kernels.h")

MAIN = r'''
int main() {
    for (int H = 16; H <= 1800; H += (H < 300 ? 1 : 97)) {
        const long n = (long)H * (H + 1) / 2;
        std::set<std::pair<int, int>> seen;
        for (long p = 0; p < n; ++p) {
            int u, v;
            blocked_tri_decode((int)p, H, u, v);
            if (u < 0 || u >= H || v < 0 || v > u) { printf("H=%d p=%ld bad (%d,%d)\n", H, p, u, v); return 1; }
            if (!seen.insert({u, v}).second) { printf("H=%d p=%ld duplicate (%d,%d)\n", H, p, u, v); return 1; }
        }
        // 64 consecutive positions inside a column of full super-tiles touch 8 rows x 8 columns of tiles
        if (H >= 64) {
            int u0, v0, u1, v1;
            blocked_tri_decode(36, H, u0, v0);
            blocked_tri_decode(36 + 63, H, u1, v1);
            if (u1 - u0 != 7 || v1 - v0 != 7) { printf("H=%d: first full super-tile is not 8 x 8\n", H); return 1; }
        }
    }
    printf("ok\n");
    return 0;
}
'''


def _function_text(name_start: str, name_end: str) -> str:
    text = open(HDR).read()
    return text[text.index(name_start):text.index(name_end)]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_blocked_tile_order_is_a_bijection(tmp_path):
    tri = _function_text("__device__ __forceinline__ void tri_decode(", "template <typename T>\nstruct KBuildArgs")
    blk = _function_text("__device__ __forceinline__ void blocked_tri_decode(",
                         "template <typename T>\n__device__ __forceinline__ void gemm_tile_decode(")
    src = tmp_path / "order.cpp"
    src.write_text("#include <cmath>\n#include <cstdio>\n#include <set>\n#include <utility>\n#define __device__\n"
                   "#define __forceinline__ inline\n#define __builtin_sqrt sqrt\n" + tri + blk + MAIN)
    exe = tmp_path / "order"
    subprocess.run(["g++", "-O2", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
