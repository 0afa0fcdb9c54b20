"""chunk_owner / chain_first (mitsuba2_amd/csrc/kernels.h): which chunk of a pass a scheduling wave generates when the waves are cut into
launch chains.  Shared by the host scheduler and the kernels; compiled here for the host and checked to be a bijection of [0, n) for
every chain count, with chain boundaries on multiples of the k_trace group size."""
import os
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "mitsuba2_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

PROGRAM = r"""
#include "kernels.h"
#include <cstdio>
#include <vector>
using namespace mtsamd;
int main() {
    const uint32_t sizes[] = { 256u, 257u, 300u, 1000u, 12345u, 53248u, 53251u };
    for (uint32_t n : sizes)
        for (uint32_t chains = 1; chains <= kMaxChains; ++chains) {
            std::vector<int> seen(n, 0);
            for (uint32_t w = 0; w < n; ++w) {
                const uint32_t c = chunk_owner(w, n, chains);
                if (c >= n || seen[c]++) { std::printf("not a bijection: n %u chains %u wave %u -> %u\n", n, chains, w, c); return 1; }
            }
            for (uint32_t k = 0; k <= chains; ++k) {
                const uint32_t lo = chain_first(k, n, chains);
                if (lo > n || (lo != n && lo % kChainAlign != 0u) || (k && lo < chain_first(k - 1, n, chains))) { std::printf("bad boundary: n %u chains %u k %u -> %u\n", n, chains, k, lo); return 1; }
            }
            if (chain_first(0, n, chains) != 0u || chain_first(chains, n, chains) != n) return 1;
        }
    // two chains: the even chunks go to the first chain, the odd ones to the second while it has waves (round-2 layout)
    for (uint32_t w = 0; w < 1024u; ++w) {
        const uint32_t c = chunk_owner(w, 1024u, 2u);
        if (c != (w < 512u ? 2u * w : 2u * (w - 512u) + 1u)) { std::printf("two-chain layout changed at wave %u: %u\n", w, c); return 1; }
    }
    std::printf("ok\n");
    return 0;
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_chunk_owner_is_a_bijection_for_every_chain_count():
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "t.cpp"), os.path.join(tmp, "t")
        with open(src, "w") as f:
            f.write(PROGRAM)
        subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-I", CSRC, "-o", exe, src],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = subprocess.check_output([exe]).decode()
        assert out.strip() == "ok", out
