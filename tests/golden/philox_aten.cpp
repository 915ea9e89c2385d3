// Prints Philox-4x32 outputs of ATen's engine (ATen/core/PhiloxRNGEngine.h, an independent
// implementation of the published Philox algorithm that ships with PyTorch) for a fixed list of
// (seed, subsequence, offset) triples at 7 and 10 rounds, as JSON.  make_philox_golden.py compiles
// and runs this to produce philox_aten.json; the oracle's Philox restatement is checked against it.
#include <ATen/core/PhiloxRNGEngine.h>
#include <cstdint>
#include <cstdio>

int main() {
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    std::printf("[\n");
    for (int i = 0; i < 48; i++) {
        uint64_t seed = i == 0 ? 0 : next(), sub = i == 0 ? 0 : next(), off = i == 0 ? 0 : next();
        if (i % 3 == 1) { sub &= 0xffff0000ffffull; off &= 0xffffffull; }  // shapes the kernels use
        for (int rounds : {7, 10}) {
            at::philox_engine e(seed, sub, off);
            uint32_t o[4];
            for (int k = 0; k < 4; k++) o[k] = e(rounds);
            std::printf("  {\"seed\": \"%llu\", \"subsequence\": \"%llu\", \"offset\": \"%llu\", \"rounds\": %d, "
                        "\"out\": [%u, %u, %u, %u]}%s\n",
                        (unsigned long long)seed, (unsigned long long)sub, (unsigned long long)off, rounds,
                        o[0], o[1], o[2], o[3], (i == 47 && rounds == 10) ? "" : ",");
        }
    }
    std::printf("]\n");
    return 0;
}
