// HBM read-rate probe (development tool): 4.3 GB read once per launch by workgroups that each follow
// one stream, with the access pattern and the number of 16-byte loads in flight per thread varied.
//   pattern 0: each workgroup's stream is contiguous
//   pattern 1: 256-byte pieces at a 2 KiB stride (one head of a (key, 8 heads, d=128) cache row)
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// DEPTH loads of 16 bytes per thread in flight; a workgroup of 256 threads covers DEPTH * 4 KiB per round.
template <int DEPTH, int PATTERN, int TWO_STAGE, int DESC>
__global__ void __launch_bounds__(256, 2) read_kernel(const char *base, size_t stream_bytes, int *sink, size_t v_skew) {
    const int tid = threadIdx.x;
    const size_t wg = blockIdx.x;
    // pattern 1: stream `wg` = (row block wg / 8, head wg % 8): piece k of the stream sits at
    // (wg / 8) * 8 * stream_bytes + k * 2048 + (wg % 8) * 256
    const size_t round_bytes = (size_t)DEPTH * 4096;
    const char *p0 = PATTERN == 0 ? base + wg * stream_bytes
                     : PATTERN == 1 ? base + (wg / 8) * 8 * stream_bytes + (wg % 8) * 256
                                    : base + (wg / 8) * 8 * (stream_bytes / 2) + (wg % 8) * 256;
    auto addr = [&](size_t off) -> const char * {  // off = byte offset within the stream
        if (DESC) off = stream_bytes - round_bytes - (off / round_bytes) * round_bytes + off % round_bytes;
        if (PATTERN == 0) return p0 + off;
        if (PATTERN == 2) {
            // first half of a round from the K half of the buffer, second half from the V half
            const size_t rnd = off / round_bytes, in = off % round_bytes;
            const size_t half = round_bytes / 2;
            const size_t o2 = rnd * half + in % half;           // offset within the half-length stream
            const char *b2 = p0 + (in >= half ? v_skew : 0);
            return b2 + (o2 / 256) * 2048 + (o2 % 256);
        }
        return p0 + (off / 256) * 2048 + (off % 256);
    };
    int acc = 0;
    i32x4 r[DEPTH];
    if (TWO_STAGE) {
        // half of the loads are re-issued while the other half is still in flight
        constexpr int H = DEPTH / 2;
#pragma unroll
        for (int i = 0; i < DEPTH; i++) r[i] = *(const i32x4 *)addr((size_t)i * 4096 + tid * 16);
        for (size_t off = round_bytes; off + round_bytes <= stream_bytes; off += round_bytes) {
#pragma unroll
            for (int i = 0; i < H; i++) acc ^= r[i][0] ^ r[i][3];
#pragma unroll
            for (int i = 0; i < H; i++) r[i] = *(const i32x4 *)addr(off + (size_t)i * 4096 + tid * 16);
#pragma unroll
            for (int i = H; i < DEPTH; i++) acc ^= r[i][0] ^ r[i][3];
#pragma unroll
            for (int i = H; i < DEPTH; i++) r[i] = *(const i32x4 *)addr(off + (size_t)i * 4096 + tid * 16);
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++) acc ^= r[i][0] ^ r[i][3];
    } else {
        for (size_t off = 0; off + round_bytes <= stream_bytes; off += round_bytes) {
#pragma unroll
            for (int i = 0; i < DEPTH; i++) r[i] = *(const i32x4 *)addr(off + (size_t)i * 4096 + tid * 16);
#pragma unroll
            for (int i = 0; i < DEPTH; i++) acc ^= r[i][0] ^ r[i][3];
        }
    }
    if (acc == 0x12345678) sink[0] = acc;
}

template <int DEPTH, int PATTERN, int TWO, int DESC>
static float run(const char *buf, size_t total, int wgs, int *sink, int lds, size_t v_skew = 0) {
    const size_t stream = total / wgs;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < 12; it++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((read_kernel<DEPTH, PATTERN, TWO, DESC>), dim3(wgs), dim3(256), lds, 0, buf, stream, sink, v_skew);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const size_t total = (size_t)128 * 8192 * 8 * 128 * 2 * 2;  // K + V of the decode shape
    char *buf; int *sink;
    if (hipMalloc(&buf, total + (2 << 20)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, total);
    hipDeviceSynchronize();
#define RUN(D, P, T, W, DESC, LDS) { float ms = run<D, P, T, DESC>(buf, total, W, sink, LDS); \
    printf("depth %2d pattern %d two-stage %d descending %d lds %5d wgs %5d: %.3f ms  %.0f GB/s\n", D, P, T, DESC, LDS, W, ms, total / ms / 1e6); fflush(stdout); }
    const int lds = 65536;
    RUN(16, 0, 0, 1024, 0, lds) RUN(16, 1, 0, 1024, 0, lds) RUN(16, 1, 0, 1024, 1, lds)
    for (size_t skew : {(size_t)0, (size_t)256, (size_t)4096, (size_t)65536, (size_t)(1 << 20), (size_t)(3 << 19)}) {
        // V half = buffer + total/2 + skew (the buffer has 2 MiB of slack at its end)
        float ms = run<16, 2, 0, 1>(buf, total, 1024, sink, lds, total / 2 + skew);
        printf("two streams (K, V = K + 2^31 + %zu), descending: %.3f ms  %.0f GB/s\n", skew, ms, total / ms / 1e6); fflush(stdout);
        ms = run<16, 2, 0, 0>(buf, total, 1024, sink, lds, total / 2 + skew);
        printf("two streams (K, V = K + 2^31 + %zu), ascending:  %.3f ms  %.0f GB/s\n", skew, ms, total / ms / 1e6); fflush(stdout);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    return 0;
}
