// Micro-benchmark: cost of v_mfma_f32_32x32x16_bf16 with the B operand in an AGPR vs a VGPR
// (C/D in AGPRs in both cases), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(unsigned long long *out, float *sink) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    f32x16 c0 = {0}, c1 = {0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; it++) {
        if (MODE == 0) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                : "+a"(c0), "+a"(c1) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                : "+a"(c0), "+a"(c1) : "v"(a), "a"(b));
        } else if (MODE == 2) {  // C/D in VGPRs
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else {  // single dependent chain, AGPR C/D, VGPR B
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                : "+a"(c0), "+a"(c1) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    float s = 0;
    for (int j = 0; j < 16; j++) s += c0[j] + c1[j];
    if (s == 12345.f) sink[0] = s;
}

int main() {
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4);
    const char *names[] = {"B in VGPR, C in AGPR, 2 chains", "B in AGPR, C in AGPR, 2 chains", "all VGPR, 2 chains", "B VGPR, C AGPR, 1 chain"};
    for (int m = 0; m < 4; m++) {
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) k<0><<<256, 256>>>(d, sink);
            if (m == 1) k<1><<<256, 256>>>(d, sink);
            if (m == 2) k<2><<<256, 256>>>(d, sink);
            if (m == 3) k<3><<<256, 256>>>(d, sink);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per MFMA\n", names[m], (double)h / (64 * 8));
    }
    return 0;
}
