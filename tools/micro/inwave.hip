// Micro-benchmark: vector ALU fillers placed between the MFMAs of the SAME wave.
// Each gap after a v_mfma_f32_32x32x16_bf16 holds NF independent VALU instructions
// (v_fma_f32, every fourth a v_exp_f32).  One or two waves per SIMD run the same program.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int NF, int NT>
__global__ void __launch_bounds__(NT, 1) k(unsigned long long *out, float *sink, float seed) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    f32x16 c0 = {0};
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = seed * (float)(threadIdx.x + j);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if ((f & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[f & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f & 7]) : "v"(seed));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[threadIdx.x >> 6] = t1 - t0; out[8 + (threadIdx.x >> 6)] = t0; out[16 + (threadIdx.x >> 6)] = t1; }
    float s = 0;
    for (int j = 0; j < 16; j++) s += c0[j];
    for (int j = 0; j < 8; j++) s += x[j];
    if (s == 12345.f) sink[0] = s;
}

template <int NF, int NT>
static void run(unsigned long long *d, float *sink) {
    for (int rep = 0; rep < 2; rep++) {
        k<NF, NT><<<256, NT>>>(d, sink, 0.37f);
        hipDeviceSynchronize();
    }
    unsigned long long h[24];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long tmin = h[8], tmax = h[16];
    for (int w = 0; w < NT / 64; w++) { if (h[8 + w] < tmin) tmin = h[8 + w]; if (h[16 + w] > tmax) tmax = h[16 + w]; }
    printf("%d waves/SIMD, %d fillers per MFMA gap: wave 0 %7.1f, wave %d %7.1f, first start to last end %7.1f cycles per 32 MFMAs of each wave (+%d VALU)\n",
           NT / 256, NF, (double)h[0] / 64, NT / 64 - 1, (double)h[NT / 64 - 1] / 64, (double)(tmax - tmin) / 64, 32 * NF);
}

int main() {
    unsigned long long *d; float *sink;
    hipMalloc(&d, 24 * 8); hipMalloc(&sink, 8);
    run<0, 256>(d, sink); run<2, 256>(d, sink); run<4, 256>(d, sink); run<5, 256>(d, sink);
    run<6, 256>(d, sink); run<8, 256>(d, sink);
    run<0, 512>(d, sink); run<2, 512>(d, sink); run<4, 512>(d, sink); run<5, 512>(d, sink); run<6, 512>(d, sink); run<8, 512>(d, sink);
    return 0;
}
