#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
// Does an out-of-range lane of buffer_load_dwordx4 ... lds write zeros to LDS, or leave it alone?
__global__ void k(const float *g, float *out, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    float *s = (float *)smem;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = NAN;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, nbytes, 0x00020000);
    const unsigned voff = (lane & 1) ? 0x80000000u : lane * 16;   // odd lanes: far out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem), 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = s[i];
}
int main() {
    float *g, *o; float h[256], gi[1024];
    for (int i = 0; i < 1024; i++) gi[i] = 1000 + i;
    hipMalloc(&g, 4096); hipMalloc(&o, 1024);
    hipMemcpy(g, gi, 4096, hipMemcpyHostToDevice);
    k<<<1, 64, 4096>>>(g, o, 40 * 16);   // lanes >= 40 are past num_records too
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l < 6 || (l > 36 && l < 44)) printf("lane %d: %g %g %g %g\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}
