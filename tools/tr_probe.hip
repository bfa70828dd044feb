// what ds_read_b64_tr_b16 delivers (gfx950): LDS image [64 rows][128 cols] of u16 = row * 128 + col; lane 4q+p of a 16-lane
// group g supplies the address of row q, columns 16g + 4p .. +3; prints each lane's four elements as (row, col)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const unsigned short* a = lds + q * 128 + 16 * g + 4 * pp;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (%d,%d)", h[l * 4 + e] / 128, h[l * 4 + e] % 128);
        printf("\n");
    }
    return 0;
}
