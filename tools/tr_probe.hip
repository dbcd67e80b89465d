#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_ __attribute__((address_space(3)))
__global__ void k(const int* perm, int* out)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned short* ls = (unsigned short*)smem;           // every short holds its own index
    for (int i = threadIdx.x; i < 4096; i += 64) ls[i] = (unsigned short)i;
    __syncthreads();
    const LDS_ unsigned char* base = (const LDS_ unsigned char*)smem;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_ s16x4*)(base + perm[threadIdx.x] * 8));
    for (int e = 0; e < 4; e++) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}
int main()
{
    int perm[64], inv[512], *dp, *dout, h[256];
    // a scattered permutation: lane L reads the 8 bytes at 8 * perm[L]
    for (int i = 0; i < 64; i++) perm[i] = (i * 37 + 11) % 64 + 64 * (i % 5);
    for (int i = 0; i < 512; i++) inv[i] = -1;
    for (int i = 0; i < 64; i++) inv[perm[i]] = i;
    (void)hipMalloc(&dp, sizeof(perm)); (void)hipMalloc(&dout, sizeof(h));
    hipMemcpy(dp, perm, sizeof(perm), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, dp, dout);
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; e++) {
            const int s = h[l * 4 + e], src = inv[s / 4], sub = s % 4;
            const int exp_src = (l & ~15) + 4 * e + ((l & 15) >> 2), exp_sub = l & 3;
            bad += (src != exp_src) | (sub != exp_sub);
            printf("  e%d <- lane %2d short %d", e, src, sub);
        }
        printf("\n");
    }
    printf("model mismatches: %d\n", bad);
    return 0;
}
