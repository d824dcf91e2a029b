// Probes the lane layout of v_mfma_f64_4x4x4_4b_f64: for every pair of lanes (pa, pb), A = e_pa, B = e_pb,
// which output lanes receive the product.  Prints, per output lane, the contributing (pa, pb) pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned char *hit) {     // hit[ld][pa][pb]
    const int l = threadIdx.x;
    for (int pa = 0; pa < 64; ++pa)
        for (int pb = 0; pb < 64; ++pb) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == pa ? 1.0 : 0.0, l == pb ? 1.0 : 0.0, 0.0, 0, 0, 0);
            hit[(l * 64 + pa) * 64 + pb] = d != 0.0;
        }
}
int main() {
    unsigned char *dh; static unsigned char h[64 * 64 * 64];
    hipMalloc(&dh, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dh);
    hipMemcpy(h, dh, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("D lane %2d <-", l);
        for (int pa = 0; pa < 64; ++pa) for (int pb = 0; pb < 64; ++pb) if (h[(l * 64 + pa) * 64 + pb]) printf(" (a%d,b%d)", pa, pb);
        printf("\n");
    }
    return 0;
}
