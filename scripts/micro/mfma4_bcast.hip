// Probes v_mfma_f64_4x4x4_4b_f64 with cbsz = 2 (one A block broadcast to the four blocks), abid = 0..3:
// for every pair of input lanes (pa, pb) with A = e_pa, B = e_pb, which output lanes receive the product.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ABID>
__global__ void k(unsigned char *hit) {     // hit[ld][pa][pb]
    const int l = threadIdx.x;
    for (int pa = 0; pa < 64; ++pa)
        for (int pb = 0; pb < 64; ++pb) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == pa ? 1.0 : 0.0, l == pb ? 1.0 : 0.0, 0.0, 2, ABID, 0);
            hit[(l * 64 + pa) * 64 + pb] = d != 0.0;
        }
}
static unsigned char h[64 * 64 * 64];
template <int ABID>
void run(unsigned char *dh) {
    hipLaunchKernelGGL(k<ABID>, dim3(1), dim3(64), 0, 0, dh);
    hipMemcpy(h, dh, sizeof h, hipMemcpyDeviceToHost);
    printf("abid %d\n", ABID);
    for (int l = 0; l < 64; l += (l < 8 ? 1 : 13)) {
        printf("D lane %2d <-", l);
        for (int pa = 0; pa < 64; ++pa) for (int pb = 0; pb < 64; ++pb) if (h[(l * 64 + pa) * 64 + pb]) printf(" (a%d,b%d)", pa, pb);
        printf("\n");
    }
}
int main() {
    unsigned char *dh;
    hipMalloc(&dh, sizeof h);
    run<0>(dh); run<1>(dh); run<2>(dh); run<3>(dh);
    return 0;
}
