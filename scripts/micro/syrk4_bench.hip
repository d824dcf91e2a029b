// Stand-alone check + timing of k_schur_syrk4 (v_mfma_f64_4x4x4_4b register tiles) against k_schur_syrk (16x16x4)
// on random Z.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cvx_amd/csrc scripts/micro/syrk4_bench.hip -o scripts/micro/syrk4_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "calib_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace cvxk;

int main(int argc, char **argv) {
    const int ZC = argc > 1 ? atoi(argv[1]) : 336, rows = argc > 2 ? atoi(argv[2]) : 60000, want_chunks = argc > 3 ? atoi(argv[3]) : 128;
    const int nb = ZC / 16, nblk = nb * (nb + 1) / 2, ZCp = ZC + ((ZC % 32 == 16) ? 0 : 16);
    int KB = 16;
    while (KB > 4 && KB * ZC > 6144) KB -= 4;     // as cvx_calib_create sizes the 16x16x4 kernel's stage
    std::vector<double> Z((size_t)rows * ZC);
    srand(3);
    for (auto &v : Z) v = rand() / (double)RAND_MAX - 0.5;
    std::vector<K4Task> tasks;
    const int parts4 = k4_build_tasks(nb, tasks);
    const int parts3 = (nblk + K3_CAP - 1) / K3_CAP;
    const int n_chunks = want_chunks;
    std::vector<int> row0(n_chunks + 1);
    for (int c = 0; c <= n_chunks; ++c) row0[c] = (int)((long long)rows * c / n_chunks);
    printf("ZC=%d nb=%d nblk=%d rows=%d chunks=%d  parts: 16x16x4 %d, 4x4x4 %d (%zu tasks)\n", ZC, nb, nblk, rows, n_chunks, parts3, parts4, tasks.size());
    double *dZ, *s3, *s4; int *dr; K4Task *dt;
    const size_t slab = (size_t)n_chunks * nblk * 256;
    CK(hipMalloc(&dZ, (Z.size() + 8 * (size_t)ZC) * 8)); CK(hipMemset(dZ, 0, (Z.size() + 8 * (size_t)ZC) * 8));
    CK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&s3, slab * 8)); CK(hipMalloc(&s4, slab * 8)); CK(hipMemset(s3, 0, slab * 8)); CK(hipMemset(s4, 0xff, slab * 8));
    CK(hipMalloc(&dr, row0.size() * 4)); CK(hipMemcpy(dr, row0.data(), row0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dt, tasks.size() * sizeof(K4Task))); CK(hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(K4Task), hipMemcpyHostToDevice));
    const size_t ldsb = 2 * (size_t)KB * ZCp * 8, ldsb4 = K4_BUFS * (size_t)K4_KB * K4_PITCH * 8 + 512;
    K3Args k3{dZ, dr, s3, ZC, ZCp, KB, n_chunks, nb, nblk, parts3};
    K4Args k4{dZ, dr, dt, s4, ZC, ZCp, KB, n_chunks, nb, nblk, parts4, rows};
    CK(hipFuncSetAttribute((const void *)k_schur_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipFuncSetAttribute((const void *)k_schur_syrk4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb4));
    const int g3 = 8 * ((n_chunks + 7) / 8) * parts3, g4 = 8 * ((n_chunks + 7) / 8) * parts4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms3 = 0, ms4 = 0;
    const int reps = 20;
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_schur_syrk, dim3(g3), dim3(64 * K3_WAVES), ldsb, 0, k3);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms3, e0, e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_schur_syrk4, dim3(g4), dim3(64 * K4_WAVES), ldsb4, 0, k4);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms4, e0, e1));
    }
    CK(hipGetLastError());
    const double flop = 2.0 * rows * 256.0 * nblk;
    printf("16x16x4: %.1f us  %.1f TFLOP/s executed    4x4x4: %.1f us  %.1f TFLOP/s executed\n", 1e3 * ms3 / reps, flop / (ms3 / reps) / 1e9,
           1e3 * ms4 / reps, flop / (ms4 / reps) / 1e9);
#ifdef K4_WALL
    { std::vector<long long> w(2 * 4096); CK(hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(cvxk::g_k4wall), w.size() * 8));
      long long t0 = w[0]; for (int b = 0; b < g4; ++b) if (w[2 * b] && w[2 * b] < t0) t0 = w[2 * b];
      for (int part = 0; part < parts4; ++part) {
          double s0 = 0, s1 = 0, e1max = 0, e1min = 1e30; int cnt = 0;
          for (int b = 0; b < g4; ++b) { if (((b >> 3) % parts4) != part || !w[2 * b + 1]) continue;
              s0 += (w[2 * b] - t0) * 0.01; s1 += (w[2 * b + 1] - t0) * 0.01; e1max = fmax(e1max, (w[2 * b + 1] - t0) * 0.01); e1min = fmin(e1min, (w[2 * b + 1] - t0) * 0.01); ++cnt; }
          printf("part %d: %d workgroups, wave 0 starts at %.2f us, ends at %.2f us on average (min %.2f, max %.2f)\n", part, cnt, s0 / cnt, s1 / cnt, e1min, e1max); } }
#endif
    std::vector<double> h3(slab), h4(slab);
    CK(hipMemcpy(h3.data(), s3, slab * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h4.data(), s4, slab * 8, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; size_t bad = 0, firstbad = (size_t)-1;
    for (size_t i = 0; i < slab; ++i) {
        const double d = fabs(h3[i] - h4[i]);
        if (!(d <= 1e-11 * (1.0 + fabs(h3[i])))) { if (!bad) firstbad = i; ++bad; }
        if (d == d) maxd = fmax(maxd, d);
        maxv = fmax(maxv, fabs(h3[i]));
    }
    printf("max |diff| %.3e (max |value| %.3e), mismatches %zu of %zu", maxd, maxv, bad, slab);
    if (bad) {
        const size_t q = (firstbad / 256) % nblk, e = firstbad % 256;
        printf("  first: chunk %zu block %zu entry (%zu,%zu): %.6g vs %.6g", firstbad / 256 / nblk, q, e / 16, e % 16, h3[firstbad], h4[firstbad]);
    }
    printf("\n");
    return bad ? 1 : 0;
}
