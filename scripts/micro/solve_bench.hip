// Stand-alone timing + correctness harness for cvx_amd/csrc/solve_kernels.h (development aid).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cvx_amd/csrc scripts/micro/solve_bench.hip -o scripts/micro/solve_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "solve_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 388;
    const int C = 8, P = 108, ncf = 8, ZC = 336, camcol0 = 324, ycol = 333;
    const int nT = (n + 1 + 31) / 32, npad = nT * 32;
    std::vector<double> S((size_t)n * n), E((size_t)n * n + 3 * n + 4, 0.0), G((size_t)n * (n + 8));
    srand(1);
    for (auto &v : G) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = 0; k < n + 8; ++k) s += G[(size_t)i * (n + 8) + k] * G[(size_t)j * (n + 8) + k];
            S[(size_t)i * n + j] = S[(size_t)j * n + i] = s;
        }
    for (size_t i = 0; i < (size_t)n * n; ++i) E[i] = S[i];
    for (int i = 0; i < n; ++i) { E[(size_t)n * n + i] = rand() / (double)RAND_MAX - 0.5; E[(size_t)n * n + n + i] = 0.1; E[(size_t)n * n + 2 * n + i] = S[(size_t)i * n + i]; }
    double *dE, *ds2, *dA, *dL, *dDs, *dintr, *dpts, *dic, *dpc, *demb, *ddelta, *dR; int *dfail;
    std::vector<double> ones(n, 1.0);
    CK(hipMalloc(&dE, E.size() * 8)); CK(hipMemcpy(dE, E.data(), E.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&ds2, n * 8)); CK(hipMemcpy(ds2, ones.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dA, 2 * (size_t)npad * npad * 8)); CK(hipMemset(dA, 0xff, 2 * (size_t)npad * npad * 8)); dL = nullptr;
    CK(hipMalloc(&dDs, n * 8)); CK(hipMalloc(&dintr, 9 * C * 8)); CK(hipMalloc(&dpts, 3 * P * 8)); CK(hipMalloc(&dic, 9 * C * 8)); CK(hipMalloc(&dpc, 3 * P * 8));
    CK(hipMemset(dintr, 0, 9 * C * 8)); CK(hipMemset(dpts, 0, 3 * P * 8));
    CK(hipMalloc(&demb, (size_t)C * ZC * 8)); CK(hipMalloc(&ddelta, n * 8)); CK(hipMalloc(&dR, 16 * 8)); CK(hipMalloc(&dfail, 4));
    hipEvent_t ev[4]; for (int i = 0; i < 4; ++i) CK(hipEventCreate(&ev[i]));
    cvxs::DampArgs da{dE, nullptr, ds2, n, npad, 1e-4, 0.0, 1e-6, 1e32, dA, dDs};
    cvxs::FinishArgs f{};
    f.n = n; f.E = dE; f.E_scal = (long long)n * n + 3 * n; f.nranks = 1; f.Ds = dDs; f.fail = dfail;
    f.C = C; f.P = P; f.ncf = ncf; f.fix_k3 = 1; f.cam_free = 1; f.pts_free = 1; f.ZC = ZC; f.camcol0 = camcol0; f.ycol = ycol;
    f.intr = dintr; f.points = dpts; f.intr_cand = dic; f.points_cand = dpc; f.ds_emb = demb; f.delta_s = ddelta; f.R = dR;
    if (n != C * ncf + 3 * P) { f.C = 0; f.P = 0; f.pts_free = 0; f.cam_free = 0; f.ncf = 0; }
    const size_t lds = 8 * cvxs::finish_lds_doubles(n);
    float t_d = 0, t_c = 0, t_f = 0;
    const int reps = 20;
    for (int rep = 0; rep < reps + 2; ++rep) {
        CK(hipMemset(dfail, 0, 4));
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(cvxs::k_damp_prepare, dim3((npad * npad + 255) / 256), dim3(256), 0, 0, da);
        CK(hipEventRecord(ev[1]));
        for (int k = 0; k < nT; ++k)
            hipLaunchKernelGGL(cvxs::k_chol_step, dim3(cvxs::chol_step_blocks(nT, k)), dim3(256), 0, 0, dA, npad, n, k, nT, dfail, ddelta);
        CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL(cvxs::k_step_finish, dim3(1), dim3(cvxs::FIN_THREADS), lds, 0, f);
        CK(hipEventRecord(ev[3]));
        CK(hipDeviceSynchronize());
        float a, b, c;
        CK(hipEventElapsedTime(&a, ev[0], ev[1])); CK(hipEventElapsedTime(&b, ev[1], ev[2])); CK(hipEventElapsedTime(&c, ev[2], ev[3]));
        if (rep >= 2) { t_d += a; t_c += b; t_f += c; }
    }
    printf("n=%d nT=%d  damp %.1f us  chol %.1f us (%.1f/step)  finish %.1f us\n", n, nT, 1e3 * t_d / reps, 1e3 * t_c / reps, 1e3 * t_c / reps / nT, 1e3 * t_f / reps);
    {
        // the same chain (damp, nT tile columns, finish) as ONE linear hipGraph, captured from a stream: what the graph
        // saves is the host's launches, which the LM loop hides behind the device's work; the gaps between dependent kernels
        // are the device's own and stay
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(cvxs::k_damp_prepare, dim3((npad * npad + 255) / 256), dim3(256), 0, st, da);
        for (int k = 0; k < nT; ++k)
            hipLaunchKernelGGL(cvxs::k_chol_step, dim3(cvxs::chol_step_blocks(nT, k)), dim3(256), 0, st, dA, npad, n, k, nT, dfail, ddelta);
        hipLaunchKernelGGL(cvxs::k_step_finish, dim3(1), dim3(cvxs::FIN_THREADS), lds, st, f);
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float tg = 0, te = 0;
        for (int rep = 0; rep < reps + 2; ++rep) {
            CK(hipEventRecord(ev[0], st));
            CK(hipGraphLaunch(exec, st));
            CK(hipEventRecord(ev[1], st));
            CK(hipStreamSynchronize(st));
            float a; CK(hipEventElapsedTime(&a, ev[0], ev[1]));
            if (rep >= 2) tg += a;
            CK(hipEventRecord(ev[0], st));
            hipLaunchKernelGGL(cvxs::k_damp_prepare, dim3((npad * npad + 255) / 256), dim3(256), 0, st, da);
            for (int k = 0; k < nT; ++k)
                hipLaunchKernelGGL(cvxs::k_chol_step, dim3(cvxs::chol_step_blocks(nT, k)), dim3(256), 0, st, dA, npad, n, k, nT, dfail, ddelta);
            hipLaunchKernelGGL(cvxs::k_step_finish, dim3(1), dim3(cvxs::FIN_THREADS), lds, st, f);
            CK(hipEventRecord(ev[1], st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&a, ev[0], ev[1]));
            if (rep >= 2) te += a;
        }
        printf("damp + %d tile columns + finish: as a hipGraph %.1f us, as %d launches on a stream %.1f us\n", nT, 1e3 * tg / reps, nT + 2, 1e3 * te / reps);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
    }
#ifdef CVX_STAMPS
    { long long cs[64]; CK(hipMemcpyFromSymbol(cs, HIP_SYMBOL(cvxs::g_cstamps), sizeof cs));
      printf("chol step 3: load %lld  factor %lld  tail %lld ns\n", (cs[1]-cs[0])*10, (cs[2]-cs[1])*10, (cs[3]-cs[2])*10); }
#endif
    // check: (S + D) x = -gred
    std::vector<double> x(n), Ds(n); int fail;
    CK(hipMemcpy(x.data(), ddelta, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Ds.data(), dDs, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, dfail, 4, hipMemcpyDeviceToHost));
    double rmax = 0, bmax = 0;
    for (int i = 0; i < n; ++i) {
        double s = Ds[i] * x[i];
        for (int j = 0; j < n; ++j) s += S[(size_t)i * n + j] * x[j];
        rmax = fmax(rmax, fabs(s + E[(size_t)n * n + i])); bmax = fmax(bmax, fabs(E[(size_t)n * n + i]));
    }
    printf("fail=%d  residual max %.3e (rhs max %.3e)\n", fail, rmax, bmax);
    return (fail || !(rmax < 1e-9 * bmax * n)) ? 1 : 0;
}
