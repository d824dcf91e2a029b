// Microbenchmarks grounding the roofline denominators used by bench.py on the box it runs on:
//   (1) v_mfma_f64_16x16x4_f64 issue rate, (2) v_fma_f64 VALU rate, (3) HBM streaming copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double seed, long long *clk = nullptr) {
    const long long c0 = clock64(), w0 = wall_clock64();
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double *out, int iters, double seed) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N>
__device__ __forceinline__ double row_ror_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the inner loop shape of k_schur_syrk: per block one A, one B, three DPP rotations, four MFMAs
template <int MODE>   // 0: registers only, 1: + DPP rotations, 2: + LDS operand reads
__global__ void __launch_bounds__(768) k_syrk_shape(double *out, int iters, double seed) {
    __shared__ double lds[16 * 336];
    for (int i = threadIdx.x; i < 16 * 336; i += 768) lds[i] = seed + i * 1e-6;
    __syncthreads();
    double acc[10][4];
    for (int i = 0; i < 10; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0;
    const int lane = threadIdx.x & 63, kq = lane >> 4, cq = lane & 15;
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
        const double *p = lds + ((it & 3) * 4 + kq) * 336 + cq;
        double pa[10], pb[10];
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 10; ++i) { pa[i] = p[16 * i]; pb[i] = p[16 * (i + 5)]; }
        }
        if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            double av = a, b0 = b;
            if (MODE >= 2) { av = pa[i]; b0 = pb[i]; }
            acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, MODE ? row_ror_f64<4>(b0) : b0, acc[i][1], 0, 0, 0);
            acc[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, MODE ? row_ror_f64<8>(b0) : b0, acc[i][2], 0, 0, 0);
            acc[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, MODE ? row_ror_f64<12>(b0) : b0, acc[i][3], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 10; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// how the LDS-fed rate depends on the number and width of operand reads per 40 MFMAs
template <int NREAD, bool WIDE>
__global__ void __launch_bounds__(768) k_syrk_reads(double *out, int iters, double seed) {
    __shared__ double lds[16 * 336];
    for (int i = threadIdx.x; i < 16 * 336; i += 768) lds[i] = seed + i * 1e-6;
    __syncthreads();
    double acc[10][4];
    for (int i = 0; i < 10; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0;
    const int lane = threadIdx.x & 63, kq = lane >> 4, cq = lane & 15;
    for (int it = 0; it < iters; ++it) {
        double v[20];
        if (WIDE) {
            const double2 *p = (const double2 *)(lds + ((it & 3) * 4 + kq) * 336) + cq;
#pragma unroll
            for (int i = 0; i < NREAD; ++i) { const double2 t = p[16 * i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
#pragma unroll
            for (int i = 2 * NREAD; i < 20; ++i) v[i] = v[i % (2 * NREAD)];
        } else {
            const double *p = lds + ((it & 3) * 4 + kq) * 336 + cq;
#pragma unroll
            for (int i = 0; i < NREAD; ++i) v[i] = p[16 * i];
#pragma unroll
            for (int i = NREAD; i < 20; ++i) v[i] = v[i % NREAD];
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const double av = v[i], b0 = v[10 + i];
            acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, row_ror_f64<4>(b0), acc[i][1], 0, 0, 0);
            acc[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, row_ror_f64<8>(b0), acc[i][2], 0, 0, 0);
            acc[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, row_ror_f64<12>(b0), acc[i][3], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 10; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA waves and VALU waves side by side: blockIdx parity picks the role
__global__ void __launch_bounds__(256) k_mix(double *out, int iters, double seed) {
    double s = 0;
    if (blockIdx.x & 1) {
        f64x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
        double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x * 1e-6;
        const double m = 1.0000001, c = 1e-9;
        for (int it = 0; it < iters * 16; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, c);
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double seed) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x * 1e-6;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, c);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_copy(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_write(double2 *__restrict__ out, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = double2{v, v};
}

__global__ void __launch_bounds__(256) k_write_nt(double2 *__restrict__ out, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { __builtin_nontemporal_store(v, &out[i].x); __builtin_nontemporal_store(v, &out[i].y); }
}
__global__ void __launch_bounds__(256) k_write_once(double2 *__restrict__ out, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = double2{v, v};
}
typedef double d2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_write_once_nt(d2v *__restrict__ out, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(d2v{v, v}, &out[i]);
}

// the store pattern of k_eval: 38 planes of N doubles, each thread writes 16 B into every plane
__global__ void __launch_bounds__(64) k_planes(double *__restrict__ out, long long N, double v) {
    const long long o = 2 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (o >= N) return;
    double2 *p = (double2 *)(out + o);
    const long long h = N >> 1;
#pragma unroll
    for (int i = 0; i < 38; ++i) p[i * h] = double2{v + i, v - i};
}
// the same bytes as tiles: [tile of 128 observations][38][128]
__global__ void __launch_bounds__(64) k_tiles(double *__restrict__ out, long long N, double v) {
    const long long o = 2 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (o >= N) return;
    double2 *p = (double2 *)(out + (size_t)blockIdx.x * 38 * 128) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 38; ++i) p[i * 64] = double2{v + i, v - i};
}

template <class F>
float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    double *out; CK(hipMalloc(&out, 2048 * 256 * sizeof(double)));
    const int iters = 4000;
    for (int wps = 1; wps <= 4; ++wps) {   // waves per SIMD
        int blocks = 256 * wps;            // 4 waves per block -> one per SIMD per block
        float ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, 5);
        double flops = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4 x4acc  %d wave/SIMD: %.3f ms  %.2f TFLOP/s\n", wps, ms, flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<9>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, 5);
        flops = (double)blocks * 4 * iters * 9 * 2048.0;
        printf("mfma_f64_16x16x4 x9acc  %d wave/SIMD: %.3f ms  %.2f TFLOP/s\n", wps, ms, flops / ms / 1e9);
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL((k_mfma4<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, 5);
        double flops = (double)blocks * 4 * iters * 8 * 512.0;
        printf("mfma_f64_4x4x4_4b x8acc  %d wave/SIMD: %.3f ms  %.2f TFLOP/s\n", wps, ms, flops / ms / 1e9);
    }
    {   // k_schur_syrk's inner loop shape, 12 waves per CU
        const int it2 = 2000;
        double flops = 256.0 * 12 * it2 * 40 * 512.0;
        float ms = time_ms([&] { hipLaunchKernelGGL((k_syrk_shape<0>), dim3(256), dim3(768), 0, 0, out, it2, 1.0); }, 5);
        printf("syrk shape, registers only: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_syrk_shape<1>), dim3(256), dim3(768), 0, 0, out, it2, 1.0); }, 5);
        printf("syrk shape, + DPP rotations: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_syrk_shape<2>), dim3(256), dim3(768), 0, 0, out, it2, 1.0); }, 5);
        printf("syrk shape, + DPP + LDS operands: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_syrk_shape<3>), dim3(256), dim3(768), 0, 0, out, it2, 1.0); }, 5);
        printf("syrk shape, + DPP + LDS operands, all 20 reads then wait: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    {
        const int it2 = 2000;
        double flops = 256.0 * 12 * it2 * 40 * 512.0;
#define RUNR(N, W) { float ms = time_ms([&] { hipLaunchKernelGGL((k_syrk_reads<N, W>), dim3(256), dim3(768), 0, 0, out, it2, 1.0); }, 5); \
        printf("syrk reads: %2d x %s per 40 MFMA: %.3f ms  %.2f TFLOP/s\n", N, W ? "b128" : "b64 ", ms, flops / ms / 1e9); }
        RUNR(20, false) RUNR(12, false) RUNR(7, false) RUNR(4, false) RUNR(2, false) RUNR(10, true) RUNR(6, true) RUNR(4, true)
    }
    {   // 2 MFMA waves + 2 VALU waves per SIMD
        int blocks = 256 * 4;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, 5);
        double fm = (double)(blocks / 2) * 4 * iters * 8 * 2048.0, fv = (double)(blocks / 2) * 256 * iters * 16.0 * 8 * 2.0;
        printf("mixed 2 MFMA + 2 VALU waves/SIMD: %.3f ms  MFMA %.2f + VALU %.2f = %.2f TFLOP/s\n", ms, fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
    }
    {   // in-kernel clock and cycles per MFMA (one wave's view), long enough for DVFS to settle
        long long *clk; CK(hipMalloc(&clk, 16)); long long h[2];
        for (int wps = 1; wps <= 4; wps *= 2) {
            int blocks = 256 * wps;
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_mfma<8>), dim3(blocks), dim3(256), 0, 0, out, iters * 8, 1.0, clk);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            double ghz = (double)h[0] / ((double)h[1] * 10.0);
            printf("mfma x8acc %d wave/SIMD: shader clock %.3f GHz, %.1f cycles per MFMA per SIMD, %.2f TFLOP/s\n", wps, ghz,
                   (double)h[0] / (iters * 8.0 * 8 * wps), 1024.0 * wps * iters * 8.0 * 8 * 2048.0 / ((double)h[1] * 10e-9) / 1e12);
        }
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters * 4, 1.0); }, 5);
        double flops = (double)blocks * 256 * iters * 4 * 8 * 2.0;
        printf("v_fma_f64  %d wave/SIMD: %.3f ms  %.2f TFLOP/s\n", wps, ms, flops / ms / 1e9);
    }
    size_t n = (size_t)1 << 26;   // 64 Mi double2 = 1 GiB
    double2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 0, n * 16));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, n); }, 10);
    { float m2 = time_ms([&] { hipLaunchKernelGGL(k_write_nt, dim3(2048), dim3(256), 0, 0, b, n, 1.0); }, 10);
      printf("write nt (2x8B) 1 GiB: %.3f ms  %.2f TB/s\n", m2, n * 16 / m2 / 1e9);
      m2 = time_ms([&] { hipLaunchKernelGGL(k_write_once, dim3((unsigned)(n / 256)), dim3(256), 0, 0, b, n, 1.0); }, 10);
      printf("write one-per-thread 1 GiB: %.3f ms  %.2f TB/s\n", m2, n * 16 / m2 / 1e9);
      m2 = time_ms([&] { hipLaunchKernelGGL(k_write_once_nt, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (d2v *)b, n, 1.0); }, 10);
      printf("write one-per-thread nt 16B 1 GiB: %.3f ms  %.2f TB/s\n", m2, n * 16 / m2 / 1e9); }
    { const long long N = 1080000; double *ev; CK(hipMalloc(&ev, 38 * N * 8));
      float m3 = time_ms([&] { hipLaunchKernelGGL(k_planes, dim3((unsigned)((N + 127) / 128)), dim3(64), 0, 0, ev, N, 1.0); }, 20);
      printf("38 planes x 1.08M (k_eval store pattern): %.3f ms  %.2f TB/s\n", m3, 38.0 * N * 8 / m3 / 1e9);
      m3 = time_ms([&] { hipLaunchKernelGGL(k_tiles, dim3((unsigned)((N + 127) / 128)), dim3(64), 0, 0, ev, N, 1.0); }, 20);
      printf("38 x 128 tiles x 1.08M: %.3f ms  %.2f TB/s\n", m3, 38.0 * N * 8 / m3 / 1e9); }
    printf("copy 1 GiB -> 1 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, b, n, 1.0); }, 10);
    printf("write 1 GiB: %.3f ms  %.2f TB/s\n", ms, 1.0 * n * 16 / ms / 1e9);
    return 0;
}
