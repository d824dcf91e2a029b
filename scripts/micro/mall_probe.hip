// Does a kernel's plain WRITE of a buffer leave it in the 256 MiB Infinity Cache for the next kernel's read?  Times a
// streaming read of B bytes after (a) a plain-store write of the buffer, (b) a non-temporal-store write, (c) a read of
// it, (d) nothing recent (a 512 MB sweep in between).  Reads: 16 B per lane, every byte summed (no elision).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mall_probe.hip -o scripts/micro/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_write(double *p, size_t n2, int nt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const d2v v = {(double)i, 1.0};
        if (nt) __builtin_nontemporal_store(v, (d2v *)p + i); else ((d2v *)p)[i] = v;
    }
}
__global__ void __launch_bounds__(256) k_read(const double *p, size_t n2, double *sink, int nt) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const d2v v = nt ? __builtin_nontemporal_load((const d2v *)p + i) : ((const d2v *)p)[i];
        s += v.x + v.y;
    }
    if (s == 1.2345) sink[0] = s;
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 161;
    const size_t bytes = mb << 20, n2 = bytes / 16, big = (size_t)512 << 20;
    double *buf, *other, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&other, big)); CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed_read = [&](int nt) { float ms = 0; hipEventRecord(e0, 0); hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, n2, sink, nt);
                                    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; };
    auto sweep = [&]() { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, other, big / 16, sink, 0); };
    hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, other, big / 16, 0);
    for (int rep = 0; rep < 3; ++rep) {
        sweep(); hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, buf, n2, 0); const float a = timed_read(0);
        sweep(); hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, buf, n2, 1); const float b = timed_read(0);
        sweep(); hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, n2, sink, 0); const float c = timed_read(0);
        sweep(); const float d = timed_read(0);
        sweep(); hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, n2, sink, 1); const float e = timed_read(0);
        CK(hipDeviceSynchronize());
        printf("%zu MB read: after plain write %.1f us (%.2f TB/s) | after nt write %.1f us | after a read %.1f us (%.2f TB/s) | cold %.1f us (%.2f TB/s) | after an nt read %.1f us\n",
               mb, 1e3 * a, bytes / (a * 1e9), 1e3 * b, 1e3 * c, bytes / (c * 1e9), 1e3 * d, bytes / (d * 1e9), 1e3 * e);
    }
    return 0;
}
