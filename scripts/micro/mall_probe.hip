// Does a kernel's WRITE of a buffer leave it in the 256 MiB Infinity Cache for the next kernel's read?  A streaming read
// cannot tell (HBM and the cache both feed it at ~6 TB/s); a latency-bound one can: one wave per CU, every lane one
// dependent 16-byte load after the other over its own slice (hit 227 ns, HBM miss ~3x that: MI355X_MICROARCH.md).
// The buffer is made by stores of a given cache policy (plain | nt | sc1 | sc0 sc1), or by a plain read (the control:
// known to leave it resident), behind a 512 MB sweep that empties the cache.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mall_probe.hip -o scripts/micro/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

template <int POLICY>
__global__ void __launch_bounds__(256) k_write(double *p, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const d2v v = {(double)i, 1.0};
        d2v *q = (d2v *)p + i;
        if (POLICY == 0) *q = v;
        else if (POLICY == 1) __builtin_nontemporal_store(v, q);
        else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
    }
}
__global__ void __launch_bounds__(256) k_stream(const double *p, size_t n2, double *sink) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const d2v v = ((const d2v *)p)[i]; s += v.x + v.y; }
    if (s == 1.2345) sink[0] = s;
}
// latency-bound: 256 workgroups of one wave; a lane walks `steps` lines of its slice, each address depending on the
// value before (the data is (double)index, so the dependency is real but the walk is a plain stride)
__global__ void __launch_bounds__(64) k_chase(const double *p, size_t n2, int steps, double *sink) {
    const size_t lanes = (size_t)gridDim.x * 64, me = (size_t)blockIdx.x * 64 + threadIdx.x;
    size_t i = me * 8 % n2;          // 128-byte lines: 8 x 16 B
    double s = 0.0;
    for (int k = 0; k < steps; ++k) {
        const d2v v = ((const d2v *)p)[i];
        s += v.y;
        i = ((size_t)v.x + lanes * 8) % n2;       // v.x == i: next line of this lane, lanes * 128 bytes further on
    }
    if (s == 1.2345) sink[0] = s;
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 161;
    const size_t bytes = mb << 20, n2 = bytes / 16, big = (size_t)512 << 20;
    double *buf, *other, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&other, big)); CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = (int)(n2 / 8 / (256 * 64));          // every line of the buffer once
    auto chase = [&]() { float ms = 0; (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(k_chase, dim3(256), dim3(64), 0, 0, buf, n2, steps, sink);
                         (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); return 1e6 * ms / steps; };
    auto sweep = [&]() { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, other, big / 16, sink); };
    hipLaunchKernelGGL(k_write<0>, dim3(2048), dim3(256), 0, 0, other, big / 16);
    hipLaunchKernelGGL(k_write<0>, dim3(2048), dim3(256), 0, 0, buf, n2);
    for (int rep = 0; rep < 3; ++rep) {
        sweep(); const float cold = chase();
        sweep(); hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, buf, n2, sink); const float rd = chase();
        sweep(); hipLaunchKernelGGL(k_write<0>, dim3(2048), dim3(256), 0, 0, buf, n2); const float w0 = chase();
        sweep(); hipLaunchKernelGGL(k_write<1>, dim3(2048), dim3(256), 0, 0, buf, n2); const float w1 = chase();
        sweep(); hipLaunchKernelGGL(k_write<2>, dim3(2048), dim3(256), 0, 0, buf, n2); const float w2 = chase();
        sweep(); hipLaunchKernelGGL(k_write<3>, dim3(2048), dim3(256), 0, 0, buf, n2); const float w3 = chase();
        CK(hipDeviceSynchronize());
        printf("%zu MB, ns per dependent load: cold %.0f | after a streaming read %.0f | after a write: plain %.0f, nt %.0f, sc1 %.0f, sc0 sc1 %.0f\n",
               mb, cold, rd, w0, w1, w2, w3);
    }
    return 0;
}
