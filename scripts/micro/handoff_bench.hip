// What a persistent reduced solve would pay per tile column on this part, measured (DESIGN.md section 7, item 3 of round
// 2's review): (a) a flag-guarded hand-off of a 16 KB panel between two workgroups inside ONE launch -- the producer's
// sc1 (write-through) stores, its s_waitcnt vmcnt(0), a workgroup barrier, one sc1 flag store; the consumer's sc1 poll of
// the flag, then sc1 loads of the whole panel -- as a ping-pong of R round trips, timed in the kernel with the 100 MHz
// wall clock; (b) a dependent kernel boundary: a chain of launches that each read what the one before wrote, timed with
// events.  The form is the guide's (MI355X_MICROARCH.md, "Valid forms", handoff-flag row).
// Every poll is bounded: a flag that never comes ends the kernel with an error count, it cannot hang.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/handoff_bench.hip -o scripts/micro/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int PANEL = 2048;         // doubles: 16 KB, a 64 x 32 panel of the solve
constexpr long long POLL_LIMIT = 4000000;

__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// pair p = blockIdx.x / 2; role = blockIdx.x & 1 (0 starts).  buf[pair][2][PANEL], flag[pair][2] on lines of their own.
__global__ void __launch_bounds__(256) k_pingpong(double *buf, int *flag, int rounds, int spread, long long *ticks, int *errors,
                                                  double *sink) {
    // `spread`: the two workgroups of a pair sit `spread` blocks apart (1: neighbours, i.e. different XCDs under
    // round-robin placement; 8: the same XCD)
    const int b = blockIdx.x;
    const int pair = (spread == 1) ? b / 2 : (b / (2 * spread)) * spread + b % spread;
    const int role = (spread == 1) ? (b & 1) : ((b / spread) & 1);
    double *mine = buf + ((size_t)pair * 2 + role) * PANEL, *theirs = buf + ((size_t)pair * 2 + (1 - role)) * PANEL;
    int *my_flag = flag + ((size_t)pair * 2 + role) * 32, *their_flag = flag + ((size_t)pair * 2 + (1 - role)) * 32;
    __shared__ int ok;
    double acc = 0.0;
    long long t0 = 0;
    if (threadIdx.x == 0) t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        {
            if (role == (r & 1)) {
                // producer of this half round: write the panel write-through, drain, barrier, raise the flag
                for (int i = threadIdx.x; i < PANEL; i += 256) st_sc1(mine + i, (double)(r * 4096 + i));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) st_flag(my_flag, r);
            } else {
                // consumer: one lane polls, the workgroup follows through a barrier, everybody loads sc1
                if (threadIdx.x == 0) {
                    long long n = 0;
                    while (ld_flag(their_flag) < r && n < POLL_LIMIT) { ++n; __builtin_amdgcn_s_sleep(1); }
                    ok = n < POLL_LIMIT;
                }
                __syncthreads();
                if (!ok) { if (threadIdx.x == 0) atomicAdd(errors, 1); break; }
                for (int i = threadIdx.x; i < PANEL; i += 256) {
                    const double v = ld_sc1(theirs + i);
                    if (v != (double)(r * 4096 + i)) atomicAdd(errors, 1);      // every word checked
                    acc += v;
                }
                __syncthreads();
            }
        }
    }
    if (threadIdx.x == 0) ticks[b] = wall_clock64() - t0;
    if (acc == 12345.678) sink[0] = acc;
}

// (b) a chain of dependent launches of the solve's size class: 90 workgroups read 16 KB each of what the launch before
// wrote and write 16 KB
__global__ void __launch_bounds__(256) k_chain(const double *in, double *out) {
    const size_t base = (size_t)blockIdx.x * PANEL;
    for (int i = threadIdx.x; i < PANEL; i += 256) out[base + i] = in[base + i] + 1.0;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    for (int pairs : {1, 8, 45}) {
        for (int spread : {1, 8}) {
            if (spread == 8 && pairs % 8 != 0) continue;
            double *buf, *sink; int *flag, *err; long long *ticks;
            CK(hipMalloc(&buf, (size_t)pairs * 2 * PANEL * 8)); CK(hipMemset(buf, 0, (size_t)pairs * 2 * PANEL * 8));
            CK(hipMalloc(&flag, (size_t)pairs * 2 * 32 * 4)); CK(hipMemset(flag, 0, (size_t)pairs * 2 * 32 * 4));
            CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4)); CK(hipMalloc(&ticks, pairs * 2 * 8)); CK(hipMalloc(&sink, 8));
            hipLaunchKernelGGL(k_pingpong, dim3(2 * pairs), dim3(256), 0, 0, buf, flag, rounds, spread, ticks, err, sink);
            CK(hipDeviceSynchronize());
            std::vector<long long> t(2 * pairs); int e = 0;
            CK(hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
            long long mx = 0; for (auto v : t) mx = v > mx ? v : mx;
            // `rounds` half round trips = `rounds` hand-offs in sequence
            printf("hand-off, %2d pair(s), partner %d block(s) away: %.2f us per 16 KB flag-guarded hand-off (write-through stores + flag + poll + sc1 loads), %d errors\n",
                   pairs, spread, mx / 100.0 / rounds, e);
            (void)hipFree(buf); (void)hipFree(flag); (void)hipFree(err); (void)hipFree(ticks); (void)hipFree(sink);
        }
    }
    {
        const int blocks = 90, n = 200;
        double *a, *b;
        CK(hipMalloc(&a, (size_t)blocks * PANEL * 8)); CK(hipMalloc(&b, (size_t)blocks * PANEL * 8));
        CK(hipMemset(a, 0, (size_t)blocks * PANEL * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("kernel boundary: %.2f us per dependent launch of 90 workgroups x 16 KB in, 16 KB out (chain of %d)\n", 1e3 * ms / n, n);
        }
    }
    return 0;
}
