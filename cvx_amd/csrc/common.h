// Shared host helpers for the HIP translation units of libcvx_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cvx_amd.h"

namespace cvx {

// thread-local last error text (cvx_last_error)
void set_last_error(const std::string &s);

inline int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    set_last_error(buf);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? CVX_ERR_NO_DEVICE : CVX_ERR_HIP;
}

#define CVX_HIP(call)                                                         \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return cvx::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// What a handle that ends leaves for the next one on its device (calib.hip keeps its device block the same way);
// CVX_ARENA_CACHE=0 switches all of it off, cvx_release_cached_memory() hands it back.
bool handle_cache_on();
// The library's own streams, never a caller's: creating a stream costs 0.1-3 ms and its first copies more (a hardware
// queue is set up behind it), which a drop-in create -> solve -> destroy would pay on every call.  One idle stream per
// device is kept; stream_give takes a stream that its owner has synchronised.
hipStream_t stream_take(int dev);
void stream_give(int dev, hipStream_t s);
void stream_cache_release();
// a handle's own stream: the kept one of the device, else a new one; and its end (kept only if it drains cleanly)
inline int own_stream_begin(int dev, hipStream_t *s) {
    *s = stream_take(dev);
    if (!*s) CVX_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    return CVX_OK;
}
inline void own_stream_end(int dev, hipStream_t s) {
    if (!s) return;
    if (hipStreamSynchronize(s) == hipSuccess) { stream_give(dev, s); return; }
    (void)hipGetLastError();
    (void)hipStreamDestroy(s);
}

// blocked Cholesky solve of an SPD system on the host: A (n x n row-major, lower triangle
// read, destroyed), b in/out.  Returns false when A is not numerically positive definite.
bool host_cholesky_solve(int n, double *A, double *b);

// One-sided (Hestenes) Jacobi SVD of an m x k matrix (row-major, k <= 8), handeye_init.cpp: on return the columns of
// W = A V are mutually orthogonal; sv[j] = |W_j| descending, V (k x k, row-major) holds the right singular vectors
// as columns.
void jacobi_svd(std::vector<double> &W, int m, int k, double *V, double *sv);

}  // namespace cvx
