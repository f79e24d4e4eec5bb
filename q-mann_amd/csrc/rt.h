// rt.h -- tiny runtime layer: error convention, allocation, launch helpers.
// Error convention of the boundary (lib/layer_cuda.h:13-22 in the reference):
// message on stderr, then exit(code).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define QM_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "[*E] HIP : %s : %s (%s:%d)\n", __func__,                 \
                    hipGetErrorString(e_), __FILE__, __LINE__);                       \
            exit((int)e_ ? (int)e_ : 1);                                              \
        }                                                                             \
    } while (0)

#define QM_LAUNCH_CHECK() QM_HIP(hipPeekAtLastError())

static inline void qm_fail(const char *fn, const char *msg)
{
    fprintf(stderr, "[*E] qmann : %s : %s\n", fn, msg);
    exit(1);
}

template <typename T>
static inline void qm_alloc(T **p, size_t n)
{
    // zero-sized requests still hand back a valid pointer, like cudaMalloc(0) would not:
    // the reference never frees a null, so keep every slot non-null
    QM_HIP(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
}

static inline unsigned qm_cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }
