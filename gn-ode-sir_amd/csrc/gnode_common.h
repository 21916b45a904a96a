// Shared host-side helpers of libgnode_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "gnode.h"

struct gnode_graph_s {
    int32_t n;
    int64_t nnz;
    int32_t max_degree;
    int32_t* rowptr;  // device [n+1]
    int32_t* col;     // device [nnz]
};

void gnode_set_error(const char* fmt, ...);

#define GN_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            gnode_set_error(__VA_ARGS__);       \
            return GNODE_ERR_ARG;               \
        }                                       \
    } while (0)

#define GN_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            gnode_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GNODE_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

#define GN_LAUNCH_CHECK() GN_HIP(hipGetLastError())

static inline size_t gn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
