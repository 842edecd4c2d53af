// Internal helpers shared by the translation units of libf5hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdio>

#include "../../include/f5_hip.h"

int f5_fail(int code, const char* fmt, ...);  // records the thread-local message returned by f5_last_error()

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return f5_fail(F5_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)                   \
    do {                            \
        int _r = (expr);            \
        if (_r != F5_OK) return _r; \
    } while (0)
#define KCHK() HIPCHK(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// Bump allocator over one hipMalloc'd block; with base == nullptr it only measures (dry run).
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    ~Arena() {
        if (base) (void)hipFree(base);
    }
    void reset() { off = 0; }
    template <typename U> U* take(size_t n) {
        off = align_up(off, 256);
        U* p = reinterpret_cast<U*>(base + off);
        off += n * sizeof(U);
        return p;
    }
};

// scratch device buffer freed at scope exit (kernel-level test entry points only)
template <typename U> struct Scratch {
    U* p = nullptr;
    ~Scratch() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n * sizeof(U), 16)); }
};
