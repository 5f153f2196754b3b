// Drop-in for HIRT_Multichannel_Convolution/ConvolveSIMD.h:3-107: the fixed four-wide float vector type the reference's convolution
// classes are written over (`FloatVector`, with `SIMDVector<T, U, n>` underneath) and the aligned allocation macros.  Hosts that used
// them for their own buffers keep compiling after the header swap.
//
// An implementation of its own over the compiler's portable vector extension (one source for x86-64 and AArch64: GCC / Clang lower it to
// SSE / NEON), not over intrinsics headers.  The MI355X engine does not use it: its arithmetic is in the HIP kernels (csrc/hcv_mac*.hip).
#pragma once

#include <cstdlib>
#include <cstring>

template <class T, class U, int vec_size>
struct SIMDVector
{
    static constexpr int size = vec_size;
    typedef T scalar_type;

    SIMDVector() {}
    SIMDVector(U a) : mVal(a) {}

    U mVal;
};

namespace hisstools_amd_detail
{
    typedef float float4_native __attribute__((vector_size(16)));
}

struct FloatVector : public SIMDVector<float, hisstools_amd_detail::float4_native, 4>
{
    typedef hisstools_amd_detail::float4_native native;

    FloatVector() {}
    FloatVector(native a) : SIMDVector(a) {}
    FloatVector(float a) : SIMDVector(native { a, a, a, a }) {}

    friend FloatVector operator + (const FloatVector& a, const FloatVector& b) { return FloatVector(a.mVal + b.mVal); }
    friend FloatVector operator - (const FloatVector& a, const FloatVector& b) { return FloatVector(a.mVal - b.mVal); }
    friend FloatVector operator * (const FloatVector& a, const FloatVector& b) { return FloatVector(a.mVal * b.mVal); }

    FloatVector operator += (const FloatVector& a)
    {
        mVal += a.mVal;
        return *this;
    }

    static FloatVector unaligned_load(const float *ptr)
    {
        native v;
        std::memcpy(&v, ptr, sizeof v);
        return FloatVector(v);
    }

    void unaligned_store(float *ptr) { std::memcpy(ptr, &mVal, sizeof mVal); }

    float sum() { return mVal[0] + mVal[1] + mVal[2] + mVal[3]; }
};

// 16-byte aligned blocks, as the reference's (the size is rounded up to the alignment, which aligned_alloc requires)
#define ALIGNED_MALLOC(x) aligned_alloc(16, (((size_t) (x)) + 15) & ~(size_t) 15)
#define ALIGNED_FREE free
