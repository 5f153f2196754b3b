// Drop-in for HIRT_Multichannel_Convolution/MemorySwap.h:19-290: a block of memory {pointer, logical size, how to free it} guarded by a
// thread_lock, handed out through a move-only `Ptr` that holds the lock for as long as it lives.  The same public members:
//
//   MemorySwap(size) / MemorySwap(alloc, free, size), movable, not copyable
//   clear()                       free now
//   access()                      blocking: the Ptr holds the lock
//   attempt()                     non-blocking: an EMPTY Ptr (get() == nullptr, getSize() == 0) when another thread holds the lock —
//                                 what an audio thread uses (MemorySwap.h:182-185; MonoConvolve.cpp:181-183)
//   swap(ptr, size)               take over memory the caller owns (never freed by this object)
//   grow(size) / equal(size)      reallocate when size > / != the current size; overloads with custom allocate / free functions
//   Ptr: clear(), swap(), grow(), equal() (the same operations on a lock already held), get(), getSize()
//
// `size` is a LOGICAL size chosen by the caller — elements for the built-in allocator, anything for custom ones (the convolver keeps IR
// lengths there).  An implementation of its own (one Block record replaced as a unit; 64-byte aligned default allocations so the memory
// suits any vector unit).  Host memory only; the MI355X engine keeps its device-side buffers differently, this
// header exists because Convolver.h:4 of the reference makes the type visible to every caller.
#pragma once

#include "ThreadLocks.hpp"

#include <cstdint>
#include <cstdlib>
#include <functional>
#include <utility>

template <class T>
class MemorySwap
{
public:

    typedef std::function<T *(uintptr_t size)> AllocFunc;
    typedef std::function<void (T *)> FreeFunc;

    class Ptr
    {
        friend MemorySwap;

    public:

        Ptr(Ptr&& p) : mOwner(p.mOwner), mPtr(p.mPtr), mSize(p.mSize) { p.forget(); }
        ~Ptr() { clear(); }

        Ptr(const Ptr&) = delete;
        Ptr& operator=(const Ptr&) = delete;

        // gives the lock back; the Ptr is empty afterwards
        void clear()
        {
            if (mOwner) mOwner->mLock.release();
            forget();
        }

        void swap(T *ptr, uintptr_t size)
        {
            if (!mOwner) return;
            mOwner->replace(ptr, size, nullptr);
            refresh();
        }

        void grow(uintptr_t size) { grow(&MemorySwap::allocate, &MemorySwap::deallocate, size); }
        void equal(uintptr_t size) { equal(&MemorySwap::allocate, &MemorySwap::deallocate, size); }

        void grow(AllocFunc allocFunction, FreeFunc freeFunction, uintptr_t size)
        {
            if (mOwner && size > mOwner->mBlock.size) reallocate(allocFunction, freeFunction, size);
        }

        void equal(AllocFunc allocFunction, FreeFunc freeFunction, uintptr_t size)
        {
            if (mOwner && size != mOwner->mBlock.size) reallocate(allocFunction, freeFunction, size);
        }

        T *get() { return mPtr; }
        uintptr_t getSize() { return mSize; }

    private:

        Ptr() : mOwner(nullptr), mPtr(nullptr), mSize(0) {}
        explicit Ptr(MemorySwap *owner) : mOwner(owner), mPtr(nullptr), mSize(0) { refresh(); }

        void forget()
        {
            mOwner = nullptr;
            mPtr = nullptr;
            mSize = 0;
        }
        void refresh()
        {
            mPtr = mOwner ? mOwner->mBlock.ptr : nullptr;
            mSize = mOwner ? mOwner->mBlock.size : 0;
        }
        void reallocate(AllocFunc allocFunction, FreeFunc freeFunction, uintptr_t size)
        {
            mOwner->replace(allocFunction(size), size, freeFunction);
            refresh();
        }

        MemorySwap *mOwner;
        T *mPtr;
        uintptr_t mSize;
    };

    MemorySwap(uintptr_t size)
    {
        if (size) replace(allocate(size), size, &MemorySwap::deallocate);
    }

    MemorySwap(AllocFunc allocFunction, FreeFunc freeFunction, uintptr_t size)
    {
        if (size) replace(allocFunction(size), size, freeFunction);
    }

    ~MemorySwap() { clear(); }

    MemorySwap(const MemorySwap&) = delete;
    MemorySwap& operator=(const MemorySwap&) = delete;

    MemorySwap(MemorySwap&& obj) { *this = std::move(obj); }

    MemorySwap& operator=(MemorySwap&& obj)
    {
        if (this != &obj)
        {
            clear();
            obj.mLock.acquire();
            mLock.acquire();
            mBlock = obj.mBlock;
            obj.mBlock = Block();
            mLock.release();
            obj.mLock.release();
        }
        return *this;
    }

    void clear() { swap(nullptr, 0); }

    Ptr access()
    {
        mLock.acquire();
        return Ptr(this);
    }

    Ptr attempt() { return mLock.attempt() ? Ptr(this) : Ptr(); }

    Ptr swap(T *ptr, uintptr_t size)
    {
        mLock.acquire();
        replace(ptr, size, nullptr);
        return Ptr(this);
    }

    Ptr grow(uintptr_t size) { return grow(&MemorySwap::allocate, &MemorySwap::deallocate, size); }
    Ptr equal(uintptr_t size) { return equal(&MemorySwap::allocate, &MemorySwap::deallocate, size); }

    Ptr grow(AllocFunc allocFunction, FreeFunc freeFunction, uintptr_t size)
    {
        Ptr p = access();
        p.grow(allocFunction, freeFunction, size);
        return p;
    }

    Ptr equal(AllocFunc allocFunction, FreeFunc freeFunction, uintptr_t size)
    {
        Ptr p = access();
        p.equal(allocFunction, freeFunction, size);
        return p;
    }

private:

    struct Block
    {
        T *ptr = nullptr;
        uintptr_t size = 0;
        FreeFunc release;           // empty: the memory is the caller's (swap)
    };

    // (lock held, or the object under construction / destruction)
    void replace(T *ptr, uintptr_t size, FreeFunc freeFunction)
    {
        if (mBlock.release) mBlock.release(mBlock.ptr);
        mBlock.ptr = ptr;
        mBlock.size = ptr ? size : 0;
        mBlock.release = ptr ? freeFunction : FreeFunc();
    }

    static T *allocate(uintptr_t size)
    {
        void *p = nullptr;
        const size_t bytes = size ? (size_t) size * sizeof(T) : 64;
        return posix_memalign(&p, 64, bytes) == 0 ? static_cast<T *>(p) : nullptr;
    }

    static void deallocate(T *ptr) { std::free(ptr); }

    thread_lock mLock;
    Block mBlock;
};
