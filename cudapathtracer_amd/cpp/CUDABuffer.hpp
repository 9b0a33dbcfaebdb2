// CUDABuffer.hpp -- the reference's device buffer layer (CUDATracer/CUDABuffer.hpp:10-117) re-created over libmipt's
// mipt_buffer_* entry points: a lazily allocated host(pinned)/device mirror with the 4-state head, zero-filled on
// first touch, plus the packed-tuple TypedBuffer view.  Class and method names are the reference's.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/mipt.h"

namespace CUDATracer {

namespace detail {
// One tracer handle per process, created on first use on device $MIPT_DEVICE (default 0): the reference's classes take
// no device/context argument (they use the current CUDA device, CUDABuffer.cpp:226-240).
inline mipt_tracer* DefaultTracer() {
    static mipt_tracer* t = [] {
        // the library writes the structs of the header it was built with: refuse one built against another version
        if (mipt_abi_version() != MIPT_ABI_VERSION)
            throw std::runtime_error("libmipt ABI version " + std::to_string(mipt_abi_version()) + " != header version " + std::to_string(MIPT_ABI_VERSION));
        const char* e = std::getenv("MIPT_DEVICE");
        mipt_tracer* h = nullptr;
        int rc = mipt_create(e ? std::atoi(e) : 0, &h);
        if (rc != MIPT_OK) throw std::runtime_error("mipt_create failed with code " + std::to_string(rc) + " (no MI355X visible?)");
        return h;
    }();
    return t;
}
inline void Check(int rc, const char* what) {
    if (rc != MIPT_OK) throw std::runtime_error(std::string(what) + ": " + mipt_last_error(DefaultTracer()));
}
}  // namespace detail

class CUDABuffer {
public:
    enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };

    CUDABuffer() : CUDABuffer(0) {}
    explicit CUDABuffer(size_t size) { detail::Check(mipt_buffer_create(detail::DefaultTracer(), size, &_b), "mipt_buffer_create"); }
    CUDABuffer(CUDABuffer&& o) noexcept : _b(o._b) { o._b = nullptr; }
    CUDABuffer& operator=(CUDABuffer&& o) noexcept { if (this != &o) { mipt_buffer_destroy(_b); _b = o._b; o._b = nullptr; } return *this; }
    CUDABuffer(const CUDABuffer&) = delete;
    CUDABuffer& operator=(const CUDABuffer&) = delete;
    virtual ~CUDABuffer() { mipt_buffer_destroy(_b); }

    const void* cpu_data() { return need(mipt_buffer_cpu_data(_b)); }
    const void* gpu_data() { return need(mipt_buffer_gpu_data(_b)); }
    void* mutable_cpu_data() { return const_cast<void*>(need(mipt_buffer_mutable_cpu_data(_b))); }
    void* mutable_gpu_data() { return const_cast<void*>(need(mipt_buffer_mutable_gpu_data(_b))); }
    SyncedHead head() const { return (SyncedHead)mipt_buffer_head(_b); }
    size_t size() const { return mipt_buffer_size(_b); }
    // non-owning adoption of caller memory (CUDABuffer.hpp:18,20): at least size() bytes; the head moves to that side
    void set_cpu_data(void* data) { detail::Check(mipt_buffer_set_cpu_data(_b, data), "mipt_buffer_set_cpu_data"); }
    void set_gpu_data(void* data) { detail::Check(mipt_buffer_set_gpu_data(_b, data), "mipt_buffer_set_gpu_data"); }
    // async_gpu_push(const cudaStream_t&) (CUDABuffer.hpp:27): Stream is the caller's hipStream_t (any pointer-sized
    // handle; this header includes no HIP); without an argument the copy goes onto the tracer's own stream
    template <typename Stream> void async_gpu_push(const Stream& stream) {
        static_assert(sizeof(Stream) == sizeof(void*), "async_gpu_push takes a hipStream_t");
        void* s = nullptr; std::memcpy(&s, &stream, sizeof s);
        detail::Check(mipt_buffer_async_gpu_push_on(_b, s), "mipt_buffer_async_gpu_push_on");
    }
    void async_gpu_push() { detail::Check(mipt_buffer_async_gpu_push(_b), "mipt_buffer_async_gpu_push"); }

private:
    static const void* need(const void* p) { if (!p) throw std::runtime_error("CUDABuffer: allocation or copy failed"); return p; }
    mipt_buffer* _b = nullptr;
};

// TypedBuffer<T...>: a CUDABuffer holding the packed tuple (T...), CUDABuffer.hpp:81-117
template <typename Ty, typename... OtherTy>
class TypedBuffer : public CUDABuffer {
    template <unsigned I, typename First, typename... Rest> struct At { using type = typename At<I - 1, Rest...>::type; static constexpr size_t offset = sizeof(First) + At<I - 1, Rest...>::offset; };
    template <typename First, typename... Rest> struct At<0, First, Rest...> { using type = First; static constexpr size_t offset = 0; };
    static constexpr size_t total() { return (sizeof(Ty) + ... + sizeof(OtherTy)); }
public:
    TypedBuffer() : CUDABuffer(total()) {}
    template <unsigned idx> const typename At<idx, Ty, OtherTy...>::type& Get() {
        return *reinterpret_cast<const typename At<idx, Ty, OtherTy...>::type*>((const std::uint8_t*)cpu_data() + At<idx, Ty, OtherTy...>::offset);
    }
    template <unsigned idx> typename At<idx, Ty, OtherTy...>::type& GetMutable() {
        return *reinterpret_cast<typename At<idx, Ty, OtherTy...>::type*>((std::uint8_t*)mutable_cpu_data() + At<idx, Ty, OtherTy...>::offset);
    }
    template <unsigned idx> void Set(const typename At<idx, Ty, OtherTy...>::type& val) { GetMutable<idx>() = val; }
};

}  // namespace CUDATracer
