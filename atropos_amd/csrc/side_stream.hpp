// side_stream.hpp -- a per-thread side stream for launches that may overlap with the caller's stream.
#ifndef ATR_SIDE_STREAM_HPP
#define ATR_SIDE_STREAM_HPP
#include <hip/hip_runtime.h>

namespace atr {

// A non-blocking side stream with its fork / join events, created on first use (per host thread, on
// the device that is current then) and kept for the life of the thread (re-created when the thread
// moves to another device).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
    bool ready() {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (stream && dev == device) return true;
        release();
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { stream = nullptr; return false; }
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) { release(); return false; }
        device = dev;
        return true;
    }
    void release() {
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; fork = join = nullptr; device = -1;
    }
    // no destructor: at thread / process exit the HIP runtime may already be gone; the handles die with it
};

}  // namespace atr
#endif
