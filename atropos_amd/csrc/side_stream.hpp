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
    // urgent: the stream is created with the device's highest priority -- for launches of FEW, LONG tasks next to a launch
    // of many short ones on the caller's stream (the window launch next to the band launch: its waves then take the
    // slots that come free first instead of queueing behind 16 k band waves)
    bool ready(bool urgent = false) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (stream && dev == device) return true;
        release();
        int least = 0, greatest = 0;
        if (urgent && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
            if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest) != hipSuccess) { stream = nullptr; (void)hipGetLastError(); }
        }
        if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { stream = nullptr; return false; }
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
