// Fork / join onto a library-owned side stream: independent kernels of one C-ABI call run beside each other instead of in
// launch order.  Users: the label-energy pass of the training forward (beside the lifting layer) and the 1x1-conv weight gradient
// of a backward phase (beside the transform and the mode-domain kernel).  BOTH ARE OFF BY DEFAULT: measured on MI355X
// (profiles/r04a_side_stream_ab.txt) a fork / join pair costs ~25-30 us of cross-queue signalling per use -- more than the 16 us
// of label-energy kernels it hides (step 1.146 ms off / 1.162 ms on at batch 256, 0.301 / 0.334 ms at batch 8) -- and the weight
// gradient does run beside the mode-domain kernel but slows it from 26 to 53 us (both want the same wave slots).  The knob
// (side_stream = mask of users) keeps the experiment reproducible.
//
// Semantics: cfd_side_fork(main) returns a stream on which work is ordered after everything enqueued on `main` so far;
// cfd_side_join(main, side) orders everything enqueued on `main` afterwards behind the side stream's work.  Both are plain
// event record / wait pairs, so they are legal inside a stream capture (the side stream joins the capture and becomes a branch
// of the graph).  The side stream is per device and shared by every caller stream (a false dependency between two engines of
// one process at worst).  Disabled -- fork returns `main` itself -- by the side_stream knob (0), while the per-kernel event
// profiler is recording (its per-kernel times must not overlap), and when stream / event creation fails.
#include <mutex>

#include "cfd_common.h"

bool cfd_prof_active();

namespace {
constexpr int kMaxDev = 16, kRing = 64;
struct Side {
    hipStream_t stream = nullptr;
    hipEvent_t ev[kRing] = {};
    int next = 0;
    bool ok = false, tried = false;
};
Side g_side[kMaxDev];
std::mutex g_mu;

hipEvent_t next_event(Side& s) {
    hipEvent_t e = s.ev[s.next];
    s.next = (s.next + 1) % kRing;
    return e;
}

Side* side_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    Side& s = g_side[dev];
    if (!s.tried) {
        s.tried = true;
        bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < kRing; ++i) ok = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) == hipSuccess;
        s.ok = ok;
        if (!ok) (void)hipGetLastError();
    }
    return s.ok ? &s : nullptr;
}
}  // namespace

// `use`: which user asks (bit 1 = label energy beside the lifting layer, bit 2 = 1x1 weight gradient beside the mode-domain
// kernels); the side_stream knob is a mask of the enabled users, default 0 = none (measured: profiles/r04a_side_stream_ab.txt).
hipStream_t cfd_side_fork(hipStream_t main, int use) {
    const int knob = cfd_tune_get(CFD_TUNE_SIDE_STREAM);
    if ((((knob < 0) ? 0 : knob) & use) == 0 || cfd_prof_active()) return main;
    std::lock_guard<std::mutex> lk(g_mu);
    Side* s = side_of_current_device();
    if (!s) return main;
    hipEvent_t e = next_event(*s);
    if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(s->stream, e, 0) != hipSuccess) {
        (void)hipGetLastError();
        return main;
    }
    return s->stream;
}

int cfd_side_join(hipStream_t main, hipStream_t side) {
    if (side == main) return CFD_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    Side* s = side_of_current_device();
    CFD_REQUIRE(s && s->stream == side, CFD_ERR_HIP, "cfd_side_join: not the side stream of the current device");
    hipEvent_t e = next_event(*s);
    const hipError_t a = hipEventRecord(e, side);
    const hipError_t b = a == hipSuccess ? hipStreamWaitEvent(main, e, 0) : a;
    CFD_REQUIRE(b == hipSuccess, CFD_ERR_HIP, "cfd_side_join: %s", hipGetErrorString(b));
    return CFD_OK;
}
