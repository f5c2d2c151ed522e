// Launch-bound inner loops as hipGraphs.  A refinement call enqueues ~170 kernels, an encoder ~50; at 256 x 256 the host
// cannot enqueue them as fast as the GPU runs them.  The part of such a sequence that touches nothing but the caller's
// workspace (fixed shapes, fixed addresses) is captured once per (shape, workspace, mode) key and replayed with one
// hipGraphLaunch:
//   first call with a key   -> plain launches (also sets the kernels' attributes: not allowed while capturing);
//   second call             -> the same launches under stream capture -> graph -> instantiate -> launch;
//   later calls             -> hipGraphLaunch.
// Same kernels, same arguments, same order, same stream dependencies (a forked side stream joins the capture through its
// events): same bits.  A key that fails to capture falls back to plain launches for good.  Per handle, no global state;
// the profiler's event brackets and a debug trace bypass it.
#pragma once
#include "common.h"
#include <cstring>
#include <vector>

namespace mftx {

struct GraphKey {
    uintptr_t v[16];
    bool operator==(const GraphKey &o) const { return memcmp(v, o.v, sizeof v) == 0; }
};

// The legacy (null) stream cannot be captured -- and it is PyTorch's default stream.  A handle that wants graphs runs a
// call that arrives on it on a private non-blocking stream instead, ordered behind the null stream's earlier work and in
// front of its later work by two events: enter() before the first launch, leave() after the last.
class StreamProxy {
    hipStream_t own_ = nullptr;
    hipEvent_t in_ = nullptr, out_ = nullptr;
public:
    ~StreamProxy() {
        if (in_) (void)hipEventDestroy(in_);
        if (out_) (void)hipEventDestroy(out_);
        if (own_) (void)hipStreamDestroy(own_);
    }
    // -> the stream to launch on (s itself unless s is the null stream); nullptr on failure (then use s)
    hipStream_t enter(hipStream_t s) {
        if (s != nullptr) return s;
        if (!own_) {
            if (hipStreamCreateWithFlags(&own_, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&in_, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&out_, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        if (hipEventRecord(in_, nullptr) != hipSuccess || hipStreamWaitEvent(own_, in_, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return own_;
    }
    void leave(hipStream_t s) {
        if (s != nullptr || !own_) return;
        if (hipEventRecord(out_, own_) == hipSuccess) (void)hipStreamWaitEvent(nullptr, out_, 0);
    }
};

class GraphCache {
    struct Entry { GraphKey k; hipGraph_t g; hipGraphExec_t x; unsigned long long stamp; hipStream_t last; };
    // an executable graph is destroyed only after the stream it was last launched on has drained: a replay may still be in flight
    static void destroy(Entry &e) {
        if (e.last) (void)hipStreamSynchronize(e.last);
        (void)hipGraphExecDestroy(e.x); (void)hipGraphDestroy(e.g);
    }
    std::vector<Entry> entries_;
    std::vector<GraphKey> seen_, refused_;
    unsigned long long clock_ = 0;
    static constexpr size_t MAX_ENTRIES = 24;

    static bool has(const std::vector<GraphKey> &v, const GraphKey &k) {
        for (const auto &e : v) if (e == k) return true;
        return false;
    }
public:
    unsigned long long replays = 0, captures = 0;
    StreamProxy proxy;

    ~GraphCache() { clear(); }
    void clear() {
        for (auto &e : entries_) destroy(e);
        entries_.clear(); seen_.clear(); refused_.clear();
    }

    // enqueue(): int(), issues the launches on stream s (and on streams forked from it by events)
    template <class F>
    int run(const GraphKey &k, hipStream_t s, F &&enqueue) {
        if (s == nullptr) return enqueue();                         // the legacy stream cannot be captured
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return enqueue();   // (inside somebody else's capture)
        for (auto &e : entries_)
            if (e.k == k) {
                e.stamp = ++clock_;
                e.last = s;
                ++replays;
                const hipError_t err = hipGraphLaunch(e.x, s);
                return err == hipSuccess ? 0 : fail((int)err, "hipGraphLaunch: %s", hipGetErrorString(err));
            }
        if (has(refused_, k)) return enqueue();
        if (!has(seen_, k)) {
            if (seen_.size() >= 256) seen_.clear();
            seen_.push_back(k);
            return enqueue();
        }
        hipError_t err = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        if (err != hipSuccess) { (void)hipGetLastError(); refused_.push_back(k); return enqueue(); }
        const int rc = enqueue();
        hipGraph_t g = nullptr;
        err = hipStreamEndCapture(s, &g);
        if (rc != 0 || err != hipSuccess || g == nullptr) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            refused_.push_back(k);
            if (rc != 0) return rc;                                 // an argument error: nothing ran, report it
            return enqueue();                                       // capture refused: run it plainly
        }
        hipGraphExec_t x = nullptr;
        err = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
        if (err != hipSuccess) {
            (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            refused_.push_back(k);
            return enqueue();
        }
        if (entries_.size() >= MAX_ENTRIES) {                       // drop the least recently used
            size_t old = 0;
            for (size_t i = 1; i < entries_.size(); ++i) if (entries_[i].stamp < entries_[old].stamp) old = i;
            destroy(entries_[old]);
            entries_.erase(entries_.begin() + (long)old);
        }
        entries_.push_back(Entry{k, g, x, ++clock_, s});
        ++captures;
        err = hipGraphLaunch(x, s);
        return err == hipSuccess ? 0 : fail((int)err, "hipGraphLaunch: %s", hipGetErrorString(err));
    }
};

}  // namespace mftx
