// stream_probe.hip -- streams that really overlap: HIP streams are multiplexed onto a few hardware queues, and the
// library's lanes (api_lanes.hip: batches in flight inside one context) need a queue each.
//
// The HIP runtime keeps GPU_MAX_HW_QUEUES hardware queues (four by default); a new stream joins the one that holds the
// fewest streams at that moment, and streams on one queue run IN ORDER.  So whether three streams created in a row end
// up on three queues depends on every stream the process has created and destroyed before (measured: 16-frame calls on
// three lanes take 8.9 us each when they do and 13.5 us when two of them share a queue; profiles/r05_lane_queues.txt).
// Nothing in the API tells which queue a stream is on, but it shows: a 200 us spin on one stream delays a no-op on the
// other exactly when they share one.  A new stream is therefore PROBED against the streams it has to overlap with, and
// candidates that share a queue with one of them are set aside (kept alive until the choice is made, so that the next
// candidate is placed elsewhere) and destroyed afterwards.  ~0.3 ms per probe, a handful of probes per stream, once.

#include "dabgpu_internal.h"

#include <vector>

namespace dabgpu {
namespace {

__global__ void probe_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();                 // (constant-rate counter, 100 MHz)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void probe_nop() {}

}  // namespace

// do kernels on a and b overlap?  (false also on any error: the caller then simply keeps what it has)
bool streams_overlap(hipStream_t a, hipStream_t b)
{
    hipEvent_t ea = nullptr, eb = nullptr;
    bool overlap = false;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess &&
        hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess) {
        hipLaunchKernelGGL(probe_spin, dim3(1), dim3(64), 0, a, 20000ull);      // 200 us
        (void)hipEventRecord(ea, a);
        hipLaunchKernelGGL(probe_nop, dim3(1), dim3(64), 0, b);
        (void)hipEventRecord(eb, b);
        if (hipEventSynchronize(eb) == hipSuccess) overlap = hipEventQuery(ea) == hipErrorNotReady;
        (void)hipEventSynchronize(ea);
        (void)hipGetLastError();
    }
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    return overlap;
}

// a new (non-blocking) stream that overlaps with every one of others[0 .. n); *own: the probe found one -- otherwise
// (fewer hardware queues than streams to keep apart) the last candidate is returned as it is
hipError_t create_stream_apart(const hipStream_t *others, int n, hipStream_t *out, bool *own)
{
    std::vector<hipStream_t> aside;
    hipStream_t pick = nullptr;
    *own = false;
    for (int tries = 0; tries < 12 && !pick; ++tries) {
        hipStream_t cand = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (hipStream_t st : aside) (void)hipStreamDestroy(st);
            return e;
        }
        bool ok = true;
        for (int j = 0; j < n && ok; ++j) ok = streams_overlap(others[j], cand);
        if (ok) { pick = cand; *own = true; } else aside.push_back(cand);
    }
    if (!pick) {
        pick = aside.back();
        aside.pop_back();
    }
    for (hipStream_t st : aside) (void)hipStreamDestroy(st);
    *out = pick;
    return hipSuccess;
}

}  // namespace dabgpu
