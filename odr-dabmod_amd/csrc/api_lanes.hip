// api_lanes.hip -- batches in flight inside ONE context (include/dabgpu.h, "batches in flight"): the lanes' streams and their
// ordering against the context's own stream and the caller's, the streaming host path (submit / collect), synchronisation.
#include "dabgpu_ctx.h"

using namespace dabgpu;
using namespace dabgpu_api;

namespace dabgpu_api {
// the stream of lane i (created on first use, lanes 1 .. i in order; lane 0 is the context's stream)
int lane_stream(dabgpu_ctx *c, int i, hipStream_t *out)
{
    if (i == 0) { *out = c->stream; return DABGPU_OK; }
    for (int k = 1; k <= i; ++k) {
        if (c->lane[k].stream) continue;
        std::vector<hipStream_t> others{c->stream};
        for (int j = 1; j < k; ++j) others.push_back(c->lane[j].stream);
        if (c->copy_stream) others.push_back(c->copy_stream);
        HIPCHK(c, create_stream_apart(others.data(), (int)others.size(), &c->lane[k].stream, &c->lane_own_queue[k]));
    }
    *out = c->lane[i].stream;
    return DABGPU_OK;
}

// A NULL-stream call that runs on the context's own stream: behind everything the other lanes have been given.
int own_stream_joins_lanes(dabgpu_ctx *c)
{
    for (int i = 1; i < (int)dabgpu_ctx::kMaxLanes; ++i) {
        if (!c->lane[i].stream || !c->lane_dirty[i]) continue;
        if (!c->lane[i].ev) HIPCHK(c, hipEventCreateWithFlags(&c->lane[i].ev, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->lane[i].ev, c->lane[i].stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->lane[i].ev, 0));
        c->lane_dirty[i] = false;
    }
    ++c->own_epoch;            // (what follows on `stream` is work the lanes have not been ordered behind)
    return DABGPU_OK;
}

// ... and a chain call that goes to lane i: behind such work of the context's own stream
int lane_joins_own_stream(dabgpu_ctx *c, int i)
{
    if (i == 0) return DABGPU_OK;
    c->lane_dirty[i] = true;
    if (c->lane_seen_epoch[i] == c->own_epoch) return DABGPU_OK;
    if (!c->own_ev) HIPCHK(c, hipEventCreateWithFlags(&c->own_ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->own_ev, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->lane[i].stream, c->own_ev, 0));
    c->lane_seen_epoch[i] = c->own_epoch;
    return DABGPU_OK;
}

// every stream of the context idle (before a table that kernels in flight on ANY lane may read is rewritten)
int drain_lanes(dabgpu_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 1; i < (int)dabgpu_ctx::kMaxLanes; ++i)
        if (c->lane[i].stream) HIPCHK(c, hipStreamSynchronize(c->lane[i].stream));
    return DABGPU_OK;
}

// Which lane a call on the context's own stream goes to: the lanes in turn while a launch alone cannot fill the chip many
// times over; lane 0 for everything that carries stream state (Resampler) and for large batches (nothing to gain, and
// the scratch of some chains grows with the batch).
int pick_lane(dabgpu_ctx *c, size_t n_frames, unsigned mask, bool *rotating)
{
    *rotating = false;
    bool resample;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        resample = (mask & DABGPU_STAGE_RESAMPLE) && c->set.rs_in != c->set.rs_out;
    }
    if (c->n_lanes <= 1 || resample || n_frames > (size_t)dabgpu_ctx::kLaneMaxFrames) return 0;
    // Every lane owns a set of per-call scratch buffers that grow to the largest call they have seen and are never trimmed.
    // The one-kernel chains need none; the others (a separate guard / FIRFilter / convert / predistorter kernel: up to two
    // native-rate frames of 8 B per sample per frame) rotate only while that stays within kLaneScratchBytes per lane --
    // the small batches the lanes exist for.
    bool scratch;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        scratch = c->set.overlap > 0 || c->set.out_format != 0 || c->set.cfr_enable || c->set.tii_enable ||
                  (c->set.gain_reference_rounding && c->set.gain_mode == DABGPU_GAIN_VAR && (mask & DABGPU_STAGE_GAIN)) ||
                  (mask & DABGPU_STAGE_POLY) || (int)c->set.taps.size() > tf_max_fused_taps();
    }
    if (scratch && n_frames * 2 * tf_samples(c->g) * sizeof(float2) > (size_t)dabgpu_ctx::kLaneScratchBytes) return 0;
    *rotating = true;
    return (int)(c->lane_seq++ % (unsigned long long)c->n_lanes);
}

int chain_dev(dabgpu_ctx *c, const void *d_in, bool from_bits, size_t n_frames, unsigned mask, void *d_iq, size_t out_cap,
              size_t *out_bytes, void *stream)
{
    int lane = 0;
    hipStream_t s = (hipStream_t)stream;
    c->call_lanes = 1;
    if (!s) {
        bool rotating = false;
        lane = pick_lane(c, n_frames, mask, &rotating);
        int rc = lane_stream(c, lane, &s);
        if (rc) return rc;
        if ((rc = lane_joins_own_stream(c, lane))) return rc;
        if (rotating) c->call_lanes = c->n_lanes;
    }
    c->clip_from_collect = false;
    TraceScope trace(c->trace_enabled ? &c->last_variant : nullptr);
    const int rc = run_chain(c, d_in, from_bits, n_frames, mask, (float2 *)d_iq, out_cap, out_bytes, s, true, lane);
    c->call_lanes = 1;
    return rc;
}

}  // namespace dabgpu_api

extern "C" {
int dabgpu_set_lanes(dabgpu_ctx *c, int lanes)
{
    CTXCHK(c);
    if (lanes < 1 || lanes > (int)dabgpu_ctx::kMaxLanes) return fail(c, DABGPU_E_INVALID, "lanes: 1 ... 4");
    const int rc = dabgpu_synchronize(c);
    if (rc) return rc;
    c->n_lanes = lanes;
    c->lane_seq = 0;
    return DABGPU_OK;
}

int dabgpu_debug_lanes(dabgpu_ctx *c, int *own_queue_mask)
{
    CTXCHK(c);
    int n = 1, mask = 1;
    for (int i = 1; i < (int)dabgpu_ctx::kMaxLanes; ++i)
        if (c->lane[i].stream) { ++n; if (c->lane_own_queue[i]) mask |= 1 << i; }
    if (own_queue_mask) *own_queue_mask = mask;
    return n;
}

int dabgpu_set_handover_frames(dabgpu_ctx *c, int frames)
{
    CTXCHK(c);
    if (frames < 0 || (frames & 1)) return fail(c, DABGPU_E_INVALID, "hand-over piece: an even number of frames, or 0");
    const int rc = dabgpu_synchronize(c);
    if (rc) return rc;
    c->handover_frames = frames;
    return DABGPU_OK;
}

// everything the context queues from now on starts after what `stream` holds now
int dabgpu_wait_for_stream(dabgpu_ctx *c, void *stream)
{
    CTXCHK(c);
    if (!c->lane[0].ev) HIPCHK(c, hipEventCreateWithFlags(&c->lane[0].ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->lane[0].ev, (hipStream_t)stream));
    for (int i = 0; i < (int)dabgpu_ctx::kMaxLanes; ++i) {
        hipStream_t ls = i ? c->lane[i].stream : c->stream;
        if (ls) HIPCHK(c, hipStreamWaitEvent(ls, c->lane[0].ev, 0));
    }
    // (a lane whose stream does not exist yet is created later, by a call the host makes after this one: it cannot start
    // before the host has seen `stream` reach this point only if the caller relies on stream order alone -- so create them)
    for (int i = 1; i < c->n_lanes; ++i)
        if (!c->lane[i].stream) {
            hipStream_t ls;
            const int rc = lane_stream(c, i, &ls);
            if (rc) return rc;
            HIPCHK(c, hipStreamWaitEvent(ls, c->lane[0].ev, 0));
        }
    return DABGPU_OK;
}

// everything queued on `stream` from now on starts after what the context has queued so far (all lanes)
int dabgpu_stream_wait_for(dabgpu_ctx *c, void *stream)
{
    CTXCHK(c);
    for (int i = 0; i < (int)dabgpu_ctx::kMaxLanes; ++i) {
        hipStream_t ls = i ? c->lane[i].stream : c->stream;
        if (!ls) continue;
        // (lane 0's `ev` is dabgpu_wait_for_stream's: the joins use events 1 ... and one more for lane 0)
        hipEvent_t &ev = i ? c->lane[i].ev : c->ho_join;
        if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(ev, ls));
        HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, ev, 0));
    }
    return DABGPU_OK;
}

// ---- asynchronous host path ------------------------------------------------------

int dabgpu_chain_submit(dabgpu_ctx *c, const uint8_t *bits, size_t n_frames, unsigned mask)
{
    CTXCHK(c);
    if (c->slot_count == 2) return fail(c, DABGPU_E_CAPACITY, "two batches are already in flight: collect one first");
    if (n_frames > (size_t)c->max_frames)
        return fail(c, DABGPU_E_CAPACITY, "n_frames exceeds max_frames of the context");
    int rc = apply_settings(c);
    if (rc) return rc;
    unsigned m2 = mask;
    if ((m2 & DABGPU_STAGE_RESAMPLE) && c->cur.rs_in == c->cur.rs_out) m2 &= ~DABGPU_STAGE_RESAMPLE;
    const size_t in_bytes = n_frames * tf_in_bytes(c->g);
    const size_t need = n_frames * out_samples_per_frame(c, m2, c->rs_L, c->rs_M) * bytes_per_sample(c->cur.out_format);
    const int slot_index = (c->slot_head + c->slot_count) & 1;
    dabgpu_ctx::Slot &sl = c->slot[slot_index];
    // the two batches in flight run on two lanes where the chain carries no stream state: their kernels overlap
    const int lane = (c->n_lanes > 1 && !(m2 & DABGPU_STAGE_RESAMPLE)) ? slot_index : 0;
    hipStream_t ls;
    if ((rc = lane_stream(c, lane, &ls))) return rc;
    if ((rc = lane_joins_own_stream(c, lane))) return rc;
    if (!c->copy_stream) {
        // the copy back must overlap with the kernels of BOTH batches in flight: apart from lanes 0 and 1
        hipStream_t l1 = nullptr;
        bool own = false;
        if (c->n_lanes > 1 && (rc = lane_stream(c, 1, &l1))) return rc;
        std::vector<hipStream_t> others{c->stream};
        if (l1) others.push_back(l1);
        HIPCHK(c, create_stream_apart(others.data(), (int)others.size(), &c->copy_stream, &own));
    }
    if (!sl.computed) {
        HIPCHK(c, hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    }
    if (sl.h_in_cap < in_bytes) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        sl.h_in = nullptr;
        sl.h_in_cap = 0;
        HIPCHK(c, hipHostMalloc(&sl.h_in, std::max<size_t>(in_bytes, 16), hipHostMallocDefault));
        sl.h_in_cap = in_bytes;
    }
    const int ho = (int)(c->submit_seq % 3);
    if (c->h_out_cap[ho] < need) {
        if (c->h_out[ho]) (void)hipHostFree(c->h_out[ho]);
        c->h_out[ho] = nullptr;
        c->h_out_cap[ho] = 0;
        HIPCHK(c, hipHostMalloc(&c->h_out[ho], std::max<size_t>(need, 16), hipHostMallocDefault));
        c->h_out_cap[ho] = need;
    }
    HIPCHK(c, sl.d_in.reserve(std::max<size_t>(in_bytes, 16)));
    HIPCHK(c, sl.d_out.reserve(std::max<size_t>(need, 16)));
    std::memcpy(sl.h_in, bits, in_bytes);                       // 28.8 kB per frame
    if (in_bytes) HIPCHK(c, hipMemcpyAsync(sl.d_in.p, sl.h_in, in_bytes, hipMemcpyHostToDevice, ls));
    size_t ob = 0;
    {
        TraceScope trace(c->trace_enabled ? &c->last_variant : nullptr);
        c->call_lanes = (c->n_lanes > 1 && !(m2 & DABGPU_STAGE_RESAMPLE)) ? 2 : 1;     // (the two batches in flight)
        rc = run_chain(c, sl.d_in.p, true, n_frames, mask, (float2 *)sl.d_out.p, need, &ob, ls, true, lane);
        c->call_lanes = 1;
    }
    if (rc) return rc;
    sl.out_format = c->cur.out_format;
    if (sl.out_format) {
        // the clip counter is one per lane and the next submit on it zeroes it: this batch's count goes to the slot now
        if (!sl.h_clip) HIPCHK(c, hipHostMalloc((void **)&sl.h_clip, 16, hipHostMallocDefault));
        LaneScope scratch(c, lane);
        HIPCHK(c, hipMemcpyAsync(sl.h_clip, c->d_clip.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ls));
    }
    HIPCHK(c, hipEventRecord(sl.computed, ls));
    // the copy back runs on its own stream: the next batch's kernels overlap it
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, sl.computed, 0));
    if (need) HIPCHK(c, hipMemcpyAsync(c->h_out[ho], sl.d_out.p, need, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(sl.copied, c->copy_stream));
    sl.out_bytes = need;
    sl.h_out_index = ho;
    sl.stream = ls;
    sl.busy = true;
    ++c->slot_count;
    ++c->submit_seq;
    return DABGPU_OK;
}

int dabgpu_chain_collect(dabgpu_ctx *c, const void **iq, size_t *out_bytes)
{
    CTXCHK(c);
    if (!iq) return fail(c, DABGPU_E_INVALID, "null argument");
    if (c->slot_count == 0) return fail(c, DABGPU_E_INVALID, "no batch in flight");
    dabgpu_ctx::Slot &sl = c->slot[c->slot_head];
    HIPCHK(c, hipEventSynchronize(sl.copied));
    // Nothing ever synchronises the lanes or the copy stream themselves (the events do the ordering), and the HIP runtime
    // retires the commands of a stream -- their signals, kernel-argument blocks, command objects -- only when somebody asks
    // about that stream: after ~800 one-frame batches it stopped for 48 ms inside one call to catch up
    // (profiles/r06_async_series.txt).  Everything this batch queued is complete here; a query is the asking.
    (void)hipStreamQuery(sl.stream);
    (void)hipStreamQuery(c->copy_stream);
    *iq = c->h_out[sl.h_out_index];
    if (out_bytes) *out_bytes = sl.out_bytes;
    c->clip_from_collect = true;
    c->collected_clipped = (sl.out_format && sl.h_clip) ? (size_t)*sl.h_clip : 0;     // (the format of ITS submit)
    sl.busy = false;
    c->slot_head ^= 1;
    --c->slot_count;
    return DABGPU_OK;
}

int dabgpu_synchronize(dabgpu_ctx *c)
{
    CTXCHK(c);
    const int rc = drain_lanes(c);
    if (rc) return rc;
    for (bool &d : c->lane_dirty) d = false;
    return DABGPU_OK;
}

}  // extern "C"
