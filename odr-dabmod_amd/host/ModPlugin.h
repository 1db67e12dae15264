// ModPlugin.h -- the operator interface of ODR-DabMod's flowgraph (reference
// src/ModPlugin.h:43-144), restated for the MI355X drop-in stages.
//
//   ModPlugin::process(vector<Buffer*> in, vector<Buffer*> out) -> int
//     0      : "no output this round", the flowgraph stops walking its nodes
//     != 0   : continue (the value itself is never interpreted)
//   fatal errors are std::runtime_error, thrown from process().
//
// ModInput / ModCodec / ModMux / ModOutput fix the arity and assert it;
// PipelinedModCodec runs internal_process() on a private thread one call
// behind the caller: call i hands frame i to the worker and returns frame i-1,
// call 0 returns 0 (that transmission frame is dropped, as in the reference,
// src/ModPlugin.cpp:90-115), and metadata is delayed by the same one call.
#pragma once

#include "Buffer.h"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

// Per-frame metadata carried beside the samples (reference src/TimestampDecoder.h).
struct frame_timestamp {
    int32_t fct = -1;
    uint32_t timestamp_sec = 0;
    uint32_t timestamp_pps = 0;
    bool timestamp_valid = false;
    bool offset_changed = false;
};

struct flowgraph_metadata {
    frame_timestamp ts;
};
using meta_vec_t = std::vector<flowgraph_metadata>;

class ModMetadata {
public:
    virtual ~ModMetadata() = default;
    virtual meta_vec_t process_metadata(const meta_vec_t &metadataIn) = 0;
};

class ModPlugin {
public:
    virtual int process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut) = 0;
    virtual const char *name() = 0;
    virtual ~ModPlugin() = default;
};

class ModInput : public ModPlugin {
public:
    int process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut) override;
    virtual int process(Buffer *dataOut) = 0;
};

class ModCodec : public ModPlugin {
public:
    int process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut) override;
    virtual int process(Buffer *const dataIn, Buffer *dataOut) = 0;
};

class ModMux : public ModPlugin {
public:
    int process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut) override;
    virtual int process(std::vector<Buffer *> dataIn, Buffer *dataOut) = 0;
};

class ModOutput : public ModPlugin {
public:
    int process(std::vector<Buffer *> dataIn, std::vector<Buffer *> dataOut) override;
    virtual int process(Buffer *dataIn) = 0;
};

class PipelinedModCodec : public ModCodec, public ModMetadata {
public:
    int process(Buffer *const dataIn, Buffer *dataOut) final;
    const char *name() override = 0;
    meta_vec_t process_metadata(const meta_vec_t &metadataIn) final;

protected:
    // the subclass calls these at the end of its constructor / start of its destructor
    void start_pipeline_thread();
    void stop_pipeline_thread();
    virtual int internal_process(Buffer *const dataIn, Buffer *dataOut) = 0;

private:
    struct Mailbox {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Buffer> q;
        void put(Buffer &&b);
        Buffer take();
    };
    void worker();

    bool m_ready_to_output_data = false;
    Mailbox m_to_worker, m_from_worker;
    std::deque<meta_vec_t> m_metadata_fifo;
    std::atomic<bool> m_running{false};
    std::thread m_thread;
};
