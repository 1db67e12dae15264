#include "ModPlugin.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace {
void arity(bool ok, const char *what, ModPlugin *p)
{
    if (!ok) throw std::runtime_error(std::string("Assertion failure: ") + what + " for " + p->name());
}
}  // namespace

int ModInput::process(std::vector<Buffer *> in, std::vector<Buffer *> out)
{
    arity(in.empty(), "dataIn.empty()", this);
    arity(out.size() == 1, "dataOut.size() == 1", this);
    return process(out[0]);
}

int ModCodec::process(std::vector<Buffer *> in, std::vector<Buffer *> out)
{
    arity(in.size() == 1, "dataIn.size() == 1", this);
    arity(out.size() == 1, "dataOut.size() == 1", this);
    return process(in[0], out[0]);
}

int ModMux::process(std::vector<Buffer *> in, std::vector<Buffer *> out)
{
    arity(!in.empty(), "not dataIn.empty()", this);
    arity(out.size() == 1, "dataOut.size() == 1", this);
    return process(in, out[0]);
}

int ModOutput::process(std::vector<Buffer *> in, std::vector<Buffer *> out)
{
    arity(in.size() == 1, "dataIn.size() == 1", this);
    arity(out.empty(), "dataOut.empty()", this);
    return process(in[0]);
}

void PipelinedModCodec::Mailbox::put(Buffer &&b)
{
    {
        std::lock_guard<std::mutex> lk(mu);
        q.push_back(std::move(b));
    }
    cv.notify_one();
}

Buffer PipelinedModCodec::Mailbox::take()
{
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !q.empty(); });
    Buffer b = std::move(q.front());
    q.pop_front();
    return b;
}

void PipelinedModCodec::start_pipeline_thread()
{
    m_running = true;
    m_thread = std::thread(&PipelinedModCodec::worker, this);
}

void PipelinedModCodec::stop_pipeline_thread()
{
    m_to_worker.put(Buffer());  // an empty buffer is the stop token
    if (m_thread.joinable()) m_thread.join();
}

int PipelinedModCodec::process(Buffer *const dataIn, Buffer *dataOut)
{
    if (!m_running) return 0;
    // the worker takes ownership of the input allocation (the producer's edge buffer is
    // left empty and re-grows on its next setLength)
    Buffer stolen;
    stolen.swap(*dataIn);
    m_to_worker.put(std::move(stolen));
    if (m_ready_to_output_data) {
        Buffer done = m_from_worker.take();
        done.swap(*dataOut);
    } else {
        // first call: nothing to hand out yet; dataIn is empty by now, so this sizes the
        // output to zero and the caller sees "no output this round"
        dataOut->setLength(dataIn->getLength());
        if (dataOut->getLength()) std::memset(dataOut->getData(), 0, dataOut->getLength());
        m_ready_to_output_data = true;
    }
    return static_cast<int>(dataOut->getLength());
}

meta_vec_t PipelinedModCodec::process_metadata(const meta_vec_t &metadataIn)
{
    m_metadata_fifo.push_back(metadataIn);
    if (m_metadata_fifo.size() < 2) return {};
    meta_vec_t r = std::move(m_metadata_fifo.front());
    m_metadata_fifo.pop_front();
    return r;
}

void PipelinedModCodec::worker()
{
    while (m_running) {
        Buffer in = m_to_worker.take();
        if (in.getLength() == 0) break;
        Buffer out;
        out.setLength(in.getLength());
        if (internal_process(&in, &out) == 0) m_running = false;
        m_from_worker.put(std::move(out));
    }
    m_running = false;
}
