#include "Flowgraph.h"

#include <algorithm>
#include <iterator>

#include <chrono>
#include <cstdio>
#include <sstream>

int Node::process()
{
    std::vector<Buffer *> in, out;
    for (auto &e : m_in) in.push_back(e->buffer().get());
    for (auto &e : m_out) out.push_back(e->buffer().get());
    const auto t0 = std::chrono::steady_clock::now();
    const int ret = m_plugin->process(in, out);
    m_time_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();

    // metadata: through the plugin when it handles metadata, otherwise straight across
    meta_vec_t all;
    for (auto &e : m_in) {
        for (auto &m : e->metadata()) all.push_back(m);
        e->metadata().clear();
    }
    if (auto *mm = dynamic_cast<ModMetadata *>(m_plugin.get())) all = mm->process_metadata(all);
    for (auto &e : m_out) e->metadata() = all;
    return ret;
}

Flowgraph::~Flowgraph()
{
    if (m_show_time) std::fputs(processTimeReport().c_str(), stderr);
    // Node -> Edge -> Node is a cycle of shared_ptr (the reference's Node keeps the edges' BUFFERS, src/Flowgraph.h:43-75, and has
    // none): without this the nodes -- and the plugins they hold -- outlived the graph, a pipelined plugin's worker thread was never
    // joined, and the last frames were still being worked on while main() returned and the HIP runtime shut down (round 6: 5 of
    // 500 runs of `host_selftest cfg4` ended in SIGSEGV or an exception from a worker thread).
    for (auto &n : m_nodes) n->dropEdges();
    m_edges.clear();
    m_nodes.clear();
}

std::shared_ptr<Node> Flowgraph::nodeFor(const std::shared_ptr<ModPlugin> &p)
{
    for (auto &n : m_nodes)
        if (n->plugin() == p) return n;
    m_nodes.push_back(std::make_shared<Node>(p));
    return m_nodes.back();
}

void Flowgraph::connect(std::shared_ptr<ModPlugin> input, std::shared_ptr<ModPlugin> output)
{
    auto src = nodeFor(input);
    auto dst = nodeFor(output);
    // Nodes run in list order.  A consumer that was already listed ahead of a producer connected
    // later (cifSig, when tii is wired after it) moves to the end of the list, exactly as the
    // reference does (src/Flowgraph.cpp:299-308: only the output node moves).
    auto is = std::find(m_nodes.begin(), m_nodes.end(), src), id = std::find(m_nodes.begin(), m_nodes.end(), dst);
    if (std::distance(m_nodes.begin(), is) > std::distance(m_nodes.begin(), id))
        m_nodes.splice(m_nodes.end(), m_nodes, id);
    auto e = std::make_shared<Edge>(src, dst);
    src->addOutputEdge(e);
    dst->addInputEdge(e);
    m_edges.push_back(e);
}

bool Flowgraph::run()
{
    for (auto &n : m_nodes)
        if (n->process() == 0) return false;
    return true;
}

std::string Flowgraph::processTimeReport() const
{
    double total = 0;
    for (auto &n : m_nodes) total += n->processTimeUs();
    std::ostringstream o;
    o << "Process time:\n";
    for (auto &n : m_nodes) {
        char line[160];
        std::snprintf(line, sizeof line, "  %30s: %10.0f us (%6.2f %%)\n", n->plugin()->name(),
                      n->processTimeUs(), total > 0 ? 100.0 * n->processTimeUs() / total : 0.0);
        o << line;
    }
    return o.str();
}
