// Flowgraph.h -- sequential DAG executor with the interface of the reference's
// Flowgraph (src/Flowgraph.h:102-104, src/Flowgraph.cpp:208-339): plugins are
// connected output-to-input by edges that own one Buffer each; run() walks the
// nodes in insertion order on the calling thread, moves metadata along, and
// stops at the first node that returns 0.
#pragma once

#include "ModPlugin.h"

#include <list>
#include <memory>
#include <string>
#include <vector>

class Edge;

class Node {
public:
    explicit Node(std::shared_ptr<ModPlugin> plugin) : m_plugin(std::move(plugin)) {}
    std::shared_ptr<ModPlugin> plugin() { return m_plugin; }
    void addInputEdge(std::shared_ptr<Edge> e) { m_in.push_back(std::move(e)); }
    void addOutputEdge(std::shared_ptr<Edge> e) { m_out.push_back(std::move(e)); }
    int process();
    double processTimeUs() const { return m_time_us; }
    // (a node's edges point back at it: the graph lets go of them when it is destroyed, Flowgraph::~Flowgraph)
    void dropEdges() { m_in.clear(); m_out.clear(); }

private:
    std::shared_ptr<ModPlugin> m_plugin;
    std::vector<std::shared_ptr<Edge>> m_in, m_out;
    double m_time_us = 0;
};

class Edge {
public:
    Edge(std::shared_ptr<Node> src, std::shared_ptr<Node> dst)
        : m_src(std::move(src)), m_dst(std::move(dst)), m_buffer(std::make_shared<Buffer>()) {}
    std::shared_ptr<Buffer> buffer() { return m_buffer; }
    meta_vec_t &metadata() { return m_meta; }

private:
    std::shared_ptr<Node> m_src, m_dst;
    std::shared_ptr<Buffer> m_buffer;
    meta_vec_t m_meta;
};

class Flowgraph {
public:
    explicit Flowgraph(bool showProcessTime = false) : m_show_time(showProcessTime) {}
    ~Flowgraph();
    void connect(std::shared_ptr<ModPlugin> input, std::shared_ptr<ModPlugin> output);
    bool run();
    std::string processTimeReport() const;

private:
    std::shared_ptr<Node> nodeFor(const std::shared_ptr<ModPlugin> &p);
    std::list<std::shared_ptr<Node>> m_nodes;
    std::vector<std::shared_ptr<Edge>> m_edges;
    bool m_show_time;
};
