// Partial-order alignment consensus of a few candidate strings (pseudo-seed of a low-quality region).  (The pairwise
// alignment of every candidate onto that seed runs on the device: np2_ond_dev.h.)  Host code: small, branchy, sequential graph
// work on strings of a few hundred bases (the reference runs them on the CPU as well and they are not on the
// per-column hot path).
//
// Restated from the behaviour of
//   poa_to_consensus   source/lib/dag.c:658-694  (graph :24-70, NW to graph :261-300, toposort :469-508,
//                                                 heaviest path with -0.5 * indegree :555-595)
// including their integer widths (16-bit node ids and score back-pointers, 8-bit degrees).
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "np2_lq.h"

namespace np2 {
namespace {

constexpr int SEQ_MAX_COUNT = 50;
constexpr long SCORE_MATCH = 1, SCORE_MISMATCH = -2, SCORE_GAP = -2;
inline long match_score(char a, char b) { return a == b ? SCORE_MATCH : SCORE_MISMATCH; }

struct PNode {
    uint8_t base = 0;
    uint8_t indegree = 0, outdegree = 0;
    uint32_t inedge[SEQ_MAX_COUNT];
    uint32_t outedge[SEQ_MAX_COUNT];
    std::vector<uint16_t> alignedto;
    int32_t best_pnode = -1;
    double best_score = 0;
};
struct PEdge {
    uint16_t innode_index = 0, outnode_index = 0;
    uint8_t lable[SEQ_MAX_COUNT];
    PEdge() { memset(lable, 0, sizeof(lable)); }
};
struct PScore {
    uint16_t x = 0, y = 0;
    long s = 0;
};
struct MatchRoute { int32_t x, y; };

struct Graph {
    std::vector<PNode> nodes;
    std::vector<PEdge> edges;
    std::vector<uint16_t> sorted_nodes;
    int32_t sorted_nodes_index = 0;
    uint16_t node_count = 0;
    uint32_t edge_count = 0;
    int32_t insert_node(char base) {
        if (nodes.size() <= node_count) nodes.resize(nodes.size() + 256);
        nodes[node_count].base = (uint8_t)base;
        ++node_count;
        if (nodes.size() <= node_count) nodes.resize(nodes.size() + 256);
        return node_count - 1;
    }
    uint32_t insert_edge(uint16_t in, uint16_t out, uint8_t lable) {
        if (edges.size() <= edge_count) edges.resize(edges.size() + 512);
        edges[edge_count].innode_index = in;
        edges[edge_count].outnode_index = out;
        edges[edge_count].lable[lable] = 1;
        ++edge_count;
        if (edges.size() <= edge_count) edges.resize(edges.size() + 512);
        return edge_count - 1;
    }
    // the graph object is kept per host thread and reused: only what the previous consensus touched is cleaned
    void reset(size_t want_nodes, size_t want_edges) {
        const size_t un = std::min(nodes.size(), (size_t)node_count + 2), ue = std::min(edges.size(), (size_t)edge_count + 2);
        for (size_t i = 0; i < un; ++i) {
            PNode& n = nodes[i];
            n.base = 0; n.indegree = 0; n.outdegree = 0; n.alignedto.clear(); n.best_pnode = -1; n.best_score = 0;
        }
        for (size_t i = 0; i < ue; ++i) { edges[i].innode_index = 0; edges[i].outnode_index = 0; memset(edges[i].lable, 0, sizeof(edges[i].lable)); }
        node_count = 0;
        edge_count = 0;
        sorted_nodes_index = 0;
        if (nodes.size() < want_nodes) nodes.resize(want_nodes);
        if (edges.size() < want_edges) edges.resize(want_edges);
        if (sorted_nodes.size() < nodes.size()) sorted_nodes.resize(nodes.size());
    }
    void link(int32_t head, int32_t node, uint32_t e) {
        nodes[(size_t)head].outedge[nodes[(size_t)head].outdegree++] = e;
        nodes[(size_t)node].inedge[nodes[(size_t)node].indegree++] = e;
    }
};

void insert_unmatched_nodes(size_t seq_index, const char* seq, size_t seq_len, Graph* g, int32_t* firstnode, int32_t* headnode) {
    for (size_t i = 0; i < seq_len; ++i) {
        const uint16_t node_index = (uint16_t)g->insert_node(seq[i]);
        if (*firstnode == -1) {
            *firstnode = node_index;
        } else {
            const uint32_t e = g->insert_edge((uint16_t)*headnode, node_index, (uint8_t)seq_index);
            g->link(*headnode, node_index, e);
        }
        *headnode = node_index;
    }
}

uint16_t predecessors(const Graph& g, uint16_t i) {
    uint16_t c = g.nodes[i].indegree;
    for (size_t j = 0; j < g.nodes[i].alignedto.size() && c == 0; ++j) c += g.nodes[g.nodes[i].alignedto[j]].indegree;
    return c;
}

void topo_visit(int32_t found, uint16_t pnid_count, const std::vector<uint16_t>& pn_to_nodes, const std::vector<int32_t>& node_to_pn,
                std::vector<int8_t>& completed, Graph* g) {
    static thread_local std::vector<int8_t> started;
    static thread_local std::vector<uint16_t> stack;
    started.assign(pnid_count, -1);
    stack.clear();
    stack.push_back((uint16_t)found);
    while (!stack.empty()) {
        const uint16_t pnid = stack.back();
        stack.pop_back();
        if (completed[pnid] == 1) continue;
        const PNode& head = g->nodes[pn_to_nodes[pnid]];
        if (started[pnid] != -1) {
            completed[pnid] = 1;
            g->sorted_nodes[(size_t)g->sorted_nodes_index--] = pn_to_nodes[pnid];
            for (size_t j = 0; j < head.alignedto.size(); ++j) g->sorted_nodes[(size_t)g->sorted_nodes_index--] = head.alignedto[j];
            started[pnid] = -1;
            continue;
        }
        started[pnid] = 1;
        stack.push_back(pnid);
        for (uint16_t k = 0; k < head.outdegree; ++k) stack.push_back((uint16_t)node_to_pn[g->edges[head.outedge[k]].outnode_index]);
        for (size_t j = 0; j < head.alignedto.size(); ++j) {
            const PNode& n = g->nodes[head.alignedto[j]];
            for (uint16_t k = 0; k < n.outdegree; ++k) stack.push_back((uint16_t)node_to_pn[g->edges[n.outedge[k]].outnode_index]);
        }
    }
}

void toposort(Graph* g) {
    static thread_local std::vector<int32_t> node_to_pn;
    static thread_local std::vector<uint16_t> pn_to_nodes;
    node_to_pn.assign(g->node_count, -1);
    pn_to_nodes.assign(g->node_count, 0);
    uint16_t cur_pnid = 0;
    for (uint16_t i = 0; i < g->node_count; ++i) {
        if (node_to_pn[i] == -1) {
            pn_to_nodes[cur_pnid] = i;
            node_to_pn[i] = cur_pnid;
            for (size_t j = 0; j < g->nodes[i].alignedto.size(); ++j) node_to_pn[g->nodes[i].alignedto[j]] = cur_pnid;
            ++cur_pnid;
        }
    }
    static thread_local std::vector<int8_t> completed;
    completed.assign(cur_pnid, -1);
    g->sorted_nodes_index = (int32_t)g->node_count - 1;
    while (g->sorted_nodes_index >= 0) {
        int32_t found = -1;
        for (uint16_t i = 0; i < cur_pnid; ++i)
            if (completed[i] == -1 && predecessors(*g, pn_to_nodes[i]) == 0) { found = i; break; }
        assert(found != -1);
        topo_visit(found, cur_pnid, pn_to_nodes, node_to_pn, completed, g);
    }
}

void align_seq_to_graph(uint16_t x, uint16_t y, size_t seq_index, const char* seq, Graph* g) {
    // ---- score table ((x + 1) x (y + 1)), row 0 = before any node (score_init, dag.c:88-138)
    const size_t W = (size_t)y + 1;
    static thread_local std::vector<PScore> tab;
    tab.assign(((size_t)x + 1) * W, PScore());
    auto S = [&](size_t i, size_t j) -> PScore& { return tab[i * W + j]; };
    for (size_t i = 0; i < W; ++i) S(0, i).s = (long)i * SCORE_GAP;
    static thread_local std::vector<uint16_t> sorted_nodes_index;
    sorted_nodes_index.assign(g->node_count, 0);
    for (uint16_t i = 0; i < g->node_count; ++i) {
        const uint16_t node_index = g->sorted_nodes[i];
        sorted_nodes_index[node_index] = i;
        long bs;
        const PNode& nd = g->nodes[node_index];
        if (nd.indegree == 0) bs = 0;
        else {
            bs = S((size_t)sorted_nodes_index[g->edges[nd.inedge[0]].innode_index] + 1, 0).s;
            for (uint16_t k = 1; k < nd.indegree; ++k) {
                const long s_ = S((size_t)sorted_nodes_index[g->edges[nd.inedge[k]].innode_index] + 1, 0).s;
                if (s_ > bs) bs = s_;
            }
        }
        S((size_t)i + 1, 0).s = bs + SCORE_GAP;
    }
    // ---- fill (align_seq_to_graph_updatescore, dag.c:261-300)
    for (g->sorted_nodes_index = 0; g->sorted_nodes_index < g->node_count; ++g->sorted_nodes_index) {
        const uint16_t node_index = g->sorted_nodes[(size_t)g->sorted_nodes_index];
        const PNode& nd = g->nodes[node_index];
        const uint16_t i = sorted_nodes_index[node_index];
        for (uint16_t j = 0; j < y; ++j) {
            long bests = S((size_t)i + 1, j).s + SCORE_GAP;
            uint16_t bestx = (uint16_t)(i + 1), besty = j;
            for (uint16_t k = 0; k < nd.indegree; ++k) {
                const int32_t pi = sorted_nodes_index[g->edges[nd.inedge[k]].innode_index];
                const long b1 = S((size_t)pi + 1, (size_t)j + 1).s + SCORE_GAP;
                const long b2 = S((size_t)pi + 1, j).s + match_score(seq[j], (char)nd.base);
                if (b1 > bests && b1 >= b2) { bests = b1; bestx = (uint16_t)(pi + 1); besty = (uint16_t)(j + 1); }
                else if (b2 > bests && b2 >= b1) { bests = b2; bestx = (uint16_t)(pi + 1); besty = j; }
            }
            if (nd.indegree == 0) {
                const long b1 = S(0, (size_t)j + 1).s + SCORE_GAP;
                const long b2 = S(0, j).s + match_score(seq[j], (char)nd.base);
                if (b1 > bests && b1 >= b2) { bests = b1; bestx = 0; besty = (uint16_t)(j + 1); }
                else if (b2 > bests && b2 >= b1) { bests = b2; bestx = 0; besty = j; }
            }
            PScore& c = S((size_t)i + 1, (size_t)j + 1);
            c.s = bests; c.x = bestx; c.y = besty;
        }
    }
    // ---- best end (dag.c:302-314)
    uint16_t bestx = 0;
    {
        long bests = 0;
        uint16_t seen = 0;
        for (uint16_t i = 0; i < g->node_count; ++i) {
            if (g->nodes[g->sorted_nodes[i]].outdegree == 0) {
                const long b = S((size_t)i + 1, y).s;
                if (seen == 0 || b > bests) { bestx = (uint16_t)(i + 1); bests = b; seen = 1; }
            }
        }
    }
    uint16_t besty = y;
    // ---- match route (dag.c:327-343)
    static thread_local std::vector<MatchRoute> route;
    route.assign((size_t)x + y + 1, MatchRoute{-1, -1});
    int64_t starty = -1, endy = -1;
    uint32_t mroute_count = 0;
    while (bestx != 0 || besty != 0) {
        const uint16_t nextx = S(bestx, besty).x, nexty = S(bestx, besty).y;
        if (nextx != bestx) route[mroute_count].x = g->sorted_nodes[(size_t)bestx - 1];
        if (nexty != besty) {
            route[mroute_count].y = (int32_t)(starty = besty - 1);
            if (endy == -1) endy = route[mroute_count].y;
        }
        bestx = nextx;
        besty = nexty;
        ++mroute_count;
    }
    for (uint32_t l = 0, r = mroute_count ? mroute_count - 1 : 0; l < r; ++l, --r) { MatchRoute t = route[l]; route[l] = route[r]; route[r] = t; }
    // ---- thread the sequence into the graph (align_seq_to_graph_updategraphy, dag.c:345-405)
    int32_t firstnode = -1, headnode = -1, tailnode = -1, node_index = -1;
    int updated_node = 1, updated_headnode = 1;
    if (starty > 0) insert_unmatched_nodes(seq_index, seq, (size_t)starty, g, &firstnode, &headnode);
    if (endy < (int64_t)y - 1) insert_unmatched_nodes(seq_index, seq + endy + 1, (size_t)((int64_t)y - endy), g, &tailnode, &node_index);   // (length as in the reference: includes the terminator)
    for (uint32_t i = 0; i < mroute_count; ++i) {
        if (route[i].y == -1) continue;
        updated_node = 0;
        const char base = seq[route[i].y];
        if (route[i].x == -1) updated_node = node_index = g->insert_node(base);
        else if ((char)g->nodes[(size_t)route[i].x].base == base) node_index = route[i].x;
        else {
            int32_t foundnode = -1;
            const PNode& mx = g->nodes[(size_t)route[i].x];
            for (size_t j = 0; j < mx.alignedto.size(); ++j)
                if ((char)g->nodes[mx.alignedto[j]].base == base) node_index = foundnode = mx.alignedto[j];
            if (foundnode == -1) {
                updated_node = node_index = g->insert_node(base);
                {   // insert_node_alignedto(g, node_index, route[i].x)
                    PNode& nn = g->nodes[(size_t)node_index];
                    const PNode& mm = g->nodes[(size_t)route[i].x];
                    nn.alignedto.push_back((uint16_t)route[i].x);
                    for (size_t j = 0; j < mm.alignedto.size(); ++j) nn.alignedto.push_back(mm.alignedto[j]);
                }
                const std::vector<uint16_t> al = g->nodes[(size_t)node_index].alignedto;
                for (size_t j = 0; j < al.size(); ++j) g->nodes[al[j]].alignedto.push_back((uint16_t)node_index);
            }
        }
        if (headnode != -1) {
            if (updated_node || updated_headnode) {
                const uint32_t e = g->insert_edge((uint16_t)headnode, (uint16_t)node_index, (uint8_t)seq_index);
                g->link(headnode, node_index, e);
            } else {
                int not_existed = 1;
                PNode& hn = g->nodes[(size_t)headnode];
                for (uint16_t q = 0; q < hn.outdegree; ++q)
                    if (g->edges[hn.outedge[q]].outnode_index == (uint16_t)node_index) { g->edges[hn.outedge[q]].lable[seq_index] = 1; not_existed = 0; }
                if (not_existed) {
                    const uint32_t e = g->insert_edge((uint16_t)headnode, (uint16_t)node_index, (uint8_t)seq_index);
                    g->link(headnode, node_index, e);
                }
            }
        }
        headnode = node_index;
        updated_headnode = updated_node;
        if (firstnode == -1) firstnode = headnode;
    }
    if (tailnode != -1) {
        const uint32_t e = g->insert_edge((uint16_t)headnode, (uint16_t)tailnode, (uint8_t)seq_index);
        g->link(headnode, tailnode, e);
    }
    if (g->sorted_nodes.size() < g->nodes.size()) g->sorted_nodes.resize(g->nodes.size());
    toposort(g);
}

}  // namespace

std::string poa_consensus(const std::vector<std::string>& seqs) {
    static thread_local Graph tl_graph;
    Graph& g = tl_graph;
    size_t total = 0;
    for (const std::string& s : seqs) total += s.size() + 2;
    g.reset(total + 16, 2 * total + 32);   // (the reference starts at 10 000 nodes and grows; capacity is not observable)
    assert((int)seqs.size() <= SEQ_MAX_COUNT);
    for (size_t si = 0; si < seqs.size(); ++si) {
        const std::string& s = seqs[si];
        if (si == 0) {
            int32_t firstnode = -1, headnode = -1;
            insert_unmatched_nodes(si, s.c_str(), s.size(), &g, &firstnode, &headnode);
            if (g.sorted_nodes.size() < g.nodes.size()) g.sorted_nodes.resize(g.nodes.size());
            for (uint16_t x = 0; x < g.node_count; ++x) g.sorted_nodes[x] = x;
        } else {
            align_seq_to_graph(g.node_count, (uint16_t)s.size(), si, s.c_str(), &g);
        }
    }
    // heaviest path (get_consensus_from_graph, dag.c:555-595)
    const int seq_count = (int)seqs.size();
    int32_t global_best_node = -1;
    double best_score = -1, global_best_score = -1;
    for (uint16_t ni = 0; ni < g.node_count; ++ni) {
        const uint16_t nodeid = g.sorted_nodes[ni];
        PNode& nd = g.nodes[nodeid];
        int32_t best_pnode = -1;
        if (nd.indegree) {
            for (uint16_t i = 0; i < nd.indegree; ++i) {
                const PEdge& e = g.edges[nd.inedge[i]];
                int cnt = 0;
                for (int q = 0; q < seq_count; ++q) cnt += e.lable[q];
                const double score = g.nodes[e.innode_index].best_score + cnt - 0.5 * nd.indegree;
                if (score > best_score || best_pnode == -1) { best_score = score; best_pnode = e.innode_index; }
            }
        } else {
            best_score = 0;
            best_pnode = -1;
        }
        nd.best_score = best_score;
        nd.best_pnode = best_pnode;
        if (best_score > global_best_score) { global_best_score = best_score; global_best_node = nodeid; }
    }
    std::string out;
    while (global_best_node != -1) {
        out.push_back((char)g.nodes[(size_t)global_best_node].base);
        global_best_node = g.nodes[(size_t)global_best_node].best_pnode;
    }
    for (size_t l = 0, r = out.size() ? out.size() - 1 : 0; l < r; ++l, --r) { char t = out[l]; out[l] = out[r]; out[r] = t; }
    // the reference returns a C string: an embedded terminator (a tail node built from the NUL of a candidate) ends it
    const size_t z = out.find('\0');
    if (z != std::string::npos) out.resize(z);
    return out;
}

}  // namespace np2
