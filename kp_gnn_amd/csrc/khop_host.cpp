// Host-side exact K-hop pre-transform + batch builder + synthetic molecule generator.
// C ABI: include/kpgnn_host.h.  Bit-identical to the reference's data_utils.py:20-241 (see header).
//
// Per source node i (independent -> parallel): walk-count rows W_1..W_K by sparse frontier
// expansion (W_{k+1}[i,:] = W_k[i,:] A, diagonal kept for the recurrence and dropped on output, as
// adj_K_order does), the spd "first reached" mask, the K-hop out-edges of i in ascending target
// order (== networkx DiGraph.edges order of the reference), and per hop the peripheral-subgraph
// statistics of i.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "kpgnn_host.h"

namespace {

thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr int64_t kCountCap = (int64_t)1 << 40;   // saturating walk counts (exactness only needed below the clamp)
constexpr int64_t kInt32Limit = (int64_t)1 << 31;  // beyond this the reference's .int() wraps

struct Graph {  // one input graph, CSR over sources with merged duplicates
    int64_t n = 0;
    std::vector<int64_t> ptr;   // [n+1]
    std::vector<int32_t> nbr;   // target
    std::vector<int64_t> mult;  // number of parallel edges (COO duplicates sum, to_scipy_sparse_matrix)
    std::vector<int64_t> type;  // summed edge attr (dense edge_attr_adj entry)
};

struct GraphResult {
    int64_t n = 0, E = 0;
    std::vector<int32_t> src, dst;        // [E] local ids, row-major (src, dst) order
    std::vector<int32_t> attr;            // [E*K]
    std::vector<int32_t> pea;             // [n*K*T*2]
    std::vector<int32_t> pca;             // [n*K*(H+1)]
    int status = KPGNN_HOST_OK;
};

void build_graph(int64_t n, int64_t E, const int64_t* src, const int64_t* dst, const int64_t* attr, Graph* g) {
    g->n = n;
    std::vector<int64_t> order(E);
    for (int64_t e = 0; e < E; ++e) order[e] = e;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        if (src[a] != src[b]) return src[a] < src[b];
        if (dst[a] != dst[b]) return dst[a] < dst[b];
        return a < b;
    });
    g->ptr.assign(n + 1, 0);
    g->nbr.clear(); g->mult.clear(); g->type.clear();
    int64_t prev_s = -1, prev_d = -1;
    for (int64_t q = 0; q < E; ++q) {
        const int64_t e = order[q];
        const int64_t t = attr ? attr[e] : 2;
        if (src[e] == prev_s && dst[e] == prev_d) {
            g->mult.back() += 1;
            g->type.back() += t;
        } else {
            g->nbr.push_back((int32_t)dst[e]);
            g->mult.push_back(1);
            g->type.push_back(t);
            g->ptr[src[e] + 1] += 1;
            prev_s = src[e]; prev_d = dst[e];
        }
    }
    for (int64_t i = 0; i < n; ++i) g->ptr[i + 1] += g->ptr[i];
}

// Scratch reused across the source nodes handled by one thread.
struct Scratch {
    std::vector<int64_t> w_cur, w_next;      // dense walk-count rows [n]
    std::vector<int32_t> nz_cur, nz_next;    // their nonzero positions
    std::vector<uint8_t> reached;            // spd: reached at an earlier hop
    std::vector<int32_t> hop_attr;           // [n*K] codes of row i per hop (0 = inactive)
    std::vector<uint8_t> any;                // [n] target is a K-hop neighbour
    std::vector<std::vector<int32_t>> sets;  // [K] peripheral node sets of row i
    // peripheral-subgraph scratch
    std::vector<int32_t> local_id;           // [n] position in S or -1
    std::vector<int32_t> dist;               // [|S|]
    std::vector<int32_t> queue;
    std::vector<int64_t> type_count;
    void init(int64_t n, int K) {
        w_cur.assign(n, 0); w_next.assign(n, 0);
        nz_cur.clear(); nz_next.clear();
        reached.assign(n, 0);
        hop_attr.assign((size_t)n * K, 0);
        any.assign(n, 0);
        sets.assign(K, {});
        local_id.assign(n, -1);
    }
};

inline int64_t sat_add(int64_t a, int64_t b) { const int64_t s = a + b; return s > kCountCap ? kCountCap : s; }
inline int64_t sat_mul(int64_t a, int64_t b) {
    if (a == 0 || b == 0) return 0;
    if (a > kCountCap / b) return kCountCap;
    const int64_t p = a * b;
    return p > kCountCap ? kCountCap : p;
}

// Peripheral statistics of the subgraph induced on S (data_utils.py:165-221).
void peripheral_stats(const Graph& g, const std::vector<int32_t>& S, const kpgnn_khop_args& a, Scratch& sc,
                      int32_t* pe_out /*[T*2]*/, int32_t* pc_out /*[H+1]*/) {
    const int T = a.max_edge_type, H = a.max_hop_num;
    const int m = (int)S.size();
    if (m < 2) return;  // :186-187
    for (int q = 0; q < m; ++q) sc.local_id[S[q]] = q;
    // edge-type histogram over the induced directed edges (nonzero dense entries) :188-197
    sc.type_count.assign((size_t)T + 2, 0);
    int64_t n_edges = 0;
    for (int q = 0; q < m; ++q) {
        const int32_t u = S[q];
        for (int64_t e = g.ptr[u]; e < g.ptr[u + 1]; ++e) {
            if (sc.local_id[g.nbr[e]] < 0 || g.type[e] == 0) continue;
            const int64_t t = g.type[e];
            if ((size_t)t >= sc.type_count.size()) sc.type_count.resize((size_t)t + 1, 0);
            sc.type_count[(size_t)t] += 1;
            ++n_edges;
        }
    }
    if (n_edges == 0) {  // :191-192
        for (int q = 0; q < m; ++q) sc.local_id[S[q]] = -1;
        return;
    }
    {   // stable descending sort of counts of types >= 2; keep the first T (:195-204)
        const int L = (int)sc.type_count.size() - 2;
        std::vector<int32_t> idx(L);
        for (int t = 0; t < L; ++t) idx[t] = t;
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) {
            return sc.type_count[(size_t)x + 2] > sc.type_count[(size_t)y + 2];
        });
        for (int t = 0; t < T; ++t) {
            int64_t c = sc.type_count[(size_t)idx[t] + 2];
            if (c > a.max_edge_count) c = a.max_edge_count;
            pe_out[t * 2 + 0] = idx[t];           // index into edge_count[2:], i.e. type-2 (Q4)
            pe_out[t * 2 + 1] = (int32_t)c;
        }
    }
    // all-pairs BFS with cutoff H inside the subgraph (:205, :224-241); per (source, distance) class sums (:206-214)
    std::vector<int64_t> conf((size_t)H + 1, 0);
    int64_t num_sub_p_edges = 0;
    sc.dist.assign(m, 0);
    std::vector<int32_t> class_size((size_t)H + 1);
    for (int s = 0; s < m; ++s) {
        std::fill(sc.dist.begin(), sc.dist.end(), -1);
        sc.queue.clear();
        sc.queue.push_back(s);
        sc.dist[s] = 0;
        for (size_t head = 0; head < sc.queue.size(); ++head) {
            const int q = sc.queue[head];
            if (sc.dist[q] >= H) continue;
            const int32_t u = S[q];
            for (int64_t e = g.ptr[u]; e < g.ptr[u + 1]; ++e) {
                if (g.type[e] == 0) continue;
                const int v = sc.local_id[g.nbr[e]];
                if (v < 0 || sc.dist[v] >= 0) continue;
                sc.dist[v] = sc.dist[q] + 1;
                sc.queue.push_back(v);
            }
        }
        std::fill(class_size.begin(), class_size.end(), 0);
        for (int q = 0; q < m; ++q)
            if (sc.dist[q] > 0) { conf[(size_t)sc.dist[q]] += 1; class_size[(size_t)sc.dist[q]] += 1; }
        // sum of edge-type VALUES inside every distance class with >= 2 nodes (Q5)
        for (int q = 0; q < m; ++q) {
            const int d = sc.dist[q];
            if (d <= 0 || class_size[(size_t)d] < 2) continue;
            const int32_t u = S[q];
            for (int64_t e = g.ptr[u]; e < g.ptr[u + 1]; ++e) {
                const int v = sc.local_id[g.nbr[e]];
                if (v >= 0 && sc.dist[v] == d) num_sub_p_edges += g.type[e];
            }
        }
    }
    conf[0] = num_sub_p_edges;
    for (int h = 0; h <= H; ++h) pc_out[h] = (int32_t)std::min<int64_t>(conf[(size_t)h], a.max_distance_count);
    for (int q = 0; q < m; ++q) sc.local_id[S[q]] = -1;
}

int transform_graph(const Graph& g, const kpgnn_khop_args& a, GraphResult* r) {
    const int64_t n = g.n;
    const int K = a.K, T = a.max_edge_type, H = a.max_hop_num;
    const bool want_periph = H > 0 && T > 0;  // :141
    r->n = n;
    r->E = 0;
    r->pea.assign(want_periph ? (size_t)n * K * T * 2 : 0, 0);
    r->pca.assign(want_periph ? (size_t)n * K * (H + 1) : 0, 0);
    r->src.clear(); r->dst.clear(); r->attr.clear();
    if (g.nbr.empty()) return KPGNN_HOST_OK;  // no edges: nothing but zeros
    Scratch sc;
    sc.init(n, K);
    for (int64_t i = 0; i < n; ++i) {
        // ---- hop 1 row = adjacency row (with multiplicities, diagonal included for the recurrence)
        sc.nz_cur.clear();
        for (int64_t e = g.ptr[i]; e < g.ptr[i + 1]; ++e) {
            sc.w_cur[g.nbr[e]] = g.mult[e];
            sc.nz_cur.push_back(g.nbr[e]);
        }
        std::vector<int32_t> touched_reached, touched_any;
        for (int k = 0; k < K; ++k) {
            if (k > 0) {  // W_{k+1}[i,:] = W_k[i,:] A
                sc.nz_next.clear();
                for (int32_t mnode : sc.nz_cur) {
                    const int64_t wv = sc.w_cur[mnode];
                    for (int64_t e = g.ptr[mnode]; e < g.ptr[mnode + 1]; ++e) {
                        const int32_t j = g.nbr[e];
                        if (sc.w_next[j] == 0) sc.nz_next.push_back(j);
                        sc.w_next[j] = sat_add(sc.w_next[j], sat_mul(wv, g.mult[e]));
                    }
                }
                for (int32_t mnode : sc.nz_cur) sc.w_cur[mnode] = 0;
                sc.w_cur.swap(sc.w_next);
                sc.nz_cur.swap(sc.nz_next);
            }
            // ---- emit hop k: entries j != i with count > 0 (and, spd, not reached earlier)
            std::vector<int32_t>& S = sc.sets[k];
            S.clear();
            for (int32_t j : sc.nz_cur) {
                if (j == i) continue;  // fill_diagonal_(0), :123
                const int64_t cnt = sc.w_cur[j];
                if (cnt >= kInt32Limit) return fail(KPGNN_HOST_ERANGE, "walk count >= 2^31 at hop %d", k + 1);
                if (a.kernel == KPGNN_KERNEL_SPD && k > 0 && sc.reached[j]) continue;  // :68-69
                int32_t code;
                if (k == 0) {
                    code = 0;  // column 0 = dense edge-type entry, filled below from the adjacency row
                } else {
                    code = (int32_t)std::min<int64_t>(cnt, a.max_edge_attr_num) + 1;  // :85-88
                }
                sc.hop_attr[(size_t)j * K + k] = code;
                if (!sc.any[j]) { sc.any[j] = 1; touched_any.push_back(j); }
                S.push_back(j);
            }
            if (a.kernel == KPGNN_KERNEL_SPD) {
                for (int32_t j : S)
                    if (!sc.reached[j]) { sc.reached[j] = 1; touched_reached.push_back(j); }
            }
            std::sort(S.begin(), S.end());
        }
        for (int32_t mnode : sc.nz_cur) sc.w_cur[mnode] = 0;
        // column 0: edge type of the 1-hop edge (0 if (i,j) is not a 1-hop edge), :80-81
        for (int64_t e = g.ptr[i]; e < g.ptr[i + 1]; ++e) {
            const int32_t j = g.nbr[e];
            if (j != i) sc.hop_attr[(size_t)j * K + 0] = (int32_t)g.type[e];
        }
        // ---- K-hop out-edges of i in ascending target order
        std::sort(touched_any.begin(), touched_any.end());
        for (int32_t j : touched_any) {
            r->src.push_back((int32_t)i);
            r->dst.push_back(j);
            for (int k = 0; k < K; ++k) r->attr.push_back(sc.hop_attr[(size_t)j * K + k]);
            for (int k = 0; k < K; ++k) sc.hop_attr[(size_t)j * K + k] = 0;
            sc.any[j] = 0;
        }
        for (int32_t j : touched_reached) sc.reached[j] = 0;
        // ---- peripheral subgraph statistics per hop
        if (want_periph) {
            for (int k = 0; k < K; ++k)
                peripheral_stats(g, sc.sets[k], a, sc, &r->pea[((size_t)i * K + k) * T * 2],
                                 &r->pca[((size_t)i * K + k) * (H + 1)]);
        }
    }
    r->E = (int64_t)r->src.size();
    return KPGNN_HOST_OK;
}

}  // namespace

struct kpgnn_khop_plan {
    kpgnn_khop_args args;
    int64_t G = 0;
    std::vector<int64_t> node_ptr;
    std::vector<GraphResult> res;
};

extern "C" int kpgnn_host_abi_version(void) { return KPGNN_HOST_ABI_VERSION; }
extern "C" const char* kpgnn_host_last_error(void) { return g_err; }

extern "C" int kpgnn_khop_plan_create(int64_t G, const int64_t* node_ptr, const int64_t* edge_ptr,
                                      const int64_t* edge_index, const int64_t* edge_attr,
                                      const kpgnn_khop_args* args, int32_t num_threads, kpgnn_khop_plan** out) {
    if (!out) return fail(KPGNN_HOST_EINVAL, "plan_create: NULL out");
    *out = nullptr;
    if (G < 0 || !node_ptr || !edge_ptr || !args) return fail(KPGNN_HOST_EINVAL, "plan_create: NULL argument");
    if (args->K < 1 || args->max_edge_attr_num < 0 || args->max_hop_num < 0 || args->max_edge_type < 0 ||
        (args->kernel != KPGNN_KERNEL_SPD && args->kernel != KPGNN_KERNEL_GD))
        return fail(KPGNN_HOST_EINVAL, "plan_create: bad args (K=%d kernel=%d)", args->K, args->kernel);
    const int64_t Etot = edge_ptr[G];
    if (Etot > 0 && !edge_index) return fail(KPGNN_HOST_EINVAL, "plan_create: NULL edge_index");
    for (int64_t gi = 0; gi < G; ++gi) {
        if (node_ptr[gi + 1] < node_ptr[gi] || edge_ptr[gi + 1] < edge_ptr[gi])
            return fail(KPGNN_HOST_EINVAL, "plan_create: node_ptr/edge_ptr not monotone at graph %lld", (long long)gi);
        const int64_t n = node_ptr[gi + 1] - node_ptr[gi];
        if (n >= kInt32Limit) return fail(KPGNN_HOST_EINVAL, "plan_create: graph too large");
        for (int64_t e = edge_ptr[gi]; e < edge_ptr[gi + 1]; ++e) {
            const int64_t s = edge_index[e], d = edge_index[Etot + e];
            if (s < 0 || s >= n || d < 0 || d >= n)
                return fail(KPGNN_HOST_EINVAL, "plan_create: edge %lld of graph %lld out of range", (long long)e, (long long)gi);
            if (edge_attr && edge_attr[e] < 0) return fail(KPGNN_HOST_EINVAL, "plan_create: negative edge type");
        }
    }
    kpgnn_khop_plan* p = new (std::nothrow) kpgnn_khop_plan();
    if (!p) return fail(KPGNN_HOST_ENOMEM, "plan_create: out of memory");
    p->args = *args;
    p->G = G;
    p->node_ptr.assign(node_ptr, node_ptr + G + 1);
    p->res.resize((size_t)G);
    int status = KPGNN_HOST_OK;
    char first_err[512] = {0};
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t gi = 0; gi < G; ++gi) {
        Graph g;
        const int64_t e0 = edge_ptr[gi], e1 = edge_ptr[gi + 1];
        build_graph(node_ptr[gi + 1] - node_ptr[gi], e1 - e0, edge_index + e0, edge_index + Etot + e0,
                    edge_attr ? edge_attr + e0 : nullptr, &g);
        const int rc = transform_graph(g, p->args, &p->res[(size_t)gi]);
        if (rc != KPGNN_HOST_OK) {
#pragma omp critical
            {
                if (status == KPGNN_HOST_OK) { status = rc; snprintf(first_err, sizeof(first_err), "graph %lld: %s", (long long)gi, g_err); }
            }
        }
    }
    if (status != KPGNN_HOST_OK) {
        delete p;
        return fail(status, "%s", first_err);
    }
    *out = p;
    return KPGNN_HOST_OK;
}

extern "C" int kpgnn_khop_plan_sizes(const kpgnn_khop_plan* plan, int64_t* out_edge_ptr) {
    if (!plan || !out_edge_ptr) return fail(KPGNN_HOST_EINVAL, "plan_sizes: NULL argument");
    out_edge_ptr[0] = 0;
    for (int64_t gi = 0; gi < plan->G; ++gi) out_edge_ptr[gi + 1] = out_edge_ptr[gi] + plan->res[(size_t)gi].E;
    return KPGNN_HOST_OK;
}

extern "C" int kpgnn_khop_plan_export(const kpgnn_khop_plan* plan, int64_t* edge_index, int64_t* edge_attr,
                                      int64_t* pe_attr, int64_t* pea, int64_t* pca, int64_t* batch) {
    if (!plan) return fail(KPGNN_HOST_EINVAL, "plan_export: NULL plan");
    const int K = plan->args.K, T = plan->args.max_edge_type, H = plan->args.max_hop_num;
    const int64_t G = plan->G, N = plan->node_ptr[(size_t)G];
    std::vector<int64_t> eptr((size_t)G + 1, 0);
    for (int64_t gi = 0; gi < G; ++gi) eptr[(size_t)gi + 1] = eptr[(size_t)gi] + plan->res[(size_t)gi].E;
    const int64_t E = eptr[(size_t)G];
    if (pe_attr && K > 1) std::memset(pe_attr, 0, sizeof(int64_t) * (size_t)N * (size_t)(K - 1));
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t gi = 0; gi < G; ++gi) {
        const GraphResult& r = plan->res[(size_t)gi];
        const int64_t n0 = plan->node_ptr[(size_t)gi], e0 = eptr[(size_t)gi];
        if (edge_index)
            for (int64_t e = 0; e < r.E; ++e) {
                edge_index[e0 + e] = n0 + r.src[(size_t)e];
                edge_index[E + e0 + e] = n0 + r.dst[(size_t)e];
            }
        if (edge_attr)
            for (int64_t q = 0; q < r.E * K; ++q) edge_attr[e0 * K + q] = r.attr[(size_t)q];
        if (pea) {
            const size_t per = (size_t)K * T * 2;
            if (r.pea.empty()) std::memset(pea + (size_t)n0 * per, 0, sizeof(int64_t) * (size_t)r.n * per);
            else for (size_t q = 0; q < (size_t)r.n * per; ++q) pea[(size_t)n0 * per + q] = r.pea[q];
        }
        if (pca) {
            const size_t per = (size_t)K * (H + 1);
            if (r.pca.empty()) std::memset(pca + (size_t)n0 * per, 0, sizeof(int64_t) * (size_t)r.n * per);
            else for (size_t q = 0; q < (size_t)r.n * per; ++q) pca[(size_t)n0 * per + q] = r.pca[q];
        }
        if (batch)
            for (int64_t v = 0; v < r.n; ++v) batch[n0 + v] = gi;
    }
    return KPGNN_HOST_OK;
}

extern "C" void kpgnn_khop_plan_destroy(kpgnn_khop_plan* plan) { delete plan; }

// ------------------------------------------------------------------------------------------------ synthetic molecules
namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    int64_t below(int64_t n) { return (int64_t)(uniform() * (double)n); }
};

struct Molecule {
    int n;
    std::vector<std::pair<int, int>> undirected;  // a < b
    std::vector<int> bond;
    std::vector<int> atom;
};

// (defaults = the ZINC-12k shape of SURVEY.md Appendix A.2)
const kpgnn_synth_shape kZincShape = {23.2, 4.5, 9, 37, 3, {0.75, 0.20, 0.05, 0, 0, 0, 0, 0}, 21};

void make_molecule(uint64_t seed, Molecule* m, const kpgnn_synth_shape& sh) {
    SplitMix64 rng(seed);
    int n = (int)std::lround(sh.mean_nodes + sh.std_nodes * rng.normal());
    n = std::max(sh.min_nodes, std::min(sh.max_nodes, n));
    m->n = n;
    m->undirected.clear(); m->bond.clear(); m->atom.clear();
    std::vector<int> deg(n, 0);
    std::vector<std::vector<int>> adj(n);
    auto add_edge = [&](int a, int b) {
        if (a > b) std::swap(a, b);
        m->undirected.push_back({a, b});
        adj[a].push_back(b); adj[b].push_back(a);
        deg[a]++; deg[b]++;
    };
    for (int v = 1; v < n; ++v) {  // tree, max degree 3, biased towards recent nodes
        double total = 0;
        for (int u = 0; u < v; ++u) if (deg[u] < 3) total += 1.0 + 3.0 * ((double)u / std::max(1, v - 1));
        double pick = rng.uniform() * total;
        int chosen = -1;
        for (int u = 0; u < v; ++u) {
            if (deg[u] >= 3) continue;
            pick -= 1.0 + 3.0 * ((double)u / std::max(1, v - 1));
            chosen = u;
            if (pick <= 0) break;
        }
        add_edge(chosen, v);
    }
    // ring closures between nodes at tree distance 4 or 5
    const int target = 1 + (int)rng.below(3);
    std::vector<std::pair<int, int>> pairs;
    std::vector<int> dist(n), q;
    for (int a = 0; a < n; ++a) {
        std::fill(dist.begin(), dist.end(), -1);
        q.clear(); q.push_back(a); dist[a] = 0;
        for (size_t h = 0; h < q.size(); ++h) {
            const int u = q[h];
            if (dist[u] >= 5) continue;
            for (int v : adj[u]) if (dist[v] < 0) { dist[v] = dist[u] + 1; q.push_back(v); }
        }
        for (int b = a + 1; b < n; ++b) if (dist[b] == 4 || dist[b] == 5) pairs.push_back({a, b});
    }
    for (size_t i = pairs.size(); i > 1; --i) std::swap(pairs[i - 1], pairs[(size_t)rng.below((int64_t)i)]);
    int added = 0;
    for (auto& pr : pairs) {
        if (added >= target) break;
        if (deg[pr.first] >= 3 || deg[pr.second] >= 3) continue;
        if (std::find(adj[pr.first].begin(), adj[pr.first].end(), pr.second) != adj[pr.first].end()) continue;
        add_edge(pr.first, pr.second);
        ++added;
    }
    for (size_t e = 0; e < m->undirected.size(); ++e) {
        const double u = rng.uniform();
        double acc = 0;
        int b = sh.num_bond_types;
        for (int t = 0; t < sh.num_bond_types; ++t) { acc += sh.bond_prob[t]; if (u < acc) { b = t + 1; break; } }
        m->bond.push_back(b);
    }
    for (int v = 0; v < n; ++v) m->atom.push_back((int)rng.below(sh.num_atom_types));
}

}  // namespace

extern "C" int kpgnn_synth_molecules(int64_t G, uint64_t seed0, int64_t* node_ptr, int64_t* edge_ptr,
                                     int64_t* edge_index, int64_t* edge_attr, int64_t* atom_type) {
    return kpgnn_synth_molecules_ex(nullptr, G, seed0, node_ptr, edge_ptr, edge_index, edge_attr, atom_type);
}

extern "C" int kpgnn_synth_molecules_ex(const kpgnn_synth_shape* shape, int64_t G, uint64_t seed0, int64_t* node_ptr,
                                        int64_t* edge_ptr, int64_t* edge_index, int64_t* edge_attr, int64_t* atom_type) {
    if (G < 0 || !node_ptr || !edge_ptr) return fail(KPGNN_HOST_EINVAL, "synth_molecules: NULL node_ptr/edge_ptr");
    const kpgnn_synth_shape sh = shape ? *shape : kZincShape;
    if (sh.min_nodes < 1 || sh.max_nodes < sh.min_nodes || sh.num_bond_types < 1 || sh.num_bond_types > 8 || sh.num_atom_types < 1)
        return fail(KPGNN_HOST_EINVAL, "synth_molecules: bad shape (nodes %d..%d, %d bond types, %d atom types)", sh.min_nodes,
                    sh.max_nodes, sh.num_bond_types, sh.num_atom_types);
    const bool fill = edge_index != nullptr;
    if (!fill) {
        node_ptr[0] = 0; edge_ptr[0] = 0;
        std::vector<int64_t> nn((size_t)G), ne((size_t)G);
#pragma omp parallel for schedule(static, 64)
        for (int64_t gi = 0; gi < G; ++gi) {
            Molecule m;
            make_molecule(seed0 + (uint64_t)gi, &m, sh);
            nn[(size_t)gi] = m.n;
            ne[(size_t)gi] = 2 * (int64_t)m.undirected.size();
        }
        for (int64_t gi = 0; gi < G; ++gi) {
            node_ptr[gi + 1] = node_ptr[gi] + nn[(size_t)gi];
            edge_ptr[gi + 1] = edge_ptr[gi] + ne[(size_t)gi];
        }
        return KPGNN_HOST_OK;
    }
    const int64_t Etot = edge_ptr[G];
    int bad = 0;
#pragma omp parallel for schedule(static, 64)
    for (int64_t gi = 0; gi < G; ++gi) {
        Molecule m;
        make_molecule(seed0 + (uint64_t)gi, &m, sh);
        if (node_ptr[gi + 1] - node_ptr[gi] != m.n || edge_ptr[gi + 1] - edge_ptr[gi] != 2 * (int64_t)m.undirected.size()) {
            bad = 1;
            continue;
        }
        // directed edge list sorted by (src, dst), like list(G.to_directed().edges) for sorted adjacency
        std::vector<std::array<int, 3>> de;
        for (size_t e = 0; e < m.undirected.size(); ++e) {
            de.push_back({m.undirected[e].first, m.undirected[e].second, m.bond[e] + 1});
            de.push_back({m.undirected[e].second, m.undirected[e].first, m.bond[e] + 1});
        }
        std::sort(de.begin(), de.end());
        const int64_t e0 = edge_ptr[gi];
        for (size_t e = 0; e < de.size(); ++e) {
            edge_index[e0 + (int64_t)e] = de[e][0];
            edge_index[Etot + e0 + (int64_t)e] = de[e][1];
            if (edge_attr) edge_attr[e0 + (int64_t)e] = de[e][2];
        }
        if (atom_type)
            for (int v = 0; v < m.n; ++v) atom_type[node_ptr[gi] + v] = m.atom[(size_t)v];
    }
    if (bad) return fail(KPGNN_HOST_EINVAL, "synth_molecules: node_ptr/edge_ptr do not match seed0/G of the sizing call");
    return KPGNN_HOST_OK;
}
