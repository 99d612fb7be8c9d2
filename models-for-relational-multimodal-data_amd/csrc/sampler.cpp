// Host-side k-hop neighbour sampler + relabel for edge-seeded mini-batches.
//
// Replaces, on the CPU and multi-threaded, what the reference does per mini-batch in Python:
//   * sample_neighbors           src/datasets/ibm_transactions_for_aml.py:61-112
//     (torch_geometric NeighborSampler.sample_from_edges, built at src/datasets/util/graph.py:38,46,53 with
//      num_neighbors=[100,100], directional, without replacement; seed edges first, in order, and sampled edges
//      that ARE seed edges dropped: ibm…py:102-110),
//   * get_graph_inputs           ibm…py:159-180  (nodes = sorted unique endpoints, O(E) Python dict relabel).
//
// Semantics kept: hop h expands every node of the current frontier once, drawing min(in-degree, fanout[h]) of its
// incoming edges uniformly without replacement; new source nodes form the next frontier; output = seed edges (in
// the given order) followed by the sampled non-seed edges in sampling order; node ids are relabelled to their
// rank among the sorted unique endpoints.  The draw of node v at hop h depends only on (seed, h, v): results are
// identical for any thread count (the reference's sampler is unseeded; only distributional parity is possible).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct InEdge { int32_t src, eid; };      // one in-edge: 8 bytes, so the in-edges of a node share cache lines
struct Graph {
  int64_t num_nodes = 0, num_edges = 0;
  std::vector<int64_t> colptr;   // [num_nodes+1]  in-edges of node v: colptr[v]..colptr[v+1]
  std::vector<InEdge> in;        // (source node, original edge id) of each in-edge, CSC order (ids < 2^31)
  // per-handle scratch (one sample() call at a time per handle), O(1) per touch, reset at the end of every call
  std::vector<uint64_t> seedbit; // [num_edges / 64]  bit e set <=> e is a seed edge of the current call (635 KB for
                                 //                   HI-Small: cache resident, where a stamp word per edge was 20 MB)
  std::vector<int32_t> local;    // [num_nodes]  -1 = not in the current subgraph, else its local id (0 until relabelled)
  // reused output staging (capacity grows to the largest batch seen)
  std::vector<int32_t> e_src, e_dst, e_id, touched, frontier, next;
  std::vector<int64_t> drawn_seeds;     // seed edge ids of the draw that awaits its tg_sampler_emit (their bits are set)
  bool pending = false;                 // a draw has been made and not yet emitted
};

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline uint64_t bounded(uint64_t& s, uint64_t n) {   // unbiased enough for sampling (n << 2^64): multiply-shift
  return (uint64_t)(((unsigned __int128)splitmix64(s) * n) >> 64);
}

// k distinct positions out of [0, deg) (Floyd), appended in increasing position order for determinism.  Membership is a
// bitmap over the positions (round 3 scanned the <= k values chosen so far: k^2 / 2 compares and a sort per hub node —
// with fan-out 100 the hubs of a B = 8192 batch cost more than all its other nodes together); the sorted order falls out
// of walking the bitmap.  Same draws, same result set as before.
void sample_positions(int64_t deg, int64_t k, uint64_t& rng, std::vector<int64_t>& out) {
  out.clear();
  if (deg <= k) {
    for (int64_t i = 0; i < deg; ++i) out.push_back(i);
    return;
  }
  thread_local std::vector<uint64_t> bits;
  const size_t words = (size_t)(deg + 63) >> 6;
  if (bits.size() < words) bits.resize(words);
  std::fill(bits.begin(), bits.begin() + (std::ptrdiff_t)words, 0ull);
  for (int64_t j = deg - k; j < deg; ++j) {
    int64_t t = (int64_t)bounded(rng, (uint64_t)j + 1);
    if ((bits[(size_t)t >> 6] >> (t & 63)) & 1ull) t = j;
    bits[(size_t)t >> 6] |= 1ull << (t & 63);
  }
  for (size_t w = 0; w < words; ++w) {
    uint64_t b = bits[w];
    while (b) {
      out.push_back((int64_t)(w << 6) + __builtin_ctzll(b));
      b &= b - 1;
    }
  }
}

thread_local char g_err[256] = "";

}  // namespace

extern "C" {

const char* tg_sampler_last_error(void) { return g_err; }

// src/dst: int64 [E] global node ids in [0, num_nodes); edge id = position.  Returns an opaque handle (NULL on error).
void* tg_sampler_create(const int64_t* src, const int64_t* dst, int64_t E, int64_t num_nodes) {
  if (E < 0 || num_nodes <= 0 || E > 2147483647LL || num_nodes > 2147483647LL) {
    snprintf(g_err, sizeof(g_err), "tg_sampler_create: bad sizes E=%lld N=%lld (ids must fit 31 bits)", (long long)E, (long long)num_nodes);
    return nullptr;
  }
  for (int64_t e = 0; e < E; ++e)
    if (src[e] < 0 || src[e] >= num_nodes || dst[e] < 0 || dst[e] >= num_nodes) {
      snprintf(g_err, sizeof(g_err), "tg_sampler_create: edge %lld has a node id outside [0, %lld)", (long long)e,
               (long long)num_nodes);
      return nullptr;
    }
  Graph* g = new Graph();
  g->num_nodes = num_nodes;
  g->num_edges = E;
  g->colptr.assign((size_t)num_nodes + 1, 0);
  for (int64_t e = 0; e < E; ++e) g->colptr[(size_t)dst[e] + 1]++;
  for (int64_t v = 0; v < num_nodes; ++v) g->colptr[(size_t)v + 1] += g->colptr[(size_t)v];
  g->in.resize((size_t)E);
  g->seedbit.assign((size_t)(E + 63) / 64, 0ull);
  g->local.assign((size_t)num_nodes, -1);
  std::vector<int64_t> cur(g->colptr.begin(), g->colptr.end() - 1);
  for (int64_t e = 0; e < E; ++e) {   // stable: in-edges of a node stay in edge-id order
    int64_t p = cur[(size_t)dst[e]]++;
    g->in[(size_t)p].src = (int32_t)src[e];
    g->in[(size_t)p].eid = (int32_t)e;
  }
  return g;
}

void tg_sampler_destroy(void* h) { delete (Graph*)h; }

int64_t tg_sampler_num_edges(void* h) { return ((Graph*)h)->num_edges; }

// Upper bound of output edges for B seed edges: B + sum over hops of (frontier bound * fanout)
int64_t tg_sampler_max_edges(int64_t B, const int32_t* fanout, int32_t hops) {
  int64_t frontier = 2 * B, total = B;
  for (int h = 0; h < hops; ++h) {
    int64_t e = frontier * (int64_t)fanout[h];
    total += e;
    frontier = e;
  }
  return total;
}

// Outputs (caller-allocated, capacity `cap` edges / 2*cap nodes):
//   out_eid [n_edges]           global edge ids, seed edges first
//   out_edge_index [2, cap]     LOCAL node ids (row 0 = src at [0..n_edges), row 1 = dst at [cap..cap+n_edges))
//   out_nodes [n_nodes]         sorted global node ids (local id = position)
// Returns 0, or non-zero with tg_sampler_last_error().
static void sampler_reset(Graph& g) {      // the per-call scratch back to its idle state
  for (int32_t v : g.touched) g.local[(size_t)v] = -1;
  for (int64_t e : g.drawn_seeds)
    if (e >= 0 && e < g.num_edges) g.seedbit[(size_t)e >> 6] = 0ull;
  g.drawn_seeds.clear();
  g.pending = false;
}

// Phase 1 of a sample: the k-hop draw into the handle's staging (global ids).  n_edges / n_nodes = the sizes of the
// outputs tg_sampler_emit will write: the caller allocates them EXACTLY (round 3 sized every call for the worst case —
// 81 MB of fresh pages for a 5 M-edge graph — and copied the used part out three times).
int tg_sampler_draw(void* h, const int64_t* seed_src, const int64_t* seed_dst, const int64_t* seed_eid, int64_t B,
                    const int32_t* fanout, int32_t hops, uint64_t rng_seed, int32_t num_threads, int64_t cap,
                    int64_t* n_edges, int64_t* n_nodes) {
  Graph& g = *(Graph*)h;
  if (B <= 0 || hops < 0 || cap < B) {
    snprintf(g_err, sizeof(g_err), "tg_sampler_sample: bad arguments B=%lld hops=%d cap=%lld", (long long)B, hops,
             (long long)cap);
    return 1;
  }
  if (g.pending) sampler_reset(g);      // a draw that was never emitted
  std::vector<int32_t>&e_src = g.e_src, &e_dst = g.e_dst, &e_id = g.e_id, &touched = g.touched, &frontier = g.frontier,
                      &next = g.next;
  e_src.clear(); e_dst.clear(); e_id.clear(); touched.clear(); frontier.clear();
  int rc = 0;
  for (int64_t i = 0; i < B; ++i) {
    if (seed_src[i] < 0 || seed_src[i] >= g.num_nodes || seed_dst[i] < 0 || seed_dst[i] >= g.num_nodes) {
      snprintf(g_err, sizeof(g_err), "tg_sampler_sample: seed edge %lld has a node id out of range", (long long)i);
      for (int64_t q = 0; q < i; ++q)
        if (seed_eid[q] >= 0 && seed_eid[q] < g.num_edges) g.seedbit[(size_t)seed_eid[q] >> 6] = 0ull;
      return 1;
    }
    e_src.push_back((int32_t)seed_src[i]); e_dst.push_back((int32_t)seed_dst[i]); e_id.push_back((int32_t)seed_eid[i]);
    if (seed_eid[i] >= 0 && seed_eid[i] < g.num_edges) g.seedbit[(size_t)seed_eid[i] >> 6] |= 1ull << (seed_eid[i] & 63);
  }
  g.drawn_seeds.assign(seed_eid, seed_eid + B);
  g.pending = true;
  // frontier 0 = sorted unique seed endpoints (torch.cat([src, dst]).unique())
  frontier.assign(e_src.begin(), e_src.end());
  frontier.insert(frontier.end(), e_dst.begin(), e_dst.end());
  std::sort(frontier.begin(), frontier.end());
  frontier.erase(std::unique(frontier.begin(), frontier.end()), frontier.end());
  touched.assign(frontier.begin(), frontier.end());          // every node of the subgraph, in first-appearance order
  for (int32_t v : frontier) g.local[(size_t)v] = 0;
  const uint64_t* seedbit = g.seedbit.data();
  int32_t* local = g.local.data();
  const InEdge* in = g.in.data();

#ifdef _OPENMP
  const int nthreads = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  const int nthreads = 1;
#endif
  std::vector<int64_t> pos;
  for (int hop = 0; hop < hops && rc == 0; ++hop) {
    const int64_t nf = (int64_t)frontier.size();
    const int64_t k = fanout[hop];
    next.clear();
    // one edge of the hop: drop it when it is a seed edge, grow the next frontier in first-appearance order
    auto take = [&](int32_t u, int32_t v, int32_t eid) {
      if (!((seedbit[(size_t)eid >> 6] >> (eid & 63)) & 1ull)) {
        e_src.push_back(u); e_dst.push_back(v); e_id.push_back(eid);
      }
      if (local[(size_t)u] < 0) {
        local[(size_t)u] = 0;
        next.push_back(u);
        touched.push_back(u);
      }
    };
    if (nthreads > 1 && nf >= 4096) {
      // parallel draw into a deterministic layout (frontier order), then the serial merge above
      std::vector<int64_t> off((size_t)nf + 1, 0);
      for (int64_t i = 0; i < nf; ++i) {
        int64_t deg = g.colptr[(size_t)frontier[(size_t)i] + 1] - g.colptr[(size_t)frontier[(size_t)i]];
        off[(size_t)i + 1] = off[(size_t)i] + (k < 0 ? deg : std::min(deg, k));
      }
      const int64_t tot = off[(size_t)nf];
      std::vector<InEdge> h_e((size_t)tot);
      std::vector<int32_t> h_dst((size_t)tot);
#pragma omp parallel num_threads(nthreads)
      {
        std::vector<int64_t> tpos;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < nf; ++i) {
          const int64_t v = frontier[(size_t)i];
          const int64_t base = g.colptr[(size_t)v], deg = g.colptr[(size_t)v + 1] - base;
          int64_t o = off[(size_t)i];
          if (k < 0 || deg <= k) {
            for (int64_t p = 0; p < deg; ++p, ++o) { h_e[(size_t)o] = in[base + p]; h_dst[(size_t)o] = (int32_t)v; }
          } else {
            uint64_t rng = rng_seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(hop + 1)) ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(v + 1));
            sample_positions(deg, k, rng, tpos);
            for (int64_t p : tpos) { h_e[(size_t)o] = in[base + p]; h_dst[(size_t)o] = (int32_t)v; ++o; }
          }
        }
      }
      for (int64_t j = 0; j < tot; ++j) take(h_e[(size_t)j].src, h_dst[(size_t)j], h_e[(size_t)j].eid);
    } else {
      // single pass: draw and merge node by node (same edge order as the parallel layout)
      for (int64_t i = 0; i < nf; ++i) {
        const int32_t v = frontier[(size_t)i];
        const int64_t base = g.colptr[(size_t)v], deg = g.colptr[(size_t)v + 1] - base;
        // two-stage: the column pointer of the node 12 ahead, then (it has arrived by then) the first in-edges of the node 4 ahead
        if (i + 12 < nf) __builtin_prefetch(&g.colptr[(size_t)frontier[(size_t)i + 12]]);
        if (i + 4 < nf) __builtin_prefetch(&in[g.colptr[(size_t)frontier[(size_t)i + 4]]]);
        // (a third stage — the touches of the first in-edges of the node 2 ahead — measured no gain: 8.9-9.6 vs 8.7 ms)
        // the two random touches of an edge (its source's slot of `local`, its bit of the seed bitmap) are requested 8
        // edges ahead: the loop was bound by those misses (local is 2 MB for HI-Small, the in-edge lists 40 MB)
        if (k < 0 || deg <= k) {
          for (int64_t p = 0; p < deg; ++p) {
            if (p + 8 < deg) {
              __builtin_prefetch(&local[(size_t)in[base + p + 8].src]);
              __builtin_prefetch(&seedbit[(size_t)in[base + p + 8].eid >> 6]);
            }
            take(in[base + p].src, v, in[base + p].eid);
          }
        } else {
          uint64_t rng = rng_seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(hop + 1)) ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(v + 1));
          sample_positions(deg, k, rng, pos);
          const int64_t np = (int64_t)pos.size();
          for (int64_t q = 0; q < np; ++q) {
            if (q + 8 < np) {
              __builtin_prefetch(&local[(size_t)in[base + pos[(size_t)q + 8]].src]);
              __builtin_prefetch(&seedbit[(size_t)in[base + pos[(size_t)q + 8]].eid >> 6]);
            }
            take(in[base + pos[(size_t)q]].src, v, in[base + pos[(size_t)q]].eid);
          }
        }
      }
    }
    if ((int64_t)e_id.size() > cap) {
      snprintf(g_err, sizeof(g_err), "tg_sampler_sample: output capacity %lld exceeded", (long long)cap);
      rc = 2;
    }
    frontier.swap(next);
  }

  if (rc) { sampler_reset(g); return rc; }
  *n_edges = (int64_t)e_id.size();
  *n_nodes = (int64_t)touched.size();
  return 0;
}

// Phase 2: relabel (sorted unique endpoints = sorted touched list, local id = rank: torch.unique) and write the outputs of
// the pending draw — out_eid [n_edges], out_edge_index [2, ld] LOCAL ids (ld >= n_edges: ld = n_edges gives the compact
// [2, E] array the wrappers take), out_nodes [n_nodes] sorted global ids — then return the handle's scratch to idle.
int tg_sampler_emit(void* h, int32_t num_threads, int64_t ld, int64_t* out_eid, int64_t* out_edge_index,
                    int64_t* out_nodes) {
  Graph& g = *(Graph*)h;
  if (!g.pending) {
    snprintf(g_err, sizeof(g_err), "tg_sampler_emit: %s", "no pending draw");
    return 1;
  }
  std::vector<int32_t>&e_src = g.e_src, &e_dst = g.e_dst, &e_id = g.e_id, &touched = g.touched;
  int32_t* local = g.local.data();
  const int64_t ne = (int64_t)e_id.size();
  const int64_t nn = (int64_t)touched.size();
  if (ld < ne || !out_eid || !out_edge_index || !out_nodes) {
    snprintf(g_err, sizeof(g_err), "tg_sampler_emit: bad arguments ld=%lld n_edges=%lld", (long long)ld, (long long)ne);
    sampler_reset(g);
    return 1;
  }
#ifdef _OPENMP
  const int nthreads = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  const int nthreads = 1;
#endif
  (void)nthreads;
  if (nn * 16 > g.num_nodes) {     // dense subgraph: one pass over the local-id array beats sorting
    int64_t w = 0;
    for (int64_t v = 0; v < g.num_nodes; ++v)
      if (local[(size_t)v] >= 0) touched[(size_t)w++] = (int32_t)v;
  } else {
    std::sort(touched.begin(), touched.end());
  }
  for (int64_t i = 0; i < nn; ++i) { local[(size_t)touched[(size_t)i]] = (int32_t)i; out_nodes[i] = touched[(size_t)i]; }
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1 && ne >= 65536)
  for (int64_t j = 0; j < ne; ++j) {
    if (j + 16 < ne) {
      __builtin_prefetch(&local[(size_t)e_src[(size_t)j + 16]]);
      __builtin_prefetch(&local[(size_t)e_dst[(size_t)j + 16]]);
    }
    out_eid[j] = e_id[(size_t)j];
    out_edge_index[j] = local[(size_t)e_src[(size_t)j]];
    out_edge_index[ld + j] = local[(size_t)e_dst[(size_t)j]];
  }
  sampler_reset(g);
  return 0;
}

// draw + emit in one call with worst-case-sized outputs (out_edge_index row stride = cap): the round-1 entry point
int tg_sampler_sample(void* h, const int64_t* seed_src, const int64_t* seed_dst, const int64_t* seed_eid, int64_t B,
                      const int32_t* fanout, int32_t hops, uint64_t rng_seed, int32_t num_threads, int64_t cap,
                      int64_t* out_eid, int64_t* out_edge_index, int64_t* out_nodes, int64_t* n_edges,
                      int64_t* n_nodes) {
  int rc = tg_sampler_draw(h, seed_src, seed_dst, seed_eid, B, fanout, hops, rng_seed, num_threads, cap, n_edges, n_nodes);
  if (rc) return rc;
  return tg_sampler_emit(h, num_threads, cap, out_eid, out_edge_index, out_nodes);
}

// ---------------------------------------------------------------------------------------------------------------
// Negative edges for link-prediction pre-training: replaces generate_negative_samples
// (src/primitives/negative_sampling/negative_sampling.cpp:10-81, called at src/utils/batch_processing.py:145).
// Reference semantics kept: candidates are node ids 0..V-1 with V = number of distinct ids in edge_index (ids are
// assumed compact, negative_sampling.cpp:34); for positive edge (s,d) a candidate is rejected when it is s, d or
// an (undirected) neighbour of s or d (:40-44); floor(k/2) accepted candidates c give (s,c), then floor(k/2) give
// (c,d) (:60-75); k <= 0 is an error (:13-15).  Differences: seedable (draws depend only on (seed, edge, slot), so
// any thread count gives the same edges; the reference seeds from std::random_device), multi-threaded over
// positive edges, and an edge whose exclusion set covers every node is reported instead of spinning forever.
int tg_negative_sample(const int64_t* src, const int64_t* dst, int64_t E, const int64_t* pos_src,
                       const int64_t* pos_dst, int64_t B, int32_t num_neg_samples, uint64_t seed, int32_t num_threads,
                       int64_t* out_src, int64_t* out_dst) {
  if (num_neg_samples <= 0) {
    snprintf(g_err, sizeof(g_err), "num_neg_samples must be greater than 0");
    return 1;
  }
  if (E <= 0 || B < 0) {
    snprintf(g_err, sizeof(g_err), "tg_negative_sample: bad sizes E=%lld B=%lld", (long long)E, (long long)B);
    return 1;
  }
  int64_t maxid = -1;
  for (int64_t e = 0; e < E; ++e) {
    if (src[e] < 0 || dst[e] < 0) {
      snprintf(g_err, sizeof(g_err), "tg_negative_sample: negative node id at edge %lld", (long long)e);
      return 1;
    }
    maxid = std::max(maxid, std::max(src[e], dst[e]));
  }
  for (int64_t i = 0; i < B; ++i)
    if (pos_src[i] < 0 || pos_dst[i] < 0 || pos_src[i] > maxid || pos_dst[i] > maxid) {
      snprintf(g_err, sizeof(g_err), "tg_negative_sample: positive edge %lld has a node outside edge_index", (long long)i);
      return 1;
    }
  const int64_t span = maxid + 1;
  // undirected adjacency (CSR over both directions; duplicates are harmless for a membership test)
  std::vector<int64_t> ptr((size_t)span + 1, 0);
  for (int64_t e = 0; e < E; ++e) { ptr[(size_t)src[e] + 1]++; ptr[(size_t)dst[e] + 1]++; }
  int64_t V = 0;                                        // |nodeset| (negative_sampling.cpp:21-29,34)
  for (int64_t v = 0; v < span; ++v) { V += ptr[(size_t)v + 1] > 0; ptr[(size_t)v + 1] += ptr[(size_t)v]; }
  std::vector<int64_t> adj((size_t)2 * E), cur(ptr.begin(), ptr.end() - 1);
  for (int64_t e = 0; e < E; ++e) {
    adj[(size_t)cur[(size_t)src[e]]++] = dst[e];
    adj[(size_t)cur[(size_t)dst[e]]++] = src[e];
  }
  const int64_t half = num_neg_samples / 2, per = 2 * half;
  int failed = 0;
#ifdef _OPENMP
  const int nthreads = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  const int nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads) if (B * per >= 16384)
  {
    std::vector<uint32_t> mark((size_t)span, 0u);      // per-thread stamp array: mark[v] == stamp <=> v unavailable
    uint32_t stamp = 0;
#pragma omp for schedule(dynamic, 16)
    for (int64_t i = 0; i < B; ++i) {
      const int64_t s = pos_src[i], d = pos_dst[i];
      ++stamp;
      int64_t blocked = 0;
      auto block = [&](int64_t v) { if (v < V && mark[(size_t)v] != stamp) { mark[(size_t)v] = stamp; ++blocked; } };
      block(s); block(d);
      for (int64_t q = ptr[(size_t)s]; q < ptr[(size_t)s + 1]; ++q) block(adj[(size_t)q]);
      for (int64_t q = ptr[(size_t)d]; q < ptr[(size_t)d + 1]; ++q) block(adj[(size_t)q]);
      if (blocked >= V) {                              // nothing left to draw (the reference would never return)
#pragma omp atomic write
        failed = 1;
        for (int64_t j = 0; j < per; ++j) { out_src[i * per + j] = s; out_dst[i * per + j] = d; }
        continue;
      }
      uint64_t rng = seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1));
      for (int64_t j = 0; j < per; ++j) {
        int64_t c;
        do { c = (int64_t)bounded(rng, (uint64_t)V); } while (mark[(size_t)c] == stamp);
        const bool corrupt_dst = j < half;             // first half keeps the source, second half the destination
        out_src[i * per + j] = corrupt_dst ? s : c;
        out_dst[i * per + j] = corrupt_dst ? c : d;
      }
    }
  }
  if (failed) {
    snprintf(g_err, sizeof(g_err), "tg_negative_sample: a positive edge excludes every node (no negative exists)");
    return 2;
  }
  return 0;
}

// Port numbering of a multigraph (src/datasets/util/graph.py:68-101: to_adj_nodes_with_times + ports + add_ports, an
// O(E) Python dict/.numpy() loop per edge there).  in_port[e] of edge e = (u -> v): rank of u among the DISTINCT
// in-neighbours of v ordered by their earliest timestamp on an edge into v; out_port[e]: rank of v among the distinct
// out-neighbours of u ordered by earliest timestamp (graph.py:99-100: the second call runs on the flipped edge_index
// with the out-adjacency).  Equal timestamps are ordered by edge position (the reference's np.argsort default is not
// a stable sort, so its order among ties is unspecified; with distinct timestamps the two agree exactly).
// ts == NULL: all timestamps 0 (graph.py:70).  One node at a time per thread, per-thread stamp arrays, no hashing.
int tg_edge_ports(const int64_t* src, const int64_t* dst, const int64_t* ts, int64_t E, int64_t num_nodes,
                  int32_t num_threads, int32_t* in_port, int32_t* out_port) {
  if (E < 0 || num_nodes < 0 || (E > 0 && (!src || !dst || !in_port || !out_port))) {
    snprintf(g_err, sizeof(g_err), "tg_edge_ports: bad arguments (E=%lld, num_nodes=%lld)", (long long)E, (long long)num_nodes);
    return 1;
  }
  for (int64_t e = 0; e < E; ++e)
    if (src[e] < 0 || dst[e] < 0 || src[e] >= num_nodes || dst[e] >= num_nodes) {
      snprintf(g_err, sizeof(g_err), "tg_edge_ports: edge %lld has a node id outside [0, %lld)", (long long)e, (long long)num_nodes);
      return 1;
    }
#ifdef _OPENMP
  const int nthreads = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  const int nthreads = 1;
#endif
  for (int dir = 0; dir < 2; ++dir) {
    const int64_t* key = dir == 0 ? dst : src;       // the node whose ports are numbered
    const int64_t* nbr = dir == 0 ? src : dst;       // the neighbour that receives a port at that node
    int32_t* out = dir == 0 ? in_port : out_port;
    std::vector<int64_t> ptr((size_t)num_nodes + 1, 0);
    for (int64_t e = 0; e < E; ++e) ptr[(size_t)key[e] + 1]++;
    for (int64_t v = 0; v < num_nodes; ++v) ptr[(size_t)v + 1] += ptr[(size_t)v];
    std::vector<int64_t> eids((size_t)E), cur(ptr.begin(), ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) eids[(size_t)cur[(size_t)key[e]]++] = e;     // edge order within a node
#pragma omp parallel num_threads(nthreads) if (E >= 65536)
    {
      std::vector<int64_t> owner((size_t)num_nodes, -1);   // owner[u] == v  <=>  u already has a port at v
      std::vector<int32_t> port((size_t)num_nodes, 0);
#pragma omp for schedule(dynamic, 256)
      for (int64_t v = 0; v < num_nodes; ++v) {
        int64_t* b = eids.data() + ptr[(size_t)v];
        int64_t* e_ = eids.data() + ptr[(size_t)v + 1];
        if (b == e_) continue;
        if (ts) std::stable_sort(b, e_, [&](int64_t a, int64_t c) { return ts[a] < ts[c]; });
        int32_t next = 0;
        for (int64_t* q = b; q < e_; ++q) {
          const int64_t u = nbr[*q];
          if (owner[(size_t)u] != v) { owner[(size_t)u] = v; port[(size_t)u] = next++; }
          out[*q] = port[(size_t)u];
        }
      }
    }
  }
  return 0;
}

}  // extern "C"

// Stable counting sort of M keys in [0, N): rowptr[N+1], perm[M] (input positions, ascending inside a segment) — the
// same structure tg_csr_build makes on the device (include/tabgnn_hip.h), produced by the sampler's host thread so
// that the aggregation kernels take the batch's CSR-by-destination / by-source as it arrives.
extern "C" int tg_host_csr(const int64_t* key, int64_t M, int64_t N, int32_t* rowptr, int32_t* perm) {
  if (M < 0 || N <= 0 || M > 2147483647LL || N > 2147483646LL || (M > 0 && (!key || !perm)) || !rowptr) {
    snprintf(g_err, sizeof(g_err), "tg_host_csr: bad arguments M=%lld N=%lld", (long long)M, (long long)N);
    return 1;
  }
  std::fill(rowptr, rowptr + N + 1, 0);
  for (int64_t i = 0; i < M; ++i) {
    if (key[i] < 0 || key[i] >= N) {
      snprintf(g_err, sizeof(g_err), "tg_host_csr: key %lld at position %lld is outside [0, %lld)", (long long)key[i],
               (long long)i, (long long)N);
      return 1;
    }
    ++rowptr[key[i] + 1];
  }
  for (int64_t n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
  std::vector<int32_t> pos(rowptr, rowptr + N);
  for (int64_t i = 0; i < M; ++i) perm[pos[(size_t)key[i]]++] = (int32_t)i;
  return 0;
}

// Every index structure the fused model reads from one sampled batch, in ONE call and ONE int32 buffer (what
// tabgnn_amd.sampler.batch_index used to assemble from three tg_host_csr calls and a dozen numpy passes):
// edge_index = int64 [2, ld] local ids, columns [0, n_seed) = seed edges, [n_seed, E) = neighbour edges (En of them).
// Parts, in order (offsets[i] = first int of part i, offsets[13] = total):
//   0 src32 [En] | 1 dst32 [En] | 2 rowptr by dst [N+1] | 3 perm by dst [max(En,1)] | 4 rowptr by src [N+1] |
//   5 perm by src [max(En,1)] | 6 seed endpoints [2B] (sources then destinations) | 7 rowptr of 6 [N+1] |
//   8 perm of 6 [max(2B,1)] | 9 dst32[perm by dst] [En] | 10 src32[perm by dst] [En] | 11 inverse of perm by dst [En] |
//   12 position in the dst-sorted layout of the edges in by-src order [En]
// (stable counting sorts: the structures tg_csr_build makes on the device).  out == NULL: only the offsets (edge_index may
// be NULL then: they are a function of E, n_seed and N).
extern "C" int tg_host_batch_index(const int64_t* edge_index, int64_t ld, int64_t E, int64_t n_seed, int64_t N,
                                   int32_t* out, int64_t* offsets) {
  if ((!edge_index && out) || !offsets || E < 0 || n_seed < 0 || n_seed > E || ld < E || N <= 0 || E > 2147483647LL ||
      N > 2147483646LL) {
    snprintf(g_err, sizeof(g_err), "tg_host_batch_index: bad arguments E=%lld n_seed=%lld N=%lld", (long long)E,
             (long long)n_seed, (long long)N);
    return 1;
  }
  const int64_t En = E - n_seed, B2 = 2 * n_seed, En1 = En > 0 ? En : 1, B21 = B2 > 0 ? B2 : 1;
  const int64_t sizes[13] = {En, En, N + 1, En1, N + 1, En1, B2, N + 1, B21, En, En, En, En};
  offsets[0] = 0;
  for (int i = 0; i < 13; ++i) offsets[i + 1] = offsets[i] + sizes[i];
  if (!out) return 0;
  int32_t *src32 = out + offsets[0], *dst32 = out + offsets[1], *rp_d = out + offsets[2], *pm_d = out + offsets[3],
          *rp_s = out + offsets[4], *pm_s = out + offsets[5], *tei = out + offsets[6], *rp_t = out + offsets[7],
          *pm_t = out + offsets[8], *dst_sorted = out + offsets[9], *src_sorted = out + offsets[10],
          *inv = out + offsets[11], *s2s = out + offsets[12];
  const int64_t *src = edge_index, *dst = edge_index + ld;
  std::fill(rp_d, rp_d + N + 1, 0); std::fill(rp_s, rp_s + N + 1, 0); std::fill(rp_t, rp_t + N + 1, 0);
  pm_d[0] = pm_s[0] = pm_t[0] = 0;
  // one pass: range check (OR of the ids against N: a single compare per edge), int32 endpoints, both histograms, and
  // whether the neighbour edges arrive GROUPED by destination (every destination's edges contiguous) — the k-hop sampler
  // emits them so (a node is expanded once, its in-edges together), and then the by-destination structures are block
  // copies instead of a random scatter per edge
  const uint64_t un = (uint64_t)N;
  bool bad = false, grouped = true;
  for (int64_t i = 0; i < n_seed; ++i) bad |= ((uint64_t)src[i] >= un) | ((uint64_t)dst[i] >= un);
  int32_t prev = -1;
  for (int64_t j = 0; j < En; ++j) {
    const int64_t s64 = src[n_seed + j], d64 = dst[n_seed + j];
    if (((uint64_t)s64 >= un) | ((uint64_t)d64 >= un)) { bad = true; break; }
    const int32_t sj = (int32_t)s64, dj = (int32_t)d64;
    src32[j] = sj; dst32[j] = dj;
    if (dj != prev) { grouped &= rp_d[dj + 1] == 0; prev = dj; }
    ++rp_d[dj + 1]; ++rp_s[sj + 1];
  }
  if (bad) {
    for (int64_t j = 0; j < E; ++j)
      if (src[j] < 0 || src[j] >= N || dst[j] < 0 || dst[j] >= N) {
        snprintf(g_err, sizeof(g_err), "tg_host_batch_index: edge %lld has a node id outside [0, %lld)", (long long)j, (long long)N);
        return 1;
      }
  }
  for (int64_t i = 0; i < n_seed; ++i) {
    tei[i] = (int32_t)src[i]; tei[n_seed + i] = (int32_t)dst[i];
    ++rp_t[tei[i] + 1]; ++rp_t[tei[n_seed + i] + 1];
  }
  for (int64_t n = 0; n < N; ++n) { rp_d[n + 1] += rp_d[n]; rp_s[n + 1] += rp_s[n]; rp_t[n + 1] += rp_t[n]; }
  // scatter with the row pointers themselves as cursors (rp[n] ends at rp[n + 1]'s old value), shifted back afterwards
  if (grouped) {
    for (int64_t j = 0; j < En;) {                       // one block per destination
      const int32_t d = dst32[j];
      const int32_t k0 = rp_d[d], c = rp_d[d + 1] - k0;
      for (int32_t q = 0; q < c; ++q) {
        pm_d[k0 + q] = (int32_t)(j + q); inv[j + q] = k0 + q;
        dst_sorted[k0 + q] = d; src_sorted[k0 + q] = src32[j + q];
      }
      j += c;
    }
  } else {
    for (int64_t j = 0; j < En; ++j) pm_d[rp_d[dst32[j]]++] = (int32_t)j;
    for (int64_t n = N; n > 0; --n) rp_d[n] = rp_d[n - 1];
    rp_d[0] = 0;
    for (int64_t k = 0; k < En; ++k) {
      const int32_t j = pm_d[k];
      dst_sorted[k] = dst32[j]; src_sorted[k] = src32[j]; inv[j] = (int32_t)k;
    }
  }
  for (int64_t j = 0; j < En; ++j) {
    if (j + 16 < En) __builtin_prefetch(&rp_s[src32[j + 16]], 1);
    pm_s[rp_s[src32[j]]++] = (int32_t)j;
  }
  for (int64_t n = N; n > 0; --n) rp_s[n] = rp_s[n - 1];
  rp_s[0] = 0;
  for (int64_t i = 0; i < B2; ++i) pm_t[rp_t[tei[i]]++] = (int32_t)i;
  for (int64_t n = N; n > 0; --n) rp_t[n] = rp_t[n - 1];
  rp_t[0] = 0;
  for (int64_t k = 0; k < En; ++k) {
    if (k + 16 < En) __builtin_prefetch(&inv[pm_s[k + 16]]);
    s2s[k] = inv[pm_s[k]];
  }
  return 0;
}
