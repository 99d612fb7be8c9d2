/* tabgnn_sampler.h — C ABI of libtabgnn_sampler.so: host-side (CPU, OpenMP) k-hop neighbour sampler + relabel for
 * edge-seeded mini-batches.  SURVEY.md §8f rank 1: replaces, per mini-batch,
 *   sample_neighbors  src/datasets/ibm_transactions_for_aml.py:61-112 (PyG NeighborSampler.sample_from_edges,
 *                     built at src/datasets/util/graph.py:38,46,53) and
 *   get_graph_inputs  src/datasets/ibm_transactions_for_aml.py:159-180 (sorted-unique relabel through a Python dict).
 * All pointers are HOST pointers.  Returns 0 / non-NULL on success; tg_sampler_last_error() has the message. */
#ifndef TABGNN_SAMPLER_H_
#define TABGNN_SAMPLER_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
const char* tg_sampler_last_error(void);
/* src/dst: int64 [E] node ids in [0, num_nodes); the edge id is the position.  Builds the in-edge CSC once. */
void* tg_sampler_create(const int64_t* src, const int64_t* dst, int64_t E, int64_t num_nodes);
void tg_sampler_destroy(void* handle);
int64_t tg_sampler_num_edges(void* handle);
int64_t tg_sampler_max_edges(int64_t B, const int32_t* fanout, int32_t hops);
/* fanout[h] < 0 = take every in-edge.  out_edge_index is [2, cap] (row stride cap) with LOCAL node ids;
 * out_nodes = sorted global ids of the n_nodes subgraph nodes; seed edges occupy the first B output slots. */
int tg_sampler_sample(void* handle, const int64_t* seed_src, const int64_t* seed_dst, const int64_t* seed_eid, int64_t B,
                      const int32_t* fanout, int32_t hops, uint64_t rng_seed, int32_t num_threads, int64_t cap,
                      int64_t* out_eid, int64_t* out_edge_index, int64_t* out_nodes, int64_t* n_edges,
                      int64_t* n_nodes);
/* The same sample in two phases, so that the caller allocates the outputs exactly: tg_sampler_draw runs the k-hop draw
 * into the handle's staging and returns the output sizes (cap = upper bound on output edges, an error beyond it);
 * tg_sampler_emit relabels and writes out_eid [n_edges], out_edge_index [2, ld] (ld >= n_edges; ld = n_edges: the
 * compact [2, E] edge_index of main.py:48) and out_nodes [n_nodes], and returns the handle to idle.  One draw may be
 * pending per handle; a new draw discards it. */
int tg_sampler_draw(void* handle, const int64_t* seed_src, const int64_t* seed_dst, const int64_t* seed_eid, int64_t B,
                    const int32_t* fanout, int32_t hops, uint64_t rng_seed, int32_t num_threads, int64_t cap,
                    int64_t* n_edges, int64_t* n_nodes);
int tg_sampler_emit(void* handle, int32_t num_threads, int64_t ld, int64_t* out_eid, int64_t* out_edge_index,
                    int64_t* out_nodes);
/* Stable counting sort of M keys in [0, N) on the host: rowptr int32 [N+1], perm int32 [M] = input positions, ascending
 * inside a segment — identical to the device's tg_csr_build (tabgnn_hip.h).  With it the sampler hands the batch's
 * CSR-by-destination / by-source to the aggregation kernels directly (get_graph_inputs + the scatter indices of
 * PNAConv.aggregate, ibm_transactions_for_aml.py:159-180; src/datasets/util/graph.py:38-53). */
int tg_host_csr(const int64_t* key, int64_t M, int64_t N, int32_t* rowptr, int32_t* perm);
/* Negative edges for link-prediction pre-training: replaces generate_negative_samples
 * (src/primitives/negative_sampling/negative_sampling.cpp:10-81; pybind11 binding :78-81; caller
 * src/utils/batch_processing.py:145).  edge_index (src,dst)[E] and the B positive edges use the same compact local
 * node ids; out_src/out_dst hold B * 2*floor(k/2) edges: per positive edge (s,d) floor(k/2) edges (s,c) then
 * floor(k/2) edges (c,d) with c uniform over the nodes that are neither s, d nor a neighbour of either.
 * k <= 0 -> error 1 with the reference's message ("num_neg_samples must be greater than 0", :13-15). */
int tg_negative_sample(const int64_t* src, const int64_t* dst, int64_t E, const int64_t* pos_src,
                       const int64_t* pos_dst, int64_t B, int32_t num_neg_samples, uint64_t seed, int32_t num_threads,
                       int64_t* out_src, int64_t* out_dst);
/* Port numbering: replaces to_adj_nodes_with_times + ports + add_ports
 * (src/datasets/util/graph.py:68-101).  in_port[e]: rank of src[e] among the distinct in-neighbours of dst[e] ordered
 * by earliest timestamp; out_port[e]: rank of dst[e] among the distinct out-neighbours of src[e].  ts may be NULL
 * (all zero, graph.py:70); equal timestamps are ordered by edge position. */
int tg_edge_ports(const int64_t* src, const int64_t* dst, const int64_t* ts, int64_t E, int64_t num_nodes,
                  int32_t num_threads, int32_t* in_port, int32_t* out_port);
/* Every index structure the fused model reads from one sampled batch ("emitting CSR-by-dst
 * directly for the aggregation kernel"), in one call and one int32 buffer: int32 endpoints of the neighbour edges, the
 * stable CSR by destination and by source, the CSR of the 2B seed endpoints, and the destination-sorted message layout
 * (parts and their order: csrc/sampler.cpp).  edge_index = int64 [2, ld] LOCAL ids, columns [0, n_seed) the seed edges.
 * offsets[14]: first int of each of the 13 parts and the total; out == NULL only fills the offsets. */
int tg_host_batch_index(const int64_t* edge_index, int64_t ld, int64_t E, int64_t n_seed, int64_t N, int32_t* out,
                        int64_t* offsets);
#ifdef __cplusplus
}
#endif
#endif
