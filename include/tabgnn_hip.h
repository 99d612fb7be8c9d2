/* tabgnn_hip.h — C ABI of libtabgnn_hip.so: the MI355X (gfx950) kernels behind the fused
 * tabular-transformer + PNA hot path of Atahanak/models-for-relational-multimodal-data.
 *
 * The reference has no FFI on this path: everything sits behind Python nn.Module.forward() surfaces
 * (SURVEY.md §8b).  This header is therefore the boundary a torch-free host would bind; the Python host in
 * models-for-relational-multimodal-data_amd/tabgnn_amd binds it with ctypes (INTEGRATION.md shows the stub).
 * Every entry point cites the reference call it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked host; no torch types; no allocation inside;
 *   - `dt`: 0 = float32 activations, 1 = bfloat16 activations; parameters/statistics are always float32;
 *   - `stream` is a hipStream_t; kernels are launched on it and never synchronise;
 *   - return 0 on success, non-zero on error (tg_last_error() has the message); nothing aborts;
 *   - indices are int32 on the device (tg_ids_to_i32 converts and range-checks torch's int64);
 *   - feature widths (C, F) must be multiples of 8 (16-byte vector access).
 */
#ifndef TABGNN_HIP_H_
#define TABGNN_HIP_H_
#include <stdint.h>

#define TABGNN_HIP_ABI_VERSION 7

/* The `uint64_t seed` argument of every entry point that draws dropout masks is either the seed itself (bit 63 clear)
 * or TG_SEED_DEVICE(p): the device address p of a 64-bit seed word that the kernel reads when it RUNS (word 0 of a
 * step record, see tg_advance_step) — what a captured HIP graph needs, since its kernel arguments are frozen.  The
 * library keeps no seed state of its own (ABI v5 had a hidden device word per translation unit). */
#define TG_SEED_DEVICE(p) (0x8000000000000000ULL | (uint64_t)(uintptr_t)(p))

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------------------------- */
const char* tg_last_error(void);
int tg_abi_version(void);
int tg_device_check(void); /* 0 when device 0 is a gfx950 */

/* ---- index structure (edge_index of the sampled subgraph; produced at
 *      src/datasets/ibm_transactions_for_aml.py:170-173, consumed at src/nn/models/fused.py:252-254) ---- */
int tg_ids_to_i32(const int64_t* ids, int64_t M, int32_t N, int32_t* out, int32_t* err_flag, void* stream);
int64_t tg_csr_workspace_ints(int64_t M, int32_t N);
/* stable counting sort of M keys in [0,N): rowptr[N+1], perm[M] (positions in input order inside a segment) */
int tg_csr_build(const int32_t* key, int64_t M, int32_t N, int32_t* rowptr, int32_t* perm, int32_t* work,
                 void* stream);

/* ---- k-hop neighbour sampler + relabel over the HBM-resident graph (the device form of
 *      tg_sampler_draw / tg_sampler_emit of tabgnn_sampler.h, i.e. of sample_neighbors + get_graph_inputs,
 *      src/datasets/ibm_transactions_for_aml.py:61-112,159-180).  The graph: esrc / edst int64 [E] = the rows of the
 *      full edge_index; its CSC colptr int32 [N+1], in_src / in_eid int32 [E] (tg_csr_build of the destination column:
 *      in-edges of v at colptr[v] .. colptr[v+1], ascending edge id).  seeds int64 [B] edge ids (device); fanout host
 *      int32 [hops], < 0 = every in-edge, else <= 128.  cap = staging capacity in edges, an upper bound on seed + drawn
 *      edges (min(tg_sampler_max_edges(B, fanout, hops), E + B), at most 8 Mi).  seedbit: tg_gsampler_seedbit_bytes(E)
 *      bytes, zero before the first draw (emit clears what the draw set); workspace: tg_gsampler_workspace_bytes(N, cap).
 *      draw: counts_out (device int64 [4]) = {output edges, output nodes, staged edges, error (0 = ok; 1/2 = seed id /
 *      endpoint out of range, 3 = cap too small)} once the stream reaches the end.  emit (after the caller has read the
 *      counts and allocated): out_eid int64 [n_edges] (seed edges first, in order), out_edge_index int64 [2, ld] LOCAL
 *      ids, out_nodes int64 [n_nodes] sorted global ids.  A sample is a pure function of (graph, seeds, fanout, rng_seed). */
int64_t tg_gsampler_seedbit_bytes(int64_t E);
int64_t tg_gsampler_workspace_bytes(int32_t N, int64_t cap);
int tg_gsampler_draw(const int64_t* seeds, int64_t B, const int64_t* esrc, const int64_t* edst, int64_t E,
                     const int32_t* colptr, const int32_t* in_src, const int32_t* in_eid, int32_t N, const int32_t* fanout,
                     int32_t hops, uint64_t rng_seed, int64_t cap, void* seedbit, void* workspace, int64_t* counts_out,
                     void* stream);
int tg_gsampler_emit(const int64_t* seeds, int64_t B, int32_t N, int64_t E, int64_t cap, int64_t ld, void* seedbit,
                     void* workspace, int64_t* out_eid, int64_t* out_edge_index, int64_t* out_nodes, void* stream);

/* ---- stype encoders (torch_frame fork EmbeddingEncoder/LinearEncoder/TimestampEncoder/ProjectionEncoder;
 *      src/datasets/ibm_transactions_for_aml.py:283-294,313-319; utils.py:357-359) -------------------- */
typedef struct {
  int32_t kind;    /* 0 numerical, 1 categorical, 2 timestamp, 3 relation */
  int32_t out_col; /* column position in the output */
  int32_t src_col; /* column inside its stype tensor */
  int32_t rows;    /* categorical: table rows = cardinality + 1 (row 0 = padding / NaN) */
  int32_t tab_off; /* categorical: first row of the column's table in the concatenated table */
  int32_t acc_off; /* backward: float offset of the column's accumulators; -1 = global atomics (big table) */
  int32_t ts_slot; /* timestamp: index among the timestamp columns of this launch */
  int32_t pad;
} tg_enc_col;
typedef struct {
  int32_t ncol, nts;
  tg_enc_col col[16];
} tg_enc_desc; /* HOST struct, passed by pointer, copied into the kernel arguments */
typedef struct {
  const float* num; int32_t nn;        /* [R, nn] */
  const int64_t* cat; int32_t nc;      /* [R, nc] */
  const int64_t* ts; int32_t nt;       /* [R, nt, 7] (year, month, day, dayofweek, hour, minute, second) */
  const float* rel; int32_t nr;        /* [R, nr] */
  const float *num_mean, *num_std, *num_w, *num_b; /* [nn] [nn] [nn,C] [nn,C] */
  const float* cat_table;              /* [sum rows, C] */
  const float *ts_min_year, *ts_w, *ts_b; /* [nt] [nt,7,8,C] [nt,C] */
  const float *rel_w, *rel_b;          /* [nr,C] */
  const int64_t* row_ids;              /* [R] or NULL.  Non-NULL: num / cat / ts / rel are the WHOLE HBM-resident raw table and
                                        * output row r encodes table row row_ids[r] (the batch is a list of edge / node ids:
                                        * TensorFrame.__getitem__ of ibm_transactions_for_aml.py:163,168 without the row copy) */
} tg_enc_ptrs; /* HOST struct of device pointers */
int tg_encode_max_cols(void);
int tg_encode_small_table_rows(void);
int tg_encode_bwd_blocks(void);
int tg_encode_fwd(const void* desc, const void* ptrs, void* out /*[R,ncols,C]*/, int64_t R, int32_t ncols, int32_t C,
                  int32_t dt, void* stream);
int tg_encode_bwd(const void* desc, const void* ptrs, const void* g /*[R,ncols,C]*/, int64_t R, int32_t ncols,
                  int32_t C, int32_t acc_floats, float* dflat, float* partials, float* big_table_grad, int32_t dt,
                  void* stream);
/* The calendar features of timestamp column src_col (TimestampEncoder: 7 fields x out_size 8 sinusoidal / cyclic
 * values, oracle/encoders.py) as a bf16 GEMM operand feats [R,128]: columns 0..55 the features, column 56 = 1 (bias
 * column), the rest 0; row r reads table row row_ids[r] when row_ids != NULL.  The encoder column is then
 * tg_gemm_nt_bf16(feats, [W | b | 0]) and its parameter gradient tg_gemm_tn_bf16(g, feats).  With w (fp32 [56,C], the
 * column's TimestampEncoder weight) the same launch also packs that GEMM weight wext = [W^T | b | 0] (bf16 [C,128]). */
int tg_encode_ts_features(const int64_t* ts, int32_t nt, int32_t src_col, const float* min_year /*[nt]*/,
                          const int64_t* row_ids, void* feats, int64_t R, const float* w, const float* b, void* wext,
                          int32_t C, void* stream);
/* adds segments of a reduced gradient vector into parameter gradient buffers in one launch: table int64 [nseg][3] on
 * the device = (destination float* as integer, offset into src, length).  Used by the encoder backward so that no
 * per-parameter zero-fill / slice copy / autograd add runs. */
int tg_scatter_add_segments(const float* src, const int64_t* table, int32_t nseg, int64_t max_len, void* stream);
/* Gradient of embedding tables too large for tg_encode_bwd's LDS accumulators (torch nn.Embedding inside the fork's
 * EmbeddingEncoder; reference call site src/datasets/ibm_transactions_for_aml.py:283-319), in a FIXED order: the caller
 * sorts keys[slot * R + r] = base[slot] + clamp(category(r, slot) + 1, 0, rows[slot] - 1) with tg_csr_build (buckets =
 * sum of the tables' rows) and one wave per (column, category) bucket sums rows of g [R, gstride] in that order into
 * dst[category, :C] (fp32; added when accumulate != 0).  cols: device int64 [ncol][4] = (base, out_col, dst pointer,
 * rows).  tg_encode_bwd with big_table_grad == NULL leaves these tables alone (non-NULL: the old atomicAdd path). */
int tg_embed_grad_sorted(const void* g, int64_t gstride, const int32_t* rowptr, const int32_t* perm, int64_t R,
                         const int64_t* cols, int32_t ncol, int32_t max_rows, int32_t C, int32_t accumulate, int32_t dt,
                         void* stream);

/* ---- column self-attention core (torch nn.MultiheadAttention inside nn.TransformerEncoderLayer;
 *      src/nn/models/fused.py:83-92,160,164,187-196,249; src/nn/models/tabgnn.py:199-208,219) ---------- */
int tg_attn_fwd(const void* qkv /*[R,S,3C]*/, void* out /*[R,S,C]*/, float* lse /*[R,H,S]*/, int64_t R, int32_t S,
                int32_t C, int32_t H, float p_drop, uint64_t seed, uint32_t rstream, int32_t dt, void* stream);
int tg_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t R, int32_t S,
                int32_t C, int32_t H, float p_drop, uint64_t seed, uint32_t rstream, int32_t dt, void* stream);

/* ---- LayerNorm with fused pre-add and residual combine:
 *      out = alpha*res + beta_c*LN(a + dropout(b + bias_b))      (norm1/norm2 of the encoder layer; tab_norm
 *      fused.py:160,164,249 / tabgnn.py:219; fuse[0], fuse_norm fused.py:224,258) -------------------------- */
int tg_ln_partials_floats(int64_t M, int32_t C);
int tg_ln_fwd(const void* a, const void* b, const float* bias_b, const float* gamma, const float* beta,
              const void* res, void* out, float* stats /*[M,2]*/, int64_t M, int32_t C, float eps, float alpha,
              float beta_c, float p_drop, uint64_t seed, uint32_t rstream, int32_t dt, void* stream);
int tg_ln_bwd(const void* a, const void* b, const float* bias_b, const float* gamma, const float* stats,
              const void* dout, void* da, void* db, void* dres, float* dparams /*[3C]: dgamma,dbeta,dbias_b*/,
              float* partials, int64_t M, int32_t C, float alpha, float beta_c, float p_drop, uint64_t seed,
              uint32_t rstream, int32_t accum_da /*1: da += (another branch's gradient is already there)*/,
              float* acc_gamma, float* acc_beta, float* acc_bias /* any non-NULL: the parameter gradients are ADDED to
              these [C] buffers (the parameters' .grad) instead of being written to dparams; a NULL one is skipped */,
              int32_t dt, void* stream);

/* The last two LayerNorm backwards of the encoder layer (tab_norm after norm2: out = alpha*x + beta_c*LN_t(x2),
 * x2 = LN_2(z2), z2 = x1 + dropout(y2 + b2), fused.py:249 over torch's post-norm layer) in one pass: the gradient of
 * x2 never reaches HBM.  Inputs x2, z2 [M,C], their statistics (mean, rstd) pairs, dout; outputs dres = alpha*dout
 * (NULL when alpha == 0), d_x1 = dL/dz2, d_y2 = d_x1 * dropout mask (same counter RNG site as the forward).
 * dparams [5C] = (dgamma_t, dbeta_t, dgamma_2, dbeta_2, dbias_2) is written, or — acc != NULL: five [C] pointers,
 * NULL entries skipped — the sums are ADDED to those buffers.  partials: 2048*5*C floats. */
int tg_ln_tail_ln_bwd(const void* x2, const void* z2, const float* gamma_t, const float* stats3, const float* gamma_2,
                      const float* stats2, const void* dout, void* dres, void* d_x1, void* d_y2, float* dparams,
                      float* partials, int64_t M, int32_t C, float alpha, float beta_c, float p_drop, uint64_t seed,
                      uint32_t rstream, float* const* acc, int32_t dt, void* stream);

/* ---- BatchNorm1d (+ReLU) with residual combine: out = alpha*res + beta_c*relu(BN(x))
 *      (torch_geometric BatchNorm -> BatchNorm1d; fused.py:214,252; tabgnn.py:172,188) ------------------- */
int tg_bn_partials_floats(int64_t N, int32_t F);
int tg_bn_act_res_fwd(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* mean, float* rstd, void* out, float* partials, int64_t N, int32_t F,
                      int32_t training, float momentum, float eps, int32_t relu, float alpha, float beta_c,
                      int64_t n_stat, int32_t phase, const int32_t* row_limit, int32_t dt, void* stream);
/* row_limit (device int32, or NULL = every row counts; training, phase 0 only): padded batches — the statistics are
 * taken over the first *row_limit rows and the padding rows get a zero input gradient.  Read by the kernels at run
 * time, so one captured graph serves batches of different true size.  (An explicit argument since ABI v6.)
 * phase 0: everything (n_stat ignored).  Synchronised BatchNorm across data-parallel ranks (optional):
 * phase 1 = local statistics only -> (sum x, sum x^2) at partials + 512*2*F, which the caller all-reduces in place;
 * phase 2 = finalize with n_stat = rows of ALL ranks (running statistics updated from the global batch) + apply.
 * Backward: phase 1 = local (sum dz, sum dz*xhat) -> dparams, caller all-reduces a COPY (the parameter gradients stay
 * local sums, the data-parallel all-reduce averages them as usual); phase 2 = dx from the global sums and n_stat. */
int tg_bn_act_res_bwd(const void* x, const void* dout, const float* gamma, const float* beta, const float* mean,
                      const float* rstd, void* dx, void* dres, float* dparams /*[2F]: dbeta,dgamma*/, float* partials,
                      int64_t N, int32_t F, int32_t training, int32_t relu, float alpha, float beta_c, int64_t n_stat,
                      int32_t phase, const int32_t* row_limit, int32_t dt, void* stream);

/* ---- activation + dropout after a Linear (encoder FFN; fuse MLP fused.py:224-231; heads decoder.py:14-15)
 *      act: 0 none, 1 relu, 2 leaky_relu(0.01) ------------------------------------------------------------ */
int tg_act_dropout_fwd(const void* x, void* y, int64_t n, int32_t act, float p_drop, uint64_t seed, uint32_t rstream,
                       int32_t dt, void* stream);
int tg_act_dropout_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t act, float p_drop, uint64_t seed,
                       uint32_t rstream, int32_t dt, void* stream);
int tg_axpby(const void* a, const void* b, void* y, int64_t n, float alpha, float beta, int32_t dt, void* stream);
/* ClassifierHead / NodeClassificationHead readout MLP (src/nn/gnn/decoder.py:5-32: Linear(D0,50) ReLU Dropout
 * Linear(50,25) ReLU Dropout Linear(25,NC)) as one forward and one backward kernel.  h0 [B,D0], w1 [50,D0], w2 [25,50]
 * in the activation dtype dt; biases, w3 [NC,25] and logits [B,NC] fp32; z1 [B,50], z2 [B,25] (dt) are the saved
 * pre-activations.  rs1 / rs2: dropout stream ids of the two Dropout sites (masks = those of tg_act_dropout_fwd on
 * the same (seed, stream, element index)).  Supported: D0 % 8 == 0, D0 <= 512, H1 == 50, H2 == 25, NC <= 10
 * (tg_head_mlp_supported).  Backward: dh0 [B,D0] (dt); parameter gradients fp32, written (accumulate 0) or added
 * (accumulate 1: .grad += semantics) in a fixed order; workspace tg_head_mlp_partial_floats() floats. */
int32_t tg_head_mlp_supported(int32_t D0, int32_t H1, int32_t H2, int32_t NC);
int64_t tg_head_mlp_partial_floats(int32_t D0, int32_t H1, int32_t H2, int32_t NC);
int tg_head_mlp_fwd(const void* h0, const void* w1, const float* b1, const void* w2, const float* b2, const float* w3,
                    const float* b3, void* z1, void* z2, float* logits, int64_t B, int32_t D0, int32_t H1, int32_t H2,
                    int32_t NC, float p_drop, uint64_t seed, uint32_t rs1, uint32_t rs2, int32_t dt, void* stream);
int tg_head_mlp_bwd(const float* g, const void* h0, const void* z1, const void* z2, const void* w1, const void* w2,
                    const float* w3, void* dh0, float* workspace, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                    float* db3, int32_t accumulate, int64_t B, int32_t D0, int32_t H1, int32_t H2, int32_t NC, float p_drop,
                    uint64_t seed, uint32_t rs1, uint32_t rs2, int32_t dt, void* stream);
/* y1 = alpha*a + beta*b (a NULL: beta*b) and y2 = gamma*b in one pass over b: the backward of the residual mixes
 * (fused.py:254) when a's gradient goes to a shared gradient buffer. */
int tg_axpby2(const void* a, const void* b, void* y1, void* y2, int64_t n, float alpha, float beta, float gamma, int32_t dt,
              void* stream);
/* out[C] fp32 (+)= column sums of x [R, C] with row stride ld (elements): the gradient of the shared CLS vector
 * (column 0 of the [R, S, C] row gradient, fused.py:158-159).  workspace: tg_col_sum_workspace_floats(R, C) floats.
 * Fixed summation order (no atomics). */
int64_t tg_col_sum_workspace_floats(int64_t R, int32_t C);
int tg_col_sum(const void* x, int64_t R, int32_t C, int64_t ld, float* out, float* workspace, int32_t accumulate, int32_t dt,
               void* stream);
/* dst[r, 0:W] = (c < C ? s_head : s_tail) * src[r, c] for c < w_src, zero beyond (src row pitch ld_src elements): the
 * row-wise pieces of the backward of the CLS merge (fused.py:259-260) and of the seed gathers (fused.py:257,
 * decoder.py:18-19) — one pass each instead of clone / slice-multiply / zeros / slice-copy. */
int tg_row_head_scale(const void* src, int64_t ld_src, int32_t w_src, void* dst, int64_t B, int32_t W, int32_t C, float s_head,
                      float s_tail, int32_t dt, void* stream);
/* CLS merge of the fused layer, fused.py:259-260 */
int tg_cls_merge_fwd(const void* xtab /*[B,S,C]*/, const void* xf /*[B,D]*/, void* out, int64_t B, int32_t S,
                     int32_t C, int32_t D, int32_t dt, void* stream);

/* ---- PNA message passing (torch_geometric 2.5.3 PNAConv as configured at fused.py:200-207) -------------- */
/* out[r] = [A[ia[r]] | B[ib[r]] | C[ic[r]]]  (index NULL = identity; optional ReLU on A,B parts):
 * PNAConv.message input [x_i,x_j,e]; edge update fused.py:254; fuse input fused.py:257; decoder.py:18-19 */
int tg_gather_concat3(const void* a, const int32_t* ia, int64_t sa, int32_t wa, int32_t relu_a, const void* b,
                      const int32_t* ib, int64_t sb, int32_t wb, int32_t relu_b, const void* c, const int32_t* ic,
                      int64_t sc, int32_t wc, void* out, int64_t rows, int32_t dt, void* stream);
/* backward of the two gathered parts as a deterministic segmented sum over CSR(s) */
int64_t tg_segment_hub_ints(int64_t total_rows); /* size of hub_work for tg_segment_sum2 */
/* pmA == NULL: segment A's rows are the CSR positions themselves (g laid out in CSR order: the destination-sorted
 * messages).  accumulate != 0: dx += the sums (rows with empty segments are not touched): dx is the gradient buffer the consumers of
 * one tensor share (every consumer adds its part in place: no autograd accumulation pass over [N,F]) */
int tg_segment_sum2(const void* g, int64_t gstride, int32_t offA, const int32_t* rpA, const int32_t* pmA, int32_t offB,
                    const int32_t* rpB, const int32_t* pmB, int32_t seedB, const void* relu_src, void* dx, int32_t N,
                    int32_t F, int32_t* hub_work, int32_t accumulate, int32_t dt, void* stream);
/* dst[r, 0:W] (+)= src[idx ? idx[r] : r, 0:W] (src row pitch sstride elements): a column block of a wider gradient,
 * optionally row-gathered, delivered into such a shared gradient buffer */
int tg_rows_add(void* dst, const void* src, const int32_t* idx, int64_t rows, int32_t W, int64_t sstride,
                int32_t accumulate, int32_t dt, void* stream);
/* mean|max|min|std of messages h[E,F] per destination -> agg[N,4F]  (PNAConv.aggregate, "the SpMM") */
/* perm == NULL: h rows are already in CSR (destination-sorted) order (E = number of rows of h) */
/* hub_work: tg_segment_hub_ints(E) ints with hub_work[0] == 0 on entry, or NULL.  With it, destinations of more than
 * 512 rows (reverse message passing: the heavy-tailed sources become destinations) are only LISTED there by these two
 * launches and reduced, a whole workgroup each, by tg_pna_aggregate_hubs (forward: dh == NULL, fills agg rows;
 * backward: fills dh rows). */
int tg_pna_aggregate_fwd(const void* h, const int32_t* rowptr, const int32_t* perm, void* agg, int32_t N, int32_t F,
                         int64_t E, int32_t* hub_work, int32_t dt, void* stream);
int tg_pna_aggregate_bwd(const void* h, const void* agg, const void* dagg, const int32_t* rowptr, const int32_t* perm,
                         void* dh, int32_t N, int32_t F, int32_t* hub_work, int32_t dt, void* stream);
int tg_pna_aggregate_hubs(const void* h, const void* agg, const void* dagg, const int32_t* rowptr, const int32_t* perm,
                          void* dh, int32_t F, const int32_t* hub_work, int32_t dt, void* stream);
/* degree scalers applied after the post GEMM: out = xw + G0 + amp*G1 + att*G2 (DegreeScalerAggregation) */
int tg_pna_scale_combine_fwd(const void* xw, const void* G /*[N,3F]*/, const int32_t* rowptr, const float* avg_log,
                             void* out, int32_t N, int32_t F, int32_t dt, void* stream);
int tg_pna_scale_combine_bwd(const void* gout, const int32_t* rowptr, const float* avg_log, void* dG, int32_t N,
                             int32_t F, int32_t dt, void* stream);
/* The same post projection with the degree scalers folded INTO the GEMMs (the [N,3F] tensor G and its gradient
 * never exist).  scales = fp32 (amp, att) pairs per node from tg_pna_degree_scalers, in a buffer of ceil(N/128)*128
 * rows (the NT kernel reads whole 128-row tiles; rows >= N are never used in a stored result) (DegreeScalerAggregation:
 * amp = log(deg+1)/avg_log, att = avg_log/log(max(deg,1)+1)).
 *  tg_gemm_nt_scaled_bf16: Y[R,N] (+)= X W_0^T + amp*X W_1^T + att*X W_2^T, X [R,kreal], W [N,3*kreal] with
 *    128-column block 3c+s = W_s[:, 128c:128c+128]
 *    (forward: X = agg [N,4F], W_s = post-projection block of scaler s; input gradient: X = dOut [N,F], W_s = W_s^T);
 *  tg_gemm_tn_scaled_bf16: out[3*mreal,N] fp32 (+)= [G | amp*G | att*G]^T X (weight gradient; G = dOut, X = agg). */
int tg_pna_degree_scalers(const int32_t* rowptr, const float* avg_log, float* out /*[N,2]*/, int32_t N, void* stream);
int tg_gemm_nt_scaled_bf16(const void* X, const void* W, const float* scales, void* Y, int64_t R, int32_t N,
                           int32_t kreal, int64_t ldx, int64_t ldy, int32_t flags /*0 | 4 (Y +=)*/, void* stream);

/* The same post projection, forward, as ONE MFMA-bound kernel with the x term and the bias inside (csrc/post_scaled.hip):
 *   out[R,128] = bias + x Wx^T + agg W_0^T + amp*(agg W_1^T) + att*(agg W_2^T)
 * three accumulator sets (one per scaler) combined in the epilogue; operands HBM/L2 -> LDS by LDS-DMA, 3-stage ring.
 * wcat [128, 3K] bf16 in tg_pna_fold_fwd's virtual-chunk order, wx [128,128] bf16, scales fp32 [>=R][2], K % 128 == 0.
 * (PNAConv.forward: post_nns + lin over [x | scalers x aggregators], torch_geometric 2.5.3, reached from fused.py:204-214) */
int tg_pna_post_dagg_bf16(const void* g, const void* wt_cat /*[K,384] = [W_0^T | W_1^T | W_2^T]*/, const float* scales,
                          void* dagg /*[R,K]*/, int64_t R, int32_t K, int64_t ld_g, int64_t ld_dagg, void* stream);
int tg_pna_post_fwd_bf16(const void* agg, const void* x, const void* wcat, const void* wx, const float* bias,
                         const float* scales, void* out, int64_t R, int32_t K, int64_t ld_agg, int64_t ld_x,
                         int64_t ld_out, void* stream);
int tg_gemm_tn_scaled_bf16(const void* G, const void* X, const float* scales, float* out, float* workspace, int64_t R,
                           int32_t mreal, int32_t N, int64_t ldg, int64_t ldx, int32_t accumulate, void* stream);
/* The weight folds of PNAConv (torch_geometric 2.5.3 PNAConv.forward: edge_encoder -> pre_nns[0], post_nns[0] -> lin,
 * as configured at src/nn/models/fused.py:200-207 / src/nn/gnn/pna.py:59-72), all on fp32 masters, F = node width,
 * Fe = raw edge width:  w_msg [F,2F+Fe] = [P[:, :2F] | P[:, 2F:] We], b_msg = pb + P[:, 2F:] be;  w_eff = Lw Qw,
 * b_eff = Lw qb + lb, w_x = w_eff[:, :F];  w_st [3F,4F]: row s*F+f, column kk*F+i = w_eff[f, F + (s*4+order[kk])*F + i]
 * (order[kk] = slot of the aggregation kernel's block kk = mean|max|min|std in the module's aggregator list).
 * The bf16 outputs (all or none) are the operand layouts of the step's GEMMs: row-major and transposed shadows of w_msg
 * and w_x, w_cat [F,12F] (128-column block 3c+s = W_s[:, 128c:128c+128]) and wt_cat [4F,3F] = [W_0^T | W_1^T | W_2^T]
 * for tg_gemm_nt_scaled_bf16.  tg_pna_fold_bwd maps the gradients of the five folded tensors (any may be NULL = zero)
 * back to the eight parameters; bit i of `accumulate` (order dP,dpb,dWe,dbe,dQw,dqb,dLw,dlb) adds into the buffer
 * instead of overwriting it; NULL outputs are skipped. */
typedef struct { const float *P, *pb, *We, *be, *Qw, *qb, *Lw, *lb; } tg_fold_params;
typedef struct {
  float *w_msg, *b_msg, *w_x, *b_eff, *w_st;
  void *w_msg_lp, *w_msg_lp_t, *w_x_lp, *w_x_lp_t, *w_cat, *wt_cat;
} tg_fold_out;
typedef struct { const float *dw_msg, *db_msg, *dw_x, *db_eff, *dw_st; } tg_fold_grads;
typedef struct { float *dP, *dpb, *dWe, *dbe, *dQw, *dqb, *dLw, *dlb, *ws /* tg_pna_fold_ws_floats(F) floats, for dLw */; int32_t accumulate; } tg_fold_dparams;
int64_t tg_pna_fold_ws_floats(int32_t F);
int tg_pna_fold_fwd(const tg_fold_params* p, const tg_fold_out* o, int32_t F, int32_t Fe, const int32_t* order /*[4], host*/,
                    void* stream);
int tg_pna_fold_bwd(const tg_fold_params* p, const tg_fold_grads* g, const tg_fold_dparams* o, int32_t F, int32_t Fe,
                    const int32_t* order /*[4], host*/, void* stream);
/* GINEConv aggregation (src/nn/gnn/gine.py:18-19,62-72 through torch_geometric 2.5.3 GINEConv.forward/message):
 * out[n] = self_scale*x[n] + sum_{e: dst[e]=n} relu(x[src[e]] + le[e]), le = lin(edge_attr) [E,F]; rowptr/perm = the
 * stable by-destination CSR (tg_csr_build); self_scale = 1+eps, or 0 for the (x, None) form of GINEConvHetero
 * (gine.py:31-32).  hub_work: tg_segment_hub_ints(E) ints (destinations with > 256 rows get a block each). */
int tg_gine_aggregate_fwd(const void* x, const void* le, const int32_t* src, const int32_t* rowptr,
                          const int32_t* perm, float self_scale, void* out, int32_t N, int32_t F, int32_t* hub_work,
                          int32_t dt, void* stream);
/* its message gradient: dle[e] = (x[src[e]] + le[e] > 0) ? dout[dst[e]] : 0; dx is then the by-source
 * tg_segment_sum2 of dle (+ self_scale*dout). */
int tg_gine_message_bwd(const void* x, const void* le, const void* dout, const int32_t* src, const int32_t* dst,
                        void* dle, int64_t E, int32_t F, int32_t dt, void* stream);
/* seed-endpoint pooling of the fused layer, fused.py:261-268 (unique / index_add_ / bincount / mean) */
int tg_seed_pool_fwd(const void* x, const void* xf, const int32_t* rowptr, const int32_t* perm, void* out, int32_t N,
                     int32_t F, int32_t B, int32_t C, int32_t dt, void* stream);
/* same update applied in place to x (only the <= 2B seed-endpoint rows are touched), as fused.py:268 does */
int tg_seed_pool_inplace(void* x, const void* xf, const int32_t* tei, const int32_t* rowptr, const int32_t* perm,
                         int32_t N, int32_t F, int32_t B, int32_t C, int32_t dt, void* stream);
int tg_seed_pool_bwd(const void* g, const int32_t* tei, const int32_t* rowptr, void* dx, void* dxf, int32_t N,
                     int32_t F, int32_t B, int32_t C, int32_t dt, void* stream);

/* ---- forward / input-gradient GEMM of the Linears on the path (torch.nn.Linear / TransformerEncoderLayer's
 *      in_proj, out_proj, linear1, linear2: fused.py:83-92; PNA and edge-update projections) with the elementwise
 *      tail fused:  Y[R,N] (bf16) = epilogue(X[R,K] W[N,K]^T);  epilogue = + bias (fp32 [N] or NULL) | flags&1 ReLU |
 *      flags&2 dropout(p_drop; element index r*N+n of stream (seed, rstream), as tg_act_dropout_*) | flags&8 gate:
 *      x 1/(1-p_drop) where gate[r,n] > 0 else 0 (backward of drop(relu(.)) from its saved output) | flags&4 Y += |
 *      flags&16 LeakyReLU(0.01) in place of ReLU (the fuse MLP, fused.py:199-202); with flags&8 a negative saved
 *      output passes 0.01/(1-p_drop) (kept on the negative slope), zero = dropped.
 *      MFMA 32x32x16 bf16, fp32 accumulation.  tg_gemm_nt_supported: N % 128 == 0 and K % 128 == 0. ------------- */
int32_t tg_gemm_nt_supported(int64_t R, int32_t N, int32_t K);
int tg_gemm_nt_bf16(const void* X, const void* W, const float* bias, const void* gate /* flags&8, else NULL */, void* Y,
                    int64_t R, int32_t N, int32_t K, int64_t ldx, int64_t ldy, int32_t flags, float p_drop, uint64_t seed,
                    uint32_t rstream, void* stream);

/* The same product with the X operand GATHERED on the fly: row r of X is [S0[i0[r]] | S1[i1[r]] | S2[i2[r]]], three
 * 128-column sources (row pitch `stride` elements; idx NULL = row r itself) — the concatenation [x_i | x_j | e] of
 * PNAConv.message (torch_geometric 2.5.3) / the edge update of fused.py:253-254 read from the node and edge embeddings
 * without materialising [E,384].  K = 384.  flags: 1 ReLU | 4 Y +=.  tg_gemm_tn_gather3_bf16 is its weight gradient:
 * out[M,384] fp32 (+)= G[R,M]^T X (M % 128 == 0), colsum = column sums of G, workspace of
 * tg_gemm_tn_gather3_workspace_floats(R, M) floats. */
typedef struct { const void* src[3]; const int32_t* idx[3]; int64_t stride[3]; } tg_gather3;
int tg_gemm_nt_gather3_bf16(const tg_gather3* gs, const void* W, const float* bias, void* Y, int64_t R, int32_t N,
                            int64_t ldy, int32_t flags, void* stream);
int64_t tg_gemm_tn_gather3_workspace_floats(int64_t R, int32_t M);
int tg_gemm_tn_gather3_bf16(const void* G, const tg_gather3* gs, float* out, float* colsum, float* workspace, int64_t R,
                            int32_t M, int64_t ldg, int32_t accumulate, void* stream);
/* GEMM + bias + dropout + residual + LayerNorm in one pass (out_proj -> norm1, linear2 -> norm2 of
 * nn.TransformerEncoderLayer, fused.py:83-92), d_model = 128:  Z = res + drop(X W^T + bias)  (kept for the backward:
 * tg_ln_bwd(a = Z, b = NULL, db != NULL) is its "z mode"),  OUT = LayerNorm(Z) * gamma + beta,  stats = (mean, rstd). */
int tg_gemm_nt_ln_bf16(const void* X, const void* W, const float* bias, const void* res, const float* gamma,
                       const float* beta, void* Z, void* OUT, float* stats, int64_t R, int32_t K, int64_t ldx, float eps,
                       float p_drop, uint64_t seed, uint32_t rstream, void* stream);

/* ---- the whole column-transformer layer in one kernel per direction (bf16, d_model = dim_feedforward = 128,
 *      4 or 8 heads, 2 <= S <= 32 tokens per table row): torch nn.TransformerEncoderLayer as configured at
 *      src/nn/models/fused.py:83-92,187-196 (post-norm, ReLU, dropout p on attention probabilities / both
 *      sub-layer outputs / the hidden activation) + the tab_norm LayerNorm and residual combine of its call sites
 *      (fused.py:160,164,249; tabgnn.py:219):  out = alpha*x + beta_c*LN_t(enc(x))  (tail = 1)  |  out = enc(x).
 *      tg_encoder_pack builds, once per call, the LDS weight images (wpack, tg_encoder_pack_bytes() bytes) and the
 *      fp32 parameter block (prm, tg_encoder_prm_floats() floats) from the layer's parameters (weights bf16 [out,in]).
 *      tg_encoder_fwd_bf16: x [R,S,128] -> out; z1 / z2 (optional) = the two pre-LayerNorm sums, all a recomputing
 *      backward needs; qkv, scores, attention output, x1 and the hidden activation never reach memory.
 *      rs[4]: dropout streams (attention, norm1, ffn, norm2), the same element indexing as the unfused kernels. ---- */
int64_t tg_encoder_pack_bytes(void);
/* Everything BEHIND the attention of the layer above, for token rows of any length (the tabgnn path's S = 130,
 * src/nn/models/tabgnn.py:127-129,219; the 64-column table's S = 65): z1 = x + drop(o Wo^T + b_o), x1 = LN1(z1),
 * z2 = x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2), out = LN2(z2) (+ tail LayerNorm and alpha x + beta_c (.)), on the flat
 * token stream as R pseudo rows of S tokens (2 <= S <= 32).  o = the attention output computed elsewhere; z1 / z2 / st1
 * optional (st1 [T][2] = LayerNorm-1 mean, rstd).  Same weight pack, dropout streams and masks as tg_encoder_fwd_bf16. */
int tg_encoder_ffn_fwd_bf16(const void* x, const void* o, void* out, void* z1, void* z2, float* st1, const void* wpack,
                            const float* prm, int64_t R, int32_t S, int32_t tail, float alpha, float beta_c, float eps,
                            float p_drop, uint64_t seed, const uint32_t* rs, void* stream);
int64_t tg_encoder_stage_bytes(void);   /* bytes of ONE LDS weight image (a [128,128] tile): the backward packs are n of them */
int64_t tg_encoder_prm_floats(void);
int32_t tg_encoder_fused_supported(int32_t S, int32_t C, int32_t H, int32_t FF);
int tg_encoder_pack(const void* w_in, const void* w_o, const void* w1, const void* w2, const float* b_in,
                    const float* b_o, const float* g1, const float* be1, const float* b1, const float* b2,
                    const float* g2, const float* be2, const float* gt /*NULL: no tail*/, const float* bt, void* wpack,
                    float* prm, void* stream);
int tg_encoder_fwd_bf16(const void* x, void* out, void* z1, void* z2, const void* wpack, const float* prm, int64_t R,
                        int32_t S, int32_t H, int32_t tail, float alpha, float beta_c, float eps, float p_drop,
                        uint64_t seed, const uint32_t* rs /*host [4]*/, void* stream);

/* backward of the fused layer, feed-forward half: everything between d out and d x1, recomputed from (z1, z2).
 * tg_encoder_pack_tiles: stage i = LDS image of a [128,128] bf16 tile (here W1, W2^T, W1^T).  Writes d_x1 and the
 * operands of the two weight-gradient GEMMs (d_y2, h) and (d_hpre, x1) (tg_gemm_tn_bf16 sums the bias gradients).
 * lnp: [tg_encoder_ln_partial_blocks(R, S)][4][128] fp32 partial sums of the LayerNorm parameter gradients
 * (d gamma2, d beta2, d gamma_t, d beta_t), one row per workgroup; tg_encoder_ln_reduce adds them up. */
int tg_encoder_pack_tiles(const void* const* tiles /*host [n]*/, const int32_t* ld /*host [n]*/, int32_t n, void* wpack,
                          void* stream);
int tg_encoder_bwd_ffn_bf16(const void* g, const void* z1, const void* z2, void* dx1, void* dy2, void* hout, void* dhpre,
                            void* x1out, const void* wpack, const float* prm, int64_t R, int32_t S, int32_t tail,
                            float beta_c, float eps, float p_drop, uint64_t seed, const uint32_t* rs, float* lnp,
                            void* stream);
/* The same half with the feed-forward WEIGHT GRADIENTS INSIDE the kernel (autograd of linear1 / linear2 of
 * nn.TransformerEncoderLayer, src/nn/models/fused.py:83-92): writes d_x1 only; d_y2, h, d_hpre, x1 are staged in LDS and
 * contracted over the workgroup's tokens by MFMA into per-workgroup fp32 partials — dwp [nblk][2][128][128] (dW1 | dW2,
 * [out][in]) and dbp [nblk][2][128] (db1 | db2), nblk = tg_encoder_dw_blocks(R, S); lnp as above with the same nblk.
 * tg_encoder_dw_reduce sums nw weights' partials in block order (deterministic) into out_w[i] / out_b[i] (NULL = skip;
 * accumulate = 1: += , .grad semantics). */
int64_t tg_encoder_dw_blocks(int64_t R, int32_t S);
int tg_encoder_bwd_ffn_dw_bf16(const void* g, const void* z1, const void* z2, void* dx1, const void* wpack,
                               const float* prm, int64_t R, int32_t S, int32_t tail, float beta_c, float eps,
                               float p_drop, uint64_t seed, const uint32_t* rs, float* lnp, float* dwp, float* dbp,
                               void* stream);
int tg_encoder_dw_reduce(const float* dwp, const float* dbp, int64_t nblk, int32_t nw, float* const* out_w /*host [nw]*/,
                         float* const* out_b /*host [nw]*/, int32_t accumulate, void* stream);
/* backward of the fused layer, attention half (4 or 8 heads): d_x1 -> LayerNorm-1 backward -> output projection
 * backward -> attention backward with q / k / v / probabilities recomputed from x.  Writes dx, the operands of the
 * weight-gradient GEMMs: dy, o [R,S,128] and dqkv [R,S,384], and (lnp, as above) the partial sums of d gamma1, d beta1
 * (rows 2, 3 zero).  w_in_t = W_in^T [128, ld_it >= 384] (bf16): the input gradient of the QKV projection,
 * d_x += d_qkv W_in, is then formed INSIDE the kernel (dx is final; wpack needs 7 stages); NULL: dx is the partial
 * gradient and the caller adds d_qkv W_in with tg_gemm_nt_bf16 (wpack: 4 stages). */
int tg_encoder_bwd_attn_bf16(const void* dx1, const void* z1, const void* x, const void* g /*d out; NULL if alpha == 0*/,
                             void* dx, void* dy, void* o, void* dqkv, const void* w_in /*[384,128]*/,
                             const void* w_o_t /*Wo^T [128, ld_ot]*/, int32_t ld_ot, const void* w_in_t, int32_t ld_it,
                             void* wpack /*4 or 7 x tg_encoder_stage_bytes()*/, const float* prm, int64_t R, int32_t S,
                             int32_t H, float alpha, float eps, float p_drop, uint64_t seed, const uint32_t* rs, float* lnp,
                             void* stream);
int64_t tg_encoder_ln_partial_blocks(int64_t R, int32_t S);
int tg_encoder_ln_reduce(const float* lnp, int64_t nblk, float* const* out /*host [4]: fp32 [128] or NULL*/,
                         int32_t accumulate, void* stream);
/* ---- weight gradient of every Linear on the path (autograd of torch.nn.Linear in the reference):
 *      out[M,N] (fp32) = G[R,M]^T X[R,N], bf16 operands, split over row slabs, deterministic slab sum ------- */
int64_t tg_gemm_tn_workspace_floats(int64_t R, int32_t M, int32_t N);
int tg_gemm_tn_bf16(const void* G, const void* X, float* out, float* colsum /*[M] bias grad, optional*/,
                    float* workspace, int64_t R, int32_t M, int32_t N, int64_t ldg, int64_t ldx,
                    int32_t accumulate /*1: out += , colsum += (gradient accumulation, .grad semantics)*/,
                    void* stream);

/* ---- train-step tail (main.py:70-75,335-336) ------------------------------------------------------------ */
int tg_weighted_ce_fwd(const void* logits, const int64_t* y, const float* w, int64_t B, int32_t K,
                       float* lossden /*[2]*/, float* partials /*[512]*/, int32_t dt, void* stream);
int tg_weighted_ce_bwd(const void* logits, const int64_t* y, const float* w, const float* lossden, const float* gloss,
                       int64_t B, int32_t K, void* dlogits, int32_t dt, void* stream);
int tg_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                 float eps, int32_t t, float grad_scale, int32_t zero_grad, void* stream);
int tg_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- device-resident step state: what lets one captured HIP graph serve every step of a shape bucket --------------
 * (the reference's loop, main.py:41-75, re-launches ~1 300 framework kernels per step; at its default --batch_size 200
 * the step is bound by that launch work, not by the GPU)
 * `state` = 4 x uint64 in device memory: [0] dropout seed word, [1] optimiser step count t, [2] two floats
 * (lr/(1-beta1^t), 1/sqrt(1-beta2^t)), [3] reserved.  tg_advance_step: one single-thread kernel moves the record to
 * the next step (seed <- LCG(seed) with bit 63 clear, t += 1, Adam's bias corrections for the new t).  Kernels of the
 * step take the seed word through seed = TG_SEED_DEVICE(state): nothing is shared between two records. */
int tg_advance_step(uint64_t* state, float lr, float beta1, float beta2, void* stream);
/* tg_adam_step with the step-dependent scalars read from `state` (written by tg_advance_step in the same stream) */
int tg_adam_step_dev(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float beta1, float beta2,
                     float eps, const uint64_t* state, float grad_scale, int32_t zero_grad, void* stream);
/* stream-ordered zero fill by a kernel (bytes % 4 == 0): the package never records a library memset node */
int tg_zero(void* p, int64_t bytes, void* stream);
/* one launch: transposed bf16 copies of all 2-D parameters (table int64 [n][3] = element offset, rows, cols inside
 * src/dst) — the input-gradient GEMMs read W^T as their row-major weight */
int tg_transpose_batched_bf16(const void* src, void* dst, const int64_t* table, int32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TABGNN_HIP_H_ */
