// Per-stype column encoders: raw table rows -> [R, ncols, C] column embeddings (the reference runs one module per
// stype and a Python loop per categorical column; pytorch-frame fork EmbeddingEncoder / LinearEncoder /
// TimestampEncoder / ProjectionEncoder, constructed at src/datasets/ibm_transactions_for_aml.py:283-294,313-319,
// called at utils.py:357-359; semantics restated in oracle/encoders.py).
//
// Forward is output-write bound: R*ncols*C*b bytes out, R*(nc*8 + nn*4 + nt*56) bytes in, tables L2-resident.
//   * numerical / categorical / relation columns: one generic kernel, 16-byte stores;
//   * timestamp columns: a [R,56] x [56,C] contraction per column.  Each thread owns two output channels and keeps
//     their 56x2 weights in registers; the 56 sin/cos features of a row chunk are staged once in LDS and
//     broadcast-read, so the weights never leave the register file.
// Backward reduces [R, ncols, C] gradients into tiny parameters without float atomics: every accumulator is
// owned by exactly one thread (register or LDS), blocks write partial vectors, a second kernel sums them in
// block order (tables with more than ENC_SMALL_TABLE rows fall back to global atomics).
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

constexpr int ENC_MAX_COLS = 16;   // columns per launch (host splits wider tables into several launches)
constexpr int ENC_RCH = 16;        // rows per chunk (generic kernels)
constexpr int TS_RCH = 32;         // rows per chunk (timestamp kernels)
constexpr int TS_F = 7, TS_O = 8, TS_K = TS_F * TS_O;  // 7 calendar fields x out_size 8
constexpr int ENC_SMALL_TABLE = 64;
constexpr int ENC_BWD_BLOCKS = 1024;

enum EncKind : int { ENC_NUM = 0, ENC_CAT = 1, ENC_TS = 2, ENC_REL = 3 };

struct EncCol {
  int kind;
  int out_col;     // position in the output [R, ncols, C]
  int src_col;     // column inside its stype tensor
  int rows;        // categorical: table rows (card + 1)
  int tab_off;     // categorical: first row of this column's table in the concatenated table
  int acc_off;     // backward: offset (floats) of this column's accumulators in the reduced vector; -1 = global atomics
  int ts_slot;     // timestamp: index among the timestamp columns of this launch
  int pad;
};
struct EncDesc {
  int ncol, nts;
  EncCol col[ENC_MAX_COLS];
};

struct EncPtrs {
  const float* num; int nn;                     // [R, nn]
  const long long* cat; int nc;                 // [R, nc]
  const long long* ts; int nt;                  // [R, nt, 7]
  const float* rel; int nr;                     // [R, nr]
  const float *num_mean, *num_std, *num_w, *num_b;   // [nn], [nn], [nn,C], [nn,C]
  const float* cat_table;                       // [sum rows, C]
  const float *ts_min_year, *ts_w, *ts_b;       // [nt], [nt,56,C], [nt,C]
  const float *rel_w, *rel_b;                   // [nr,C]
  const long long* row_ids;                     // [R] or NULL: output row r reads raw-table row row_ids[r] (batch = ids into the
                                                // HBM-resident table; tg_enc_ptrs.row_ids)
};
// raw-table row of output row r
__device__ __forceinline__ long long enc_src_row(const long long* __restrict__ row_ids, long long r) {
  return row_ids ? row_ids[r] : r;
}

__device__ __forceinline__ void ts_features(const long long* t7, float min_year, float* f /*[56]*/, int field) {
  // field 0: year -> sinusoidal positional encoding; fields 1..6: value / {12,31,7,24,60,60} -> cyclic encoding
  const float div[6] = {12.f, 31.f, 7.f, 24.f, 60.f, 60.f};
  float v = (float)t7[field];
  float* o = f + field * TS_O;
  if (field == 0) {
    float y = v - min_year;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float mult = powf(1.f / 10000.f, (float)(2 * i) / (float)TS_O);
      float a = y * mult;
      o[i] = sinf(a);
      o[4 + i] = cosf(a);
    }
  } else {
    float x = v / div[field - 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float b = x * (float)(i + 1);
      o[i] = sinf(b * 3.14159265358979323846f);
      o[4 + i] = cosf(b * 2.f * 3.14159265358979323846f);
    }
  }
}

// ------------------------------------------------------------------ generic columns (num / cat / rel)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_encode_fwd(EncDesc d, EncPtrs p, T* __restrict__ out, long long R, int ncols,
                                                     int C, int ngen /* columns to enumerate: the timestamp columns
                                                     are skipped, or not even enumerated when they come last */) {
  const int vpr = C / VEC;
  for (long long r0 = (long long)blockIdx.x * ENC_RCH; r0 < R; r0 += (long long)gridDim.x * ENC_RCH) {
    int nrows = (int)((R - r0) < ENC_RCH ? (R - r0) : ENC_RCH);
    int items = nrows * ngen * vpr;
    for (int it = threadIdx.x; it < items; it += 256) {
      int cv = (it % vpr) * VEC, ci = (it / vpr) % ngen, rr = it / (vpr * ngen);
      const EncCol& c = d.col[ci];
      if (c.kind == ENC_TS) continue;
      long long r = r0 + rr;
      const long long sr = enc_src_row(p.row_ids, r);
      float o[VEC];
      if (c.kind == ENC_NUM) {
        float z = (p.num[sr * p.nn + c.src_col] - p.num_mean[c.src_col]) / p.num_std[c.src_col];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = z * p.num_w[c.src_col * C + cv + j] + p.num_b[c.src_col * C + cv + j];
      } else if (c.kind == ENC_REL) {
        float z = p.rel[sr * p.nr + c.src_col];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = z * p.rel_w[c.src_col * C + cv + j] + p.rel_b[c.src_col * C + cv + j];
      } else {
        long long idx = p.cat[sr * p.nc + c.src_col] + 1;  // NaN index -1 -> padding row 0
        idx = idx < 0 ? 0 : (idx >= c.rows ? c.rows - 1 : idx);
        const float* row = p.cat_table + ((long long)c.tab_off + idx) * C + cv;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = row[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = isnan(o[j]) ? 0.f : o[j];   // nan_to_num(nan=0)
      storev<T, VEC>(out + (r * ncols + c.out_col) * C + cv, o);
    }
  }
}

// accumulator layout of one column (floats): NUM/REL: w[C], b[C];  TS: w[56*C], b[C];  CAT(small): table[rows*C]
// (timestamp columns are reduced by k_encode_ts_bwd; their slots in this kernel's partial vector stay zero)
template <typename T>
__global__ void __launch_bounds__(256) k_encode_bwd(EncDesc d, EncPtrs p, const T* __restrict__ g, long long R,
                                                     int ncols, int C, int acc_floats, float* __restrict__ partials,
                                                     float* __restrict__ big_table_grad, int groups, int ngen) {
  // acc_floats here = the accumulators THIS kernel owns: when the timestamp columns come last in the descriptor (the
  // host plans them so) their 57*C slots are neither allocated in LDS nor enumerated as work items (ngen = the
  // non-timestamp columns) — with the AML edge table (3 cat + 1 num + 1 ts) that is 21 KB instead of 50 KB of LDS
  // per row group and 128 instead of 160 items, so two row groups fit and all 256 threads work (they were 128).
  extern __shared__ __align__(16) float acc_all[];   // [groups][acc_floats]
  for (int i = threadIdx.x; i < acc_floats * groups; i += 256) acc_all[i] = 0.f;
  __syncthreads();
  const long long rowstride = (long long)ncols * C;
  // each (column, 4-channel group) is owned by one thread: 8-byte (bf16) / 16-byte (fp32) loads — one channel per
  // lane (2-byte loads) ran at 0.76 TB/s, request-bound.  With few columns the block is split into `groups` row
  // groups (each with its own LDS accumulators, summed in group order at the end) so that all 256 threads work.
  constexpr int V = 4;
  const int nitems = ngen * (C / V);
  const int grp = groups > 1 ? threadIdx.x / nitems : 0;
  const int first = groups > 1 ? threadIdx.x % nitems : threadIdx.x;
  const int step = groups > 1 ? nitems : 256;                  // groups > 1  =>  one item per thread
  float* acc = acc_all + (grp < groups ? grp : 0) * acc_floats;
  for (long long r0 = ((long long)blockIdx.x * groups + grp) * ENC_RCH; grp < groups && r0 < R;
       r0 += (long long)gridDim.x * groups * ENC_RCH) {
    int nrows = (int)((R - r0) < ENC_RCH ? (R - r0) : ENC_RCH);
    for (int pr = first; pr < nitems; pr += step) {
      int ch = (pr % (C / V)) * V, ci = pr / (C / V);
      const EncCol& c = d.col[ci];
      if (c.kind == ENC_TS) continue;
      const T* gp = g + (r0 * ncols + c.out_col) * C + ch;
      float gv[ENC_RCH][V];
      // loads are UNCONDITIONAL (row index clamped into the chunk) so that all 16 fly together: a per-element
      // "load or zero" on a runtime bound makes hipcc branch and wait around every load (guide §5, item 4c)
#pragma unroll
      for (int rr = 0; rr < ENC_RCH; ++rr) {
        int rc = rr < nrows ? rr : nrows - 1;
        float t[V];
        loadv<T, V>(gp + rc * rowstride, t);
#pragma unroll
        for (int j = 0; j < V; ++j) gv[rr][j] = rr < nrows ? t[j] : 0.f;
      }
      if (c.kind == ENC_NUM || c.kind == ENC_REL) {
        // all 16 raw values are fetched before any arithmetic (no branch between the loads: they fly together)
        const bool isnum = c.kind == ENC_NUM;
        const float* src = isnum ? p.num + c.src_col : p.rel + c.src_col;
        const int sstride = isnum ? p.nn : p.nr;
        float zv[ENC_RCH];
#pragma unroll
        for (int rr = 0; rr < ENC_RCH; ++rr)                                                         // gv is 0 past nrows
          zv[rr] = src[enc_src_row(p.row_ids, r0 + (rr < nrows ? rr : nrows - 1)) * sstride];
        const float mu = isnum ? p.num_mean[c.src_col] : 0.f, sd = isnum ? p.num_std[c.src_col] : 1.f;
        float aw[V], ab[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { aw[j] = 0.f; ab[j] = 0.f; }
#pragma unroll
        for (int rr = 0; rr < ENC_RCH; ++rr) {
          float z = (zv[rr] - mu) / sd;
          bool ok = !isnan(z);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            aw[j] += ok ? gv[rr][j] * z : 0.f;
            ab[j] += ok ? gv[rr][j] : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
          acc[c.acc_off + ch + j] += aw[j];
          acc[c.acc_off + C + ch + j] += ab[j];
        }
      } else {
        const long long* src = p.cat + c.src_col;
        int rowi[ENC_RCH];
#pragma unroll
        for (int rr = 0; rr < ENC_RCH; ++rr) {
          long long idx = src[enc_src_row(p.row_ids, r0 + (rr < nrows ? rr : nrows - 1)) * p.nc] + 1;   // gv is 0 past nrows
          rowi[rr] = (int)(idx < 0 ? 0 : (idx >= c.rows ? c.rows - 1 : idx));
        }
        if (c.acc_off >= 0) {
          // predicated read-modify-write of the thread-owned LDS columns; row 0 (padding_idx) only ever gets += 0
#pragma unroll
          for (int rr = 0; rr < ENC_RCH; ++rr) {
            float4* a4 = reinterpret_cast<float4*>(acc + c.acc_off + rowi[rr] * C + ch);
            float4 a = *a4;
            const bool on = rowi[rr] != 0;
            a.x += on ? gv[rr][0] : 0.f; a.y += on ? gv[rr][1] : 0.f; a.z += on ? gv[rr][2] : 0.f; a.w += on ? gv[rr][3] : 0.f;
            *a4 = a;
          }
        } else if (big_table_grad) {      // (NULL: the caller reduces the big tables by tg_embed_grad_sorted, in a fixed order)
#pragma unroll
          for (int rr = 0; rr < ENC_RCH; ++rr)
            if (rowi[rr] != 0)
#pragma unroll
              for (int j = 0; j < V; ++j)
                atomicAdd(big_table_grad + ((long long)c.tab_off + rowi[rr]) * C + ch + j, gv[rr][j]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < acc_floats; i += 256) {
    float t = acc_all[i];
    for (int gq = 1; gq < groups; ++gq) t += acc_all[gq * acc_floats + i];
    partials[(long long)blockIdx.x * acc_floats + i] = t;
  }
}

// ------------------------------------------------------------------ timestamp columns
// thread = (row lane rl, channel pair cg); 256 threads = RL x (C/2)
template <typename T>
__global__ void __launch_bounds__(256) k_encode_ts_fwd(const long long* __restrict__ ts, int nt, int src_col,
                                                        const float* __restrict__ min_year_p, const float* __restrict__ w /*[56,C]*/,
                                                        const float* __restrict__ bias /*[C]*/, T* __restrict__ out,
                                                        long long R, int ncols, int out_col, int C,
                                                        const long long* __restrict__ row_ids) {
  __shared__ __attribute__((aligned(16))) float feats[TS_RCH * TS_K];
  const int ngrp = C / 2, cg = threadIdx.x % ngrp, rl = threadIdx.x / ngrp, RL = 256 / ngrp;
  const int c0 = cg * 2;
  const float min_year = min_year_p[0];
  float w0[TS_K], w1[TS_K];
#pragma unroll
  for (int k = 0; k < TS_K; ++k) { w0[k] = w[k * C + c0]; w1[k] = w[k * C + c0 + 1]; }
  const float b0 = bias[c0], b1 = bias[c0 + 1];
  for (long long r0 = (long long)blockIdx.x * TS_RCH; r0 < R; r0 += (long long)gridDim.x * TS_RCH) {
    int nrows = (int)((R - r0) < TS_RCH ? (R - r0) : TS_RCH);
    __syncthreads();
    if (threadIdx.x < nrows * TS_F) {
      int rr = threadIdx.x / TS_F, field = threadIdx.x % TS_F;
      ts_features(ts + (enc_src_row(row_ids, r0 + rr) * nt + src_col) * TS_F, min_year, feats + rr * TS_K, field);
    }
    __syncthreads();
    for (int rr = rl; rr < nrows; rr += RL) {
      const float4* f4 = reinterpret_cast<const float4*>(feats + rr * TS_K);
      float a0 = b0, a1 = b1;
#pragma unroll
      for (int k4 = 0; k4 < TS_K / 4; ++k4) {
        float4 f = f4[k4];
        a0 += f.x * w0[4 * k4] + f.y * w0[4 * k4 + 1] + f.z * w0[4 * k4 + 2] + f.w * w0[4 * k4 + 3];
        a1 += f.x * w1[4 * k4] + f.y * w1[4 * k4 + 1] + f.z * w1[4 * k4 + 2] + f.w * w1[4 * k4 + 3];
      }
      a0 = isnan(a0) ? 0.f : a0;
      a1 = isnan(a1) ? 0.f : a1;
      T* o = out + ((r0 + rr) * ncols + out_col) * C + c0;
      o[0] = from_f<T>(a0);
      o[1] = from_f<T>(a1);
    }
  }
}

// The 56 calendar features of a timestamp column as a bf16 GEMM operand [R,128]: columns 0..55 the features, column 56
// the constant 1 (so the bias rides in the weight matrix), the rest 0.  With it the timestamp encoder is
// tg_gemm_nt_bf16(feats, [W | b | 0]) forward and tg_gemm_tn_bf16(g, feats) backward — MFMA at the HBM rate instead of
// 56 x C multiply-adds per row on the vector ALU (180 / 259 us -> ~60 us each on the 430 k-row edge table).
__global__ void __launch_bounds__(256) k_encode_ts_feats(const long long* __restrict__ ts, int nt, int src_col,
                                                          const float* __restrict__ min_year_p,
                                                          const long long* __restrict__ row_ids,
                                                          unsigned short* __restrict__ out, long long R,
                                                          const float* __restrict__ w /*[56,C] or NULL*/,
                                                          const float* __restrict__ b /*[C]*/,
                                                          unsigned short* __restrict__ wext /*[C,128]*/, int C) {
  __shared__ __attribute__((aligned(16))) float feats[TS_RCH * TS_K];
  if (w != nullptr && blockIdx.x == gridDim.x - 1) {       // the extra last block packs the GEMM weight [W | b | 0] (bf16)
    for (int i = threadIdx.x; i < C * 128; i += 256) {
      const int c = i >> 7, k = i & 127;
      const float v = k < TS_K ? w[(long long)k * C + c] : (k == TS_K ? b[c] : 0.f);
      wext[i] = f2bf(v);
    }
    return;
  }
  const float min_year = min_year_p[0];
  const int nrow_blocks = w != nullptr ? (int)gridDim.x - 1 : (int)gridDim.x;
  for (long long r0 = (long long)blockIdx.x * TS_RCH; r0 < R; r0 += (long long)nrow_blocks * TS_RCH) {
    const int nrows = (int)((R - r0) < TS_RCH ? (R - r0) : TS_RCH);
    __syncthreads();
    if (threadIdx.x < nrows * TS_F) {
      const int rr = threadIdx.x / TS_F, field = threadIdx.x % TS_F;
      ts_features(ts + (enc_src_row(row_ids, r0 + rr) * nt + src_col) * TS_F, min_year, feats + rr * TS_K, field);
    }
    __syncthreads();
    const int rr = threadIdx.x >> 3, seg = threadIdx.x & 7;            // 32 rows x 8 segments of 16 columns
    if (rr < nrows) {
      unsigned w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c0 = seg * 16 + 2 * j;
        float a = c0 < TS_K ? feats[rr * TS_K + c0] : (c0 == TS_K ? 1.f : 0.f);
        float b = c0 + 1 < TS_K ? feats[rr * TS_K + c0 + 1] : (c0 + 1 == TS_K ? 1.f : 0.f);
        a = isnan(a) ? 0.f : a;
        b = isnan(b) ? 0.f : b;
        w[j] = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
      }
      uint4* o = reinterpret_cast<uint4*>(out + (r0 + rr) * 128 + seg * 16);
      o[0] = make_uint4(w[0], w[1], w[2], w[3]);
      o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// partial[blk][57*C] = (dW[56][C], db[C]) of this block's rows
template <typename T>
__global__ void __launch_bounds__(256) k_encode_ts_bwd(const long long* __restrict__ ts, int nt, int src_col,
                                                        const float* __restrict__ min_year_p, const T* __restrict__ g, long long R, int ncols,
                                                        int out_col, int C, float* __restrict__ partials,
                                                        const long long* __restrict__ row_ids) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // feats [TS_RCH*56] | acc [57*C]
  float* feats = sm;
  float* acc = sm + TS_RCH * TS_K;
  const int ngrp = C / 2, cg = threadIdx.x % ngrp, rl = threadIdx.x / ngrp, RL = 256 / ngrp;
  const int c0 = cg * 2;
  const float min_year = min_year_p[0];
  float a0[TS_K], a1[TS_K], s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k = 0; k < TS_K; ++k) { a0[k] = 0.f; a1[k] = 0.f; }
  for (long long r0 = (long long)blockIdx.x * TS_RCH; r0 < R; r0 += (long long)gridDim.x * TS_RCH) {
    int nrows = (int)((R - r0) < TS_RCH ? (R - r0) : TS_RCH);
    __syncthreads();
    if (threadIdx.x < nrows * TS_F) {
      int rr = threadIdx.x / TS_F, field = threadIdx.x % TS_F;
      ts_features(ts + (enc_src_row(row_ids, r0 + rr) * nt + src_col) * TS_F, min_year, feats + rr * TS_K, field);
    }
    __syncthreads();
    for (int rr = rl; rr < nrows; rr += RL) {
      const T* gp = g + ((r0 + rr) * ncols + out_col) * C + c0;
      float g0 = to_f<T>(gp[0]), g1 = to_f<T>(gp[1]);
      const float4* f4 = reinterpret_cast<const float4*>(feats + rr * TS_K);
      s0 += g0;
      s1 += g1;
#pragma unroll
      for (int k4 = 0; k4 < TS_K / 4; ++k4) {
        float4 f = f4[k4];
        a0[4 * k4] += g0 * f.x; a0[4 * k4 + 1] += g0 * f.y; a0[4 * k4 + 2] += g0 * f.z; a0[4 * k4 + 3] += g0 * f.w;
        a1[4 * k4] += g1 * f.x; a1[4 * k4 + 1] += g1 * f.y; a1[4 * k4 + 2] += g1 * f.z; a1[4 * k4 + 3] += g1 * f.w;
      }
    }
  }
  // row lanes add their registers into the block vector one after another (fixed order)
  for (int turn = 0; turn < RL; ++turn) {
    __syncthreads();
    if (rl == turn) {
#pragma unroll
      for (int k = 0; k < TS_K; ++k) {
        if (turn == 0) { acc[k * C + c0] = a0[k]; acc[k * C + c0 + 1] = a1[k]; }
        else { acc[k * C + c0] += a0[k]; acc[k * C + c0 + 1] += a1[k]; }
      }
      if (turn == 0) { acc[TS_K * C + c0] = s0; acc[TS_K * C + c0 + 1] = s1; }
      else { acc[TS_K * C + c0] += s0; acc[TS_K * C + c0 + 1] += s1; }
    }
  }
  __syncthreads();
  const int width = (TS_K + 1) * C;
  for (int i = threadIdx.x; i < width; i += 256) partials[(long long)blockIdx.x * width + i] = acc[i];
}

// 256 threads = 64 columns x 4 strips of blocks; strips meet in LDS in strip order
__global__ void __launch_bounds__(256) k_enc_reduce(const float* __restrict__ partials, int nblk, int width,
                                                     float* __restrict__ out, const int* __restrict__ skip_lo,
                                                     int nskip) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float t = 0.f;
  if (col < width) {
    int per = (nblk + 3) / 4, b0 = strip * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      t0 += partials[(long long)b * width + col];
      t1 += partials[(long long)(b + 1) * width + col];
      t2 += partials[(long long)(b + 2) * width + col];
      t3 += partials[(long long)(b + 3) * width + col];
    }
    for (; b < b1; ++b) t0 += partials[(long long)b * width + col];
    t = (t0 + t1) + (t2 + t3);
  }
  red[strip][lane] = t;
  __syncthreads();
  if (strip == 0 && col < width) out[col] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// dst_k[0..len_k) += src[off_k .. off_k+len_k) for every segment k of a table (dst pointer, src offset, length):
// the reduced encoder gradient vector is added into the encoder parameters' gradient buffers in ONE launch
// (instead of a zero-fill, a slice copy and an autograd add per parameter column).
__global__ void __launch_bounds__(256) k_scatter_add_segments(const float* __restrict__ src,
                                                               const long long* __restrict__ table, int nseg) {
  const int k = blockIdx.y;
  if (k >= nseg) return;
  float* __restrict__ dst = reinterpret_cast<float*>(table[3 * k]);
  const long long off = table[3 * k + 1], len = table[3 * k + 2];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
    dst[i] += src[off + i];
}

}  // namespace tg

using namespace tg;

static_assert(sizeof(EncCol) == 32, "EncCol layout is part of the C ABI");

extern "C" int tg_encode_max_cols(void) { return ENC_MAX_COLS; }
extern "C" int tg_encode_small_table_rows(void) { return ENC_SMALL_TABLE; }
extern "C" int tg_encode_bwd_blocks(void) { return ENC_BWD_BLOCKS; }

static int check_desc(const EncDesc* d, int C, const char* who) {
  TG_CHECK(d && d->ncol > 0 && d->ncol <= ENC_MAX_COLS, "%s: ncol out of range", who);
  TG_CHECK(C % 8 == 0, "%s: C must be a multiple of 8 (C=%d)", who, C);
  if (d->nts > 0) TG_CHECK(256 % (C / 2) == 0 && C <= 512, "%s: timestamp columns need C/2 | 256 (C=%d)", who, C);
  return 0;
}

// number of leading non-timestamp columns when ALL timestamp columns come last in the descriptor (the host plans them
// so), else every column (the generic kernels then skip the timestamp ones item by item)
static int enc_ngen(const tg::EncDesc* d) {
  int first_ts = -1;
  for (int i = 0; i < d->ncol; ++i) {
    if (d->col[i].kind == tg::ENC_TS) { if (first_ts < 0) first_ts = i; }
    else if (first_ts >= 0) return d->ncol;
  }
  return first_ts < 0 ? d->ncol : first_ts;
}

extern "C" int tg_encode_fwd(const void* desc, const void* ptrs, void* out, int64_t R, int32_t ncols, int32_t C,
                             int32_t dt, void* stream) {
  const EncDesc* d = (const EncDesc*)desc;
  const EncPtrs* p = (const EncPtrs*)ptrs;
  if (check_desc(d, C, "tg_encode_fwd")) return 1;
  if (R == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (d->nts < d->ncol) {
    int grid = grid_cap(ceil_div(R, ENC_RCH), 256 * 8);
    if (dt == F32)
      hipLaunchKernelGGL((k_encode_fwd<float, 4>), dim3(grid), dim3(256), 0, st, *d, *p, (float*)out, (long long)R,
                         ncols, C, enc_ngen(d));
    else
      hipLaunchKernelGGL((k_encode_fwd<bf16_t, 8>), dim3(grid), dim3(256), 0, st, *d, *p, (bf16_t*)out, (long long)R,
                         ncols, C, enc_ngen(d));
  }
  for (int i = 0; i < d->ncol; ++i) {
    const EncCol& c = d->col[i];
    if (c.kind != ENC_TS) continue;
    TG_CHECK(p->ts && p->ts_w && p->ts_b && p->ts_min_year, "tg_encode_fwd: timestamp pointers missing");
    int grid = grid_cap(ceil_div(R, TS_RCH), 256 * 8);
    const float* w = p->ts_w + (long long)c.src_col * TS_K * C;
    const float* b = p->ts_b + (long long)c.src_col * C;
    const float* min_year_host = p->ts_min_year + c.src_col;   // device pointer
    if (dt == F32)
      hipLaunchKernelGGL((k_encode_ts_fwd<float>), dim3(grid), dim3(256), 0, st, (const long long*)p->ts, p->nt,
                         c.src_col, min_year_host, w, b, (float*)out, (long long)R, ncols, c.out_col, C, p->row_ids);
    else
      hipLaunchKernelGGL((k_encode_ts_fwd<bf16_t>), dim3(grid), dim3(256), 0, st, (const long long*)p->ts, p->nt,
                         c.src_col, min_year_host, w, b, (bf16_t*)out, (long long)R, ncols, c.out_col, C, p->row_ids);
  }
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_encode_bwd(const void* desc, const void* ptrs, const void* g, int64_t R, int32_t ncols, int32_t C,
                             int32_t acc_floats, float* dflat, float* partials, float* big_table_grad, int32_t dt,
                             void* stream) {
  const EncDesc* d = (const EncDesc*)desc;
  const EncPtrs* p = (const EncPtrs*)ptrs;
  if (check_desc(d, C, "tg_encode_bwd")) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (R == 0) {
    zero_async(dflat, (size_t)acc_floats * sizeof(float), st);
    return 0;
  }
  if (d->nts < d->ncol) {
    TG_CHECK((size_t)acc_floats * sizeof(float) <= 150 * 1024,
             "tg_encode_bwd: column group needs %zu B of LDS (> 150 KiB); split the launch", (size_t)acc_floats * 4);
    // timestamp columns last in the descriptor => the generic kernel neither allocates nor enumerates them
    const int ngen = enc_ngen(d);
    if (ngen < d->ncol) acc_floats = d->col[ngen].acc_off;         // the prefix of the reduced vector this kernel owns
    const int nitems = ngen * (C / 4);
    int groups = nitems <= 128 ? 256 / nitems : 1;                 // all 256 threads busy when the columns are few
    while (groups > 1 && (size_t)groups * acc_floats * sizeof(float) > 48 * 1024) --groups;
    size_t shm = (size_t)groups * acc_floats * sizeof(float);
    int grid = grid_cap(ceil_div(R, (long long)ENC_RCH * groups), ENC_BWD_BLOCKS);
    if (dt == F32) {
      (void)hipFuncSetAttribute((const void*)k_encode_bwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_encode_bwd<float>), dim3(grid), dim3(256), shm, st, *d, *p, (const float*)g, (long long)R,
                         ncols, C, acc_floats, partials, big_table_grad, groups, ngen);
    } else {
      (void)hipFuncSetAttribute((const void*)k_encode_bwd<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_encode_bwd<bf16_t>), dim3(grid), dim3(256), shm, st, *d, *p, (const bf16_t*)g,
                         (long long)R, ncols, C, acc_floats, partials, big_table_grad, groups, ngen);
    }
    if (acc_floats > 0)
    hipLaunchKernelGGL(k_enc_reduce, dim3(ceil_div(acc_floats, 64)), dim3(256), 0, st, partials, grid, acc_floats,
                       dflat, (const int*)nullptr, 0);
  }
  for (int i = 0; i < d->ncol; ++i) {
    const EncCol& c = d->col[i];
    if (c.kind != ENC_TS) continue;
    const float* min_year_host = p->ts_min_year + c.src_col;   // device pointer
    int width = (TS_K + 1) * C;
    size_t shm = ((size_t)TS_RCH * TS_K + width) * sizeof(float);
    int grid = grid_cap(ceil_div(R, TS_RCH), ENC_BWD_BLOCKS);
    if (dt == F32) {
      (void)hipFuncSetAttribute((const void*)k_encode_ts_bwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_encode_ts_bwd<float>), dim3(grid), dim3(256), shm, st, (const long long*)p->ts, p->nt,
                         c.src_col, min_year_host, (const float*)g, (long long)R, ncols, c.out_col, C, partials, p->row_ids);
    } else {
      (void)hipFuncSetAttribute((const void*)k_encode_ts_bwd<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_encode_ts_bwd<bf16_t>), dim3(grid), dim3(256), shm, st, (const long long*)p->ts, p->nt,
                         c.src_col, min_year_host, (const bf16_t*)g, (long long)R, ncols, c.out_col, C, partials, p->row_ids);
    }
    hipLaunchKernelGGL(k_enc_reduce, dim3(ceil_div(width, 64)), dim3(256), 0, st, partials, grid, width,
                       dflat + c.acc_off, (const int*)nullptr, 0);
  }
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_encode_ts_features(const int64_t* ts, int32_t nt, int32_t src_col, const float* min_year,
                                     const int64_t* row_ids, void* feats, int64_t R, const float* w, const float* b,
                                     void* wext, int32_t C, void* stream) {
  if (R == 0) return 0;
  TG_CHECK(ts && min_year && feats && nt > 0 && src_col >= 0 && src_col < nt, "tg_encode_ts_features: bad argument");
  TG_CHECK((reinterpret_cast<uintptr_t>(feats) & 15) == 0, "tg_encode_ts_features: feats must be 16-byte aligned");
  TG_CHECK(!w || (b && wext && C > 0), "tg_encode_ts_features: the weight pack needs w, b, wext and C");
  const int grid = grid_cap(ceil_div(R, TS_RCH), 256 * 16) + (w ? 1 : 0);
  hipLaunchKernelGGL(k_encode_ts_feats, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long long*)ts, nt, src_col,
                     min_year + src_col, (const long long*)row_ids, (unsigned short*)feats, (long long)R, w, b,
                     (unsigned short*)wext, C);
  TG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ big embedding tables, deterministic (round 5)
// Tables with more rows than the LDS accumulators hold (> tg_encode_small_table_rows) used float atomicAdd: the order of
// the additions — hence the last bits of the gradient — changed from run to run.  Here the batch's (column, category)
// pairs are counting-sorted first (tg_csr_build over keys = bucket base of the column + clamped category, key index
// i = column slot * R + row: a stable sort, rows ascending inside a bucket) and ONE wave sums a bucket's gradient rows in
// that order into the table's gradient row: every element has exactly one writer and one order.
// cols: int64 [ncol][4] on the device = (first bucket of the column, out_col of the column in g's rows, destination
// float* of the column's table gradient (row 0), table rows).  Bucket 0 of a column is padding_idx: no gradient.
struct EmbCol { long long base, out_col, dst, rows; };
template <typename T>
__global__ void __launch_bounds__(256) k_embed_grad_sorted(const T* __restrict__ g, long long gstride, const int* __restrict__ rowptr,
                                                          const int* __restrict__ perm, long long R, const EmbCol* __restrict__ cols,
                                                          int C, int accumulate) {
  const EmbCol c = cols[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* dst = reinterpret_cast<float*>(c.dst);
  const long long slot_base = (long long)blockIdx.y * R;          // key index of row r of this column = slot_base + r
  for (long long cat = 1 + (long long)blockIdx.x * 4 + wave; cat < c.rows; cat += (long long)gridDim.x * 4) {
    const int b0 = rowptr[c.base + cat], b1 = rowptr[c.base + cat + 1];
    for (int ch = 2 * lane; ch < C; ch += 128) {                   // two channels per lane and pass (C = 128: one pass)
      float a0 = 0.f, a1 = 0.f;
      int k = b0;
      for (; k + 4 <= b1; k += 4) {                                // four rows in flight; summed in order
        float v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long r = (long long)perm[k + u] - slot_base;
          float t[2];
          loadv<T, 2>(g + r * gstride + c.out_col * C + ch, t);
          v[u][0] = t[0]; v[u][1] = t[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 += v[u][0]; a1 += v[u][1]; }
      }
      for (; k < b1; ++k) {
        const long long r = (long long)perm[k] - slot_base;
        float t[2];
        loadv<T, 2>(g + r * gstride + c.out_col * C + ch, t);
        a0 += t[0]; a1 += t[1];
      }
      float* o = dst + cat * C + ch;
      if (accumulate) { o[0] += a0; o[1] += a1; }
      else { o[0] = a0; o[1] = a1; }
    }
  }
}

extern "C" int tg_embed_grad_sorted(const void* g, int64_t gstride, const int32_t* rowptr, const int32_t* perm, int64_t R,
                                    const int64_t* cols, int32_t ncol, int32_t max_rows, int32_t C, int32_t accumulate,
                                    int32_t dt, void* stream) {
  if (ncol <= 0 || R <= 0) return 0;
  TG_CHECK(g && rowptr && perm && cols && C > 0 && C % 2 == 0 && max_rows > 0, "tg_embed_grad_sorted: bad argument");
  TG_CHECK((long long)ncol * R <= 2147483647LL, "tg_embed_grad_sorted: %d columns x %lld rows exceed the int32 key index", ncol, (long long)R);
  int bx = (max_rows + 3) / 4;
  if (bx > 2048) bx = 2048;
  if (dt == F32)
    hipLaunchKernelGGL((k_embed_grad_sorted<float>), dim3(bx, ncol), dim3(256), 0, (hipStream_t)stream, (const float*)g,
                       (long long)gstride, rowptr, perm, (long long)R, (const EmbCol*)cols, C, accumulate);
  else
    hipLaunchKernelGGL((k_embed_grad_sorted<bf16_t>), dim3(bx, ncol), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g,
                       (long long)gstride, rowptr, perm, (long long)R, (const EmbCol*)cols, C, accumulate);
  TG_LAUNCH_CHECK();
  return 0;
}

// table: int64 [nseg][3] on the device = (destination float* as integer, offset into src, length); segments must not
// overlap in their destinations (each parameter column is one segment).
extern "C" int tg_scatter_add_segments(const float* src, const int64_t* table, int32_t nseg, int64_t max_len,
                                       void* stream) {
  if (nseg <= 0) return 0;
  TG_CHECK(src && table && max_len > 0, "tg_scatter_add_segments: null argument");
  int bx = (int)((max_len + 255) / 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_scatter_add_segments, dim3(bx, nseg), dim3(256), 0, (hipStream_t)stream, src,
                     (const long long*)table, nseg);
  TG_LAUNCH_CHECK();
  return 0;
}
