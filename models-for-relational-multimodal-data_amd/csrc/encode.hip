// Per-stype column encoders: raw table rows -> [R, ncols, C] column embeddings, all stypes in ONE
// kernel (the reference runs one module per stype and a Python loop per categorical column;
// pytorch-frame fork EmbeddingEncoder / LinearEncoder / TimestampEncoder / ProjectionEncoder,
// constructed at src/datasets/ibm_transactions_for_aml.py:283-294,313-319, called at utils.py:357-359;
// semantics restated in oracle/encoders.py).
//
// Forward is output-write bound: R*ncols*C*b bytes out, R*(nc*8 + nn*4 + nt*56) bytes in, tables L2-resident.
// Backward reduces [R, ncols, C] gradients into tiny parameters: every accumulator (weight/bias element,
// small-table element) is owned by exactly one thread of a block and lives in LDS, blocks write partial
// vectors, a second kernel sums them in block order -> deterministic, no float atomics
// (tables with more than ENC_SMALL_TABLE rows fall back to global atomics).
#include "common.hpp"
#include "../../include/tabgnn_hip.h"

namespace tg {

constexpr int ENC_MAX_COLS = 16;   // columns per launch (host splits wider tables into several launches)
constexpr int ENC_RCH = 16;        // rows per chunk
constexpr int TS_F = 7, TS_O = 8, TS_K = TS_F * TS_O;  // 7 calendar fields x out_size 8
constexpr int ENC_SMALL_TABLE = 64;

enum EncKind : int { ENC_NUM = 0, ENC_CAT = 1, ENC_TS = 2, ENC_REL = 3 };

struct EncCol {
  int kind;
  int out_col;     // position in the output [R, ncols, C]
  int src_col;     // column inside its stype tensor
  int rows;        // categorical: table rows (card + 1)
  int tab_off;     // categorical: first row of this column's table in the concatenated table
  int acc_off;     // backward: offset (floats) of this column's accumulators in the per-block vector; -1 = global atomics
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
};

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

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_encode_fwd(EncDesc d, EncPtrs p, T* __restrict__ out, long long R, int ncols,
                                                     int C) {
  extern __shared__ float feats[];  // [ENC_RCH][nts][56]
  const int vpr = C / VEC;
  for (long long r0 = (long long)blockIdx.x * ENC_RCH; r0 < R; r0 += (long long)gridDim.x * ENC_RCH) {
    int nrows = (int)((R - r0) < ENC_RCH ? (R - r0) : ENC_RCH);
    if (d.nts > 0) {
      __syncthreads();
      for (int it = threadIdx.x; it < nrows * d.ncol * TS_F; it += 256) {
        int field = it % TS_F, ci = (it / TS_F) % d.ncol, rr = it / (TS_F * d.ncol);
        const EncCol& c = d.col[ci];
        if (c.kind == ENC_TS)
          ts_features(p.ts + ((r0 + rr) * p.nt + c.src_col) * TS_F, p.ts_min_year[c.src_col],
                      feats + (rr * d.nts + c.ts_slot) * TS_K, field);
      }
      __syncthreads();
    }
    int items = nrows * d.ncol * vpr;
    for (int it = threadIdx.x; it < items; it += 256) {
      int cv = (it % vpr) * VEC, ci = (it / vpr) % d.ncol, rr = it / (vpr * d.ncol);
      const EncCol& c = d.col[ci];
      long long r = r0 + rr;
      float o[VEC];
      if (c.kind == ENC_NUM) {
        float z = (p.num[r * p.nn + c.src_col] - p.num_mean[c.src_col]) / p.num_std[c.src_col];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = z * p.num_w[c.src_col * C + cv + j] + p.num_b[c.src_col * C + cv + j];
      } else if (c.kind == ENC_REL) {
        float z = p.rel[r * p.nr + c.src_col];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = z * p.rel_w[c.src_col * C + cv + j] + p.rel_b[c.src_col * C + cv + j];
      } else if (c.kind == ENC_CAT) {
        long long idx = p.cat[r * p.nc + c.src_col] + 1;   // NaN index -1 -> padding row 0
        idx = idx < 0 ? 0 : (idx >= c.rows ? c.rows - 1 : idx);
        const float* row = p.cat_table + ((long long)c.tab_off + idx) * C + cv;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = row[j];
      } else {
        const float* f = feats + (rr * d.nts + c.ts_slot) * TS_K;
        const float* w = p.ts_w + (long long)c.src_col * TS_K * C + cv;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = p.ts_b[c.src_col * C + cv + j];
        for (int k = 0; k < TS_K; ++k) {
          float fk = f[k];
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] += fk * w[(long long)k * C + j];
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = isnan(o[j]) ? 0.f : o[j];   // nan_to_num(nan=0)
      storev<T, VEC>(out + (r * ncols + c.out_col) * C + cv, o);
    }
  }
}

// accumulator layout of one column (floats): NUM/REL: w[C], b[C];  TS: w[56*C], b[C];  CAT(small): table[rows*C]
template <typename T>
__global__ void __launch_bounds__(256) k_encode_bwd(EncDesc d, EncPtrs p, const T* __restrict__ g, long long R,
                                                     int ncols, int C, int acc_floats, float* __restrict__ partials,
                                                     float* __restrict__ big_table_grad) {
  extern __shared__ float lds[];
  float* acc = lds;                 // [acc_floats]
  float* feats = lds + acc_floats;  // [ENC_RCH][nts][56]
  for (int i = threadIdx.x; i < acc_floats; i += 256) acc[i] = 0.f;
  __syncthreads();
  for (long long r0 = (long long)blockIdx.x * ENC_RCH; r0 < R; r0 += (long long)gridDim.x * ENC_RCH) {
    int nrows = (int)((R - r0) < ENC_RCH ? (R - r0) : ENC_RCH);
    if (d.nts > 0) {
      __syncthreads();
      for (int it = threadIdx.x; it < nrows * d.ncol * TS_F; it += 256) {
        int field = it % TS_F, ci = (it / TS_F) % d.ncol, rr = it / (TS_F * d.ncol);
        const EncCol& c = d.col[ci];
        if (c.kind == ENC_TS)
          ts_features(p.ts + ((r0 + rr) * p.nt + c.src_col) * TS_F, p.ts_min_year[c.src_col],
                      feats + (rr * d.nts + c.ts_slot) * TS_K, field);
      }
      __syncthreads();
    }
    // each (column, channel) pair is owned by one thread of the block
    for (int pr = threadIdx.x; pr < d.ncol * C; pr += 256) {
      int ch = pr % C, ci = pr / C;
      const EncCol& c = d.col[ci];
      const T* gp = g + (r0 * ncols + c.out_col) * C + ch;
      if (c.kind == ENC_NUM || c.kind == ENC_REL) {
        float aw = 0.f, ab = 0.f;
        for (int rr = 0; rr < nrows; ++rr) {
          float z = c.kind == ENC_NUM
                        ? (p.num[(r0 + rr) * p.nn + c.src_col] - p.num_mean[c.src_col]) / p.num_std[c.src_col]
                        : p.rel[(r0 + rr) * p.nr + c.src_col];
          float gv = to_f<T>(gp[(long long)rr * ncols * C]);
          if (!isnan(z)) { aw += gv * z; ab += gv; }
        }
        acc[c.acc_off + ch] += aw;
        acc[c.acc_off + C + ch] += ab;
      } else if (c.kind == ENC_CAT) {
        for (int rr = 0; rr < nrows; ++rr) {
          long long idx = p.cat[(r0 + rr) * p.nc + c.src_col] + 1;
          idx = idx < 0 ? 0 : (idx >= c.rows ? c.rows - 1 : idx);
          if (idx == 0) continue;  // padding_idx row receives no gradient
          float gv = to_f<T>(gp[(long long)rr * ncols * C]);
          if (c.acc_off >= 0) acc[c.acc_off + (int)idx * C + ch] += gv;
          else atomicAdd(big_table_grad + ((long long)c.tab_off + idx) * C + ch, gv);
        }
      } else {
        float aw[TS_K], ab = 0.f;
#pragma unroll
        for (int k = 0; k < TS_K; ++k) aw[k] = 0.f;
        for (int rr = 0; rr < nrows; ++rr) {
          float gv = to_f<T>(gp[(long long)rr * ncols * C]);
          const float* f = feats + (rr * d.nts + c.ts_slot) * TS_K;
          ab += gv;
#pragma unroll
          for (int k = 0; k < TS_K; ++k) aw[k] += gv * f[k];
        }
#pragma unroll
        for (int k = 0; k < TS_K; ++k) acc[c.acc_off + k * C + ch] += aw[k];
        acc[c.acc_off + TS_K * C + ch] += ab;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < acc_floats; i += 256) partials[(long long)blockIdx.x * acc_floats + i] = acc[i];
}

__global__ void k_enc_reduce(const float* __restrict__ partials, int nblk, int width, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= width) return;
  float t = 0.f;
  for (int b = 0; b < nblk; ++b) t += partials[(long long)b * width + i];
  out[i] = t;
}

}  // namespace tg

using namespace tg;

// Host-side description of one launch, filled by the Python host (ctypes.Structure mirrors of these).
static_assert(sizeof(EncCol) == 32, "EncCol layout is part of the C ABI");

extern "C" int tg_encode_max_cols(void) { return ENC_MAX_COLS; }
extern "C" int tg_encode_small_table_rows(void) { return ENC_SMALL_TABLE; }
extern "C" int tg_encode_bwd_blocks(void) { return 256; }

static int check_desc(const EncDesc* d, const char* who) {
  TG_CHECK(d && d->ncol > 0 && d->ncol <= ENC_MAX_COLS, "%s: ncol out of range", who);
  return 0;
}

extern "C" int tg_encode_fwd(const void* desc, const void* ptrs, void* out, int64_t R, int32_t ncols, int32_t C,
                             int32_t dt, void* stream) {
  const EncDesc* d = (const EncDesc*)desc;
  const EncPtrs* p = (const EncPtrs*)ptrs;
  if (check_desc(d, "tg_encode_fwd")) return 1;
  TG_CHECK(C % 8 == 0, "tg_encode_fwd: C must be a multiple of 8 (C=%d)", C);
  if (R == 0) return 0;
  size_t shm = (size_t)ENC_RCH * (d->nts > 0 ? d->nts : 1) * TS_K * sizeof(float);
  int grid = grid_cap(ceil_div(R, ENC_RCH), 256 * 8);
  if (dt == F32)
    hipLaunchKernelGGL((k_encode_fwd<float, 4>), dim3(grid), dim3(256), shm, (hipStream_t)stream, *d, *p, (float*)out,
                       (long long)R, ncols, C);
  else
    hipLaunchKernelGGL((k_encode_fwd<bf16_t, 8>), dim3(grid), dim3(256), shm, (hipStream_t)stream, *d, *p,
                       (bf16_t*)out, (long long)R, ncols, C);
  TG_LAUNCH_CHECK();
  return 0;
}

// dflat [acc_floats]: reduced accumulators in the layout given by EncCol.acc_off; partials [256*acc_floats].
extern "C" int tg_encode_bwd(const void* desc, const void* ptrs, const void* g, int64_t R, int32_t ncols, int32_t C,
                             int32_t acc_floats, float* dflat, float* partials, float* big_table_grad, int32_t dt,
                             void* stream) {
  const EncDesc* d = (const EncDesc*)desc;
  const EncPtrs* p = (const EncPtrs*)ptrs;
  if (check_desc(d, "tg_encode_bwd")) return 1;
  hipStream_t st = (hipStream_t)stream;
  size_t shm = ((size_t)acc_floats + (size_t)ENC_RCH * (d->nts > 0 ? d->nts : 1) * TS_K) * sizeof(float);
  TG_CHECK(shm <= 150 * 1024, "tg_encode_bwd: column group needs %zu B of LDS (> 150 KiB); split the launch", shm);
  if (R == 0) {
    (void)hipMemsetAsync(dflat, 0, (size_t)acc_floats * sizeof(float), st);
    return 0;
  }
  int grid = grid_cap(ceil_div(R, ENC_RCH), 256);
  if (dt == F32) {
    (void)hipFuncSetAttribute((const void*)k_encode_bwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_encode_bwd<float>), dim3(grid), dim3(256), shm, st, *d, *p, (const float*)g, (long long)R,
                       ncols, C, acc_floats, partials, big_table_grad);
  } else {
    (void)hipFuncSetAttribute((const void*)k_encode_bwd<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_encode_bwd<bf16_t>), dim3(grid), dim3(256), shm, st, *d, *p, (const bf16_t*)g, (long long)R,
                       ncols, C, acc_floats, partials, big_table_grad);
  }
  hipLaunchKernelGGL(k_enc_reduce, dim3(ceil_div(acc_floats, 256)), dim3(256), 0, st, partials, grid, acc_floats, dflat);
  TG_LAUNCH_CHECK();
  return 0;
}
