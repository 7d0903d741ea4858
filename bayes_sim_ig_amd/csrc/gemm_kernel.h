// The fp32 MFMA GEMM kernel template and its launcher (see gemm_f32.hip for the design).
// Included by gemm_f32.hip (epilogue helpers for its reduce kernel) and by the four
// gemm_tile_*.hip translation units, one per tile shape, so that the 64 kernel variants
// compile in parallel.
#pragma once
#include "gemm.h"

#include <algorithm>
#include <cstdlib>

namespace bsig {


typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int BKP = BK + 4;

__device__ inline float act_fwd(float v, int act) {
  switch (act) {
    case BSIG_ACT_TANH: return tanhf(v);
    case BSIG_ACT_RELU: return v > 0.f ? v : 0.f;
    case BSIG_ACT_LEAKY_RELU: return v > 0.f ? v : 0.01f * v;
    case BSIG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// derivative expressed through the activation OUTPUT h
__device__ inline float act_bwd_from_out(float h, int act) {
  switch (act) {
    case BSIG_ACT_TANH: return 1.f - h * h;
    case BSIG_ACT_RELU: return h > 0.f ? 1.f : 0.f;
    case BSIG_ACT_LEAKY_RELU: return h > 0.f ? 1.f : 0.01f;
    case BSIG_ACT_SIGMOID: return h * (1.f - h);
    default: return 1.f;
  }
}

template <int EPI>
__device__ inline void epilogue_one(const GemmParams& p, int row, int col, float v) {
  float* dst = p.c + (int64_t)row * p.ldc + col;
  if constexpr (EPI == BSIG_EPI_NONE) {
    *dst = v;
  } else if constexpr (EPI == BSIG_EPI_BIAS) {
    *dst = v + p.bias[col];
  } else if constexpr (EPI == BSIG_EPI_BIAS_ACT) {
    *dst = act_fwd(v + p.bias[col], p.act);
  } else if constexpr (EPI == BSIG_EPI_COS_SIN) {
    float sn, cs;
    sincosf(v, &sn, &cs);
    dst[0] = p.alpha * cs;
    dst[p.n] = p.alpha * sn;
  } else if constexpr (EPI == BSIG_EPI_COS_OFF) {
    *dst = p.alpha * cosf(v + p.bias[col]);
  } else {
    *dst = v * act_bwd_from_out(p.aux[(int64_t)row * p.ldaux + col], p.act);
  }
}

__device__ inline void epilogue_store(const GemmParams& p, int row, int col, float v) {
  switch (p.epilogue) {
    case BSIG_EPI_NONE: epilogue_one<BSIG_EPI_NONE>(p, row, col, v); break;
    case BSIG_EPI_BIAS: epilogue_one<BSIG_EPI_BIAS>(p, row, col, v); break;
    case BSIG_EPI_BIAS_ACT: epilogue_one<BSIG_EPI_BIAS_ACT>(p, row, col, v); break;
    case BSIG_EPI_COS_SIN: epilogue_one<BSIG_EPI_COS_SIN>(p, row, col, v); break;
    case BSIG_EPI_COS_OFF: epilogue_one<BSIG_EPI_COS_OFF>(p, row, col, v); break;
    case BSIG_EPI_MUL_DACT: epilogue_one<BSIG_EPI_MUL_DACT>(p, row, col, v); break;
    default: {  // EPI_ADAM: v is the gradient of c[row, col]
      const int64_t e = (int64_t)row * p.ldc + col;
      if (p.grad_out) p.grad_out[e] = v;
      const float m1 = p.adam_m[e] + (v - p.adam_m[e]) * (1.0f - p.beta1);
      const float v1 = p.adam_v[e] * p.beta2 + (1.0f - p.beta2) * v * v;
      p.adam_m[e] = m1;
      p.adam_v[e] = v1;
      p.c[e] = p.c[e] - p.adam_dyn[0] * (m1 / (sqrtf(v1) * p.adam_dyn[1] + p.adam_eps));
      if (col == 0 && p.bias_p) {
        const float g = p.bias_g[row];
        const float bm = p.bias_m[row] + (g - p.bias_m[row]) * (1.0f - p.beta1);
        const float bv = p.bias_v[row] * p.beta2 + (1.0f - p.beta2) * g * g;
        p.bias_m[row] = bm;
        p.bias_v[row] = bv;
        p.bias_p[row] = p.bias_p[row] - p.adam_dyn[0] * (bm / (sqrtf(bv) * p.adam_dyn[1] + p.adam_eps));
      }
      break;
    }
  }
}

// Epilogue of one 32x32 accumulator tile read back from the wave's LDS patch.
// Vector form (tile fully inside N, 16-byte aligned pitches): lane l owns the
// four consecutive columns 4*(l&7).. of rows (l>>3) + 8*it, it = 0..3 — every
// global access is a 16-byte access, eight lanes cover one 128-byte row segment.
// Scalar form (edge tiles / unaligned outputs): lane (h, l31) owns column l31 of
// rows h + 2*it, it = 0..15.  In both, all operand loads are issued before the
// first store so that they overlap (the pointers may alias for the compiler).
struct F4 { float v[4]; };
__device__ inline F4 ld4(const float* p) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  return F4{{q.x, q.y, q.z, q.w}};
}
__device__ inline void st4(float* p, const F4& a) {
  *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}

template <int EPI>
__device__ inline void tile_epilogue_vec(const GemmParams& p, const float* __restrict__ patch,
                                         int row0, int col0, int lane, float& exp_acc,
                                         float adam_ss, float adam_ib) {
  const int r8 = lane >> 3, c4 = (lane & 7) * 4;
  const int col = col0 + c4;
  F4 v[4];
  bool ok[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    v[it] = ld4(patch + (it * 8 + r8) * 32 + c4);
    ok[it] = row0 + it * 8 + r8 < p.m;
  }
  if constexpr (EPI == EPI_ADAM) {
    // every optimizer-state load (weights and, for the column-0 lanes, the
    // layer bias) is in flight before the first dependent instruction
    const bool own_bias = col == 0 && p.bias_p != nullptr;
    float bg[4] = {0.f, 0.f, 0.f, 0.f}, bm0[4] = {0.f, 0.f, 0.f, 0.f};
    float bv0[4] = {0.f, 0.f, 0.f, 0.f}, bp0[4] = {0.f, 0.f, 0.f, 0.f};
    if (own_bias) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = ok[it] ? row0 + it * 8 + r8 : 0;
        bg[it] = p.bias_g[row]; bm0[it] = p.bias_m[row];
        bv0[it] = p.bias_v[row]; bp0[it] = p.bias_p[row];
      }
    }
    const float ss = adam_ss, ib = adam_ib;
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // two rows' state in flight: bounded registers
      F4 pm[2], pv[2], pp[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int it = half * 2 + u;
        const int64_t e = ok[it] ? (int64_t)(row0 + it * 8 + r8) * p.ldc + col : 0;
        pm[u] = ld4(p.adam_m + e); pv[u] = ld4(p.adam_v + e); pp[u] = ld4(p.c + e);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int it = half * 2 + u;
        if (ok[it]) {
          const int64_t e = (int64_t)(row0 + it * 8 + r8) * p.ldc + col;
          if (p.grad_out) st4(p.grad_out + e, v[it]);
          F4 m1, v1, p1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float g = v[it].v[q];
            m1.v[q] = pm[u].v[q] + (g - pm[u].v[q]) * (1.0f - p.beta1);
            v1.v[q] = pv[u].v[q] * p.beta2 + (1.0f - p.beta2) * g * g;
            p1.v[q] = pp[u].v[q] - ss * (m1.v[q] / (sqrtf(v1.v[q]) * ib + p.adam_eps));
          }
          st4(p.adam_m + e, m1); st4(p.adam_v + e, v1); st4(p.c + e, p1);
        }
      }
    }
    if (own_bias) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (ok[it]) {
          const int row = row0 + it * 8 + r8;
          const float bm = bm0[it] + (bg[it] - bm0[it]) * (1.0f - p.beta1);
          const float bv = bv0[it] * p.beta2 + (1.0f - p.beta2) * bg[it] * bg[it];
          p.bias_m[row] = bm;
          p.bias_v[row] = bv;
          p.bias_p[row] = bp0[it] - ss * (bm / (sqrtf(bv) * ib + p.adam_eps));
        }
      }
    }
  } else if constexpr (EPI == BSIG_EPI_MUL_DACT) {
    F4 hv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
      hv[it] = ld4(p.aux + (ok[it] ? (int64_t)(row0 + it * 8 + r8) * p.ldaux + col : 0));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (ok[it]) {
        F4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o.v[q] = v[it].v[q] * act_bwd_from_out(hv[it].v[q], p.act);
        st4(p.c + (int64_t)(row0 + it * 8 + r8) * p.ldc + col, o);
      }
    }
  } else if constexpr (EPI < 0) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (ok[it])
        st4(p.partial + ((int64_t)p.bid_z * p.m + row0 + it * 8 + r8) * p.n + col, v[it]);
  } else {
    F4 bias{{0.f, 0.f, 0.f, 0.f}};
    if constexpr (EPI == BSIG_EPI_BIAS || EPI == BSIG_EPI_BIAS_ACT || EPI == BSIG_EPI_COS_OFF)
      bias = ld4(p.bias + col);
    const bool want_exp = EPI == BSIG_EPI_BIAS && p.expsum != nullptr;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (ok[it]) {
        float* dst = p.c + (int64_t)(row0 + it * 8 + r8) * p.ldc + col;
        F4 o, o2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = v[it].v[q];
          if constexpr (EPI == BSIG_EPI_NONE) o.v[q] = x;
          else if constexpr (EPI == BSIG_EPI_BIAS) {
            o.v[q] = x + bias.v[q];
            if (want_exp && col + q >= p.expsum_col0 && col + q < p.expsum_col0 + p.expsum_ncols)
              exp_acc += expf(o.v[q]);
          } else if constexpr (EPI == BSIG_EPI_BIAS_ACT) o.v[q] = act_fwd(x + bias.v[q], p.act);
          else if constexpr (EPI == BSIG_EPI_COS_OFF) o.v[q] = p.alpha * cosf(x + bias.v[q]);
          else {
            float sn, cs;
            sincosf(x, &sn, &cs);
            o.v[q] = p.alpha * cs;
            o2.v[q] = p.alpha * sn;
          }
        }
        st4(dst, o);
        if constexpr (EPI == BSIG_EPI_COS_SIN) st4(dst + p.n, o2);
      }
    }
  }
}

template <int EPI>
__device__ inline void tile_epilogue(const GemmParams& p, const float* __restrict__ patch,
                                     int rbase, int col, int h, int l31, float& exp_acc,
                                     float adam_ss, float adam_ib) {
  float v[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) v[it] = patch[(2 * it + h) * 32 + l31];
  const bool colok = col < p.n;
  if constexpr (EPI == EPI_ADAM) {
    const float ss = adam_ss, ib = adam_ib;
#pragma unroll
    for (int half = 0; half < 4; ++half) {     // 4 rows' optimizer state in flight
      float pm[4], pv[4], pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = rbase + 2 * (half * 4 + u);
        const int64_t e = (colok && row < p.m) ? (int64_t)row * p.ldc + col : 0;
        pm[u] = p.adam_m[e]; pv[u] = p.adam_v[e]; pp[u] = p.c[e];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = rbase + 2 * (half * 4 + u);
        if (colok && row < p.m) {
          const int64_t e = (int64_t)row * p.ldc + col;
          const float g = v[half * 4 + u];
          if (p.grad_out) p.grad_out[e] = g;
          const float m1 = pm[u] + (g - pm[u]) * (1.0f - p.beta1);
          const float v1 = pv[u] * p.beta2 + (1.0f - p.beta2) * g * g;
          p.adam_m[e] = m1;
          p.adam_v[e] = v1;
          p.c[e] = pp[u] - ss * (m1 / (sqrtf(v1) * ib + p.adam_eps));
        }
      }
    }
    if (col == 0 && p.bias_p) {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int row = rbase + 2 * it;
        if (row < p.m) {
          const float g = p.bias_g[row];
          const float bm = p.bias_m[row] + (g - p.bias_m[row]) * (1.0f - p.beta1);
          const float bv = p.bias_v[row] * p.beta2 + (1.0f - p.beta2) * g * g;
          p.bias_m[row] = bm;
          p.bias_v[row] = bv;
          p.bias_p[row] = p.bias_p[row] - ss * (bm / (sqrtf(bv) * ib + p.adam_eps));
        }
      }
    }
  } else if constexpr (EPI == BSIG_EPI_MUL_DACT) {
    float hv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = rbase + 2 * it;
      hv[it] = (colok && row < p.m) ? p.aux[(int64_t)row * p.ldaux + col] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = rbase + 2 * it;
      if (colok && row < p.m)
        p.c[(int64_t)row * p.ldc + col] = v[it] * act_bwd_from_out(hv[it], p.act);
    }
  } else if constexpr (EPI < 0) {   // raw split-K partial slab
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = rbase + 2 * it;
      if (colok && row < p.m)
        p.partial[((int64_t)p.bid_z * p.m + row) * p.n + col] = v[it];
    }
  } else {
    float bias = 0.f;
    if constexpr (EPI == BSIG_EPI_BIAS || EPI == BSIG_EPI_BIAS_ACT || EPI == BSIG_EPI_COS_OFF)
      bias = colok ? p.bias[col] : 0.f;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = rbase + 2 * it;
      if (colok && row < p.m) {
        float* dst = p.c + (int64_t)row * p.ldc + col;
        if constexpr (EPI == BSIG_EPI_NONE) *dst = v[it];
        else if constexpr (EPI == BSIG_EPI_BIAS) {
          const float o = v[it] + bias;
          *dst = o;
          if (p.expsum && col >= p.expsum_col0 && col < p.expsum_col0 + p.expsum_ncols)
            exp_acc += expf(o);
        }
        else if constexpr (EPI == BSIG_EPI_BIAS_ACT) *dst = act_fwd(v[it] + bias, p.act);
        else if constexpr (EPI == BSIG_EPI_COS_OFF) *dst = p.alpha * cosf(v[it] + bias);
        else {
          float sn, cs;
          sincosf(v[it], &sn, &cs);
          dst[0] = p.alpha * cs;
          dst[p.n] = p.alpha * sn;
        }
      }
    }
  }
}

// can this launch use 16-byte epilogue accesses?  (uniform over the grid)
__device__ inline bool epilogue_vec_ok(const GemmParams& p) {
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.splits > 1) return (p.n & 3) == 0 && al(p.partial);
  bool ok = (p.ldc & 3) == 0 && al(p.c);
  switch (p.epilogue) {
    case BSIG_EPI_BIAS: case BSIG_EPI_BIAS_ACT: case BSIG_EPI_COS_OFF: ok = ok && al(p.bias); break;
    case BSIG_EPI_COS_SIN: ok = ok && (p.n & 3) == 0; break;
    case BSIG_EPI_MUL_DACT: ok = ok && (p.ldaux & 3) == 0 && al(p.aux); break;
    case EPI_ADAM:
      ok = ok && al(p.adam_m) && al(p.adam_v) && (!p.grad_out || al(p.grad_out));
      break;
    default: break;
  }
  return ok;
}

#define BSIG_EPI_DISPATCH(FN, ...)                                                   \
  switch (p.splits > 1 ? -1 : p.epilogue) {                                          \
    case -1: FN<-1>(__VA_ARGS__); break;                                             \
    case BSIG_EPI_NONE: FN<BSIG_EPI_NONE>(__VA_ARGS__); break;                       \
    case BSIG_EPI_BIAS: FN<BSIG_EPI_BIAS>(__VA_ARGS__); break;                       \
    case BSIG_EPI_BIAS_ACT: FN<BSIG_EPI_BIAS_ACT>(__VA_ARGS__); break;               \
    case BSIG_EPI_COS_SIN: FN<BSIG_EPI_COS_SIN>(__VA_ARGS__); break;                 \
    case BSIG_EPI_COS_OFF: FN<BSIG_EPI_COS_OFF>(__VA_ARGS__); break;                 \
    case BSIG_EPI_MUL_DACT: FN<BSIG_EPI_MUL_DACT>(__VA_ARGS__); break;               \
    default: FN<EPI_ADAM>(__VA_ARGS__); break;                                       \
  }

__device__ __forceinline__ void run_tile_epilogue(const GemmParams& p, const float* patch,
                                                  int row0, int col0, int lane, bool vec,
                                                  float& exp_acc, float adam_ss, float adam_ib) {
  if (vec && col0 + 32 <= p.n) {
    BSIG_EPI_DISPATCH(tile_epilogue_vec, p, patch, row0, col0, lane, exp_acc, adam_ss, adam_ib)
  } else {
    const int h = lane >> 5, l31 = lane & 31;
    BSIG_EPI_DISPATCH(tile_epilogue, p, patch, row0 + h, col0 + l31, h, l31, exp_acc, adam_ss,
                      adam_ib)
  }
}

// ---- global -> register fetch and register -> LDS commit of one operand tile
// All loads of a tile are unconditional (addresses clamped into the operand,
// out-of-range elements zeroed afterwards) so they issue back to back and are
// waited for once, at the commit.  Only the contraction tail must be zeroed:
// rows beyond M / N only feed output rows / columns that are never stored.
template <int ROWS, bool KMAJOR, int VEC, int NT>
struct TileLoader {
  static constexpr int kItems = ROWS * BK / (NT * VEC);
  static_assert(ROWS * BK % (NT * VEC) == 0, "tile not divisible");
  struct Regs { float r[kItems][VEC]; };   // one tile in flight
  int srow[kItems];   // k-contiguous operand: gathered source row of each item

  int64_t off;        // device-resolved row offset of the gathered dimension

  __device__ inline void init(const int32_t* __restrict__ idx, int row0, int nrows, int tid,
                              int64_t row_off) {
    off = row_off;
    if constexpr (!KMAJOR) {
      constexpr int per_row = BK / VEC;
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int64_t gr = min(row0 + (tid + it * NT) / per_row, nrows - 1) + row_off;
        srow[it] = idx ? idx[gr] : (int)gr;
      }
    }
  }

  // `unal`: rows of a k-contiguous operand are not 16-byte aligned (ld % 4 != 0, e.g. the
  // cross-correlation widths S*A + 2): the quads are still fetched with one 16-byte load each
  // (gfx950 takes dwordx4 at any 4-byte address at full rate, tools/micro/stream_pattern_bench.hip);
  // only the last quad of a row, clamped back inside the row, has to be shifted into place.
  __device__ inline void fetch(Regs& t, const float* __restrict__ g, int64_t ld,
                               const int32_t* __restrict__ idx, int row0, int nrows,
                               int k0, int kend, int ktot, int tid, bool unal = false) const {
    auto& r = t.r;
    if constexpr (!KMAJOR) {
      constexpr int per_row = BK / VEC;
      const int cc = (tid % per_row) * VEC;   // NT % per_row == 0: same for every item
      const int gk = k0 + cc;
      // stay inside the row pitch (ld % 4 == 0 when VEC == 4 unless `unal`)
      const int gkc = (VEC == 4) ? min(gk, (int)ld - 4) : min(gk, ktot - 1);
      typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const float* src = g + (int64_t)srow[it] * ld + gkc;
        if constexpr (VEC == 4) {
          const f32x4u_t q = *reinterpret_cast<const f32x4u_t*>(src);
          r[it][0] = q.x; r[it][1] = q.y; r[it][2] = q.z; r[it][3] = q.w;
        } else {
          r[it][0] = src[0];
        }
      }
      if constexpr (VEC == 4) {
        if (unal) {
          // slot v holds column gkc + v and stands for column gk + v: move slot v + s to v
          // (slots that run past the row stand for columns >= ld >= kend: zeroed below)
          const int sft = gk - gkc;
#pragma unroll
          for (int it = 0; it < kItems; ++it) {
            const float q0 = r[it][0], q1 = r[it][1], q2 = r[it][2], q3 = r[it][3];
            r[it][0] = sft == 0 ? q0 : (sft == 1 ? q1 : (sft == 2 ? q2 : q3));
            r[it][1] = sft == 0 ? q1 : (sft == 1 ? q2 : q3);
            r[it][2] = sft == 0 ? q2 : q3;
          }
        }
      }
#pragma unroll
      for (int it = 0; it < kItems; ++it)
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (gk + v >= kend) r[it][v] = 0.f;
    } else {
      constexpr int per_k = ROWS / VEC;
      // (96-row tiles: per_k = 24 does not divide the thread count, the column of an item
      // then depends on the item)
      constexpr bool kSameCol = NT % per_k == 0;
      auto col_of = [&](int it) {
        const int gr = row0 + ((kSameCol ? tid : tid + it * NT) % per_k) * VEC;
        return (VEC == 4) ? min(gr, (int)ld - 4) : min(gr, nrows - 1);
      };
      const int grc0 = col_of(0);
      int krow[kItems];
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int64_t gkc = min(k0 + (tid + it * NT) / per_k, ktot - 1) + off;
        krow[it] = idx ? idx[gkc] : (int)gkc;
      }
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const float* src = g + (int64_t)krow[it] * ld + (kSameCol ? grc0 : col_of(it));
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          r[it][0] = q.x; r[it][1] = q.y; r[it][2] = q.z; r[it][3] = q.w;
        } else {
          r[it][0] = src[0];
        }
      }
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const bool dead = k0 + (tid + it * NT) / per_k >= kend;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (dead) r[it][v] = 0.f;
      }
    }
  }

  __device__ inline void commit(const Regs& t, float* __restrict__ lds, int tid) const {
    const auto& r = t.r;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int item = tid + it * NT;
      int off;
      if constexpr (!KMAJOR) {
        constexpr int per_row = BK / VEC;
        off = (item / per_row) * BKP + (item % per_row) * VEC;
      } else {
        constexpr int per_k = ROWS / VEC;
        off = (item / per_k) * ROWS + (item % per_k) * VEC;
      }
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(lds + off) =
            make_float4(r[it][0], r[it][1], r[it][2], r[it][3]);
      } else {
        lds[off] = r[it][0];
      }
    }
  }
};

template <int ROWS, bool KMAJOR>
constexpr int lds_floats() { return KMAJOR ? BK * ROWS : ROWS * BKP; }

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, int AVEC, int BVEC>
__global__ __launch_bounds__(WM * WN * 64) void gemm_mfma_kernel(GemmParams p) {
#ifndef BSIG_HOST_SAN_BUILD   // (the host-sanitizer build launches nothing: fit_persistent_mdnn.hip)
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int kAs = lds_floats<BM, AKM>(), kBs = lds_floats<BN, BKM>();
  constexpr int kEpi = WM * WN * 32 * 32;  // per-wave [32][32] epilogue patches
#ifdef BSIG_LDS_DB
#ifndef BSIG_LDS_DB_MIN
#define BSIG_LDS_DB_MIN 4
#endif
  constexpr bool kDoubleLds = TM * TN >= BSIG_LDS_DB_MIN;   // large tiles: two LDS images, one barrier per K step
#else
  constexpr bool kDoubleLds = false;
#endif
  constexpr int kOps = (kDoubleLds ? 2 : 1) * (kAs + kBs);
  __shared__ __attribute__((aligned(16))) float smem[kOps > kEpi ? kOps : kEpi];
  float* As = smem;
  float* Bs = smem + kAs;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  // Workgroup -> tile.  The dispatcher deals consecutive workgroups round the 8 XCDs (block b on
  // XCD b % 8, each with a private L2), so consecutive tiles -- which share an operand panel -- would
  // each fetch it from HBM: with the remap an XCD works on a contiguous run of tiles (all column tiles
  // of a row panel, all tiles of a K split) and the panel is read once per XCD.  A speed choice only:
  // every tile is still computed exactly once (the remap is a bijection on [0, nwg)).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_swz) {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * (int)gridDim.z;
    const int lin = bx + gx * (by + gy * bz);
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = wgid % gx; by = (wgid / gx) % gy; bz = wgid / (gx * gy);
  }
  p.bid_z = bz;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * p.k_chunk;
  const int kend = min(p.k, kbeg + p.k_chunk);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // Software pipeline: PD operand tiles in flight in registers.  The
  // minibatch-sized problems (M or K ~ 100) are latency bound — a dependent
  // global round trip costs more than a whole K step of MFMAs — so the loads of
  // PD K steps are issued back to back and consumed in order.
#ifndef BSIG_PD_BIG
#define BSIG_PD_BIG 1
#endif
#ifndef BSIG_PD_SMALL
#define BSIG_PD_SMALL 4
#endif
  constexpr int PD = (TM * TN == 1) ? BSIG_PD_SMALL : BSIG_PD_BIG;
  TileLoader<BM, AKM, AVEC, NT> la;
  TileLoader<BN, BKM, BVEC, NT> lb;
  typename TileLoader<BM, AKM, AVEC, NT>::Regs ra[PD];
  typename TileLoader<BN, BKM, BVEC, NT>::Regs rb[PD];
  const int nkt = (kend - kbeg + BK - 1) / BK;
  // device-resolved scalars first: their round trips overlap the operand loads
  float adam_ss = 0.f, adam_ib = 0.f;
  if (p.epilogue == EPI_ADAM) { adam_ss = p.adam_dyn[0]; adam_ib = p.adam_dyn[1]; }
  const int64_t dstep = p.dyn ? (int64_t)(p.dyn[0] + p.dyn_delta) : 0;
  la.init(p.a_rows, m0, p.m, tid, dstep * p.a_dyn_stride + p.a_dyn_base);
  lb.init(p.b_rows, n0, p.n, tid, dstep * p.b_dyn_stride + p.b_dyn_base);
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < nkt) {
      la.fetch(ra[s], p.a, p.lda, p.a_rows, m0, p.m, kbeg + s * BK, kend, p.k, tid, p.a_unal != 0);
      lb.fetch(rb[s], p.b, p.ldb, p.b_rows, n0, p.n, kbeg + s * BK, kend, p.k, tid, p.b_unal != 0);
    }
  // MFMAs of one staged K tile (BK = 32 -> 4 groups of 8 k; TM*TN*4 MFMAs each)
  auto compute_tile = [&](const float* __restrict__ At, const float* __restrict__ Bt) {
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + l31;
        if constexpr (!AKM) {
          const float4 q = *reinterpret_cast<const float4*>(&At[row * BKP + kg * 8 + h * 4]);
          af[i][0] = q.x; af[i][1] = q.y; af[i][2] = q.z; af[i][3] = q.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) af[i][u] = At[(kg * 8 + h * 4 + u) * BM + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + l31;
        if constexpr (!BKM) {
          const float4 q = *reinterpret_cast<const float4*>(&Bt[col * BKP + kg * 8 + h * 4]);
          bf[j][0] = q.x; bf[j][1] = q.y; bf[j][2] = q.z; bf[j][3] = q.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) bf[j][u] = Bt[(kg * 8 + h * 4 + u) * BN + col];
        }
      }
#ifdef BSIG_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][u], bf[j][u], acc[i][j],
                                                             0, 0, 0);
#ifdef BSIG_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };

  if constexpr (kDoubleLds) {
    // two LDS images: the next tile is committed while the others still read
    // the current one -> one barrier per K step
    if (nkt > 0) {
      la.commit(ra[0], As, tid);
      lb.commit(rb[0], Bs, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      float* cur = smem + (kt & 1) * (kAs + kBs);
      float* nxt = smem + ((kt + 1) & 1) * (kAs + kBs);
      if (kt + 1 < nkt) {
        const int k0 = kbeg + (kt + 1) * BK;
        la.fetch(ra[0], p.a, p.lda, p.a_rows, m0, p.m, k0, kend, p.k, tid, p.a_unal != 0);
        lb.fetch(rb[0], p.b, p.ldb, p.b_rows, n0, p.n, k0, kend, p.k, tid, p.b_unal != 0);
      }
      compute_tile(cur, cur + kAs);
      if (kt + 1 < nkt) {
        la.commit(ra[0], nxt, tid);
        lb.commit(rb[0], nxt + kAs, tid);
      }
      __syncthreads();
    }
  } else {
    for (int kt0 = 0; kt0 < nkt; kt0 += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
        const int kt = kt0 + s;
        if (kt < nkt) {          // block-uniform
          la.commit(ra[s], As, tid);
          lb.commit(rb[s], Bs, tid);
          __syncthreads();
          if (kt + PD < nkt) {   // refill this register slot
            const int k0 = kbeg + (kt + PD) * BK;
            la.fetch(ra[s], p.a, p.lda, p.a_rows, m0, p.m, k0, kend, p.k, tid, p.a_unal != 0);
            lb.fetch(rb[s], p.b, p.ldb, p.b_rows, n0, p.n, k0, kend, p.k, tid, p.b_unal != 0);
          }
          compute_tile(As, Bs);
          __syncthreads();
        }
      }
    }
  }

  // Epilogue.  C/D fragment of a 32x32 tile: col = lane & 31,
  // row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5).  Each wave bounces one tile
  // at a time through its private [32][32] LDS patch (the operand tiles are dead after the last
  // barrier) so that the accumulator indices stay compile-time constants while
  // the epilogue itself is a runtime switch.
  float* patch = smem + wid * (32 * 32);
  float exp_acc = 0.f;
  const bool vec_epi = epilogue_vec_ok(p);
  // (i, j) enumerated explicitly: the accumulator indices must be constants
#define BSIG_TILE_EPILOGUE(I, J)                                                          \
  if constexpr ((I) < TM && (J) < TN) {                                                   \
    _Pragma("unroll") for (int q = 0; q < 16; ++q)                                        \
        patch[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + l31] = acc[I][J][q];                \
    __builtin_amdgcn_wave_barrier();                                                      \
    run_tile_epilogue(p, patch, m0 + (wm * TM + (I)) * 32, n0 + (wn * TN + (J)) * 32,     \
                      lane, vec_epi, exp_acc, adam_ss, adam_ib);                          \
    __builtin_amdgcn_wave_barrier();                                                      \
  }
  BSIG_TILE_EPILOGUE(0, 0)
  BSIG_TILE_EPILOGUE(0, 1)
  BSIG_TILE_EPILOGUE(0, 2)
  BSIG_TILE_EPILOGUE(1, 0)
  BSIG_TILE_EPILOGUE(1, 1)
  BSIG_TILE_EPILOGUE(1, 2)
  BSIG_TILE_EPILOGUE(2, 0)
  BSIG_TILE_EPILOGUE(2, 1)
  BSIG_TILE_EPILOGUE(2, 2)
  // (one row or one column of up to nine tiles: the whole-width tiles 128 x 288 / 288 x 128)
  BSIG_TILE_EPILOGUE(0, 3) BSIG_TILE_EPILOGUE(0, 4) BSIG_TILE_EPILOGUE(0, 5)
  BSIG_TILE_EPILOGUE(0, 6) BSIG_TILE_EPILOGUE(0, 7) BSIG_TILE_EPILOGUE(0, 8)
  BSIG_TILE_EPILOGUE(3, 0) BSIG_TILE_EPILOGUE(4, 0) BSIG_TILE_EPILOGUE(5, 0)
  BSIG_TILE_EPILOGUE(6, 0) BSIG_TILE_EPILOGUE(7, 0) BSIG_TILE_EPILOGUE(8, 0)
#undef BSIG_TILE_EPILOGUE
  static_assert((TM <= 3 && TN <= 3) || (TM == 1 && TN <= 9) || (TN == 1 && TM <= 9), "extend the tile enumeration");
  if (p.expsum && p.splits == 1) {   // one partial per workgroup, fixed order
    const float s = block_sum(exp_acc, smem);
    if (tid == 0) p.expsum[by * gridDim.x + bx] = s;
  }
#endif
}

template <int WM, int WN, int TM, int TN>
inline int launch_tile(const GemmParams& p, bool akm, bool bkm, int avec, int bvec,
                       hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const dim3 grid(ceil_div(p.n, BN), ceil_div(p.m, BM), p.splits);
  const dim3 block(WM * WN * 64);
#define BSIG_GEMM_CASE(AK, BKm, AV, BV)                                            \
  if (akm == AK && bkm == BKm && avec == AV && bvec == BV) {                       \
    hipLaunchKernelGGL((gemm_mfma_kernel<WM, WN, TM, TN, AK, BKm, AV, BV>), grid,  \
                       block, 0, st, p);                                           \
    return BSIG_OK;                                                                \
  }
  BSIG_GEMM_CASE(false, false, 4, 4) BSIG_GEMM_CASE(false, false, 4, 1)
  BSIG_GEMM_CASE(false, false, 1, 4) BSIG_GEMM_CASE(false, false, 1, 1)
  BSIG_GEMM_CASE(false, true, 4, 4) BSIG_GEMM_CASE(false, true, 4, 1)
  BSIG_GEMM_CASE(false, true, 1, 4) BSIG_GEMM_CASE(false, true, 1, 1)
  BSIG_GEMM_CASE(true, false, 4, 4) BSIG_GEMM_CASE(true, false, 4, 1)
  BSIG_GEMM_CASE(true, false, 1, 4) BSIG_GEMM_CASE(true, false, 1, 1)
  BSIG_GEMM_CASE(true, true, 4, 4) BSIG_GEMM_CASE(true, true, 4, 1)
  BSIG_GEMM_CASE(true, true, 1, 4) BSIG_GEMM_CASE(true, true, 1, 1)
#undef BSIG_GEMM_CASE
  set_error("gemm: no kernel for this operand layout");
  return BSIG_EUNSUPPORTED;
}


// one per tile shape (gemm_tile_*.hip)
int launch_tile_64(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_128(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_128x32(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_128x64(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_128x96(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_96x128(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_128x288(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);
int launch_tile_288x128(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st);

}  // namespace bsig
