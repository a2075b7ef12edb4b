// K-E (generic, layer-at-a-time): fp32 MFMA linear layer over a never-materialised
// concatenation, weight-norm, masked mean/sum pooling and the embedder.
// Reference: encoders/epic.py:71-107,145-159,187-208; encoders/embedder.py:63-97.
//
// mmb_linear_cat tiling (gfx950, wave64): a block of WAVES waves owns WAVES*32 rows x BN
// columns; each wave owns 32 rows and BN/32 accumulator tiles of v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD).  Activation rows are read ONCE by exactly one wave, so
// they go straight from global memory to the A fragments (no LDS round trip); the weight
// chunk [BN x BK] is shared by all waves of the block and is double-buffered through LDS.
// Inside a K chunk the MFMA k index is permuted (k = half*BK/2 + step) so that every lane
// reads BK/2 CONTIGUOUS floats for its fragment (dwordx4 global loads, ds_read_b128).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <utility>

namespace mmb {

// compile-time loop: guarantees constant indices into the accumulator vectors (a '#pragma
// unroll' that the optimizer declines leaves them runtime-indexed, i.e. in scratch memory)
template <int I> struct IC { static constexpr int v = I; };
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Seg {
  const float* ptr;
  int width;
  int kind;
  int ld;
  int k_begin;
};
struct Segs {
  Seg s[4];
  int n;
};

// second-stage epilogue: Y2 = dropout(post_act(Y1)) + res2   (epic.py:196-206 + :89-90)
struct Post {
  float* Y2;
  int64_t y2_ld;
  const float* res2;
  int64_t res2_ld;
  int act;
  float drop_p;
  uint64_t seed, offset;
  // LayerNorm as the second stage (mmb_linear_cat_ln; row-streaming kernel with the whole row in one wave):
  // Y2 = (Y1 - mean) * rstd * ln_gamma + ln_beta, ln_stats (M,2) | NULL = (mean, rstd)
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  float* ln_stats;
  // gate of the first stage (mmb_linear_cat_gate: the reverse pass of an activation folded into the GEMM that
  // produces the gradient of its output): v *= act'(gate), applied after the residual.  gate = the pre-activation
  // (GELU) or anything with its sign (ReLU, leaky ReLU; SELU: the output itself)
  const float* gate;
  int64_t gate_ld;
  int gate_act;
};

// inverted dropout with a Philox keep-mask indexed by the element (row*Nout+col)
__device__ inline float dropout_apply(float v, float p, uint64_t seed, uint64_t offset, uint64_t idx) {
  uint32_t r[4];
  philox4(seed, offset, idx >> 2, 0xD0D0u, r);
  const float u = u01(r[idx & 3]);
  return u < p ? 0.0f : v / (1.0f - p);
}

__device__ inline float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float apply_act(int act, float x) {
  if (act == MMB_ACT_GELU) return gelu_exact(x);
  if (act == MMB_ACT_LRELU) return x > 0.0f ? x : 0.01f * x;
  if (act == MMB_ACT_RELU) return x > 0.0f ? x : 0.0f;
  if (act == MMB_ACT_SELU)          // torch.nn.SELU: scale * (max(0,x) + min(0, alpha * (exp(x) - 1)))
    return x > 0.0f ? 1.0507009873554804934193349852946f * x
                    : (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * (expf(x) - 1.0f);
  return x;
}

// act'(z) from ref: the pre-activation z for GELU; anything with the sign of z for (leaky) ReLU; the OUTPUT y for SELU
// (y > 0 -> scale, else scale*alpha*exp(z) = y + scale*alpha) -- the same function as backward_kernels.hip: act_grad
__device__ inline float act_grad(int act, float ref) {
  if (act == MMB_ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(ref * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * ref * ref);
    return cdf + ref * pdf;
  }
  if (act == MMB_ACT_LRELU) return ref > 0.0f ? 1.0f : 0.01f;
  if (act == MMB_ACT_RELU) return ref > 0.0f ? 1.0f : 0.0f;
  if (act == MMB_ACT_SELU)
    return ref > 0.0f ? 1.0507009873554804934193349852946f
                      : ref + 1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f;
  return 1.0f;
}

// generic element fetch of the concatenated input (slow path only)
__device__ inline float fetch_cat(const Segs& sg, int row, int jet, float m, int k) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < sg.n) {
      const Seg& s = sg.s[i];
      if (k >= s.k_begin && k < s.k_begin + s.width) {
        int kk = k - s.k_begin;
        if (s.kind == MMB_SEG_ROW) return s.ptr[(int64_t)row * s.ld + kk];
        float v = s.ptr[(int64_t)jet * s.ld + kk];
        return s.kind == MMB_SEG_JET_MASK ? v * m : v;
      }
    }
  }
  return 0.0f;
}

// common epilogue of one 32x32 accumulator tile (C/D layout of 32x32x2: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)):  y1 = act(z + b) + res1 [* row_scale];  y2 = dropout(post(y1)) + res2
// act over the 16 values of a tile, ONE uniform switch
__device__ __forceinline__ void apply_act16(int act, f32x16& v) {
  switch (act) {
    case MMB_ACT_GELU: static_for<16>([&](auto ic) { v[decltype(ic)::v] = apply_act(MMB_ACT_GELU, v[decltype(ic)::v]); }); break;
    case MMB_ACT_LRELU: static_for<16>([&](auto ic) { v[decltype(ic)::v] = apply_act(MMB_ACT_LRELU, v[decltype(ic)::v]); }); break;
    case MMB_ACT_RELU: static_for<16>([&](auto ic) { v[decltype(ic)::v] = apply_act(MMB_ACT_RELU, v[decltype(ic)::v]); }); break;
    case MMB_ACT_SELU: static_for<16>([&](auto ic) { v[decltype(ic)::v] = apply_act(MMB_ACT_SELU, v[decltype(ic)::v]); }); break;
    default: break;
  }
}

// Epilogue of one 32x32 accumulator tile.  One pass per optional stream over the tile's 16 values (uniform branches
// OUTSIDE the element loops), each stream addressed from ONE 64-bit base per lane plus compile-time multiples of its
// row stride.  FULL: all 32 rows of the tile exist -- no per-row guards, so the loads of a stream are issued back to
// back and awaited once (the per-element form compiled to ~500 branches and one awaited load per element).
template <bool FULL>
struct TileIO {
  int row0, half, nrow, col;                   // rows of this lane's 16 that exist: offsets (i&3) + 8*(i>>2) < nrow
  __device__ __forceinline__ int rbase() const { return row0 + 4 * half; }
  __device__ __forceinline__ void load16(const float* base, int64_t ld, f32x16& r) const {
    const float* p = base + (int64_t)rbase() * ld + col;
    static_for<16>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int off = (i & 3) + 8 * (i >> 2);
      r[i] = (FULL || off < nrow) ? p[off * ld] : 0.0f;
    });
  }
  __device__ __forceinline__ void store16(float* base, int64_t ld, const f32x16& r) const {
    float* p = base + (int64_t)rbase() * ld + col;
    static_for<16>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int off = (i & 3) + 8 * (i >> 2);
      if (FULL || off < nrow) p[off * ld] = r[i];
    });
  }
};

// first stage: v = act(v + bias) + residual [* row_scale] -> Y (and the pre-activation)
template <bool FULL>
__device__ __forceinline__ void
epilogue_stage1(f32x16& v, const TileIO<FULL>& io, int Nout, float bv, int act,
                const float* __restrict__ residual, int64_t res_ld, const float* __restrict__ row_scale,
                float* __restrict__ Y, int64_t y_ld, float* __restrict__ pre_act, const Post& post) {
  f32x16 t;                                      // bv: the bias of this lane's column (loaded by the caller: a load here
  static_for<16>([&](auto ic) { v[decltype(ic)::v] += bv; });   // would wait for every load in flight, prefetches included)
  if (pre_act) io.store16(pre_act, (int64_t)Nout, v);
  apply_act16(act, v);
  if (residual) {
    io.load16(residual, res_ld, t);
    v += t;
  }
  if (row_scale) {
    io.load16(row_scale - io.col, (int64_t)1, t);
    v *= t;
  }
  if (post.gate) {
    io.load16(post.gate, post.gate_ld, t);
    switch (post.gate_act) {                     // (one uniform switch, as apply_act16)
      case MMB_ACT_GELU: static_for<16>([&](auto ic) { constexpr int i = decltype(ic)::v; v[i] *= act_grad(MMB_ACT_GELU, t[i]); }); break;
      case MMB_ACT_LRELU: static_for<16>([&](auto ic) { constexpr int i = decltype(ic)::v; v[i] *= act_grad(MMB_ACT_LRELU, t[i]); }); break;
      case MMB_ACT_SELU: static_for<16>([&](auto ic) { constexpr int i = decltype(ic)::v; v[i] *= act_grad(MMB_ACT_SELU, t[i]); }); break;
      default: static_for<16>([&](auto ic) { constexpr int i = decltype(ic)::v; v[i] = t[i] > 0.0f ? v[i] : 0.0f; }); break;
    }
  }
  if (Y) io.store16(Y, y_ld, v);
}

template <bool FULL>
__device__ __forceinline__ void
epilogue_tile_body(f32x16 v, int row0, int col, int half, int M, int Nout, float bv, int act,
                   const float* __restrict__ residual, int64_t res_ld, const float* __restrict__ row_scale,
                   float* __restrict__ Y, int64_t y_ld, float* __restrict__ pre_act, const Post& post) {
  const TileIO<FULL> io{row0, half, M - (row0 + 4 * half), col};
  epilogue_stage1<FULL>(v, io, Nout, bv, act, residual, res_ld, row_scale, Y, y_ld, pre_act, post);
  if (post.Y2) {
    f32x16 t;
    apply_act16(post.act, v);
    if (post.drop_p > 0.0f)
      static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        v[i] = dropout_apply(v[i], post.drop_p, post.seed, post.offset,
                             (uint64_t)(io.rbase() + (i & 3) + 8 * (i >> 2)) * Nout + col);
      });
    if (post.res2) {
      io.load16(post.res2, post.res2_ld, t);
      v += t;
    }
    io.store16(post.Y2, post.y2_ld, v);
  }
}

// bv: bias of column col, already in a register
__device__ __forceinline__ void epilogue_tile_b(const f32x16& v, int row0, int col, int half, int M, int Nout, float bv,
                                                int act, const float* __restrict__ residual, int64_t res_ld,
                                                const float* __restrict__ row_scale, float* __restrict__ Y,
                                                int64_t y_ld, float* __restrict__ pre_act, const Post& post) {
  if (col >= Nout) return;
  row0 = __builtin_amdgcn_readfirstlane(row0);   // (wave-uniform at every caller: a tile belongs to one wave)
  if (row0 + 32 <= M)
    epilogue_tile_body<true>(v, row0, col, half, M, Nout, bv, act, residual, res_ld, row_scale, Y, y_ld, pre_act, post);
  else
    epilogue_tile_body<false>(v, row0, col, half, M, Nout, bv, act, residual, res_ld, row_scale, Y, y_ld, pre_act, post);
}
__device__ __forceinline__ void epilogue_tile(const f32x16& v, int row0, int col, int half, int M, int Nout,
                                              const float* __restrict__ bias, int act,
                                              const float* __restrict__ residual, int64_t res_ld,
                                              const float* __restrict__ row_scale, float* __restrict__ Y,
                                              int64_t y_ld, float* __restrict__ pre_act, const Post& post) {
  if (col >= Nout) return;
  epilogue_tile_b(v, row0, col, half, M, Nout, bias ? bias[col] : 0.0f, act, residual, res_ld, row_scale, Y, y_ld,
                  pre_act, post);
}

// The wave owns WHOLE rows (Nout = 32 NT): first stage of every tile, then LayerNorm over the row -- a row lives in
// register r of the 32 lanes of one lane half, NT tiles wide: two butterfly sums over the half (mean, then the
// variance of the centred values, as mmb_layernorm_fwd) -- and the normalised row leaves as the second stage.
// sum over the 32 lanes of a lane half, delivered to all of them: four DPP steps inside the rows of 16 (no LDS
// crossbar: ds_bpermute for all five steps made the fused epilogue slower than a separate LayerNorm launch) and one
// exchange between the two rows
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_sum32(float s) {
  s += dpp_f<0xB1>(s);                         // quad_perm [1,0,3,2]
  s += dpp_f<0x4E>(s);                         // quad_perm [2,3,0,1]
  s += dpp_f<0x141>(s);                        // row_half_mirror: the other quad of the 8
  s += dpp_f<0x140>(s);                        // row_mirror: the other 8 of the 16
  s += __shfl_xor(s, 16, 64);
  return s;
}
template <int NT, bool FULL>
__device__ __forceinline__ void
epilogue_rows_ln(f32x16 (&v)[NT], int row0, int col0, int r32, int half, int M, int Nout,
                 const float (&bvj)[NT], int act, const float* __restrict__ residual, int64_t res_ld,
                 const float* __restrict__ row_scale, float* __restrict__ Y, int64_t y_ld,
                 float* __restrict__ pre_act, const Post& post) {
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::v;
    const TileIO<FULL> io{row0, half, M - (row0 + 4 * half), col0 + 32 * j + r32};
    epilogue_stage1<FULL>(v[j], io, Nout, bvj[j], act, residual, res_ld, row_scale, Y, y_ld, pre_act, post);
  });
  const float inv_n = 1.0f / (float)(32 * NT);
  f32x16 mean, rstd;
  static_for<16>([&](auto ic) {
    constexpr int i = decltype(ic)::v;
    float s = 0.0f;
    static_for<NT>([&](auto jc) { s += v[decltype(jc)::v][i]; });
    mean[i] = half_sum32(s) * inv_n;
  });
  static_for<16>([&](auto ic) {
    constexpr int i = decltype(ic)::v;
    float q = 0.0f;
    static_for<NT>([&](auto jc) { constexpr int j = decltype(jc)::v;
      v[j][i] -= mean[i];
      q = fmaf(v[j][i], v[j][i], q); });
    rstd[i] = rsqrtf(half_sum32(q) * inv_n + post.ln_eps);
  });
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::v;
    const int col = col0 + 32 * j + r32;
    const TileIO<FULL> io{row0, half, M - (row0 + 4 * half), col};
    const float gm = post.ln_gamma[col], bt = post.ln_beta[col];
    static_for<16>([&](auto ic) { constexpr int i = decltype(ic)::v; v[j][i] = fmaf(v[j][i] * rstd[i], gm, bt); });
    io.store16(post.Y2, post.y2_ld, v[j]);
  });
  if (post.ln_stats && r32 == 0) {
    const int rbase = row0 + 4 * half;
    static_for<16>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int off = (i & 3) + 8 * (i >> 2);
      if (FULL || rbase + off < M) {
        post.ln_stats[2 * (int64_t)(rbase + off)] = mean[i];
        post.ln_stats[2 * (int64_t)(rbase + off) + 1] = rstd[i];
      }
    });
  }
}

template <int WAVES, int NT, int BK, bool ALIGNED>
__global__ void __launch_bounds__(WAVES * 64)
linear_cat_kernel(int M, int rows_per_jet, Segs sg, int K, const float* __restrict__ maskf,
                  const float* __restrict__ W, int64_t w_ld, const float* __restrict__ bias, int Nout,
                  int act, const float* __restrict__ residual, int64_t res_ld,
                  const float* __restrict__ row_scale, float* __restrict__ Y, int64_t y_ld,
                  float* __restrict__ pre_act, Post post) {
  constexpr int BN = NT * 32;
  constexpr int HALF = BK / 2;
  constexpr int PITCH = BK + 4;                 // conflict-free for ds_read_b128
  constexpr int THREADS = WAVES * 64;
  constexpr int W_F4 = BN * BK / 4;             // float4 per W chunk
  constexpr int W_PER_T = (W_F4 + THREADS - 1) / THREADS;
  __shared__ __attribute__((aligned(16))) float Ws[2][BN * PITCH];

  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int r32 = lane & 31;
  const int row0 = blockIdx.x * (WAVES * 32) + wave * 32;
  const int col0 = blockIdx.y * BN;
  const int my_row = row0 + r32;
  const int ld_row = my_row < M ? my_row : M - 1;     // clamp loads; stores are guarded
  const int jet = ld_row / rows_per_jet;
  const float m = maskf ? maskf[ld_row] : 1.0f;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;

  const int nchunks = (K + BK - 1) / BK;
  float a_next[HALF];        // RAW prefetched values; the mask scale is applied when they are consumed
  float a_scale_next = 1.0f;  // (multiplying at load time would force a vmcnt(0) right behind the load)
  float4 w_next[W_PER_T];

  auto load_a = [&](int chunk) {
    const int kh = chunk * BK + half * HALF;
    if constexpr (ALIGNED) {
      // segments are HALF-aligned: this lane's HALF values live in one segment
      const float* src = nullptr;
      float scale = 1.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < sg.n && kh >= sg.s[i].k_begin && kh < sg.s[i].k_begin + sg.s[i].width) {
          const Seg& s = sg.s[i];
          const int64_t base = (s.kind == MMB_SEG_ROW) ? (int64_t)ld_row * s.ld : (int64_t)jet * s.ld;
          src = s.ptr + base + (kh - s.k_begin);
          scale = (s.kind == MMB_SEG_JET_MASK) ? m : 1.0f;
        }
      }
      if (src != nullptr) {
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
          float4 v = reinterpret_cast<const float4*>(src)[q];
          a_next[4 * q + 0] = v.x; a_next[4 * q + 1] = v.y;
          a_next[4 * q + 2] = v.z; a_next[4 * q + 3] = v.w;
        }
        a_scale_next = scale;
      } else {
        a_scale_next = 1.0f;
#pragma unroll
        for (int q = 0; q < HALF; ++q) a_next[q] = 0.0f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < HALF; ++q) a_next[q] = fetch_cat(sg, ld_row, jet, m, kh + q);
      a_scale_next = 1.0f;
    }
  };

  auto load_w = [&](int chunk) {
    const int k0 = chunk * BK;
#pragma unroll
    for (int q = 0; q < W_PER_T; ++q) {
      const int idx = tid + q * THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < W_F4) {
        const int o = idx / (BK / 4);
        const int c4 = idx % (BK / 4);
        const int go = col0 + o;
        const int gk = k0 + 4 * c4;
        if (go < Nout) {
          const float* wp = W + (int64_t)go * w_ld + gk;
          if constexpr (ALIGNED) {
            if (gk < K) v = *reinterpret_cast<const float4*>(wp);
          } else {
            v.x = gk + 0 < K ? wp[0] : 0.f; v.y = gk + 1 < K ? wp[1] : 0.f;
            v.z = gk + 2 < K ? wp[2] : 0.f; v.w = gk + 3 < K ? wp[3] : 0.f;
          }
        }
      }
      w_next[q] = v;
    }
  };

  auto store_w = [&](int buf) {
#pragma unroll
    for (int q = 0; q < W_PER_T; ++q) {
      const int idx = tid + q * THREADS;
      if (idx < W_F4) {
        const int o = idx / (BK / 4);
        const int c4 = idx % (BK / 4);
        *reinterpret_cast<float4*>(&Ws[buf][o * PITCH + 4 * c4]) = w_next[q];
      }
    }
  };

  load_a(0);
  load_w(0);
  store_w(0);

  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                              // Ws[c&1] complete; Ws[(c+1)&1] free
    float a_cur[HALF];
#pragma unroll
    for (int q = 0; q < HALF; ++q) a_cur[q] = a_next[q] * a_scale_next;
    if (c + 1 < nchunks) {
      load_a(c + 1);
      load_w(c + 1);
    }
    const float* wbuf = Ws[c & 1];
#pragma unroll
    for (int s4 = 0; s4 < HALF / 4; ++s4) {
      float4 b[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        b[j] = *reinterpret_cast<const float4*>(&wbuf[(j * 32 + r32) * PITCH + half * HALF + 4 * s4]);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 0], b[j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 1], b[j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 2], b[j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 3], b[j].w, acc[j], 0, 0, 0);
    }
    if (c + 1 < nchunks) store_w((c + 1) & 1);
  }

  // epilogue.  C/D layout of 32x32x2: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::v;
    epilogue_tile(acc[j], row0, col0 + j * 32 + r32, half, M, Nout, bias, act, residual, res_ld, row_scale, Y, y_ld,
                  pre_act, post);
  });
}

// ---- large-M variant with the weight block RESIDENT in LDS ("row streaming") ---------------------------------
// The per-particle layers are tall and skinny (M = 10^5 rows, K and Nout of 64..384): a 128-row block of the kernel
// above lives for two or three K chunks, most of which it spends waiting for its first loads and at barriers.  Here
// ONE block of 16 waves per CU stages its [NT*32 x K] weight block in LDS once (up to 160 KB) and after that single
// barrier every wave streams 32-row tiles on its own: A fragments straight from global memory, the next chunk (or
// the next tile's first chunk) in flight during the MFMAs and the epilogue.  Tiles are dealt to SIMDs (wave & 3),
// not to waves, so that the matrix pipes -- the shared resource -- carry equal loads.
template <int NT>
__global__ void __launch_bounds__(1024)
linear_stream_kernel(int M, int rows_per_jet, Segs sg, int K, const float* __restrict__ maskf,
                     const float* __restrict__ W, int64_t w_ld, const float* __restrict__ bias, int Nout, int act,
                     const float* __restrict__ residual, int64_t res_ld, const float* __restrict__ row_scale,
                     float* __restrict__ Y, int64_t y_ld, float* __restrict__ pre_act, Post post) {
  constexpr int BN = NT * 32;
  constexpr int HALF = 16;                      // BK = 32
  extern __shared__ __attribute__((aligned(16))) float Wl[];
  const int KP = (K + 31) & ~31, PITCH = KP + 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  const int col0 = blockIdx.y * BN;
  for (int idx = tid; idx < BN * (KP / 4); idx += blockDim.x) {
    const int o = idx / (KP / 4), gk = 4 * (idx - o * (KP / 4));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col0 + o < Nout && gk < K) v = *reinterpret_cast<const float4*>(W + (int64_t)(col0 + o) * w_ld + gk);
    *reinterpret_cast<float4*>(&Wl[o * PITCH + gk]) = v;
  }
  __syncthreads();                              // the only barrier: waves are on their own from here

  const int nchunks = KP / 32, ntiles = (M + 31) / 32;
  const int nsimd = gridDim.x * 4, tstep = nsimd * (blockDim.x >> 8);
  int t = blockIdx.x * 4 + (wave & 3) + nsimd * (wave >> 2);
  if (t >= ntiles) return;

  int ld_row, jet;
  float m;
  float a_next[HALF];
  float a_scale_next = 1.0f;
  auto setup = [&](int tile) {
    const int my_row = tile * 32 + r32;
    ld_row = my_row < M ? my_row : M - 1;       // clamp loads; stores are guarded
    jet = ld_row / rows_per_jet;
    m = maskf ? maskf[ld_row] : 1.0f;
  };
  auto load_a = [&](int chunk) {                // (segments are HALF-aligned: the lane's 16 values live in one of them)
    const int kh = chunk * 32 + half * HALF;
    const float* src = nullptr;
    float scale = 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < sg.n && kh >= sg.s[i].k_begin && kh < sg.s[i].k_begin + sg.s[i].width) {
        const Seg& sgm = sg.s[i];
        const int64_t base = (sgm.kind == MMB_SEG_ROW) ? (int64_t)ld_row * sgm.ld : (int64_t)jet * sgm.ld;
        src = sgm.ptr + base + (kh - sgm.k_begin);
        scale = (sgm.kind == MMB_SEG_JET_MASK) ? m : 1.0f;
      }
    }
    if (src != nullptr) {
#pragma unroll
      for (int q = 0; q < HALF / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(src)[q];
        a_next[4 * q + 0] = v.x; a_next[4 * q + 1] = v.y;
        a_next[4 * q + 2] = v.z; a_next[4 * q + 3] = v.w;
      }
      a_scale_next = scale;
    } else {
      a_scale_next = 1.0f;
#pragma unroll
      for (int q = 0; q < HALF; ++q) a_next[q] = 0.0f;
    }
  };

  setup(t);
  load_a(0);
  while (true) {
    const int row0 = t * 32;
    const int tn = t + tstep;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    for (int c = 0; c < nchunks; ++c) {
      float a_cur[HALF];
#pragma unroll
      for (int q = 0; q < HALF; ++q) a_cur[q] = a_next[q] * a_scale_next;
      if (c + 1 < nchunks) load_a(c + 1);
      const float* wbuf = Wl + c * 32 + half * HALF;
#pragma unroll
      for (int s4 = 0; s4 < HALF / 4; ++s4) {
        float4 b[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(&wbuf[(j * 32 + r32) * PITCH + 4 * s4]);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 0], b[j].x, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 1], b[j].y, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 2], b[j].z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[4 * s4 + 3], b[j].w, acc[j], 0, 0, 0);
      }
    }
    static_for<NT>([&](auto jc) {
      constexpr int j = decltype(jc)::v;
      epilogue_tile(acc[j], row0, col0 + j * 32 + r32, half, M, Nout, bias, act, residual, res_ld, row_scale, Y, y_ld,
                    pre_act, post);
    });
    if (tn >= ntiles) break;
    t = tn;
    setup(t);                                   // (fetching before the epilogue would keep 16 more registers live
    load_a(0);                                  //  across it: the other three waves of the SIMD cover this latency)
  }
}

// ---- the row-streaming kernel on the bf16 matrix cores: fp32 products as six bf16-plane products -------------
// Same structure as linear_stream_kernel; every fp32 value is split EXACTLY into three bf16 planes (hi + mid + lo,
// 24 mantissa bits) and X W^T = Xl Wh + Xm Wm + Xh Wl + Xm Wh + Xh Wm + Xh Wh (each bf16 x bf16 product is exact in
// fp32, fp32 accumulation, small terms first; the three dropped products are below 2^-24 of the result) -- the scheme
// of the fused kernels (fused_common.h: mma_x).  Six v_mfma_f32_32x32x16_bf16 retire 16 k-values in 192 cycles
// against 512 for eight v_mfma_f32_32x32x2_f32.  The weight block is split once, when it is staged into LDS
// (three bf16 arrays [NT*32][K + 8]); an activation value is split once per column block, in registers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split8(const float* v, bf16x8& h, bf16x8& m, bf16x8& l) {
  uint32_t ph[4], pm[4], pq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    ph[j] = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, ph[j] << 16), rb = b - __builtin_bit_cast(float, ph[j] & 0xFFFF0000u);
    pm[j] = pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, pm[j] << 16), sb = rb - __builtin_bit_cast(float, pm[j] & 0xFFFF0000u);
    pq[j] = pk_bf16(sa, sb);
  }
  const u32x4 H = {ph[0], ph[1], ph[2], ph[3]}, M = {pm[0], pm[1], pm[2], pm[3]}, L = {pq[0], pq[1], pq[2], pq[3]};
  h = __builtin_bit_cast(bf16x8, H);
  m = __builtin_bit_cast(bf16x8, M);
  l = __builtin_bit_cast(bf16x8, L);
}

template <int NT, int THREADS, bool SIMPLE, bool LN>   // SIMPLE: one ROW segment (no segment search, no scale);
                                                       // LN: LayerNorm second stage (its own instantiation: the
                                                       // row statistics cost registers the plain kernel must not pay)
__global__ void __launch_bounds__(THREADS)
linear_stream_x6_kernel(int M, int rows_per_jet, Segs sg, int K, const float* __restrict__ maskf,
                        const float* __restrict__ W, int64_t w_ld, const float* __restrict__ bias, int Nout, int act,
                        const float* __restrict__ residual, int64_t res_ld, const float* __restrict__ row_scale,
                        float* __restrict__ Y, int64_t y_ld, float* __restrict__ pre_act, Post post) {
  constexpr int BN = NT * 32;
  constexpr int HALF = 16;                      // a lane's share of a 32-wide K chunk
  extern __shared__ __attribute__((aligned(16))) unsigned short Wp[];     // [3][BN][PK] bf16
  const int KP = (K + 63) & ~63, PK = KP + 8;    // K chunks of 32 come in PAIRS (buffers A, B): pad to 64
  const int PLANE = BN * PK;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  const int col0 = blockIdx.y * BN;
  for (int idx = tid; idx < BN * (KP / 4); idx += blockDim.x) {
    const int o = idx / (KP / 4), gk = 4 * (idx - o * (KP / 4));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col0 + o < Nout && gk < K) v = *reinterpret_cast<const float4*>(W + (int64_t)(col0 + o) * w_ld + gk);
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint32_t ph[2], pm[2], pq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a = f[2 * j], b = f[2 * j + 1];
      ph[j] = pk_bf16(a, b);
      const float ra = a - __builtin_bit_cast(float, ph[j] << 16), rb = b - __builtin_bit_cast(float, ph[j] & 0xFFFF0000u);
      pm[j] = pk_bf16(ra, rb);
      const float sa = ra - __builtin_bit_cast(float, pm[j] << 16), sb = rb - __builtin_bit_cast(float, pm[j] & 0xFFFF0000u);
      pq[j] = pk_bf16(sa, sb);
    }
    unsigned short* dst = Wp + o * PK + gk;
    *reinterpret_cast<uint2*>(dst) = make_uint2(ph[0], ph[1]);
    *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(pm[0], pm[1]);
    *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(pq[0], pq[1]);
  }
  __syncthreads();                              // the only barrier

  const int nchunks = KP / 32, npairs = KP / 64, ntiles = (M + 31) / 32;
  const int nsimd = gridDim.x * 4, tstep = nsimd * (blockDim.x >> 8);
  int t = blockIdx.x * 4 + (wave & 3) + nsimd * (wave >> 2);
  if (t >= ntiles) return;

  // The (tile, chunk) sequence of this wave is walked by TWO cursors: the fetch cursor runs two chunks ahead of the
  // compute cursor (buffers A and B, filled alternately), across tile boundaries -- 128 bytes per lane in flight,
  // which is what a CU needs outstanding to keep its share of HBM busy with 8..12 waves.
  int ld_row, jet;
  float m;
  int ft = t, fc = 0;                            // fetch cursor
  auto setup = [&](int tile) {
    const int my_row = tile * 32 + r32;
    ld_row = my_row < M ? my_row : M - 1;
    jet = ld_row / rows_per_jet;
    m = (!SIMPLE && maskf) ? maskf[ld_row] : 1.0f;
  };
  // branch-free: ALWAYS four 16-byte loads (past the last tile the previous rows are fetched again, past K the
  // start of the row -- both land in valid memory and are multiplied by 0), so that the number of loads in flight
  // is static and the compiler can wait for the OLDER buffer only (vmcnt(4)) instead of draining the queue
  const bool kfull = SIMPLE && (K & 31) == 0;    // then every chunk is whole and bscale is not needed at all
  auto fetch = [&](float (&buf)[HALF], float& bscale) __attribute__((always_inline)) {
    const int kh = fc * 32 + half * HALF;
    const float* src;
    float scale = ft < ntiles ? 1.0f : 0.0f;
    if constexpr (SIMPLE) {
      const bool in = kh < K;
      src = sg.s[0].ptr + (int64_t)ld_row * sg.s[0].ld + (in ? kh : 0);
      if (!in) scale = 0.0f;
    } else {
      src = sg.s[0].ptr + ((sg.s[0].kind == MMB_SEG_ROW) ? (int64_t)ld_row * sg.s[0].ld : (int64_t)jet * sg.s[0].ld);
      float sc = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < sg.n && kh >= sg.s[i].k_begin && kh < sg.s[i].k_begin + sg.s[i].width) {
          const Seg& sgm = sg.s[i];
          const int64_t base = (sgm.kind == MMB_SEG_ROW) ? (int64_t)ld_row * sgm.ld : (int64_t)jet * sgm.ld;
          src = sgm.ptr + base + (kh - sgm.k_begin);
          sc = (sgm.kind == MMB_SEG_JET_MASK) ? m : 1.0f;
        }
      }
      scale *= sc;
    }
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 v = reinterpret_cast<const float4*>(src)[q];
      buf[4 * q + 0] = v.x; buf[4 * q + 1] = v.y;
      buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
    }
    bscale = scale;
    if (++fc == nchunks) {
      fc = 0;
      ft += tstep;
      if (ft < ntiles) setup(ft);
    }
  };

  f32x16 acc[NT];
  auto zero_acc = [&]() __attribute__((always_inline)) {                        // (static_for: constant indices keep acc in registers)
    static_for<NT>([&](auto jc) { static_for<16>([&](auto ic) { acc[decltype(jc)::v][decltype(ic)::v] = 0.0f; }); });
  };
  auto consume = [&](float (&buf)[HALF], float& bscale, int cc) __attribute__((always_inline)) {   // (else acc lives in scratch)
    bf16x8 ahs[2], ams[2], als[2];               // split first: the buffer is free for the next fetch, no copy
    if (!kfull) {
#pragma unroll
      for (int q = 0; q < HALF; ++q) buf[q] *= bscale;
    }
    split8(buf, ahs[0], ams[0], als[0]);
    split8(buf + 8, ahs[1], ams[1], als[1]);
    fetch(buf, bscale);
    const unsigned short* wb = Wp + r32 * PK + cc * 32 + half * HALF;
#pragma unroll
    for (int s8 = 0; s8 < 2; ++s8) {
      const bf16x8 ah = ahs[s8], am = ams[s8], al = als[s8];
      static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::v;
        const unsigned short* wj = wb + j * 32 * PK + 8 * s8;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wj));
        const bf16x8 bm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wj + PLANE));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wj + 2 * PLANE));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      });
    }
  };

  float bvj[NT];                                // this lane's bias per column tile: loaded ONCE, before the stream
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::v;
    const int c = col0 + j * 32 + r32;
    bvj[j] = (bias != nullptr && c < Nout) ? bias[c] : 0.0f;
  });
  float bufA[HALF], bufB[HALF];
  float scaleA = 1.0f, scaleB = 1.0f;
  setup(t);
  fetch(bufA, scaleA);
  fetch(bufB, scaleB);
  for (; t < ntiles; t += tstep) {
    zero_acc();
    for (int p = 0; p < npairs; ++p) {
      consume(bufA, scaleA, 2 * p);
      consume(bufB, scaleB, 2 * p + 1);
    }
    const int row0 = __builtin_amdgcn_readfirstlane(t) * 32;
    if constexpr (LN) {                         // (the host launches this only with Nout == 32 NT, one column block)
      if (row0 + 32 <= M)
        epilogue_rows_ln<NT, true>(acc, row0, col0, r32, half, M, Nout, bvj, act, residual, res_ld, row_scale, Y, y_ld,
                                   pre_act, post);
      else
        epilogue_rows_ln<NT, false>(acc, row0, col0, r32, half, M, Nout, bvj, act, residual, res_ld, row_scale, Y, y_ld,
                                    pre_act, post);
    } else {
      static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::v;
        epilogue_tile_b(acc[j], row0, col0 + j * 32 + r32, half, M, Nout, bvj[j], act, residual, res_ld, row_scale, Y,
                        y_ld, pre_act, post);
      });
    }
  }
}

// ---- small-M variant (per-jet layers, M = B rows): split-K over the 4 waves of a block ----------
// A per-jet layer is a (B x K) x (K x Nout) product with B ~ 1024: a handful of 32x32 tiles whose cost
// is the LENGTH of the serial k loop, not flops.  Here one block owns ONE 32x32 output tile and its
// four waves each take a quarter of K; both operands go straight from global memory to the MFMA
// fragments with every load of the wave in flight at once (k is permuted inside the wave's slice so
// that each lane reads one contiguous run of X[row][.] / W[col][.]), the four partial tiles meet in
// LDS and wave 0 runs the common epilogue.  One ROW segment, K % 4 == 0, 16-byte aligned rows.
template <int RL4>                                   // float4 per lane and operand: slice = 8*RL4 k values
__global__ void __launch_bounds__(256)
linear_splitk_kernel(int M, const float* __restrict__ X, int64_t x_ld, int K, const float* __restrict__ W,
                     int64_t w_ld, const float* __restrict__ bias, int Nout, int act,
                     const float* __restrict__ residual, int64_t res_ld, const float* __restrict__ row_scale,
                     float* __restrict__ Y, int64_t y_ld, float* __restrict__ pre_act, Post post) {
  constexpr int RL = 4 * RL4;                        // contiguous floats per lane
  __shared__ float red[3][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r32 = lane & 31;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int row = min(row0 + r32, M - 1), col = min(col0 + r32, Nout - 1);      // clamped loads, guarded stores
  const int kb = wave * (2 * RL) + half * RL;        // this lane's run inside the block's K range
  const float* xp = X + (int64_t)row * x_ld + kb;
  const float* wp = W + (int64_t)col * w_ld + kb;
  float4 a[RL4], b[RL4];
#pragma unroll
  for (int q = 0; q < RL4; ++q) {
    const bool ok = kb + 4 * q < K;                  // K % 4 == 0: a float4 is all in or all out
    a[q] = ok ? *reinterpret_cast<const float4*>(xp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    b[q] = ok ? *reinterpret_cast<const float4*>(wp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
  for (int q = 0; q < RL4; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave - 1][i * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += (red[0][i * 64 + lane] + red[1][i * 64 + lane]) + red[2][i * 64 + lane];
    epilogue_tile(acc, row0, col0 + r32, half, M, Nout, bias, act, residual, res_ld, row_scale, Y, y_ld, pre_act,
                  post);
  }
}

// ---------------------------------------------------------------- weight norm ----------
__global__ void __launch_bounds__(64)
weight_norm_kernel(int n_rows, const int32_t* __restrict__ row_off, const int32_t* __restrict__ row_len,
                   const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ W,
                   float* __restrict__ inv_norm) {
  const int r = blockIdx.x;
  const int off = row_off[r], len = row_len[r];
  float ss = 0.0f;
  for (int k = threadIdx.x; k < len; k += 64) { float a = v[off + k]; ss += a * a; }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float scale = g[r] / nrm;
  for (int k = threadIdx.x; k < len; k += 64) W[off + k] = v[off + k] * scale;
  if (inv_norm && threadIdx.x == 0) inv_norm[r] = 1.0f / nrm;
}

// dW (and db) may arrive as `reps` partial replicas (rep_stride / db_stride floats apart): the
// weight-gradient kernel spreads its atomics over replicas because same-address atomics serialise
// at the memory side; they are summed here, on the way in.
__global__ void __launch_bounds__(64)
weight_norm_bwd_kernel(int n_rows, const int32_t* __restrict__ row_off, const int32_t* __restrict__ row_len,
                       const float* __restrict__ v, const float* __restrict__ g,
                       const float* __restrict__ inv_norm, const float* __restrict__ dW, int reps,
                       int64_t rep_stride, const float* __restrict__ db_rep, int64_t db_stride,
                       float* __restrict__ dv, float* __restrict__ dg, float* __restrict__ db_out) {
  const int r = blockIdx.x;
  const int off = row_off[r], len = row_len[r];
  auto dw_at = [&](int k) {
    float s = dW[off + k];
    for (int q = 1; q < reps; ++q) s += dW[q * rep_stride + off + k];
    return s;
  };
  float dot = 0.0f;
  for (int k = threadIdx.x; k < len; k += 64) dot += dw_at(k) * v[off + k];
  dot = wave_sum(dot);
  const float inv = inv_norm[r];
  const float gg = g[r];
  const float c1 = gg * inv;
  const float c2 = gg * dot * inv * inv * inv;
  for (int k = threadIdx.x; k < len; k += 64) dv[off + k] = c1 * dw_at(k) - c2 * v[off + k];
  if (threadIdx.x == 0) {
    dg[r] = dot * inv;
    if (db_out != nullptr) {
      float s = 0.0f;
      for (int q = 0; q < reps; ++q) s += db_rep[q * db_stride + r];
      db_out[r] = s;
    }
  }
}

// ---- all layers of a model in ONE launch: a device table with one record per weight row ----------
// (training re-normalises ~60 small matrices per step; 60 launches of a few hundred waves each are
// launch-bound on the host AND on the GPU)
struct WnRow {            // all fields 8 bytes: the table is built on the host as an int64 tensor
  const float* v;         // source row
  const float* g;         // its gain, or NULL: plain copy (biases and other un-normalised parameters)
  float* w;               // destination row
  float* inv;             // 1/||v|| destination, or NULL
  int64_t len;
};

__global__ void __launch_bounds__(64)
weight_norm_multi_kernel(int n_rows, const WnRow* __restrict__ rows) {
  const WnRow r = rows[blockIdx.x];
  const int len = (int)r.len;
  if (r.g == nullptr) {
    for (int k = threadIdx.x; k < len; k += 64) r.w[k] = r.v[k];
    return;
  }
  float ss = 0.0f;
  for (int k = threadIdx.x; k < len; k += 64) { float a = r.v[k]; ss += a * a; }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float scale = r.g[0] / nrm;
  for (int k = threadIdx.x; k < len; k += 64) r.w[k] = r.v[k] * scale;
  if (r.inv && threadIdx.x == 0) r.inv[0] = 1.0f / nrm;
}

struct WnBwdRow {
  const float* v;         // weight_v row
  const float* g;         // its gain
  const float* inv;       // its 1/||v||
  const float* dW;        // gradient of the effective row, replica 0
  const float* db_rep;    // bias-gradient element of this row, replica 0 (or NULL)
  float* dv;              // outputs: ACCUMULATED (+=) like every other gradient of the step, so that a second
  float* dg;              // backward() without zeroing sums, as torch.autograd does
  float* db_out;          // (or NULL)
  int64_t len, reps, rep_stride, db_stride;
};

__global__ void __launch_bounds__(64)
weight_norm_bwd_multi_kernel(int n_rows, const WnBwdRow* __restrict__ rows) {
  const WnBwdRow r = rows[blockIdx.x];
  const int len = (int)r.len, reps = (int)r.reps;
  auto dw_at = [&](int k) {
    float s = r.dW[k];
    for (int q = 1; q < reps; ++q) s += r.dW[q * r.rep_stride + k];
    return s;
  };
  float dot = 0.0f;
  for (int k = threadIdx.x; k < len; k += 64) dot += dw_at(k) * r.v[k];
  dot = wave_sum(dot);
  const float inv = r.inv[0];
  const float gg = r.g[0];
  const float c1 = gg * inv;
  const float c2 = gg * dot * inv * inv * inv;
  for (int k = threadIdx.x; k < len; k += 64) r.dv[k] += c1 * dw_at(k) - c2 * r.v[k];
  if (threadIdx.x == 0) {
    r.dg[0] += dot * inv;
    if (r.db_out != nullptr) {
      float s = 0.0f;
      for (int q = 0; q < reps; ++q) s += r.db_rep[q * r.db_stride];
      r.db_out[0] += s;
    }
  }
}

// ---------------------------------------------------------------- pooling --------------
__global__ void __launch_bounds__(256)
meansum_pool_kernel(int N, int H, const float* __restrict__ x, int64_t x_ld, const float* __restrict__ maskf,
                    const int32_t* __restrict__ counts, float* __restrict__ pooled, int64_t pooled_ld) {
  __shared__ float red[4][64];
  const int b = blockIdx.x;
  const int col = threadIdx.x & 63;
  const int rg = threadIdx.x >> 6;
  const float cnt = (float)counts[b];
  for (int c0 = 0; c0 < H; c0 += 64) {
    const int c = c0 + col;
    float s = 0.0f;
    if (c < H)
      for (int n = rg; n < N; n += 4) {
        const int64_t p = (int64_t)b * N + n;
        s += x[p * x_ld + c] * maskf[p];
      }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && c < H) {
      const float tot = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
      pooled[(int64_t)b * pooled_ld + c] = tot / cnt;
      pooled[(int64_t)b * pooled_ld + H + c] = tot * 0.01f;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- embedder -------------
__global__ void __launch_bounds__(64)
time_embed_kernel(int Et, const float* __restrict__ t, float t_scalar, const float* __restrict__ fW,
                  const float* __restrict__ lw, const float* __restrict__ lb, float* __restrict__ temb,
                  float* __restrict__ feat_out) {
  extern __shared__ float feat[];                  // Et fourier features
  const int b = blockIdx.x;
  const float tt = t ? t[b] : t_scalar;
  const int hE = Et / 2;
  for (int i = threadIdx.x; i < hE; i += 64) {
    float a = tt * fW[i];
    a = a * 2.0f;
    a = a * 3.14159265358979323846f;
    feat[i] = sinf(a);
    feat[hE + i] = cosf(a);
  }
  __syncthreads();
  if (feat_out)
    for (int i = threadIdx.x; i < 2 * hE; i += 64) feat_out[(int64_t)b * Et + i] = feat[i];
  for (int o = threadIdx.x; o < Et; o += 64) {
    float s = 0.0f;
    for (int k = 0; k < 2 * hE; ++k) s += feat[k] * lw[o * Et + k];
    temb[(int64_t)b * Et + o] = s + lb[o];
  }
}

__global__ void __launch_bounds__(256)
particle_embed_kernel(int P, int Dc, int Ec, int Ed, const float* __restrict__ x, const int64_t* __restrict__ k,
                      const float* __restrict__ maskf, const float* __restrict__ cW,
                      const float* __restrict__ cb, const float* __restrict__ table,
                      float* __restrict__ cont, float* __restrict__ disc) {
  // one lane per (particle, output feature); consecutive lanes -> consecutive features
  const int F = Ec + Ed;
  const int64_t total = (int64_t)P * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i / F);
    const int f = (int)(i - (int64_t)p * F);
    const float m = maskf[p];
    if (f < Ec) {
      float s = 0.0f;
      for (int d = 0; d < Dc; ++d) s += x[(int64_t)p * Dc + d] * cW[f * Dc + d];
      cont[(int64_t)p * Ec + f] = (s + cb[f]) * m;
    } else {
      const int e = f - Ec;
      disc[(int64_t)p * Ed + e] = truncf(table[(int)k[p] * Ed + e] * m);      // quirk Q1
    }
  }
}

template <int WAVES, int NT, int BK, bool ALIGNED>
static void launch_linear(dim3 grid, hipStream_t st, int M, int rpj, const Segs& sg, int K, const float* maskf,
                          const float* W, int64_t w_ld, const float* bias, int Nout, int act, const float* residual,
                          int64_t res_ld, const float* row_scale, float* Y, int64_t y_ld, float* pre_act,
                          const Post& post) {
  hipLaunchKernelGGL((linear_cat_kernel<WAVES, NT, BK, ALIGNED>), grid, dim3(WAVES * 64), 0, st, M, rpj, sg, K,
                     maskf, W, w_ld, bias, Nout, act, residual, res_ld, row_scale, Y, y_ld, pre_act, post);
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}
// large-M kernel choice (A/B switch for tests and profiles): MMB_LINEAR_PATH = tiled (0) | stream (1: row streaming
// on the fp32 MFMA) | x6 (2, the default: row streaming, six bf16-plane products)
static int linear_path() {
  const char* e = getenv("MMB_LINEAR_PATH");
  if (e && !strcmp(e, "tiled")) return 0;
  if (e && !strcmp(e, "stream")) return 1;
  return 2;
}

}  // namespace mmb

using namespace mmb;

// shared body of mmb_linear_cat / mmb_linear_cat_ln.  ln: a LayerNorm second stage is wanted (post.ln_*, post.Y2);
// *ln_done tells whether the launched kernel did it (row-streaming kernel, whole rows per wave) -- if not, the
// first stage went to Y (or, without a Y, to post.Y2 as scratch) and the caller runs mmb_layernorm_fwd on it.
static int linear_cat_impl(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr, const int* seg_width,
                           const int* seg_kind, const int* seg_ld, const float* maskf, const float* W, int64_t w_ld,
                           const float* bias, int Nout, int act, const float* residual, int64_t res_ld,
                           const float* row_scale, float* Y, int64_t y_ld, float* pre_act, Post post, bool* ln_done,
                           void* stream) {
  const bool want_ln = post.ln_gamma != nullptr;
  if (ln_done) *ln_done = false;
  MMB_REQUIRE(rows_per_jet > 0, "mmb_linear_cat: rows_per_jet must be > 0");
  MMB_REQUIRE(n_seg >= 1 && n_seg <= 4, "mmb_linear_cat: n_seg=%d outside [1,4]", n_seg);
  MMB_REQUIRE(seg_ptr && seg_width && seg_kind && seg_ld && W, "mmb_linear_cat: NULL argument");
  MMB_REQUIRE(act >= 0 && act <= 4, "mmb_linear_cat: bad activation %d", act);
  Segs sg;
  sg.n = n_seg;
  int K = 0;
  bool al16 = true, al8 = true;
  for (int i = 0; i < 4; ++i) {
    if (i < n_seg) {
      MMB_REQUIRE(seg_ptr[i] && seg_width[i] > 0, "mmb_linear_cat: segment %d empty", i);
      MMB_REQUIRE(seg_kind[i] >= 0 && seg_kind[i] <= 2, "mmb_linear_cat: segment %d bad kind %d", i, seg_kind[i]);
      MMB_REQUIRE(seg_ld[i] >= seg_width[i], "mmb_linear_cat: segment %d ld < width", i);
      if (seg_kind[i] == MMB_SEG_JET_MASK) MMB_REQUIRE(maskf, "mmb_linear_cat: JET_MASK segment needs maskf");
      sg.s[i] = Seg{(const float*)seg_ptr[i], seg_width[i], seg_kind[i], seg_ld[i], K};
      K += seg_width[i];
      bool a4 = (seg_ld[i] % 4 == 0) && ((reinterpret_cast<uintptr_t>(seg_ptr[i]) & 15) == 0);
      al16 = al16 && a4 && (seg_width[i] % 16 == 0);
      al8 = al8 && a4 && (seg_width[i] % 8 == 0);
    } else {
      sg.s[i] = Seg{nullptr, 0, 0, 0, 0};
    }
  }
  MMB_REQUIRE(w_ld >= K, "mmb_linear_cat: w_ld=%lld < K=%d", (long long)w_ld, K);
  bool w_al = (w_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  al16 = al16 && w_al;
  al8 = al8 && w_al;
  hipStream_t st = as_stream(stream);
  const int lpath = linear_path();
  bool ln_stream = false;
  if (want_ln) {       // only the bf16x6 row-streaming kernel normalises, and only with the whole row in one wave
    const int KP2 = (K + 63) & ~63;
    const char* e = getenv("MMB_LN_FUSE");       // A/B switch: "0" = always linear kernel + mmb_layernorm_fwd
    ln_stream = !(e && e[0] == '0') && M > 4096 && al16 && lpath == 2 && (Nout == 32 || Nout == 64 || Nout == 128) &&
                (size_t)3 * Nout * (KP2 + 8) * sizeof(unsigned short) <= 160 * 1024;
    if (!ln_stream) {  // first stage only: into Y, or into the LN output as scratch (the caller normalises in place)
      if (!Y) { Y = post.Y2; y_ld = post.y2_ld; }
      post.Y2 = nullptr;
      post.ln_gamma = post.ln_beta = nullptr;
      post.ln_stats = nullptr;
    }
  }
  // per-jet layers (one ROW segment, few rows): split-K kernel, one 32x32 tile per block
  const bool al4 = (seg_ld[0] % 4 == 0) && ((reinterpret_cast<uintptr_t>(seg_ptr[0]) & 15) == 0) && w_al;
  if (M <= 4096 && n_seg == 1 && seg_kind[0] == MMB_SEG_ROW && K % 4 == 0 && K <= 8 * 4 * 12 && al4) {
    const int rl4 = (K + 31) / 32;                 // float4 per lane: 4 waves x 2 halves x 4 floats
    dim3 g2((M + 31) / 32, (Nout + 31) / 32);
    const float* Xp = (const float*)seg_ptr[0];
#define MMB_SK(R_) hipLaunchKernelGGL((linear_splitk_kernel<R_>), g2, dim3(256), 0, st, M, Xp, (int64_t)seg_ld[0], K, W, w_ld, bias, Nout, act, residual, res_ld, row_scale, Y, y_ld, pre_act, post)
    switch (rl4) {
      case 1: MMB_SK(1); break;
      case 2: MMB_SK(2); break;
      case 3: MMB_SK(3); break;
      case 4: MMB_SK(4); break;
      case 5: MMB_SK(5); break;
      case 6: MMB_SK(6); break;
      case 7: MMB_SK(7); break;
      case 8: MMB_SK(8); break;
      case 9: MMB_SK(9); break;
      case 10: MMB_SK(10); break;
      case 11: MMB_SK(11); break;
      default: MMB_SK(12); break;
    }
#undef MMB_SK
    return check_launch("mmb_linear_cat");
  }
  const bool small_m = M <= 4096;                // per-jet rows: one wave per block for parallelism
  if (!small_m && al16 && lpath != 0) {
    // row-streaming kernels: the widest column block (tiles of 32 columns) that pads Nout least and fits LDS
    const bool x6 = lpath == 2;
    const int KP = (K + 31) & ~31, KP2 = (K + 63) & ~63;
    auto lds_of = [&](int c) {
      return x6 ? (size_t)3 * 32 * c * (KP2 + 8) * sizeof(unsigned short) : (size_t)32 * c * (KP + 4) * sizeof(float);
    };
    int snt = 0;
    int best_pad = 1 << 30;
    int cmax = x6 ? 4 : 2;                       // (fp32 MFMA: 4 tiles per wave do not fit 128 registers)
    if (const char* e = getenv("MMB_X6_NT")) cmax = atoi(e) >= 1 && atoi(e) <= cmax ? atoi(e) : cmax;   // tuning knob
    for (int c = cmax; c >= 1; c >>= 1) {
      const int pad = (Nout + 32 * c - 1) / (32 * c) * (32 * c);
      if (lds_of(c) <= 160 * 1024 && pad < best_pad) {
        best_pad = pad;
        snt = c;
      }
    }
    if (ln_stream) {                             // whole rows in one wave: Nout = 32 * snt exactly, one column block
      snt = Nout / 32;
      best_pad = Nout;
    }
    if (snt) {
      if (ln_stream && ln_done) *ln_done = true;
      const int gy = best_pad / (32 * snt), ntiles = (M + 31) / 32, cus = cu_count();
      int gx = cus / gy / 8 * 8;                 // same-x blocks (same rows, other columns) land on one XCD's L2
      if (gx < 8) gx = cus / gy > 0 ? cus / gy : 1;
      if (gx > (ntiles + 3) / 4) gx = (ntiles + 3) / 4;
      const size_t lds = lds_of(snt);
      dim3 g3(gx, gy);
#define MMB_STREAM(KERNEL, THREADS_)                                                                             \
  do {                                                                                                           \
    static PerDeviceOnce lds_limit;          /* per instantiation AND per device (the runtime keeps it per device) */ \
    MMB_REQUIRE(lds_limit.run([&] {                                                                              \
                  return hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                             160 * 1024) == hipSuccess; }),                                      \
                "mmb_linear_cat: cannot raise the dynamic LDS limit");                                           \
    hipLaunchKernelGGL(KERNEL, g3, dim3(THREADS_), lds, st, M, rows_per_jet, sg, K, maskf, W, w_ld, bias, Nout,  \
                       act, residual, res_ld, row_scale, Y, y_ld, pre_act, post);                                \
  } while (0)
      if (x6) {                                  // wider column blocks need more registers: fewer waves per block
        const bool simple = n_seg == 1 && seg_kind[0] == MMB_SEG_ROW;
#define MMB_X6(NT_, TH_)                                                                               \
  do {                                                                                                 \
    if (ln_stream && simple) MMB_STREAM((linear_stream_x6_kernel<NT_, TH_, true, true>), TH_);         \
    else if (ln_stream) MMB_STREAM((linear_stream_x6_kernel<NT_, TH_, false, true>), TH_);             \
    else if (simple) MMB_STREAM((linear_stream_x6_kernel<NT_, TH_, true, false>), TH_);                \
    else MMB_STREAM((linear_stream_x6_kernel<NT_, TH_, false, false>), TH_);                           \
  } while (0)
        if (snt == 4) MMB_X6(4, 512);
        else if (snt == 2) MMB_X6(2, 768);
        else MMB_X6(1, 768);
#undef MMB_X6
      } else {
        if (snt == 2) MMB_STREAM((linear_stream_kernel<2>), 1024);
        else MMB_STREAM((linear_stream_kernel<1>), 1024);
      }
#undef MMB_STREAM
      return check_launch("mmb_linear_cat");
    }
  }
  const int nt = Nout <= 32 ? 1 : (Nout <= 64 ? 2 : 4);
  const int bm = small_m ? 32 : 128;
  dim3 grid((M + bm - 1) / bm, (Nout + nt * 32 - 1) / (nt * 32));
#define MMB_LAUNCH(WV, NT_, BK_, AL)                                                                           \
  launch_linear<WV, NT_, BK_, AL>(grid, st, M, rows_per_jet, sg, K, maskf, W, w_ld, bias, Nout, act, residual, \
                                  res_ld, row_scale, Y, y_ld, pre_act, post)
#define MMB_PICK_NT(WV, BK_, AL)                \
  do {                                          \
    if (nt == 1) MMB_LAUNCH(WV, 1, BK_, AL);    \
    else if (nt == 2) MMB_LAUNCH(WV, 2, BK_, AL); \
    else MMB_LAUNCH(WV, 4, BK_, AL);            \
  } while (0)
  if (small_m) {
    if (al16) MMB_PICK_NT(1, 32, true);
    else if (al8) MMB_PICK_NT(1, 16, true);
    else MMB_PICK_NT(1, 16, false);
  } else {
    if (al16) MMB_PICK_NT(4, 32, true);
    else if (al8) MMB_PICK_NT(4, 16, true);
    else MMB_PICK_NT(4, 16, false);
  }
#undef MMB_PICK_NT
#undef MMB_LAUNCH
  return check_launch("mmb_linear_cat");
}

extern "C" int mmb_linear_cat(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr, const int* seg_width,
                              const int* seg_kind, const int* seg_ld, const float* maskf, const float* W,
                              int64_t w_ld, const float* bias, int Nout, int act, const float* residual,
                              int64_t res_ld, const float* row_scale, float* Y, int64_t y_ld, float* pre_act,
                              int post_act, float drop_p, uint64_t seed, uint64_t offset, const float* res2,
                              int64_t res2_ld, float* Y2, int64_t y2_ld, void* stream) {
  MMB_REQUIRE(M > 0 && Nout > 0, "mmb_linear_cat: empty shape M=%d Nout=%d", M, Nout);
  MMB_REQUIRE(Y || Y2, "mmb_linear_cat: no output given");
  MMB_REQUIRE(post_act >= 0 && post_act <= 3, "mmb_linear_cat: bad post activation %d", post_act);   // (SELU: first stage only)
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_linear_cat: dropout p=%f outside [0,1)", drop_p);
  Post post{Y2, y2_ld, res2, res2_ld, post_act, drop_p, seed, offset, nullptr, nullptr, 0.0f, nullptr, nullptr, 0, MMB_ACT_NONE};
  return linear_cat_impl(M, rows_per_jet, n_seg, seg_ptr, seg_width, seg_kind, seg_ld, maskf, W, w_ld, bias, Nout, act,
                         residual, res_ld, row_scale, Y, y_ld, pre_act, post, nullptr, stream);
}

extern "C" int mmb_linear_cat_gate(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr,
                                   const int* seg_width, const int* seg_kind, const int* seg_ld, const float* maskf,
                                   const float* W, int64_t w_ld, int Nout, int gate_act, const float* gate,
                                   int64_t gate_ld, float* Y, int64_t y_ld, void* stream) {
  MMB_REQUIRE(M > 0 && Nout > 0, "mmb_linear_cat_gate: empty shape M=%d Nout=%d", M, Nout);
  MMB_REQUIRE(gate_act >= 1 && gate_act <= 4, "mmb_linear_cat_gate: bad activation %d", gate_act);
  MMB_REQUIRE(gate && Y && gate_ld >= Nout && y_ld >= Nout, "mmb_linear_cat_gate: NULL gate / output or row stride < Nout");
  Post post{nullptr, 0, nullptr, 0, MMB_ACT_NONE, 0.0f, 0, 0, nullptr, nullptr, 0.0f, nullptr, gate, gate_ld, gate_act};
  return linear_cat_impl(M, rows_per_jet, n_seg, seg_ptr, seg_width, seg_kind, seg_ld, maskf, W, w_ld, nullptr, Nout,
                         MMB_ACT_NONE, nullptr, 0, nullptr, Y, y_ld, nullptr, post, nullptr, stream);
}

extern "C" int mmb_linear_cat_ln(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr, const int* seg_width,
                                 const int* seg_kind, const int* seg_ld, const float* maskf, const float* W,
                                 int64_t w_ld, const float* bias, int Nout, int act, const float* residual,
                                 int64_t res_ld, float* Y1, int64_t y1_ld, const float* gamma, const float* beta,
                                 float eps, float* Yln, int64_t yln_ld, float* stats, void* stream) {
  MMB_REQUIRE(M > 0 && Nout > 0, "mmb_linear_cat_ln: empty shape M=%d Nout=%d", M, Nout);
  MMB_REQUIRE(gamma && beta && Yln && yln_ld >= Nout, "mmb_linear_cat_ln: NULL gamma / beta / output");
  MMB_REQUIRE(!Y1 || y1_ld >= Nout, "mmb_linear_cat_ln: y1 row stride < Nout");
  Post post{Yln, yln_ld, nullptr, 0, MMB_ACT_NONE, 0.0f, 0, 0, gamma, beta, eps, stats, nullptr, 0, MMB_ACT_NONE};
  bool done = false;
  if (int rc = linear_cat_impl(M, rows_per_jet, n_seg, seg_ptr, seg_width, seg_kind, seg_ld, maskf, W, w_ld, bias, Nout,
                               act, residual, res_ld, nullptr, Y1, y1_ld, nullptr, post, &done, stream))
    return rc;
  if (done) return MMB_OK;
  // this shape is not served by the fused epilogue: the first stage is in Y1 (or in Yln), normalise it (in place)
  return mmb_layernorm_fwd(M, Nout, Y1 ? Y1 : Yln, Y1 ? y1_ld : yln_ld, gamma, beta, eps, Yln, yln_ld, stats, stream);
}

extern "C" int mmb_weight_norm(int n_rows, const int32_t* row_off, const int32_t* row_len, const float* v,
                               const float* g, float* W, float* inv_norm, void* stream) {
  MMB_REQUIRE(n_rows > 0, "mmb_weight_norm: n_rows=%d", n_rows);
  MMB_REQUIRE(row_off && row_len && v && g && W, "mmb_weight_norm: NULL argument");
  hipLaunchKernelGGL(weight_norm_kernel, dim3(n_rows), dim3(64), 0, as_stream(stream), n_rows, row_off, row_len, v,
                     g, W, inv_norm);
  return check_launch("mmb_weight_norm");
}

extern "C" int mmb_weight_norm_multi(int n_rows, const void* rows, void* stream) {
  MMB_REQUIRE(n_rows > 0 && rows, "mmb_weight_norm_multi: empty table");
  hipLaunchKernelGGL(weight_norm_multi_kernel, dim3(n_rows), dim3(64), 0, as_stream(stream), n_rows,
                     reinterpret_cast<const WnRow*>(rows));
  return check_launch("mmb_weight_norm_multi");
}

// Parameter block of the fused kernels from the flat effective-weight buffer, one launch: the fp32 part is a gather
// (+ a second gather for the layers that see the time embedding twice), the MFMA operand part a gather followed by
// the exact three-plane bf16 split of fused_common.h (mma_x): fragment f of 512 values -> [f][plane][512] bf16.
__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ flat, const int64_t* __restrict__ g1,
                                                          const int64_t* __restrict__ g2, int64_t n_main,
                                                          const int64_t* __restrict__ gx, int64_t n_x,
                                                          float* __restrict__ dst) {
  const int64_t n_main4 = n_main >> 2, n_x8 = n_x >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_main4) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = flat[g1[4 * i + e]];
      if (g2) v[e] += flat[g2[4 * i + e]];
    }
    reinterpret_cast<float4*>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
  } else if (i - n_main4 < n_x8) {
    const int64_t j = (i - n_main4) * 8, f = j >> 9, e0 = j & 511;
    uint32_t pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = flat[gx[j + 2 * e]], b = flat[gx[j + 2 * e + 1]];
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      auto pk = [](float x, float y) { const bf16x2_t t = {(__bf16)x, (__bf16)y}; return __builtin_bit_cast(uint32_t, t); };
      const uint32_t h = pk(a, b);
      const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xFFFF0000u);
      const uint32_t m = pk(ra, rb);
      const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xFFFF0000u);
      pl[0][e] = h;
      pl[1][e] = m;
      pl[2][e] = pk(sa, sb);
    }
    uint16_t* out = reinterpret_cast<uint16_t*>(dst + n_main);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<uint4*>(out + (f * 3 + q) * 512 + e0) = make_uint4(pl[q][0], pl[q][1], pl[q][2], pl[q][3]);
  }
}

extern "C" int mmb_pack_params(const float* flat, const int64_t* gather, const int64_t* gather2, int64_t n_main,
                               const int64_t* gather_x, int64_t n_x, float* dst, void* stream) {
  MMB_REQUIRE(flat && gather && gather_x && dst, "mmb_pack_params: null pointer");
  MMB_REQUIRE(n_main >= 0 && n_main % 4 == 0, "mmb_pack_params: the fp32 part must be a multiple of 4 floats");
  MMB_REQUIRE(n_x > 0 && n_x % 512 == 0, "mmb_pack_params: the operand part is whole fragments of 512 values");
  MMB_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 15) == 0, "mmb_pack_params: dst must be 16-byte aligned");
  const int64_t threads = n_main / 4 + n_x / 8;
  hipLaunchKernelGGL(pack_params_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), flat,
                     gather, gather2, n_main, gather_x, n_x, dst);
  return check_launch("mmb_pack_params");
}

extern "C" int mmb_weight_norm_bwd_multi(int n_rows, const void* rows, void* stream) {
  MMB_REQUIRE(n_rows > 0 && rows, "mmb_weight_norm_bwd_multi: empty table");
  hipLaunchKernelGGL(weight_norm_bwd_multi_kernel, dim3(n_rows), dim3(64), 0, as_stream(stream), n_rows,
                     reinterpret_cast<const WnBwdRow*>(rows));
  return check_launch("mmb_weight_norm_bwd_multi");
}

extern "C" int mmb_weight_norm_bwd_rep(int n_rows, const int32_t* row_off, const int32_t* row_len, const float* v,
                                       const float* g, const float* inv_norm, const float* dW, int replicas,
                                       int64_t rep_stride, const float* db_rep, int64_t db_stride, float* dv,
                                       float* dg, float* db_out, void* stream) {
  MMB_REQUIRE(n_rows > 0, "mmb_weight_norm_bwd: n_rows=%d", n_rows);
  MMB_REQUIRE(row_off && row_len && v && g && inv_norm && dW && dv && dg, "mmb_weight_norm_bwd: NULL argument");
  MMB_REQUIRE(replicas >= 1 && replicas <= 64, "mmb_weight_norm_bwd: replicas=%d outside [1,64]", replicas);
  MMB_REQUIRE((db_out == nullptr) == (db_rep == nullptr), "mmb_weight_norm_bwd: db_rep and db_out go together");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(n_rows), dim3(64), 0, as_stream(stream), n_rows, row_off,
                     row_len, v, g, inv_norm, dW, replicas, rep_stride, db_rep, db_stride, dv, dg, db_out);
  return check_launch("mmb_weight_norm_bwd");
}

extern "C" int mmb_weight_norm_bwd(int n_rows, const int32_t* row_off, const int32_t* row_len, const float* v,
                                   const float* g, const float* inv_norm, const float* dW, float* dv, float* dg,
                                   void* stream) {
  return mmb_weight_norm_bwd_rep(n_rows, row_off, row_len, v, g, inv_norm, dW, 1, 0, nullptr, 0, dv, dg, nullptr,
                                 stream);
}

extern "C" int mmb_meansum_pool(int B, int N, int H, const float* x, int64_t x_ld, const float* maskf,
                                const int32_t* counts, float* pooled, int64_t pooled_ld, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && H > 0, "mmb_meansum_pool: empty shape");
  MMB_REQUIRE(x && maskf && counts && pooled, "mmb_meansum_pool: NULL argument");
  MMB_REQUIRE(x_ld >= H && pooled_ld >= 2 * H, "mmb_meansum_pool: leading dimensions too small");
  hipLaunchKernelGGL(meansum_pool_kernel, dim3(B), dim3(256), 0, as_stream(stream), N, H, x, x_ld, maskf, counts,
                     pooled, pooled_ld);
  return check_launch("mmb_meansum_pool");
}

extern "C" int mmb_embed(int B, int N, int Dc, int V, int Et, int Ec, int Ed, const float* t, float t_scalar,
                         const float* x, const int64_t* k, const float* maskf, const float* fourier_W,
                         const float* time_W, const float* time_b, const float* cont_W, const float* cont_b,
                         const float* table, float* temb, float* cont, float* disc, float* fourier_feat,
                         void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_embed: empty shape");
  MMB_REQUIRE(Et > 0 && Et % 2 == 0, "mmb_embed: dim_emb_time=%d must be even and > 0", Et);
  MMB_REQUIRE(fourier_W && time_W && time_b && temb && maskf, "mmb_embed: NULL argument");
  const bool has_c = cont != nullptr, has_d = disc != nullptr;
  if (has_c) MMB_REQUIRE(x && cont_W && cont_b && Dc > 0 && Ec > 0, "mmb_embed: continuous mode needs x,cont_W,cont_b");
  if (has_d) MMB_REQUIRE(k && table && Ed > 0 && V > 0, "mmb_embed: discrete mode needs k,table");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(time_embed_kernel, dim3(B), dim3(64), Et * sizeof(float), st, Et, t, t_scalar, fourier_W,
                     time_W, time_b, temb, fourier_feat);
  int rc = check_launch("mmb_embed(time)");
  if (rc) return rc;
  if (has_c || has_d) {
    const int P = B * N;
    const int ec = has_c ? Ec : 0, ed = has_d ? Ed : 0;
    hipLaunchKernelGGL(particle_embed_kernel, dim3(stream_grid((int64_t)P * (ec + ed), 256)), dim3(256), 0, st, P,
                       Dc, ec, ed, x, k, maskf, cont_W, cont_b, table, cont, disc);
    rc = check_launch("mmb_embed(particles)");
  }
  return rc;
}
