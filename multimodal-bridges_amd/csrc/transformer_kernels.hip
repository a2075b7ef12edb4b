// Transformer set encoder (reference: encoders/particle_transformer.py:8-50, i.e. torch's
// nn.TransformerEncoderLayer, post-LN, batch_first): masked multi-head attention over one jet's particles
// (forward + backward) and LayerNorm (forward + backward).  The Linear layers around them are mmb_linear_cat.
//
// A jet has N <= 150 particles and a head is 4..64 wide, so one (jet, head) pair is a 150 x 150 x 16 problem:
// K and V of the jet's VALID keys are compacted into LDS once (the reference passes src_key_padding_mask =
// mask > 0, quirk Q4, so the valid keys are the PADDED slots; which side is valid is an argument here), every
// thread owns one query and streams the keys with an online softmax (LDS reads are wave-wide broadcasts).
// A query with no valid key gets zeros, as torch's need_weights=False path does (safe softmax).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace mmb {

// compact the indices of the valid keys of jet b into idx[] (LDS); returns their number.  One wave does the scan.
__device__ inline int compact_keys(const float* __restrict__ maskf, int N, int valid_is_zero, int* idx, int* nk_sh) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int base = 0;
    for (int c = 0; c < N; c += 64) {
      const int n = c + lane;
      bool ok = false;
      if (n < N) ok = ((maskf[n] == 0.0f) == (valid_is_zero != 0));
      const unsigned long long bal = __ballot(ok);
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (ok) idx[pos] = n;
      base += __popcll(bal);
    }
    if (lane == 0) *nk_sh = base;
  }
  __syncthreads();
  return *nk_sh;
}

// Dropout on the attention probabilities (torch's MultiheadAttention: dropout(softmax(.)) @ V, active in training;
// nn.TransformerEncoderLayer(dropout=p) at particle_transformer.py:29-35).  The keep decision of (jet b, head h, query q,
// j-th VALID key) is word j & 3 of Philox4(seed, offset, index = ((b*nh + h)*N + q) * ceil(N/4) + (j >> 2)): a thread
// that walks the keys of one query makes one call per four keys; the reverse pass regenerates the same bits.
struct AttnDrop {
  float p, inv_keep;
  uint64_t seed, offset;
};
constexpr uint32_t kAttnDropTag = 0xA77Eu;

__device__ __forceinline__ uint32_t attn_keep4(const AttnDrop& d, int64_t bh, int N, int q, int g) {
  uint32_t r[4];
  philox4(d.seed, d.offset, (uint64_t)(bh * N + q) * (uint64_t)((N + 3) >> 2) + (uint64_t)g, kAttnDropTag, r);
  return (u01(r[0]) >= d.p ? 1u : 0u) | (u01(r[1]) >= d.p ? 2u : 0u) | (u01(r[2]) >= d.p ? 4u : 0u) |
         (u01(r[3]) >= d.p ? 8u : 0u);
}
// 1/(1-p) if the j-th valid key of query q is kept, else 0
__device__ __forceinline__ float attn_keep1(const AttnDrop& d, int64_t bh, int N, int q, int j) {
  return ((attn_keep4(d, bh, N, q, j >> 2) >> (j & 3)) & 1u) ? d.inv_keep : 0.0f;
}

// Uniform keys.  The reference restricts the keys to the PADDED slots (quirk Q4), and in its model every padded slot of
// a jet carries the same row in every layer (all inputs are multiplied by the mask: embedder.py:63-97), so in eval mode
// all valid keys of a (jet, head) have bit-identical K and V rows: the scores of a query are then all equal, the softmax
// is uniform and the output is that one V row whatever the query.  The forward kernels TEST this on the staged rows
// (nk * dh compares against row 0, one block-wide vote) and, when it holds and no log-sum-exp is asked for, write the
// row out for every query instead of walking the keys; any other input takes the general path in the same launch.
// K, V rows of pitch P in LDS; returns true for nk > 0 identical rows.
__device__ inline bool keys_uniform(const float* Ks, const float* Vs, int nk, int dh, int P) {
  bool same = nk > 0;
  for (int i = threadIdx.x; i < nk * dh; i += blockDim.x) {
    const int j = i / dh, d = i - j * dh;
    same = same && __float_as_uint(Ks[j * P + d]) == __float_as_uint(Ks[d]) &&
           __float_as_uint(Vs[j * P + d]) == __float_as_uint(Vs[d]);
  }
  return __syncthreads_and(same ? 1 : 0) != 0;
}
__device__ inline void broadcast_row(const float* Vs, int N, int dh, float* __restrict__ o, int64_t out_ld) {
  for (int i = threadIdx.x; i < N * dh; i += blockDim.x) {
    const int q = i / dh, d = i - q * dh;
    o[(int64_t)q * out_ld + d] = Vs[d];
  }
}
static bool attn_uniform_enabled() {
  static const bool on = [] { const char* e = getenv("MMB_ATTN_UNIFORM"); return !(e && e[0] == '0'); }();
  return on;
}

// out[q, h*dh + d] = sum_j softmax_j(scale * q.k_j) v_j over the valid keys j;  lse[(b*nh+h)*N + q] = log sum exp
template <int DHP, bool DROP>
__global__ void __launch_bounds__(256)
attn_fwd_kernel(int N, int H, int nh, int dh, const float* __restrict__ qkv, int64_t ld,
                const float* __restrict__ maskf, int valid_is_zero, float scale, float* __restrict__ out,
                int64_t out_ld, float* __restrict__ lse, AttnDrop drop, int uniform_ok) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / nh, h = blockIdx.x - b * nh;
  int* idx = reinterpret_cast<int*>(sm);
  int* nk_sh = idx + N;
  float* Ks = sm + ((N + 1 + 3) & ~3);
  float* Vs = Ks + (size_t)N * DHP;
  const int nk = compact_keys(maskf + (int64_t)b * N, N, valid_is_zero, idx, nk_sh);
  const float* base = qkv + (int64_t)b * N * ld + h * dh;
  for (int i = threadIdx.x; i < nk * DHP; i += blockDim.x) {
    const int j = i / DHP, d = i - j * DHP;
    const int64_t row = (int64_t)idx[j] * ld;
    Ks[i] = d < dh ? base[row + H + d] : 0.0f;
    Vs[i] = d < dh ? base[row + 2 * H + d] : 0.0f;
  }
  __syncthreads();
  if constexpr (!DROP) {
    if (uniform_ok && lse == nullptr && keys_uniform(Ks, Vs, nk, dh, DHP)) {
      broadcast_row(Vs, N, dh, out + (int64_t)b * N * out_ld + h * dh, out_ld);
      return;
    }
  }
  for (int q = threadIdx.x; q < N; q += blockDim.x) {
    float qv[DHP], acc[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) {
      qv[d] = d < dh ? base[(int64_t)q * ld + d] * scale : 0.0f;
      acc[d] = 0.0f;
    }
    float m = -INFINITY, l = 0.0f;
    uint32_t kb = 0;
    for (int j = 0; j < nk; ++j) {
      const float* kr = Ks + j * DHP;
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) s = fmaf(qv[d], kr[d], s);
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn);       // exp(-inf) = 0 on the first key
      const float p = __expf(s - mn);
      l = fmaf(l, alpha, p);                    // the softmax normaliser is that of the UNdropped probabilities
      float pk = p;
      if constexpr (DROP) {
        if ((j & 3) == 0) kb = attn_keep4(drop, blockIdx.x, N, q, j >> 2);
        pk = ((kb >> (j & 3)) & 1u) ? p * drop.inv_keep : 0.0f;
      }
      const float* vr = Vs + j * DHP;
#pragma unroll
      for (int d = 0; d < DHP; ++d) acc[d] = fmaf(acc[d], alpha, pk * vr[d]);
      m = mn;
    }
    const float inv = nk > 0 ? 1.0f / l : 0.0f;
    float* o = out + ((int64_t)b * N + q) * out_ld + h * dh;
#pragma unroll
    for (int d = 0; d < DHP; ++d)
      if (d < dh) o[d] = acc[d] * inv;
    if (lse) lse[((int64_t)b * nh + h) * N + q] = nk > 0 ? m + __logf(l) : 0.0f;
  }
}

// Reverse of the above.  With P = softmax, D_q = dO_q . O_q:
//   dV_j = sum_q P_qj dO_q ; dS_qj = P_qj (dO_q . V_j - D_q) ; dQ_q = scale sum_j dS_qj K_j ; dK_j = scale sum_q dS_qj Q_q
// Phase A: thread = query (keys from LDS).  Phase B: thread = valid key, queries (Q, dO, lse, D) from the SAME LDS
// area re-filled.  Keys that are not valid get zero dK, dV.
// With dropout (keep factor c_qj in {0, 1/(1-p)}): O = sum_j c_qj P_qj V_j, so dV_j = sum_q c_qj P_qj dO_q and
// dS_qj = P_qj (c_qj dO_q . V_j - D_q) with the same D_q = dO_q . O_q.
template <int DHP, bool DROP>
__global__ void __launch_bounds__(256)
attn_bwd_kernel(int N, int H, int nh, int dh, const float* __restrict__ qkv, int64_t ld,
                const float* __restrict__ maskf, int valid_is_zero, float scale, const float* __restrict__ out,
                int64_t out_ld, const float* __restrict__ dout, int64_t dout_ld, const float* __restrict__ lse,
                float* __restrict__ dqkv, int64_t dld, AttnDrop drop) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / nh, h = blockIdx.x - b * nh;
  int* idx = reinterpret_cast<int*>(sm);
  int* nk_sh = idx + N;
  float* Ls = sm + ((N + 1 + 3) & ~3);
  float* Ds = Ls + N;
  float* A0 = Ds + ((N + 3) & ~3);           // K (phase A) / Q (phase B)
  float* A1 = A0 + (size_t)N * DHP;          // V (phase A) / dO (phase B)
  const int nk = compact_keys(maskf + (int64_t)b * N, N, valid_is_zero, idx, nk_sh);
  const float* base = qkv + (int64_t)b * N * ld + h * dh;
  float* dbase = dqkv + (int64_t)b * N * dld + h * dh;
  const float* lrow = lse + ((int64_t)b * nh + h) * N;
  for (int i = threadIdx.x; i < nk * DHP; i += blockDim.x) {
    const int j = i / DHP, d = i - j * DHP;
    const int64_t row = (int64_t)idx[j] * ld;
    A0[i] = d < dh ? base[row + H + d] : 0.0f;
    A1[i] = d < dh ? base[row + 2 * H + d] : 0.0f;
  }
  __syncthreads();
  // ---- phase A: dQ, D
  for (int q = threadIdx.x; q < N; q += blockDim.x) {
    float qv[DHP], go[DHP], dq[DHP];
    const float* orow = out + ((int64_t)b * N + q) * out_ld + h * dh;
    const float* grow = dout + ((int64_t)b * N + q) * dout_ld + h * dh;
    float D = 0.0f;
#pragma unroll
    for (int d = 0; d < DHP; ++d) {
      qv[d] = d < dh ? base[(int64_t)q * ld + d] * scale : 0.0f;
      go[d] = d < dh ? grow[d] : 0.0f;
      if (d < dh) D = fmaf(go[d], orow[d], D);
      dq[d] = 0.0f;
    }
    const float L = lrow[q];
    uint32_t kb = 0;
    for (int j = 0; j < nk; ++j) {
      const float* kr = A0 + j * DHP;
      const float* vr = A1 + j * DHP;
      float s = 0.0f, dp = 0.0f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) {
        s = fmaf(qv[d], kr[d], s);
        dp = fmaf(go[d], vr[d], dp);
      }
      if constexpr (DROP) {
        if ((j & 3) == 0) kb = attn_keep4(drop, blockIdx.x, N, q, j >> 2);
        dp = ((kb >> (j & 3)) & 1u) ? dp * drop.inv_keep : 0.0f;
      }
      const float ds = __expf(s - L) * (dp - D);
#pragma unroll
      for (int d = 0; d < DHP; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
    }
    Ls[q] = L;
    Ds[q] = D;
#pragma unroll
    for (int d = 0; d < DHP; ++d)
      if (d < dh) dbase[(int64_t)q * dld + d] = dq[d] * scale;
  }
  // ---- phase B: dK, dV.  Each valid key keeps its own K, V rows in registers; the LDS area is re-filled with Q, dO
  for (int j0 = 0; j0 < (nk > 0 ? nk : 1); j0 += blockDim.x) {       // (executed at least once: every thread reaches the barriers)
    const int j = j0 + threadIdx.x;
    float kv[DHP], vv[DHP], dk[DHP], dv[DHP];
    const int64_t krow = j < nk ? (int64_t)idx[j] * ld : 0;
#pragma unroll
    for (int d = 0; d < DHP; ++d) {
      kv[d] = (j < nk && d < dh) ? base[krow + H + d] * scale : 0.0f;
      vv[d] = (j < nk && d < dh) ? base[krow + 2 * H + d] : 0.0f;
      dk[d] = dv[d] = 0.0f;
    }
    __syncthreads();
    if (j0 == 0)
      for (int i = threadIdx.x; i < N * DHP; i += blockDim.x) {
        const int q = i / DHP, d = i - q * DHP;
        A0[i] = d < dh ? base[(int64_t)q * ld + d] : 0.0f;
        A1[i] = d < dh ? dout[((int64_t)b * N + q) * dout_ld + h * dh + d] : 0.0f;
      }
    __syncthreads();
    if (j < nk) {
      for (int q = 0; q < N; ++q) {
        const float* qr = A0 + q * DHP;
        const float* gr = A1 + q * DHP;
        float s = 0.0f, dp = 0.0f;
#pragma unroll
        for (int d = 0; d < DHP; ++d) {
          s = fmaf(qr[d], kv[d], s);
          dp = fmaf(gr[d], vv[d], dp);
        }
        const float p = __expf(s - Ls[q]);
        float pk = p;
        if constexpr (DROP) {
          const float c = attn_keep1(drop, blockIdx.x, N, q, j);
          dp *= c;
          pk = p * c;
        }
        const float ds = p * (dp - Ds[q]);
#pragma unroll
        for (int d = 0; d < DHP; ++d) {
          dv[d] = fmaf(pk, gr[d], dv[d]);
          dk[d] = fmaf(ds, qr[d], dk[d]);
        }
      }
      const int64_t row = (int64_t)idx[j] * dld;
#pragma unroll
      for (int d = 0; d < DHP; ++d)
        if (d < dh) {
          dbase[row + H + d] = dk[d] * scale;
          dbase[row + 2 * H + d] = dv[d];
        }
    }
  }
  // keys outside the valid set take no part: zero gradient
  for (int i = threadIdx.x; i < N * dh; i += blockDim.x) {
    const int n = i / dh, d = i - n * dh;
    const bool ok = ((maskf[(int64_t)b * N + n] == 0.0f) == (valid_is_zero != 0));
    if (!ok) {
      dbase[(int64_t)n * dld + H + d] = 0.0f;
      dbase[(int64_t)n * dld + 2 * H + d] = 0.0f;
    }
  }
}

// ---- the same two ops on the matrix cores (fp32 MFMA, v_mfma_f32_32x32x2_f32) ---------------------------------
// One wave owns a tile of 32 queries (lane & 31 = query) and walks the valid keys 32 at a time.  Everything is kept
// TRANSPOSED so that no tile ever changes layout between two products:
//   S^T (keys x queries) = K Q^T      A = K rows from LDS (lane = key, its half of the head dims), B = Q^T registers
//   softmax over keys    = over the 16 registers of the lane + one exchange with lane ^ 32
//   O^T (dims x queries) += V^T P^T   B = the registers of P^T AS THEY ARE: step r contracts the two keys that
//                                     register r holds in the two lane halves, A = those two rows of V (lane = dim)
// The backward uses the same two shapes five times: phase A (wave = query tile) S^T, dP^T = V dO^T, dQ^T += K^T dS^T;
// phase B (wave = key tile, lane & 31 = key) S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
// LDS rows have a pitch of DH + 4 floats (conflict-free 16-byte row reads); rows past the last valid key / query
// and columns past the head width are zeros.  For DH = 16 the dim-tile of the second product is half empty (lanes
// 16..31 read the next row: finite junk that only reaches output rows nobody stores).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
#define MMB_MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// the valid keys of a jet of N <= 256 particles: every lane fetches its (up to four) mask entries BEFORE the first
// ballot, so the scan costs one memory latency instead of one per 64 particles
__device__ inline void compact_keys256(const float* __restrict__ maskf, int N, int valid_is_zero, int* idx,
                                       int* pos, int* nk_sh) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float mv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) mv[c] = c * 64 + lane < N ? maskf[c * 64 + lane] : 0.0f;
    int base = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool ok = c * 64 + lane < N && ((mv[c] == 0.0f) == (valid_is_zero != 0));
      const unsigned long long bal = __ballot(ok);
      const int p = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (ok) idx[p] = c * 64 + lane;
      if (c * 64 + lane < N) pos[c * 64 + lane] = ok ? p : -1;
      base += __popcll(bal);
    }
    if (lane == 0) *nk_sh = base;
  }
}

// K and V rows of the jet's valid keys, compacted, into LDS (pitch DH + 4; rows in [nk, pad32(nk)) and columns >= dh
// are zeros); returns nk.  VEC: every thread fetches its 16-byte pieces of ALL N rows while wave 0 scans the mask,
// so the whole prologue costs one memory latency; the pieces of keys that take no part are dropped.
template <int DH, bool VEC>
__device__ inline int stage_keys(const float* __restrict__ maskf, int N, int valid_is_zero, const float* __restrict__ k0,
                                 const float* __restrict__ v0, int64_t ld, int dh, int* idx, int* pos, int* nk_sh,
                                 float* Ks, float* Vs) {
  constexpr int P = DH + 4;
  if constexpr (VEC) {
    constexpr int C = DH / 4, U = C / 2;              // blockDim >= 2 N threads: U pieces per thread cover N * C
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = threadIdx.x + u * blockDim.x, j = i / C, c = i - j * C;
      a[u] = b[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (j < N && 4 * c < dh) {
        a[u] = *reinterpret_cast<const float4*>(k0 + (int64_t)j * ld + 4 * c);
        b[u] = *reinterpret_cast<const float4*>(v0 + (int64_t)j * ld + 4 * c);
      }
    }
    compact_keys256(maskf, N, valid_is_zero, idx, pos, nk_sh);
    __syncthreads();
    const int nk = *nk_sh, nkp = (nk + 31) & ~31;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = threadIdx.x + u * blockDim.x, j = i / C, c = i - j * C;
      if (j < N) {
        const int pj = pos[j];
        if (pj >= 0) {
          *reinterpret_cast<float4*>(Ks + pj * P + 4 * c) = a[u];
          *reinterpret_cast<float4*>(Vs + pj * P + 4 * c) = b[u];
        }
      }
    }
    for (int i = threadIdx.x; i < (nkp - nk) * C; i += blockDim.x) {
      const int j = nk + i / C, c = i % C;
      *reinterpret_cast<float4*>(Ks + j * P + 4 * c) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      *reinterpret_cast<float4*>(Vs + j * P + 4 * c) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    return nk;
  } else {
    compact_keys256(maskf, N, valid_is_zero, idx, pos, nk_sh);
    __syncthreads();
    const int nk = *nk_sh, nkp = (nk + 31) & ~31;
    for (int i = threadIdx.x; i < nkp * DH; i += blockDim.x) {
      const int j = i / DH, d = i - j * DH;
      float a = 0.0f, b = 0.0f;
      if (j < nk && d < dh) {
        const int64_t row = idx[j];
        a = k0[row * ld + d];
        b = v0[row * ld + d];
      }
      Ks[j * P + d] = a;
      Vs[j * P + d] = b;
    }
    __syncthreads();
    return nk;
  }
}

// the lane's half of one row (HD consecutive floats from p, those past `lim` are zeros), times scale
template <int HD, bool VEC>
__device__ __forceinline__ void load_half(float (&r)[HD], const float* __restrict__ p, bool ok, int lim, float scale) {
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < HD; i += 4) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (ok && i < lim) v = *reinterpret_cast<const float4*>(p + i);
      r[i] = v.x * scale; r[i + 1] = v.y * scale; r[i + 2] = v.z * scale; r[i + 3] = v.w * scale;
    }
  } else {
#pragma unroll
    for (int i = 0; i < HD; ++i) r[i] = (ok && i < lim) ? p[i] * scale : 0.0f;
  }
}

// a transposed result tile (lane & 31 = row of the output matrix, registers = head dims) to global memory
template <int DT, bool VEC>
__device__ __forceinline__ void store_tile_t(float* __restrict__ o, const f32x16 (&acc)[DT], int par, int dh, float mul) {
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    if constexpr (VEC) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * par;
        if (d < dh)
          *reinterpret_cast<float4*>(o + d) = make_float4(acc[t][4 * g] * mul, acc[t][4 * g + 1] * mul,
                                                          acc[t][4 * g + 2] * mul, acc[t][4 * g + 3] * mul);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = t * 32 + crow(r, par);
        if (d < dh) o[d] = acc[t][r] * mul;
      }
    }
  }
}

// (jet, head) of a workgroup.  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2; the
// heads of one jet read the SAME rows of qkv (64..128 bytes each per row), so they are given ids 8 apart: same XCD,
// dispatched together -- with heads on different XCDs every 128-byte line of a d_h = 16 model was fetched from HBM
// twice (PMC: 239 MB against 118 MB algorithmic).
__device__ __forceinline__ void jet_head_of_block(int nh, int& b, int& h) {
  const int g = blockIdx.x;
  if (gridDim.x % (8 * nh) == 0) {
    const int s = g >> 3;
    h = s % nh;
    b = (s / nh) * 8 + (g & 7);
  } else {
    b = g / nh;
    h = g - b * nh;
  }
}

template <int DH, bool VEC, bool DROP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(DH == 16 ? 8 : 6)))
attn_fwd_mfma_kernel(int N, int H, int nh, int dh, const float* __restrict__ qkv, int64_t ld,
                     const float* __restrict__ maskf, int valid_is_zero, float scale, float* __restrict__ out,
                     int64_t out_ld, float* __restrict__ lse, AttnDrop drop, int uniform_ok, uint32_t* __restrict__ keep_bits) {
  constexpr int P = DH + 4, HD = DH / 2, DT = (DH + 31) / 32;
  extern __shared__ float sm[];
  int b, h;
  jet_head_of_block(nh, b, h);
  const int NP = (N + 31) & ~31;
  int* idx = reinterpret_cast<int*>(sm);
  int* pos = idx + N;
  int* nk_sh = pos + N;
  float* Ks = sm + ((2 * N + 1 + 3) & ~3);
  float* Vs = Ks + (size_t)NP * P;
  const float* base = qkv + (int64_t)b * N * ld + h * dh;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, lane = threadIdx.x & 63, col = lane & 31,
            par = lane >> 5;
  float qr[HD];                                  // the first query tile's rows are fetched beside the keys
  load_half<HD, VEC>(qr, base + (int64_t)(wave * 32 + col) * ld + par * HD, wave * 32 + col < N, dh - par * HD, scale * LOG2E);
  const int nk = stage_keys<DH, VEC>(maskf + (int64_t)b * N, N, valid_is_zero, base + H, base + 2 * H, ld, dh, idx, pos,
                                     nk_sh, Ks, Vs);
  const int nkp = (nk + 31) & ~31;
  if constexpr (!DROP) {
    if (uniform_ok && lse == nullptr && keys_uniform(Ks, Vs, nk, dh, P)) {
      broadcast_row(Vs, N, dh, out + (int64_t)b * N * out_ld + h * dh, out_ld);
      return;
    }
  }
  for (int qt = wave * 32; qt < N; qt += nwaves * 32) {
    const int q = qt + col;
    if (qt != wave * 32)
      load_half<HD, VEC>(qr, base + (int64_t)q * ld + par * HD, q < N, dh - par * HD, scale * LOG2E);
    f32x16 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;              // (scores and m in units of log 2: qr carries scale * log2(e))
    for (int kt = 0; kt < nkp; kt += 32) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
      const float* kr = Ks + (kt + col) * P + par * HD;
#pragma unroll
      for (int i = 0; i < HD; ++i) s = MMB_MFMA_F32(kr[i], qr[i], s);
      if (kt + 32 > nk) {                        // only the last tile has keys to mask
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt + crow(r, par) >= nk) s[r] = -INFINITY;
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));             // key kt of this tile is valid: finite
      const float mn = fmaxf(m, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float ps = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - mn);
        ps += s[r];
      }
      ps += __shfl_xor(ps, 32);
      l = fmaf(l, alpha, ps);
      m = mn;
      if constexpr (DROP) {                      // registers 4g..4g+3 of a lane hold four consecutive keys of its query
        uint32_t word = 0;                         // bit c = key kt + c of this query
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t kb = attn_keep4(drop, (int64_t)b * nh + h, N, q < N ? q : 0, (kt + 8 * g + 4 * par) >> 2);
          word |= kb << (8 * g + 4 * par);
#pragma unroll
          for (int i = 0; i < 4; ++i) s[4 * g + i] = ((kb >> i) & 1u) ? s[4 * g + i] * drop.inv_keep : 0.0f;
        }
        if (keep_bits != nullptr) {                // for the reverse pass: one word per (query, 32-key tile)
          word |= (uint32_t)__shfl_xor((int)word, 32, 64);
          if (par == 0 && q < N) keep_bits[(((int64_t)b * nh + h) * N + q) * ((N + 31) >> 5) + (kt >> 5)] = word;
        }
      }
#pragma unroll
      for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[t] = MMB_MFMA_F32(Vs[(kt + crow(r, par)) * P + t * 32 + col], s[r], acc[t]);
      }
    }
    if (q < N) {
      store_tile_t<DT, VEC>(out + ((int64_t)b * N + q) * out_ld + h * dh, acc, par, dh, nk > 0 ? 1.0f / l : 0.0f);
      if (lse && par == 0) lse[((int64_t)b * nh + h) * N + q] = nk > 0 ? (m + __log2f(l)) * LN2 : 0.0f;
    }
  }
}

template <int DH, bool VEC, bool DROP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(DH == 16 ? 4 : 3)))
attn_bwd_mfma_kernel(int N, int H, int nh, int dh, const float* __restrict__ qkv, int64_t ld,
                     const float* __restrict__ maskf, int valid_is_zero, float scale, const float* __restrict__ out,
                     int64_t out_ld, const float* __restrict__ dout, int64_t dout_ld, const float* __restrict__ lse,
                     float* __restrict__ dqkv, int64_t dld, AttnDrop drop, const uint32_t* __restrict__ keep_bits) {
  constexpr int P = DH + 4, HD = DH / 2, DT = (DH + 31) / 32;
  extern __shared__ float sm[];
  int b, h;
  jet_head_of_block(nh, b, h);
  const int NP = (N + 31) & ~31;
  int* idx = reinterpret_cast<int*>(sm);
  int* pos = idx + N;
  int* nk_sh = pos + N;
  float* Ls = sm + ((2 * N + 1 + 3) & ~3);
  float* Ds = Ls + NP;
  float* A0 = Ds + NP;                         // K (phase A) / scale * Q (phase B)
  float* A1 = A0 + (size_t)NP * P;             // V (phase A) / dO (phase B)
  const float* base = qkv + (int64_t)b * N * ld + h * dh;
  const float* obase = out + (int64_t)b * N * out_ld + h * dh;
  const float* gbase = dout + (int64_t)b * N * dout_ld + h * dh;
  float* dbase = dqkv + (int64_t)b * N * dld + h * dh;
  const float* lrow = lse + ((int64_t)b * nh + h) * N;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, lane = threadIdx.x & 63, col = lane & 31,
            par = lane >> 5;
  // the block has one wave per query tile (host): the tile's rows of Q, dO, O and its L are fetched beside the keys,
  // and stay in registers until phase B takes them over through LDS -- one memory latency for the whole kernel
  const int q = wave * 32 + col;
  float qr[HD], gr[HD];
  float D = 0.0f;
  {
    float orow[HD];
    load_half<HD, VEC>(qr, base + (int64_t)q * ld + par * HD, q < N, dh - par * HD, scale * LOG2E);
    load_half<HD, VEC>(gr, gbase + (int64_t)q * dout_ld + par * HD, q < N, dh - par * HD, 1.0f);
    load_half<HD, VEC>(orow, obase + (int64_t)q * out_ld + par * HD, q < N, dh - par * HD, 1.0f);
    const float L0 = q < N ? lrow[q] * LOG2E : 0.0f;      // (units of log 2, as the scores)
    if (par == 0) Ls[q] = L0;                  // (Ls, Ds are not touched by stage_keys)
#pragma unroll
    for (int i = 0; i < HD; ++i) D = fmaf(gr[i], orow[i], D);
  }
  const int nk = stage_keys<DH, VEC>(maskf + (int64_t)b * N, N, valid_is_zero, base + H, base + 2 * H, ld, dh, idx, pos,
                                     nk_sh, A0, A1);
  const int nkp = (nk + 31) & ~31;
  // ---- phase A: D, dQ (wave = query tile)
  {
    D += __shfl_xor(D, 32);
    const float L = Ls[q];
    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[t][r] = 0.0f;
    for (int kt = 0; kt < nkp; kt += 32) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
      const float* kr = A0 + (kt + col) * P + par * HD;
      const float* vr = A1 + (kt + col) * P + par * HD;
#pragma unroll
      for (int i = 0; i < HD; ++i) {
        s = MMB_MFMA_F32(kr[i], qr[i], s);
        dp = MMB_MFMA_F32(vr[i], gr[i], dp);
      }
      if constexpr (DROP) {
        if (keep_bits != nullptr) {                // the forward's words (bit c = key kt + c)
          const uint32_t word = keep_bits[(((int64_t)b * nh + h) * N + (q < N ? q : 0)) * ((N + 31) >> 5) + (kt >> 5)];
#pragma unroll
          for (int r = 0; r < 16; ++r) dp[r] = ((word >> crow(r, par)) & 1u) ? dp[r] * drop.inv_keep : 0.0f;
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t kb = attn_keep4(drop, (int64_t)b * nh + h, N, q < N ? q : 0, (kt + 8 * g + 4 * par) >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) dp[4 * g + i] = ((kb >> i) & 1u) ? dp[4 * g + i] * drop.inv_keep : 0.0f;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = kt + crow(r, par) < nk ? __builtin_amdgcn_exp2f(s[r] - L) : 0.0f;
        s[r] = p * (dp[r] - D);
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dq[t] = MMB_MFMA_F32(A0[(kt + crow(r, par)) * P + t * 32 + col], s[r], dq[t]);
    }
    if (par == 0) Ds[q] = D;
    if (q < N) store_tile_t<DT, VEC>(dbase + (int64_t)q * dld, dq, par, dh, scale);
  }
  // ---- phase B: dK, dV (wave = key tile; its K, V rows move to registers, the LDS area is re-filled with Q, dO)
  for (int kt0 = 0; kt0 < nkp; kt0 += nwaves * 32) {          // nkp is uniform over the block: so are the barriers
    const int kt = kt0 + wave * 32;
    const bool active = kt < nkp;
    float kreg[HD], vreg[HD];
#pragma unroll
    for (int i = 0; i < HD; ++i) {
      kreg[i] = active ? A0[(kt + col) * P + par * HD + i] : 0.0f;
      vreg[i] = active ? A1[(kt + col) * P + par * HD + i] : 0.0f;
    }
    __syncthreads();
    if (kt0 == 0) {                            // scale * Q and dO rows of this wave's tile: registers -> LDS
#pragma unroll
      for (int i = 0; i < HD; ++i) {
        A0[q * P + par * HD + i] = qr[i];
        A1[q * P + par * HD + i] = gr[i];
      }
    }
    __syncthreads();
    if (active) {
      f32x16 dk[DT], dv[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.0f;
      for (int qt = 0; qt < NP; qt += 32) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
        const float* qrow = A0 + (qt + col) * P + par * HD;
        const float* grow = A1 + (qt + col) * P + par * HD;
#pragma unroll
        for (int i = 0; i < HD; ++i) {
          s = MMB_MFMA_F32(qrow[i], kreg[i], s);
          dp = MMB_MFMA_F32(grow[i], vreg[i], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qt + crow(r, par);
          const float p = qq < N ? __builtin_amdgcn_exp2f(s[r] - Ls[qq]) : 0.0f;
          float c = 1.0f;
          if constexpr (DROP) {                                                  // (lane = key kt + col)
            if (keep_bits != nullptr)          // one broadcast word per query instead of one Philox call per element
              c = ((keep_bits[(((int64_t)b * nh + h) * N + (qq < N ? qq : 0)) * ((N + 31) >> 5) + (kt >> 5)] >> col) & 1u)
                      ? drop.inv_keep : 0.0f;
            else
              c = attn_keep1(drop, (int64_t)b * nh + h, N, qq < N ? qq : 0, kt + col);
          }
          dp[r] = p * (dp[r] * c - Ds[qq]);
          s[r] = p * c;
        }
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = (qt + crow(r, par)) * P + t * 32 + col;
            dv[t] = MMB_MFMA_F32(A1[o], s[r], dv[t]);
            dk[t] = MMB_MFMA_F32(A0[o], dp[r], dk[t]);
          }
      }
      const int key = kt + col;
      if (key < nk) {
        float* drow = dbase + (int64_t)idx[key] * dld;
        store_tile_t<DT, VEC>(drow + H, dk, par, dh, LN2);         // Q in LDS carries scale * log2(e)
        store_tile_t<DT, VEC>(drow + 2 * H, dv, par, dh, 1.0f);
      }
    }
  }
  for (int i = threadIdx.x; i < N * dh; i += blockDim.x) {     // keys outside the valid set: zero gradient
    const int n = i / dh, d = i - n * dh;
    const bool ok = ((maskf[(int64_t)b * N + n] == 0.0f) == (valid_is_zero != 0));
    if (!ok) {
      dbase[(int64_t)n * dld + H + d] = 0.0f;
      dbase[(int64_t)n * dld + 2 * H + d] = 0.0f;
    }
  }
}

// LDS of the MFMA kernels: idx + nk, (L, D), two row arrays of pitch DH + 4, 32 floats of slack
static size_t attn_mfma_lds_bytes(int N, int DH, bool bwd) {
  const size_t NP = (N + 31) & ~31;
  size_t fl = ((2 * N + 1 + 3) & ~3) + 2 * NP * (DH + 4) + 32;
  if (bwd) fl += 2 * NP;
  return fl * sizeof(float);
}
// the matrix-core kernels serve head widths <= 32 and jets of <= 256 particles that fit 64 KiB of LDS
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool attn_use_mfma(int N, int dh, bool bwd) {
  if (dh > 32 || N > 256) return false;
  const char* e = getenv("MMB_ATTN_PATH");
  if (e && !strcmp(e, "valu")) return false;                // A/B switch for tests and profiles
  return attn_mfma_lds_bytes(N, dh <= 16 ? 16 : 32, bwd) <= 64 * 1024;
}

// ---- LayerNorm over the last dim, one wave per row --------------------------------------------------------
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(int64_t M, int D, const float* __restrict__ x, int64_t x_ld, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float eps, float* __restrict__ y, int64_t y_ld,
                     float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < M; r += nw) {
    const float* xr = x + r * x_ld;
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mu = wave_sum(s) / (float)D;
    float v = 0.0f;
    for (int c = lane; c < D; c += 64) {
      const float d = xr[c] - mu;
      v = fmaf(d, d, v);
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
    for (int c = lane; c < D; c += 64) y[r * y_ld + c] = (xr[c] - mu) * rstd * gamma[c] + beta[c];
    if (stats && lane == 0) {
      stats[2 * r] = mu;
      stats[2 * r + 1] = rstd;
    }
  }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;  dgamma += sum_rows dy xhat ; dbeta += sum_rows dy
// (dgamma/dbeta: per-lane partial sums over the block's rows, folded over the 4 waves in LDS, one atomic per column
// and block; the buffers must be pre-zeroed.)  D <= 64*LN_MAXC.
constexpr int LN_MAXC = 8;
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(int64_t M, int D, const float* __restrict__ x, int64_t x_ld, const float* __restrict__ gamma,
                     const float* __restrict__ stats, const float* __restrict__ dy, int64_t dy_ld,
                     float* __restrict__ dx, int64_t dx_ld, int accumulate, float* __restrict__ dgamma,
                     float* __restrict__ dbeta) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = blockIdx.x * (int64_t)4 + w;
  const int64_t nw = (int64_t)gridDim.x * 4;
  float pg[LN_MAXC], pb[LN_MAXC];
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) pg[i] = pb[i] = 0.0f;
  for (int64_t r = wave; r < M; r += nw) {
    const float mu = stats[2 * r], rstd = stats[2 * r + 1];
    const float* xr = x + r * x_ld;
    const float* gr = dy + r * dy_ld;
    float xh[LN_MAXC], g[LN_MAXC];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        xh[i] = (xr[c] - mu) * rstd;
        const float d = gr[c];
        g[i] = d * gamma[c];
        pg[i] = fmaf(d, xh[i], pg[i]);
        pb[i] += d;
        s1 += g[i];
        s2 = fmaf(g[i], xh[i], s2);
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        const float v = rstd * (g[i] - s1 - xh[i] * s2);
        float* o = dx + r * dx_ld + c;
        *o = accumulate ? *o + v : v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    if (64 * i >= D) break;
    red[0][w][lane] = pg[i];
    red[1][w][lane] = pb[i];
    __syncthreads();
    const int c = lane + 64 * i;
    if (w == 0 && c < D) {
      atomicAdd(dgamma + c, (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]));
      atomicAdd(dbeta + c, (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
    }
    __syncthreads();
  }
}

// ---- LayerNorm, D = 64 / 128 / 256 (the transformer's widths): G = D/4 lanes per row, a float4 per lane, 64/G rows
// per wave and U row groups in flight per iteration -- the one-row-per-wave kernels above keep 256 bytes per wave in
// flight and are latency-bound at these widths (87 us for 39 MB in, 39 MB out)
template <int G>
__device__ __forceinline__ float group_sum(float s) {
#pragma unroll
  for (int o = G / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  return s;
}

template <int G>
__global__ void __launch_bounds__(256)
layernorm_fwd_vec_kernel(int64_t M, const float* __restrict__ x, int64_t x_ld, const float* __restrict__ gamma,
                         const float* __restrict__ beta, float eps, float* __restrict__ y, int64_t y_ld,
                         float* __restrict__ stats) {
  constexpr int RPW = 64 / G, U = 4;
  constexpr float INV_D = 1.0f / (4 * G);
  const int lane = threadIdx.x & 63, sub = lane % G, rsub = lane / G;
  const int64_t wave = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const float4 gm = reinterpret_cast<const float4*>(gamma)[sub], bt = reinterpret_cast<const float4*>(beta)[sub];
  for (int64_t r0 = wave * (RPW * U); r0 < M; r0 += nw * (RPW * U)) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * RPW + rsub;
      v[u] = r < M ? *reinterpret_cast<const float4*>(x + r * x_ld + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * RPW + rsub;
      const float mu = group_sum<G>((v[u].x + v[u].y) + (v[u].z + v[u].w)) * INV_D;
      const float dx = v[u].x - mu, dy = v[u].y - mu, dz = v[u].z - mu, dw = v[u].w - mu;
      const float rstd = rsqrtf(group_sum<G>(fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, dw * dw)))) * INV_D + eps);
      if (r < M) {
        *reinterpret_cast<float4*>(y + r * y_ld + 4 * sub) =
            make_float4(fmaf(dx * rstd, gm.x, bt.x), fmaf(dy * rstd, gm.y, bt.y), fmaf(dz * rstd, gm.z, bt.z),
                        fmaf(dw * rstd, gm.w, bt.w));
        if (stats && sub == 0) *reinterpret_cast<float2*>(stats + 2 * r) = make_float2(mu, rstd);
      }
    }
  }
}

template <int G>
__global__ void __launch_bounds__(256)
layernorm_bwd_vec_kernel(int64_t M, const float* __restrict__ x, int64_t x_ld, const float* __restrict__ gamma,
                         const float* __restrict__ stats, const float* __restrict__ dy, int64_t dy_ld,
                         float* __restrict__ dx, int64_t dx_ld, int accumulate, float* __restrict__ dgamma,
                         float* __restrict__ dbeta) {
  constexpr int RPW = 64 / G, U = 2;
  constexpr float INV_D = 1.0f / (4 * G);
  __shared__ float4 red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane % G, rsub = lane / G;
  const int64_t wave = blockIdx.x * (int64_t)4 + w, nw = (int64_t)gridDim.x * 4;
  const float4 gm = reinterpret_cast<const float4*>(gamma)[sub];
  float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = pg;
  for (int64_t r0 = wave * (RPW * U); r0 < M; r0 += nw * (RPW * U)) {
    float4 xv[U], dv[U], old[U];
    float2 st[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * RPW + rsub;
      const bool ok = r < M;
      xv[u] = ok ? *reinterpret_cast<const float4*>(x + r * x_ld + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
      dv[u] = ok ? *reinterpret_cast<const float4*>(dy + r * dy_ld + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
      st[u] = ok ? *reinterpret_cast<const float2*>(stats + 2 * r) : make_float2(0.f, 0.f);
      if (accumulate) old[u] = ok ? *reinterpret_cast<const float4*>(dx + r * dx_ld + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * RPW + rsub;
      const float mu = st[u].x, rstd = st[u].y;
      const float4 xh = make_float4((xv[u].x - mu) * rstd, (xv[u].y - mu) * rstd, (xv[u].z - mu) * rstd,
                                    (xv[u].w - mu) * rstd);
      const float4 d = dv[u];
      const float4 g = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      pg.x = fmaf(d.x, xh.x, pg.x); pg.y = fmaf(d.y, xh.y, pg.y); pg.z = fmaf(d.z, xh.z, pg.z); pg.w = fmaf(d.w, xh.w, pg.w);
      pb.x += d.x; pb.y += d.y; pb.z += d.z; pb.w += d.w;
      const float s1 = group_sum<G>((g.x + g.y) + (g.z + g.w)) * INV_D;
      const float s2 = group_sum<G>(fmaf(g.x, xh.x, fmaf(g.y, xh.y, fmaf(g.z, xh.z, g.w * xh.w)))) * INV_D;
      if (r < M) {
        float4 o = make_float4(rstd * (g.x - s1 - xh.x * s2), rstd * (g.y - s1 - xh.y * s2),
                               rstd * (g.z - s1 - xh.z * s2), rstd * (g.w - s1 - xh.w * s2));
        if (accumulate) { o.x += old[u].x; o.y += old[u].y; o.z += old[u].z; o.w += old[u].w; }
        *reinterpret_cast<float4*>(dx + r * dx_ld + 4 * sub) = o;
      }
    }
  }
  red[0][w][lane] = pg;
  red[1][w][lane] = pb;
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * 4 * G; c += 256) {           // one thread per (array, column)
    const int which = c / (4 * G), col = c - which * (4 * G), sb = col >> 2, comp = col & 3;
    float s = 0.0f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww)
#pragma unroll
      for (int rs = 0; rs < RPW; ++rs) s += reinterpret_cast<const float*>(&red[which][ww][rs * G + sb])[comp];
    atomicAdd((which ? dbeta : dgamma) + col, s);
  }
}

// dst[i] (+)= sum_r src[r*stride + i]
__global__ void __launch_bounds__(256)
sum_replicas_kernel(int64_t n, int R, int64_t stride, const float* __restrict__ src, float* __restrict__ dst,
                    int accumulate) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int r = 0; r < R; ++r) s += src[r * stride + i];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}

// the keep factors as a dense (B, nh, N, N) tensor [b,h,q,n]: 1/(1-p) or 0 at the valid keys n, 0 elsewhere (tests)
__global__ void __launch_bounds__(256)
attn_dropout_mask_kernel(int64_t total, int N, int nh, const float* __restrict__ maskf, int valid_is_zero, AttnDrop drop,
                         float* __restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(e % N);
    const int64_t bh = e / N, b = bh / nh;
    int j = 0;
    for (int n = 0; n < N; ++n) {
      const bool ok = ((maskf[b * N + n] == 0.0f) == (valid_is_zero != 0));
      out[e * N + n] = ok ? attn_keep1(drop, bh, N, q, j) : 0.0f;
      j += ok ? 1 : 0;
    }
  }
}

static size_t attn_lds_bytes(int N, int dhp, bool bwd) {
  size_t fl = ((N + 1 + 3) & ~3) + 2 * (size_t)N * dhp;
  if (bwd) fl += N + ((N + 3) & ~3);
  return fl * sizeof(float);
}

static int pad_dh(int dh) { return dh <= 4 ? 4 : dh <= 8 ? 8 : dh <= 16 ? 16 : dh <= 32 ? 32 : 64; }

}  // namespace mmb

using namespace mmb;

#define MMB_ATTN_DISPATCH_D(KERNEL, DROPV, ...)                                                       \
  switch (dhp) {                                                                                    \
    case 4: hipLaunchKernelGGL((KERNEL<4, DROPV>), grid, block, lds, st, __VA_ARGS__); break;       \
    case 8: hipLaunchKernelGGL((KERNEL<8, DROPV>), grid, block, lds, st, __VA_ARGS__); break;       \
    case 16: hipLaunchKernelGGL((KERNEL<16, DROPV>), grid, block, lds, st, __VA_ARGS__); break;     \
    case 32: hipLaunchKernelGGL((KERNEL<32, DROPV>), grid, block, lds, st, __VA_ARGS__); break;     \
    default: hipLaunchKernelGGL((KERNEL<64, DROPV>), grid, block, lds, st, __VA_ARGS__); break;     \
  }
#define MMB_ATTN_DISPATCH(KERNEL, ...)                                      \
  if (drop.p > 0.0f) { MMB_ATTN_DISPATCH_D(KERNEL, true, __VA_ARGS__) }     \
  else { MMB_ATTN_DISPATCH_D(KERNEL, false, __VA_ARGS__) }

static int attn_drop_args(const char* fn, float p, uint64_t seed, uint64_t offset, AttnDrop& d) {
  MMB_REQUIRE(p >= 0.0f && p < 1.0f, "%s: dropout probability %g outside [0, 1)", fn, (double)p);
  d = AttnDrop{p, 1.0f / (1.0f - p), seed, offset};
  return MMB_OK;
}

static int attn_check(const char* fn, int B, int N, int H, int nheads, int64_t ld, size_t lds) {
  MMB_REQUIRE(B > 0 && N > 0 && H > 0 && nheads > 0, "%s: empty shape", fn);
  MMB_REQUIRE(H % nheads == 0, "%s: model width %d is not a multiple of %d heads", fn, H, nheads);
  MMB_REQUIRE(H / nheads <= 64, "%s: head width %d > 64 is not supported", fn, H / nheads);
  MMB_REQUIRE(ld >= 3 * (int64_t)H, "%s: qkv row stride %lld < 3*%d", fn, (long long)ld, H);
  MMB_REQUIRE(lds <= 64 * 1024, "%s: N=%d particles x head width %d needs %zu B of LDS (> 64 KiB)", fn, N, H / nheads,
              lds);
  MMB_REQUIRE((int64_t)B * nheads < (1ll << 31), "%s: too many (jet, head) pairs", fn);
  return MMB_OK;
}

static int attention_fwd_impl(const char* fn, int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                              const float* maskf, int valid_is_zero, float* out, int64_t out_ld, float* lse,
                              const AttnDrop& drop, uint32_t* keep_bits, void* stream) {
  MMB_REQUIRE(qkv && maskf && out, "%s: NULL argument", fn);
  const int dh = nheads > 0 ? H / nheads : 0, dhp = pad_dh(dh);
  const bool mfma = N > 0 && dh > 0 && attn_use_mfma(N, dh, false);
  const size_t lds = mfma ? attn_mfma_lds_bytes(N, dh <= 16 ? 16 : 32, false) : attn_lds_bytes(N, dhp, false);
  if (int rc = attn_check(fn, B, N, H, nheads, qkv_ld, lds)) return rc;
  MMB_REQUIRE(out_ld >= H, "%s: out row stride < H", fn);
  dim3 grid(B * nheads), block(N <= 64 ? 64 : N <= 128 ? 128 : N <= 192 ? 192 : 256);
  hipStream_t st = as_stream(stream);
  const float scale = 1.0f / sqrtf((float)dh);
  const int uniform_ok = attn_uniform_enabled() ? 1 : 0;
  if (mfma) {
    block = dim3(64 * ((N + 31) / 32));
    const bool vec = dh % 4 == 0 && qkv_ld % 4 == 0 && out_ld % 4 == 0 && aligned16(qkv) && aligned16(out);
#define MMB_FWD_MFMA(DHV, VECV, DROPV)                                                                                \
  hipLaunchKernelGGL((attn_fwd_mfma_kernel<DHV, VECV, DROPV>), grid, block, lds, st, N, H, nheads, dh, qkv, qkv_ld,    \
                     maskf, valid_is_zero, scale, out, out_ld, lse, drop, uniform_ok, keep_bits)
#define MMB_FWD_MFMA2(DHV, VECV) do { if (drop.p > 0.0f) MMB_FWD_MFMA(DHV, VECV, true); else MMB_FWD_MFMA(DHV, VECV, false); } while (0)
    if (dh <= 16) { if (vec) MMB_FWD_MFMA2(16, true); else MMB_FWD_MFMA2(16, false); }
    else { if (vec) MMB_FWD_MFMA2(32, true); else MMB_FWD_MFMA2(32, false); }
#undef MMB_FWD_MFMA2
#undef MMB_FWD_MFMA
    return check_launch(fn);
  }
  MMB_ATTN_DISPATCH(attn_fwd_kernel, N, H, nheads, dh, qkv, qkv_ld, maskf, valid_is_zero, scale, out, out_ld, lse, drop,
                    uniform_ok);
  return check_launch(fn);
}

extern "C" int mmb_attention_fwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                                 const float* maskf, int valid_is_zero, float* out, int64_t out_ld, float* lse,
                                 void* stream) {
  return attention_fwd_impl("mmb_attention_fwd", B, N, H, nheads, qkv, qkv_ld, maskf, valid_is_zero, out, out_ld, lse,
                            AttnDrop{0.0f, 1.0f, 0, 0}, nullptr, stream);
}

extern "C" int mmb_attention_dropout_fwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                                         const float* maskf, int valid_is_zero, float* out, int64_t out_ld, float* lse,
                                         float drop_p, uint64_t seed, uint64_t offset, uint32_t* keep_bits, void* stream) {
  AttnDrop d;
  if (int rc = attn_drop_args("mmb_attention_dropout_fwd", drop_p, seed, offset, d)) return rc;
  return attention_fwd_impl("mmb_attention_dropout_fwd", B, N, H, nheads, qkv, qkv_ld, maskf, valid_is_zero, out, out_ld,
                            lse, d, keep_bits, stream);
}

extern "C" int mmb_attention_dropout_mask(int B, int N, int nheads, const float* maskf, int valid_is_zero, float drop_p,
                                          uint64_t seed, uint64_t offset, float* out, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && nheads > 0 && maskf && out, "mmb_attention_dropout_mask: bad argument");
  AttnDrop d;
  if (int rc = attn_drop_args("mmb_attention_dropout_mask", drop_p, seed, offset, d)) return rc;
  const int64_t total = (int64_t)B * nheads * N;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), total, N,
                     nheads, maskf, valid_is_zero, d, out);
  return check_launch("mmb_attention_dropout_mask");
}

static int attention_bwd_impl(const char* fn, int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                              const float* maskf, int valid_is_zero, const float* out, int64_t out_ld,
                              const float* dout, int64_t dout_ld, const float* lse, float* dqkv, int64_t dqkv_ld,
                              const AttnDrop& drop, const uint32_t* keep_bits, void* stream) {
  MMB_REQUIRE(qkv && maskf && out && dout && lse && dqkv, "%s: NULL argument", fn);
  const int dh = nheads > 0 ? H / nheads : 0, dhp = pad_dh(dh);
  const bool mfma = N > 0 && dh > 0 && attn_use_mfma(N, dh, true);
  const size_t lds = mfma ? attn_mfma_lds_bytes(N, dh <= 16 ? 16 : 32, true) : attn_lds_bytes(N, dhp, true);
  if (int rc = attn_check(fn, B, N, H, nheads, qkv_ld, lds)) return rc;
  MMB_REQUIRE(out_ld >= H && dout_ld >= H && dqkv_ld >= 3 * (int64_t)H, "%s: row stride too small", fn);
  dim3 grid(B * nheads), block(N <= 64 ? 64 : N <= 128 ? 128 : N <= 192 ? 192 : 256);
  hipStream_t st = as_stream(stream);
  const float scale = 1.0f / sqrtf((float)dh);
  if (mfma) {
    block = dim3(64 * ((N + 31) / 32));
    const bool vec = dh % 4 == 0 && qkv_ld % 4 == 0 && out_ld % 4 == 0 && dout_ld % 4 == 0 && dqkv_ld % 4 == 0 &&
                     aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv);
#define MMB_BWD_MFMA(DHV, VECV, DROPV)                                                                                \
  hipLaunchKernelGGL((attn_bwd_mfma_kernel<DHV, VECV, DROPV>), grid, block, lds, st, N, H, nheads, dh, qkv, qkv_ld,    \
                     maskf, valid_is_zero, scale, out, out_ld, dout, dout_ld, lse, dqkv, dqkv_ld, drop, keep_bits)
#define MMB_BWD_MFMA2(DHV, VECV) do { if (drop.p > 0.0f) MMB_BWD_MFMA(DHV, VECV, true); else MMB_BWD_MFMA(DHV, VECV, false); } while (0)
    if (dh <= 16) { if (vec) MMB_BWD_MFMA2(16, true); else MMB_BWD_MFMA2(16, false); }
    else { if (vec) MMB_BWD_MFMA2(32, true); else MMB_BWD_MFMA2(32, false); }
#undef MMB_BWD_MFMA2
#undef MMB_BWD_MFMA
    return check_launch(fn);
  }
  MMB_ATTN_DISPATCH(attn_bwd_kernel, N, H, nheads, dh, qkv, qkv_ld, maskf, valid_is_zero, scale, out, out_ld, dout,
                    dout_ld, lse, dqkv, dqkv_ld, drop);
  return check_launch(fn);
}

extern "C" int mmb_attention_bwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                                 const float* maskf, int valid_is_zero, const float* out, int64_t out_ld,
                                 const float* dout, int64_t dout_ld, const float* lse, float* dqkv, int64_t dqkv_ld,
                                 void* stream) {
  return attention_bwd_impl("mmb_attention_bwd", B, N, H, nheads, qkv, qkv_ld, maskf, valid_is_zero, out, out_ld, dout,
                            dout_ld, lse, dqkv, dqkv_ld, AttnDrop{0.0f, 1.0f, 0, 0}, nullptr, stream);
}

extern "C" int mmb_attention_dropout_bwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                                         const float* maskf, int valid_is_zero, const float* out, int64_t out_ld,
                                         const float* dout, int64_t dout_ld, const float* lse, float* dqkv,
                                         int64_t dqkv_ld, float drop_p, uint64_t seed, uint64_t offset,
                                         const uint32_t* keep_bits, void* stream) {
  AttnDrop d;
  if (int rc = attn_drop_args("mmb_attention_dropout_bwd", drop_p, seed, offset, d)) return rc;
  return attention_bwd_impl("mmb_attention_dropout_bwd", B, N, H, nheads, qkv, qkv_ld, maskf, valid_is_zero, out, out_ld,
                            dout, dout_ld, lse, dqkv, dqkv_ld, d, keep_bits, stream);
}

extern "C" int mmb_layernorm_fwd(int64_t M, int D, const float* x, int64_t x_ld, const float* gamma, const float* beta,
                                 float eps, float* y, int64_t y_ld, float* stats, void* stream) {
  MMB_REQUIRE(M > 0 && D > 0, "mmb_layernorm_fwd: empty shape");
  MMB_REQUIRE(x && gamma && beta && y, "mmb_layernorm_fwd: NULL argument");
  MMB_REQUIRE(x_ld >= D && y_ld >= D, "mmb_layernorm_fwd: row stride < D");
  if ((D == 64 || D == 128 || D == 256) && x_ld % 4 == 0 && y_ld % 4 == 0 && aligned16(x) && aligned16(y) &&
      aligned16(gamma) && aligned16(beta) && (!stats || (reinterpret_cast<uintptr_t>(stats) & 7) == 0)) {
    const int rpb = 4 * (256 / D) * 4 * 4;         // rows per block and iteration: 4 waves x 64/G rows x U = 4
    const int64_t nb = (M + rpb - 1) / rpb;
    const dim3 grid((unsigned)(nb > 2048 ? 2048 : nb));
    hipStream_t st = as_stream(stream);
#define MMB_LN_F(G_) hipLaunchKernelGGL((layernorm_fwd_vec_kernel<G_>), grid, dim3(256), 0, st, M, x, x_ld, gamma, beta, eps, y, y_ld, stats)
    if (D == 64) MMB_LN_F(16); else if (D == 128) MMB_LN_F(32); else MMB_LN_F(64);
#undef MMB_LN_F
    return check_launch("mmb_layernorm_fwd");
  }
  const int64_t blocks = (M + 3) / 4;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
                     as_stream(stream), M, D, x, x_ld, gamma, beta, eps, y, y_ld, stats);
  return check_launch("mmb_layernorm_fwd");
}

extern "C" int mmb_layernorm_bwd(int64_t M, int D, const float* x, int64_t x_ld, const float* gamma,
                                 const float* stats, const float* dy, int64_t dy_ld, float* dx, int64_t dx_ld,
                                 int accumulate, float* dgamma, float* dbeta, void* stream) {
  MMB_REQUIRE(M > 0 && D > 0, "mmb_layernorm_bwd: empty shape");
  MMB_REQUIRE(D <= 64 * LN_MAXC, "mmb_layernorm_bwd: D=%d > %d is not supported", D, 64 * LN_MAXC);
  MMB_REQUIRE(x && gamma && stats && dy && dx && dgamma && dbeta, "mmb_layernorm_bwd: NULL argument");
  MMB_REQUIRE(x_ld >= D && dy_ld >= D && dx_ld >= D, "mmb_layernorm_bwd: row stride < D");
  if ((D == 64 || D == 128 || D == 256) && x_ld % 4 == 0 && dy_ld % 4 == 0 && dx_ld % 4 == 0 && aligned16(x) &&
      aligned16(dy) && aligned16(dx) && aligned16(gamma) && (reinterpret_cast<uintptr_t>(stats) & 7) == 0) {
    const int rpb = 4 * (256 / D) * 4 * 2;         // 4 waves x 64/G rows x U = 2
    const int64_t nb = (M + rpb - 1) / rpb;
    const dim3 grid((unsigned)(nb > 1024 ? 1024 : nb));   // <= 1024 same-address atomics per column
    hipStream_t st = as_stream(stream);
#define MMB_LN_B(G_) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<G_>), grid, dim3(256), 0, st, M, x, x_ld, gamma, stats, dy, dy_ld, dx, dx_ld, accumulate, dgamma, dbeta)
    if (D == 64) MMB_LN_B(16); else if (D == 128) MMB_LN_B(32); else MMB_LN_B(64);
#undef MMB_LN_B
    return check_launch("mmb_layernorm_bwd");
  }
  const int64_t blocks = (M + 63) / 64;            // 16 rows per wave: few atomics per column
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                     as_stream(stream), M, D, x, x_ld, gamma, stats, dy, dy_ld, dx, dx_ld, accumulate, dgamma, dbeta);
  return check_launch("mmb_layernorm_bwd");
}

extern "C" int mmb_sum_replicas(int64_t n, int replicas, int64_t stride, const float* src, float* dst, int accumulate,
                                void* stream) {
  MMB_REQUIRE(n > 0 && replicas > 0 && src && dst, "mmb_sum_replicas: bad argument");
  hipLaunchKernelGGL(sum_replicas_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), n, replicas,
                     stride, src, dst, accumulate);
  return check_launch("mmb_sum_replicas");
}
