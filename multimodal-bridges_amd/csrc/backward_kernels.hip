// Backward kernels of the layered EPiC path (training): activation/epilogue gradient,
// per-jet column sums, weight gradient (fp32 MFMA, dZ^T @ cat(X) with the concatenation never
// materialised), pooling gradient.  Reference: the autograd of encoders/epic.py layers.
#include "common.h"
#include <stdlib.h>
#include <utility>

namespace mmb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I> struct ICb { static constexpr int v = I; };
template <typename F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ICb<Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

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

__device__ inline float act_grad(int act, float ref) {
  // ref: pre-activation z for GELU; any value with the sign of z for leaky-relu
  if (act == MMB_ACT_GELU) {
    const float c = 0.70710678118654752440f;
    const float cdf = 0.5f * (1.0f + erff(ref * c));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * ref * ref);
    return cdf + ref * pdf;
  }
  if (act == MMB_ACT_LRELU) return ref > 0.0f ? 1.0f : 0.01f;
  if (act == MMB_ACT_RELU) return ref > 0.0f ? 1.0f : 0.0f;
  if (act == MMB_ACT_SELU)          // ref = the OUTPUT y (same sign as z): y > 0 -> scale, else scale*alpha*exp(z) = y + scale*alpha
    return ref > 0.0f ? 1.0507009873554804934193349852946f
                      : ref + 1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f;
  return 1.0f;
}

// forward: a = act(z); y1 = a + res1; y2 = dropout(post(y1)) + res2
// backward: g1 = gY1 + gY2 * keep/(1-p) * post'(y1);  dZ = g1 * act'(z)
struct ActBwd {
  const float *gY1, *gY2, *y1, *y2, *res2, *pre;
  int64_t gY1_ld, gY2_ld, y1_ld, y2_ld, res2_ld, pre_ld, dz_ld;
  int act, post;
  float drop_p;
  uint64_t seed, offset;
};

__global__ void __launch_bounds__(256)
act_bwd_kernel(int64_t M, int Nout, ActBwd a, float* __restrict__ dZ) {
  const int64_t total = M * (int64_t)Nout;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / Nout;
    const int col = (int)(e - row * Nout);
    float g = a.gY1 ? a.gY1[row * a.gY1_ld + col] : 0.0f;
    if (a.gY2) {
      float g2 = a.gY2[row * a.gY2_ld + col];
      if (a.drop_p > 0.0f) {
        uint32_t r[4];
        philox4(a.seed, a.offset, (uint64_t)e >> 2, 0xD0D0u, r);
        g2 = (u01(r[e & 3]) < a.drop_p) ? 0.0f : g2 / (1.0f - a.drop_p);
      }
      float s;
      if (a.y1) s = a.y1[row * a.y1_ld + col];
      else s = a.y2[row * a.y2_ld + col] - (a.res2 ? a.res2[row * a.res2_ld + col] : 0.0f);
      g += g2 * act_grad(a.post, s);
    }
    if (a.act != MMB_ACT_NONE) {
      // (relu followed by a plain dropout, nothing else: y2 = keep * relu(z) has the sign of z wherever g is not 0)
      const float ref = a.pre ? a.pre[row * a.pre_ld + col] : a.y1 ? a.y1[row * a.y1_ld + col] : a.y2[row * a.y2_ld + col];
      g *= act_grad(a.act, ref);
    }
    dZ[row * a.dz_ld + col] = g;
  }
}

// out[b, c] (+)= sum_n x[b,n,c] * (maskf ? maskf[b,n] : 1)
__global__ void __launch_bounds__(256)
jet_sum_kernel(int N, int H, const float* __restrict__ x, int64_t x_ld, const float* __restrict__ maskf,
               float* __restrict__ out, int64_t out_ld, int accumulate) {
  __shared__ float red[4][64];
  const int b = blockIdx.x;
  const int col = threadIdx.x & 63;
  const int rg = threadIdx.x >> 6;
  for (int c0 = 0; c0 < H; c0 += 64) {
    const int c = c0 + col;
    float s = 0.0f;
    if (c < H)
      for (int n = rg; n < N; n += 4) {
        const int64_t p = (int64_t)b * N + n;
        const float v = x[p * x_ld + c];
        s += maskf ? v * maskf[p] : v;
      }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && c < H) {
      const float tot = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
      float* o = out + (int64_t)b * out_ld + c;
      *o = accumulate ? *o + tot : tot;
    }
    __syncthreads();
  }
}

// dX[b,n,c] (+)= m[b,n] * ( dP[b,c]/cnt[b] + 0.01*dP[b,H+c] )      (epic.py:100-107 backward)
__global__ void __launch_bounds__(256)
pool_bwd_kernel(int64_t P, int N, int H, const float* __restrict__ dP, int64_t dp_ld,
                const float* __restrict__ maskf, const int32_t* __restrict__ counts, float* __restrict__ dX,
                int64_t dx_ld, int accumulate) {
  const int64_t total = P * (int64_t)H;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = e / H;
    const int c = (int)(e - p * H);
    const int64_t b = p / N;
    const float g = maskf[p] * (dP[b * dp_ld + c] / (float)counts[b] + 0.01f * dP[b * dp_ld + H + c]);
    float* o = dX + p * dx_ld + c;
    *o = accumulate ? *o + g : g;
  }
}

// dW[o, k] += sum_m dZ[m, o] * cat(X)[m, k] ;  db[o] += sum_m dZ[m, o]  (bias = extra column k == K)
// grid.x: row slabs, grid.y: 32-wide tiles of o.  Wave w of the block owns column tiles
// jt = w + 4q (q < NJ).  Operands go straight from global memory to the MFMA fragments: for the
// transposed product lane l reads dZ[row][o0 + (l&31)] / X[row][j0 + (l&31)] -- 128-byte
// coalesced rows -- and lane half (l>>5) takes the odd/even row of the pair.
template <int NJ>
__global__ void __launch_bounds__(256)
grad_weight_kernel(int M, int rows_per_jet, Segs sg, int K, const float* __restrict__ maskf,
                   const float* __restrict__ dZ, int64_t dz_ld, int Nout, float* __restrict__ dW,
                   int64_t dw_ld, float* __restrict__ db, int rows_per_block) {
  constexpr int U = 8;                            // row pairs in flight per step
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const int i32 = lane & 31;
  const int o0 = blockIdx.y * 32;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(M, r_begin + rows_per_block);
  const bool o_ok = (o0 + i32) < Nout;

  // resolve this lane's input column for each of its tiles once.  Every (row, tile) element is
  // ONE unconditional load from base + (per-row ? row : jet) * ld; masks and the bias column are
  // applied when the value is consumed, so no load is followed by a dependent wait or branch.
  const float* base[NJ];
  int ld[NJ];
  bool per_row[NJ], use_mask[NJ], is_one[NJ], is_zero[NJ];
  sfor<NJ>([&](auto qc) {
    constexpr int q = decltype(qc)::v;
    const int col = (wave + 4 * q) * 32 + i32;
    base[q] = dZ; ld[q] = 0; per_row[q] = false; use_mask[q] = false;
    is_one[q] = (col == K);                                      // bias column: constant one
    is_zero[q] = (col > K);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < sg.n && col >= sg.s[s].k_begin && col < sg.s[s].k_begin + sg.s[s].width) {
        base[q] = sg.s[s].ptr + (col - sg.s[s].k_begin);
        ld[q] = sg.s[s].ld;
        per_row[q] = sg.s[s].kind == MMB_SEG_ROW;
        use_mask[q] = sg.s[s].kind == MMB_SEG_JET_MASK;
      }
  });
  const float* mptr = maskf ? maskf : dZ;          // always-valid address; ignored unless use_mask

  f32x16 acc[NJ];
  sfor<NJ>([&](auto qc) {
    constexpr int q = decltype(qc)::v;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;
  });

  // software pipeline: the loads of row group g+1 are in flight while group g feeds the MFMAs
  float a_nx[U], m_nx[U];
  float b_nx[U][NJ];
  auto load_group = [&](int m0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = m0 + 2 * u + half;
      const int rr = row < r_end ? row : r_end - 1;
      a_nx[u] = dZ[(int64_t)rr * dz_ld + (o_ok ? o0 + i32 : 0)];
      m_nx[u] = mptr[maskf ? rr : 0];
      const int jet = rr / rows_per_jet;
      sfor<NJ>([&](auto qc) {
        constexpr int q = decltype(qc)::v;
        b_nx[u][q] = base[q][(int64_t)(per_row[q] ? rr : jet) * ld[q]];
      });
    }
  };
  load_group(r_begin);
  for (int m0 = r_begin; m0 < r_end; m0 += 2 * U) {
    float a[U];
    float b[U][NJ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float okf = (m0 + 2 * u + half) < r_end ? 1.0f : 0.0f;
      a[u] = o_ok ? a_nx[u] * okf : 0.0f;
      sfor<NJ>([&](auto qc) {
        constexpr int q = decltype(qc)::v;
        float v = use_mask[q] ? b_nx[u][q] * m_nx[u] : b_nx[u][q];
        v = is_one[q] ? 1.0f : v;
        b[u][q] = is_zero[q] ? 0.0f : v;
      });
    }
    if (m0 + 2 * U < r_end) load_group(m0 + 2 * U);
#pragma unroll
    for (int u = 0; u < U; ++u)
      sfor<NJ>([&](auto qc) {
        constexpr int q = decltype(qc)::v;
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][q], acc[q], 0, 0, 0);
      });
  }

  sfor<NJ>([&](auto qc) {
    constexpr int q = decltype(qc)::v;
    const int col = (wave + 4 * q) * 32 + i32;
    sfor<16>([&](auto ec) {
      constexpr int e = decltype(ec)::v;
      const int o = o0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      if (o < Nout) {
        if (col < K) atomicAdd(&dW[(int64_t)o * dw_ld + col], acc[q][e]);
        else if (col == K && db) atomicAdd(&db[o], acc[q][e]);
      }
    });
  });
}

// ---- weight gradient, row-streaming form ------------------------------------------------------
// dW[o][k] (+)= sum_rows dZ[row][o] * X[row][k] for ONE row-major input X (a per-particle ROW
// segment with M = B*N rows, or a per-jet matrix with M = B rows: the broadcast segments of a layer
// enter through the jet sums of dZ).  A 64x64 layer is a 67 MB stream against 7 us of MFMA work, so
// the kernel is built around bytes in flight: every wave streams its OWN 16-row chunks with
// dwordx4 loads (1 KiB per instruction, each operand byte fetched exactly once per block), parks
// them in a wave-private LDS tile and reads the MFMA fragments back transposed: lane l takes
// dZ[row][o0 + 32 ot + (l&31)] as the A operand (m = o) and X[row][k0 + 32 kt + (l&31)] as the B
// operand (n = k) of v_mfma_f32_32x32x2_f32, lane half l>>5 = row parity.  No block barrier inside
// the stream (a wave only reads what it wrote).  The four waves' partial tiles are summed through
// LDS and leave as one atomicAdd per element and block; db is the column sum of the A values.
// Requires K and Nout to be multiples of 4 and 16-byte aligned rows (ragged last windows are zero-filled).
// exact three-plane bf16 split of 8 floats (fused_common.h: split8): hi + mid + lo = the fp32 value
typedef __bf16 gw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gw_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t gw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t gw_pk(float a, float b) {
  const gw_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void gw_split8(const float (&v)[8], gw_bf16x8& h, gw_bf16x8& m, gw_bf16x8& l) {
  uint32_t ph[4], pm[4], pq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    ph[j] = gw_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, ph[j] << 16), rb = b - __builtin_bit_cast(float, ph[j] & 0xFFFF0000u);
    pm[j] = gw_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, pm[j] << 16), sb = rb - __builtin_bit_cast(float, pm[j] & 0xFFFF0000u);
    pq[j] = gw_pk(sa, sb);
  }
  const gw_u32x4 H = {ph[0], ph[1], ph[2], ph[3]}, M = {pm[0], pm[1], pm[2], pm[3]}, L = {pq[0], pq[1], pq[2], pq[3]};
  h = __builtin_bit_cast(gw_bf16x8, H);
  m = __builtin_bit_cast(gw_bf16x8, M);
  l = __builtin_bit_cast(gw_bf16x8, L);
}

// X6: the 16 rows of a chunk are ONE k-extent of v_mfma_f32_32x32x16_bf16 (lane half h takes rows 8h..8h+7 of its
// column); both operands are split in registers into three exact bf16 planes and every fp32 product is six plane
// products (small terms first) -- 6 x 32 cycles per tile and chunk instead of 8 x 64 on the fp32 MFMA.
template <int OT, int KT, bool X6>
__device__ __forceinline__ void
grad_weight_rows_body(int M, const float* __restrict__ X, int64_t x_ld, int K, const float* __restrict__ dZ,
                      int64_t dz_ld, int Nout, float* __restrict__ dW, int64_t dw_ld, float* __restrict__ db,
                      int rows_per_block, int replicas, int64_t rep_stride, int64_t db_stride, int bx, int by, int bz,
                      const int32_t* __restrict__ rows = nullptr, const int32_t* __restrict__ n_rows = nullptr,
                      int nbx = 1) {
  // rows / n_rows (both or neither): the reduction runs over rows[0 .. *n_rows) of X and dZ instead of 0 .. M -- the
  // rows of real particles (mmb_mask_rows).  Rows of padded slots carry gradient rows that are exactly zero (the losses
  // are masked): on AOJ multiplicities that is 62 % of a per-particle problem.  The grid stays sized for M rows, and
  // the list's length is only known on the device: the 16-row chunks are dealt out round-robin over the nbx row blocks
  // (block bx takes chunk groups bx, bx + nbx, ...), so that every block gets its share of however many rows there are.
  const bool listed = rows != nullptr && n_rows != nullptr;
  if (listed) M = min(M, *n_rows);
  if ((listed ? bx * 4 * 16 : bx * rows_per_block) >= M) return;
  constexpr int R = 16;                                  // rows per chunk
  {                                                      // this block's replica of the accumulators
    const int rep = bx % replicas;
    dW += rep * rep_stride;
    if (db != nullptr) db += rep * db_stride;
  }
  constexpr int KW = 32 * KT, OW = 32 * OT;              // window widths
  constexpr int XP = KW + 4, ZP = OW + 4;                // LDS pitches (floats)
  constexpr int XF4 = R * KW / 4 / 64, ZF4 = R * OW / 4 / 64;   // dwordx4 per lane and chunk
  constexpr int TILE = R * (XP + ZP);                    // floats per wave
  constexpr int REDN = OT * KT * 16 * 64;
  constexpr int LDSN = 4 * TILE > REDN ? 4 * TILE : REDN;
  __shared__ __attribute__((aligned(16))) float lds[LDSN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, i32 = lane & 31;
  const int o0 = bz * OW, k0 = by * KW;
  const int r_begin = listed ? bx * 4 * R : bx * rows_per_block;
  const int r_end = listed ? M : min(M, r_begin + rows_per_block);
  float* xs = lds + wave * TILE;
  float* zs = xs + R * XP;

  f32x16 acc[OT][KT];
  sfor<OT>([&](auto oc) { sfor<KT>([&](auto kc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[decltype(oc)::v][decltype(kc)::v][e] = 0.0f; }); });
  float dbs[OT];
  sfor<OT>([&](auto c) { dbs[decltype(c)::v] = 0.0f; });

  // per-lane position inside a chunk for each of its dwordx4 loads
  int xr[XF4], xc[XF4], zr[ZF4], zc[ZF4];
  sfor<XF4>([&](auto c) { constexpr int i = decltype(c)::v; const int f = i * 64 + lane; xr[i] = f / (KW / 4); xc[i] = 4 * (f % (KW / 4)); });
  sfor<ZF4>([&](auto c) { constexpr int i = decltype(c)::v; const int f = i * 64 + lane; zr[i] = f / (OW / 4); zc[i] = 4 * (f % (OW / 4)); });

  // two chunks in flight per wave (buffers A, B): the stream is latency-bound with one.  Loads are unconditional
  // (rows past the end are clamped here and zeroed at the park), so their number in flight is static and the park of
  // the older buffer waits for that buffer only.
  float4 xa[XF4], za[ZF4], xb[XF4], zb[ZF4];
  // Row list: the 16 row numbers of a chunk are ONE load by lanes 0..15 (row_ids), requested a whole iteration before
  // the chunk's data loads need them and handed to the lanes by a bpermute -- a per-lane rows[] load in front of every
  // data load put two dependent latencies into a stream that is latency-bound already.
  auto row_ids = [&](int row0) __attribute__((always_inline)) -> int {
    return listed ? rows[min(row0 + (lane & 15), r_end - 1)] : 0;
  };
  auto load_chunk = [&](int row0, int ids, float4 (&xn)[XF4], float4 (&zn)[ZF4]) __attribute__((always_inline)) {
    sfor<XF4>([&](auto c) { constexpr int i = decltype(c)::v;          // (no branch: rows AND columns clamped into
      int64_t rr = min(row0 + xr[i], r_end - 1);                       //  valid memory, zeroed at the park)
      if (listed) rr = __shfl(ids, xr[i], 64);
      xn[i] = *reinterpret_cast<const float4*>(X + rr * x_ld + min(k0 + xc[i], K - 4)); });
    sfor<ZF4>([&](auto c) { constexpr int i = decltype(c)::v;
      int64_t rr = min(row0 + zr[i], r_end - 1);
      if (listed) rr = __shfl(ids, zr[i], 64);
      zn[i] = *reinterpret_cast<const float4*>(dZ + rr * dz_ld + min(o0 + zc[i], Nout - 4)); });
  };
  auto park_chunk = [&](int row0, const float4 (&xn)[XF4], const float4 (&zn)[ZF4]) __attribute__((always_inline)) {
    sfor<XF4>([&](auto c) { constexpr int i = decltype(c)::v;
      float4 v = xn[i];
      if (k0 + xc[i] >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);        // columns past K (ragged last window)
      *reinterpret_cast<float4*>(xs + xr[i] * XP + xc[i]) = v; });
    sfor<ZF4>([&](auto c) { constexpr int i = decltype(c)::v;
      float4 v = zn[i];
      if (row0 + zr[i] >= r_end || o0 + zc[i] >= Nout) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(zs + zr[i] * ZP + zc[i]) = v; });
  };

  // wave w owns chunks w, w+4, ... of the block
  const int stride = listed ? 4 * R * nbx : 4 * R;
  const bool two = listed || rows_per_block % (2 * stride) == 0;
  int row0 = r_begin + wave * R;
  int ida = row_ids(row0), idb = two ? row_ids(row0 + stride) : 0;
  load_chunk(row0, ida, xa, za);
  if (two) load_chunk(row0 + stride, idb, xb, zb);
  ida = row_ids(row0 + (two ? 2 : 1) * stride);
  if (two) idb = row_ids(row0 + 3 * stride);
  auto mma_chunk = [&]() __attribute__((always_inline)) {
    if constexpr (X6) {
      gw_bf16x8 ah[OT], am[OT], al[OT];
      sfor<OT>([&](auto c) { constexpr int t = decltype(c)::v;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = zs[(8 * half + j) * ZP + 32 * t + i32];
        dbs[t] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        gw_split8(v, ah[t], am[t], al[t]); });
      sfor<KT>([&](auto kc) { constexpr int kt = decltype(kc)::v;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xs[(8 * half + j) * XP + 32 * kt + i32];
        gw_bf16x8 bh, bm, bl;
        gw_split8(v, bh, bm, bl);
        sfor<OT>([&](auto oc) { constexpr int ot = decltype(oc)::v;
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ot], bh, acc[ot][kt], 0, 0, 0);
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ot], bm, acc[ot][kt], 0, 0, 0);
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ot], bl, acc[ot][kt], 0, 0, 0);
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ot], bh, acc[ot][kt], 0, 0, 0);
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ot], bm, acc[ot][kt], 0, 0, 0);
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ot], bh, acc[ot][kt], 0, 0, 0); }); });
    } else {
#pragma unroll
    for (int p = 0; p < R / 2; ++p) {
      float a[OT], b[KT];
      sfor<OT>([&](auto c) { constexpr int t = decltype(c)::v; a[t] = zs[(2 * p + half) * ZP + 32 * t + i32]; });
      sfor<KT>([&](auto c) { constexpr int t = decltype(c)::v; b[t] = xs[(2 * p + half) * XP + 32 * t + i32]; });
      sfor<OT>([&](auto oc) { constexpr int ot = decltype(oc)::v;
        dbs[ot] += a[ot];
        sfor<KT>([&](auto kc) { constexpr int kt = decltype(kc)::v;
          acc[ot][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ot], b[kt], acc[ot][kt], 0, 0, 0); }); });
    }
    }
  };
  if (two) {
    // every wave owns an EVEN number of chunks (the last block's may lie past the end: clamped loads, zeroed dZ),
    // so both halves run unconditionally and the waits stay static
    for (; row0 < r_end; row0 += 2 * stride) {
      park_chunk(row0, xa, za);                         // (waits for buffer A only)
      load_chunk(row0 + 2 * stride, ida, xa, za);
      ida = row_ids(row0 + 4 * stride);
      mma_chunk();
      park_chunk(row0 + stride, xb, zb);
      load_chunk(row0 + 3 * stride, idb, xb, zb);
      idb = row_ids(row0 + 5 * stride);
      mma_chunk();
    }
  } else {                                              // short blocks (per-jet problems): one buffer
    for (; row0 < r_end; row0 += stride) {
      park_chunk(row0, xa, za);
      load_chunk(row0 + stride, ida, xa, za);
      ida = row_ids(row0 + 2 * stride);
      mma_chunk();
    }
  }

  // block reduction of the four waves' tiles: wave 0 stores, waves 1..2 add, wave 3 adds and flushes
  __syncthreads();                                       // every wave is done with its private tile
  auto each = [&](auto&& fn) {                           // value = fn(slot, value) over this lane's accumulators
    sfor<OT>([&](auto oc) { sfor<KT>([&](auto kc) {
      constexpr int ot = decltype(oc)::v, kt = decltype(kc)::v;
      sfor<16>([&](auto ec) { constexpr int e = decltype(ec)::v;
        acc[ot][kt][e] = fn(&lds[((ot * KT + kt) * 16 + e) * 64 + lane], acc[ot][kt][e]); }); }); });
  };
  if (wave == 0) each([](float* slot, float v) { *slot = v; return v; });
  __syncthreads();
  if (wave == 1) each([](float* slot, float v) { *slot += v; return v; });
  __syncthreads();
  if (wave == 2) each([](float* slot, float v) { *slot += v; return v; });
  __syncthreads();
  if (wave == 3) each([](float* slot, float v) { *slot += v; return v; });
  __syncthreads();
  // flush: the OT*KT summed tiles are shared out over the four waves (one wave alone issued 16*OT*KT
  // atomics per lane back to back: a third of the kernel's time at 512 rows per block)
  sfor<OT>([&](auto oc) { sfor<KT>([&](auto kc) {
    constexpr int ot = decltype(oc)::v, kt = decltype(kc)::v;
    if (((ot * KT + kt) & 3) == wave) {
      const int col = k0 + 32 * kt + i32;
      sfor<16>([&](auto ec) { constexpr int e = decltype(ec)::v;
        const int o = o0 + 32 * ot + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (o < Nout && col < K) atomicAdd(&dW[(int64_t)o * dw_ld + col], lds[((ot * KT + kt) * 16 + e) * 64 + lane]); });
    } }); });
  __syncthreads();                                       // (db below reuses lds)
  // db: one atomic per column and BLOCK (same-address atomics serialise at the memory side: ~60 ns each)
  if (db != nullptr && by == 0) {
    sfor<OT>([&](auto oc) { constexpr int ot = decltype(oc)::v;
      float v = dbs[ot];
      v += __shfl_xor(v, 32, 64);
      if (half == 0) lds[wave * OW + 32 * ot + i32] = v; });
    __syncthreads();
    if (threadIdx.x < OW && o0 + (int)threadIdx.x < Nout) {
      const int c = threadIdx.x;
      atomicAdd(&db[o0 + c], (lds[c] + lds[OW + c]) + (lds[2 * OW + c] + lds[3 * OW + c]));
    }
  }
}

template <int OT, int KT, bool X6>
__global__ void __launch_bounds__(256)
grad_weight_rows_kernel(int M, const float* __restrict__ X, int64_t x_ld, int K, const float* __restrict__ dZ,
                        int64_t dz_ld, int Nout, float* __restrict__ dW, int64_t dw_ld, float* __restrict__ db,
                        int rows_per_block, int replicas, int64_t rep_stride, int64_t db_stride, int nbx, int nwin,
                        int nby) {
  // 1-D grid, XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2); the
  // nwin = nby * nbz workgroups that stream the SAME rows (one per K window x output window) get ids 8 apart -- same
  // XCD, dispatched together -- so the rows come from HBM once instead of once per window (128 -> 384 at OT x KT =
  // 4 x 2: 708 MB of re-reads against 315 MB)
  const int g = blockIdx.x, slot = g >> 3;
  const int win = slot % nwin, bx = (slot / nwin) * 8 + (g & 7);
  if (bx >= nbx) return;
  grad_weight_rows_body<OT, KT, X6>(M, X, x_ld, K, dZ, dz_ld, Nout, dW, dw_ld, db, rows_per_block, replicas, rep_stride,
                                db_stride, bx, win % nby, win / nby);
}

// Many independent weight-gradient problems of one tile class in ONE launch (a training step has ~100 of
// them, most far too small to fill the chip alone).  The device table holds one record per problem with its
// range of linear block ids; a block finds its problem by a scalar scan.
struct GwProb {            // all fields 8 bytes: built on the host as an int64 tensor
  const float* X;
  const float* dZ;
  float* dW;
  float* db;               // or NULL
  int64_t x_ld, dz_ld, dw_ld, rep_stride, db_stride;
  int64_t M, K, Nout, rpb, replicas;
  int64_t nbx, nby;        // blocks along the rows / the K windows (the rest of the range: output windows)
  int64_t block_begin, block_end;
  const int32_t* rows;     // row list of the reduction (mmb_mask_rows) or NULL: all M rows
  const int32_t* n_rows;   // its length, on the device
};

template <int OT, int KT, bool X6>
__global__ void __launch_bounds__(256)
grad_weight_rows_multi_kernel(const GwProb* __restrict__ probs, int n_probs) {
  const int64_t id = blockIdx.x;
  int i = 0;
  while (i + 1 < n_probs && id >= probs[i].block_end) ++i;
  const GwProb q = probs[i];
  if (id < q.block_begin || id >= q.block_end) return;
  const int local = (int)(id - q.block_begin);
  const int nbx = (int)q.nbx, nby = (int)q.nby;
  int bx, by, bz;
  if ((nbx & 7) == 0 && (q.block_begin & 7) == 0) {
    // XCD-aware order: consecutive block ids go round-robin over the 8 XCDs, each with its own L2.  The nby
    // blocks that stream the SAME rows of dZ (one per K window) are given ids 8 apart -- same XCD, dispatched
    // together -- so the second reader finds the rows in that L2 instead of fetching them again.
    const int xcd = local & 7, slot = local >> 3;
    by = slot % nby;
    const int t = slot / nby;
    bx = (t % (nbx >> 3)) * 8 + xcd;
    bz = t / (nbx >> 3);
  } else {
    bx = local % nbx;
    const int rest = local / nbx;
    by = rest % nby;
    bz = rest / nby;
  }
  grad_weight_rows_body<OT, KT, X6>((int)q.M, q.X, q.x_ld, (int)q.K, q.dZ, q.dz_ld, (int)q.Nout, q.dW, q.dw_ld, q.db,
                                (int)q.rpb, (int)q.replicas, q.rep_stride, q.db_stride, bx, by, bz, q.rows, q.n_rows, nbx);
}

// rows[offs[b] + i] = b N + (slot of the i-th real particle of jet b): the flat rows of the real particles, ascending.
// offs: exclusive prefix sum of the per-jet particle counts.  One wave per jet.
__global__ void __launch_bounds__(256)
mask_rows_kernel(int B, int N, const float* __restrict__ maskf, const int32_t* __restrict__ offs, int32_t* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int at = offs[b];
  for (int p0 = 0; p0 < N; p0 += 64) {
    const int p = p0 + lane;
    const bool real = p < N && maskf[(int64_t)b * N + p] != 0.0f;
    const unsigned long long bal = __ballot(real);
    if (real) rows[at + __popcll(bal & ((1ull << lane) - 1ull))] = b * N + p;
    at += __popcll(bal);
  }
}

// dW[o][k] += sum_rows dZ[row][o] X[row][k], db[o] += sum_rows dZ[row][o] for a VERY narrow input (K <= 4: the
// continuous embedding, Linear(3 -> Ec), embedder.py:57): an MFMA tile would be 29/32 padding, so this is a plain
// streaming reduction -- one (o, row-slice) per lane, K+1 accumulators, one atomic per lane group at the end.
__global__ void __launch_bounds__(256)
grad_weight_smallk_kernel(int M, const float* __restrict__ X, int64_t x_ld, int K, const float* __restrict__ dZ,
                          int64_t dz_ld, int Nout, float* __restrict__ dW, int64_t dw_ld, float* __restrict__ db) {
  // thread = (row lane r of 256/NP, output o): NP = Nout rounded up to a power of two <= 32
  int NP = 1;
  while (NP < Nout) NP <<= 1;
  const int o = threadIdx.x % NP, r = threadIdx.x / NP, RL = 256 / NP;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (o < Nout)
    for (int64_t row = (int64_t)blockIdx.x * RL + r; row < M; row += (int64_t)gridDim.x * RL) {
      const float z = dZ[row * dz_ld + o];
      acc[4] += z;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < K) acc[k] += z * X[row * x_ld + k];
    }
  __shared__ float red[5][256];
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  if (r == 0 && o < Nout) {
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < RL; ++j)
#pragma unroll
      for (int k = 0; k < 5; ++k) s[k] += red[k][j * NP + o];
    for (int k = 0; k < K; ++k) atomicAdd(&dW[(int64_t)o * dw_ld + k], s[k]);
    if (db != nullptr) atomicAdd(&db[o], s[4]);
  }
}

// weight-gradient arithmetic (A/B switch for tests and profiles): MMB_GW_PATH = f32 (fp32 MFMA) | x6 (default: six
// bf16-plane products per fp32 product)
static bool gw_x6() {
  const char* e = getenv("MMB_GW_PATH");
  return !(e && e[0] == 'f');
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_mask_rows(int B, int N, const float* maskf, const int32_t* offs, int32_t* rows, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && maskf && offs && rows, "mmb_mask_rows: bad argument (B=%d N=%d)", B, N);
  MMB_REQUIRE((int64_t)B * N < (1ll << 31), "mmb_mask_rows: B*N = %lld rows do not fit 32-bit row indices", (long long)B * N);
  hipLaunchKernelGGL(mask_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), B, N, maskf, offs, rows);
  return check_launch("mmb_mask_rows");
}

extern "C" int mmb_grad_weight_smallk(int M, const float* X, int64_t x_ld, int K, const float* dZ, int64_t dz_ld,
                                      int Nout, float* dW, int64_t dw_ld, float* db, void* stream) {
  MMB_REQUIRE(M > 0 && K >= 1 && K <= 4 && Nout >= 1 && Nout <= 32, "mmb_grad_weight_smallk: needs K <= 4, Nout <= 32 (K=%d Nout=%d)", K, Nout);
  MMB_REQUIRE(X && dZ && dW && x_ld >= K && dz_ld >= Nout && dw_ld >= K, "mmb_grad_weight_smallk: bad argument");
  int NP = 1;
  while (NP < Nout) NP <<= 1;
  const int rows_per_pass = 256 / NP;                       // rows a block covers per loop iteration
  // every block ends with one atomic per (o, k) on the SAME few addresses (they serialise at the memory side,
  // ~60 ns each), so few, long blocks: ~32 iterations per thread, at most 256 blocks
  int grid = (M + rows_per_pass * 32 - 1) / (rows_per_pass * 32);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(grad_weight_smallk_kernel, dim3(grid), dim3(256), 0, as_stream(stream), M, X, x_ld, K, dZ, dz_ld,
                     Nout, dW, dw_ld, db);
  return check_launch("mmb_grad_weight_smallk");
}

// probs: device table of n_probs GwProb records of tile class (OTb, KTb) (see mmb_hip.ops.gw_class), ranges of
// linear block ids ascending and contiguous from 0 to total_blocks.
extern "C" int mmb_grad_weight_rows_multi(int OTb, int KTb, const void* probs, int n_probs, int total_blocks,
                                          void* stream) {
  MMB_REQUIRE(probs && n_probs > 0 && total_blocks > 0, "mmb_grad_weight_rows_multi: empty table");
  const GwProb* P = reinterpret_cast<const GwProb*>(probs);
  hipStream_t st = as_stream(stream);
  const bool x6 = gw_x6();
#define GWM(O_, K_)                                                                                                    \
  do {                                                                                                                 \
    if (x6) hipLaunchKernelGGL((grad_weight_rows_multi_kernel<O_, K_, true>), dim3(total_blocks), dim3(256), 0, st, P, n_probs); \
    else hipLaunchKernelGGL((grad_weight_rows_multi_kernel<O_, K_, false>), dim3(total_blocks), dim3(256), 0, st, P, n_probs);   \
  } while (0)
  if (OTb == 1 && KTb == 1) GWM(1, 1);
  else if (OTb == 1 && KTb == 2) GWM(1, 2);
  else if (OTb == 1 && KTb == 4) GWM(1, 4);
  else if (OTb == 2 && KTb == 1) GWM(2, 1);
  else if (OTb == 2 && KTb == 2) GWM(2, 2);
  else if (OTb == 2 && KTb == 4) GWM(2, 4);
  else if (OTb == 4 && KTb == 1) GWM(4, 1);
  else if (OTb == 4 && KTb == 2) GWM(4, 2);
  else MMB_REQUIRE(false, "mmb_grad_weight_rows_multi: no tile class (%d,%d)", OTb, KTb);
#undef GWM
  return check_launch("mmb_grad_weight_rows_multi");
}

extern "C" int mmb_grad_weight_rows(int M, const float* X, int64_t x_ld, int K, const float* dZ, int64_t dz_ld,
                                    int Nout, float* dW, int64_t dw_ld, float* db, int replicas, int64_t rep_stride,
                                    int64_t db_stride, void* stream) {
  MMB_REQUIRE(replicas >= 1 && replicas <= 64, "mmb_grad_weight_rows: replicas=%d outside [1,64]", replicas);
  MMB_REQUIRE(M > 0 && K > 0 && Nout > 0, "mmb_grad_weight_rows: empty shape M=%d K=%d Nout=%d", M, K, Nout);
  MMB_REQUIRE(X && dZ && dW, "mmb_grad_weight_rows: NULL argument");
  MMB_REQUIRE(x_ld >= K && dz_ld >= Nout && dw_ld >= K, "mmb_grad_weight_rows: leading dimension too small");
  const bool fast = (K % 4 == 0) && (Nout % 4 == 0) && (x_ld % 4 == 0) && (dz_ld % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dZ) & 15) == 0);
  if (!fast) {                                           // ragged widths: the generic segment kernel
    const void* ptr[1] = {X};
    int width[1] = {K}, kind[1] = {MMB_SEG_ROW}, ld[1] = {(int)x_ld};
    return mmb_grad_weight(M, 1, 1, ptr, width, kind, ld, nullptr, dZ, dz_ld, Nout, dW, dw_ld, db, stream);
  }
  const int ot_all = (Nout + 31) / 32, kt_all = (K + 31) / 32;
  int OTb = ot_all >= 4 ? 4 : (ot_all >= 2 ? 2 : 1);
  int KTb = kt_all >= 4 ? 4 : (kt_all >= 2 ? 2 : 1);
  // tiles per block: 8 keep each row on chip longest but leave one wave per SIMD (182 registers), and the stream is
  // latency-bound (one 16-row chunk in flight per wave); 4 tiles run 3-4 waves per SIMD -- the extra windows re-read
  // the rows from L2, not HBM (XCD-aware order).  Sweep on the S4 shapes: 1362 / 953 / 1037 us at 8 / 4 / 2 tiles.
  int max_tiles = M >= 65536 ? 4 : 8;
  if (const char* e = getenv("MMB_GW_TILES")) max_tiles = atoi(e) > 0 ? atoi(e) : max_tiles;      // tuning knob
  while (OTb * KTb > max_tiles) { if (KTb >= OTb) KTb /= 2; else OTb /= 2; }
  int rpb = M >= 65536 ? 1024 : 512;                     // 16 / 8 chunks per wave
  if (M <= 16384) rpb = 64;
  if (const char* e = getenv("MMB_GW_RPB")) rpb = atoi(e) > 0 ? atoi(e) : rpb;      // tuning knob
  const int nbx = (M + rpb - 1) / rpb, nby = (kt_all + KTb - 1) / KTb, nbz = (ot_all + OTb - 1) / OTb;
  const int nwin = nby * nbz;
  MMB_REQUIRE((int64_t)((nbx + 7) / 8) * 8 * nwin < (1ll << 31), "mmb_grad_weight_rows: too many blocks");
  dim3 grid((nbx + 7) / 8 * 8 * nwin);
  hipStream_t st = as_stream(stream);
  const bool x6 = gw_x6();
#define GWR(O_, K_)                                                                                                    \
  do {                                                                                                                 \
    if (x6) hipLaunchKernelGGL((grad_weight_rows_kernel<O_, K_, true>), grid, dim3(256), 0, st, M, X, x_ld, K, dZ, dz_ld, Nout, dW, dw_ld, db, rpb, replicas, rep_stride, db_stride, nbx, nwin, nby); \
    else hipLaunchKernelGGL((grad_weight_rows_kernel<O_, K_, false>), grid, dim3(256), 0, st, M, X, x_ld, K, dZ, dz_ld, Nout, dW, dw_ld, db, rpb, replicas, rep_stride, db_stride, nbx, nwin, nby);   \
  } while (0)
  if (OTb == 1 && KTb == 1) GWR(1, 1);
  else if (OTb == 1 && KTb == 2) GWR(1, 2);
  else if (OTb == 1 && KTb == 4) GWR(1, 4);
  else if (OTb == 2 && KTb == 1) GWR(2, 1);
  else if (OTb == 2 && KTb == 2) GWR(2, 2);
  else if (OTb == 2 && KTb == 4) GWR(2, 4);
  else if (OTb == 4 && KTb == 1) GWR(4, 1);
  else GWR(4, 2);
#undef GWR
  return check_launch("mmb_grad_weight_rows");
}

extern "C" int mmb_act_bwd(int64_t M, int Nout, const float* gY1, int64_t gY1_ld, const float* gY2, int64_t gY2_ld,
                           const float* y1, int64_t y1_ld, const float* y2, int64_t y2_ld, const float* res2,
                           int64_t res2_ld, const float* pre, int64_t pre_ld, int act, int post, float drop_p,
                           uint64_t seed, uint64_t offset, float* dZ, int64_t dz_ld, void* stream) {
  MMB_REQUIRE(M > 0 && Nout > 0, "mmb_act_bwd: empty shape");
  MMB_REQUIRE(dZ && (gY1 || gY2), "mmb_act_bwd: need dZ and at least one incoming gradient");
  MMB_REQUIRE(act >= 0 && act <= 4 && post >= 0 && post <= 3, "mmb_act_bwd: bad activation");
  if (gY2 && post != MMB_ACT_NONE) MMB_REQUIRE(y1 || y2, "mmb_act_bwd: post activation needs y1 or y2");
  if (act == MMB_ACT_GELU) MMB_REQUIRE(pre, "mmb_act_bwd: GELU needs the pre-activation");
  const bool relu_from_y2 = act == MMB_ACT_RELU && !pre && !y1 && y2 && !gY1 && !res2 && post == MMB_ACT_NONE;
  if (act == MMB_ACT_LRELU || act == MMB_ACT_RELU)
    MMB_REQUIRE(pre || y1 || relu_from_y2, "mmb_act_bwd: (leaky-)relu needs pre or y1 (or, for relu + dropout alone, y2)");
  if (act == MMB_ACT_SELU) MMB_REQUIRE(y1 && !pre, "mmb_act_bwd: SELU takes its derivative from the output y1 (no residual on it)");
  ActBwd a{gY1, gY2, y1, y2, res2, pre, gY1_ld, gY2_ld, y1_ld, y2_ld, res2_ld, pre_ld, dz_ld, act, post, drop_p, seed, offset};
  hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(M * Nout, 256)), dim3(256), 0, as_stream(stream), M, Nout, a, dZ);
  return check_launch("mmb_act_bwd");
}

extern "C" int mmb_jet_sum(int B, int N, int H, const float* x, int64_t x_ld, const float* maskf, float* out,
                           int64_t out_ld, int accumulate, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && H > 0, "mmb_jet_sum: empty shape");
  MMB_REQUIRE(x && out && x_ld >= H && out_ld >= H, "mmb_jet_sum: bad argument");
  hipLaunchKernelGGL(jet_sum_kernel, dim3(B), dim3(256), 0, as_stream(stream), N, H, x, x_ld, maskf, out, out_ld,
                     accumulate);
  return check_launch("mmb_jet_sum");
}

extern "C" int mmb_pool_bwd(int B, int N, int H, const float* dP, int64_t dp_ld, const float* maskf,
                            const int32_t* counts, float* dX, int64_t dx_ld, int accumulate, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && H > 0, "mmb_pool_bwd: empty shape");
  MMB_REQUIRE(dP && maskf && counts && dX && dp_ld >= 2 * H && dx_ld >= H, "mmb_pool_bwd: bad argument");
  const int64_t P = (int64_t)B * N;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(stream_grid(P * H, 256)), dim3(256), 0, as_stream(stream), P, N, H, dP,
                     dp_ld, maskf, counts, dX, dx_ld, accumulate);
  return check_launch("mmb_pool_bwd");
}

extern "C" int mmb_grad_weight(int M, int rows_per_jet, int n_seg, const void* const* seg_ptr, const int* seg_width,
                               const int* seg_kind, const int* seg_ld, const float* maskf, const float* dZ,
                               int64_t dz_ld, int Nout, float* dW, int64_t dw_ld, float* db, void* stream) {
  MMB_REQUIRE(M > 0 && Nout > 0 && rows_per_jet > 0, "mmb_grad_weight: bad shape");
  MMB_REQUIRE(n_seg >= 1 && n_seg <= 4 && seg_ptr && seg_width && seg_kind && seg_ld, "mmb_grad_weight: bad segments");
  MMB_REQUIRE(dZ && dW && dz_ld >= Nout, "mmb_grad_weight: bad argument");
  Segs sg;
  sg.n = n_seg;
  int K = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < n_seg) {
      MMB_REQUIRE(seg_ptr[i] && seg_width[i] > 0 && seg_kind[i] >= 0 && seg_kind[i] <= 2 && seg_ld[i] >= seg_width[i],
                  "mmb_grad_weight: segment %d invalid", i);
      if (seg_kind[i] == MMB_SEG_JET_MASK) MMB_REQUIRE(maskf, "mmb_grad_weight: JET_MASK segment needs maskf");
      sg.s[i] = Seg{(const float*)seg_ptr[i], seg_width[i], seg_kind[i], seg_ld[i], K};
      K += seg_width[i];
    } else {
      sg.s[i] = Seg{nullptr, 0, 0, 0, 0};
    }
  }
  MMB_REQUIRE(dw_ld >= K, "mmb_grad_weight: dw_ld < K");
  const int TJ = (K + 1 + 31) / 32;              // +1: the bias column
  const int NJ = (TJ + 3) / 4;
  MMB_REQUIRE(NJ <= 5, "mmb_grad_weight: K=%d too wide (max 639)", K);
  int rpb = 1024;
  if (M < 64 * 1024) rpb = 256;
  if (M < 4096) rpb = 64;
  dim3 grid((M + rpb - 1) / rpb, (Nout + 31) / 32);
  hipStream_t st = as_stream(stream);
#define GW(NJ_) hipLaunchKernelGGL((grad_weight_kernel<NJ_>), grid, dim3(256), 0, st, M, rows_per_jet, sg, K, maskf, dZ, dz_ld, Nout, dW, dw_ld, db, rpb)
  switch (NJ) {
    case 1: GW(1); break;
    case 2: GW(2); break;
    case 3: GW(3); break;
    case 4: GW(4); break;
    default: GW(5); break;
  }
#undef GW
  return check_launch("mmb_grad_weight");
}
