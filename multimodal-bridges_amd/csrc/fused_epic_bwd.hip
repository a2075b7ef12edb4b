// K-E fused, training backward: the reverse pass of the whole MultiModalEPiC for one jet inside one
// workgroup.  Mirror of fused_epic.hip's forward_jet: gradients of activations stay TRANSPOSED in the
// 32x32 MFMA C/D layout (lane&31 = particle), so dX^T = W^T dZ^T is the same MFMA stream
// as the forward with the transposed weight pre-packed in fragment order, and nothing but the per-layer
// dZ (row-major, what the weight-gradient kernel streams) and the per-jet column sums of dZ (the
// gradients of the broadcast time / global-vector inputs) ever goes to HBM.  Saved activations come
// back from the forward's slots with buffer loads (lanes of padded particles read 0).
// Reference: autograd of encoders/multimodal_epic.py:158-194, encoders/epic.py:71-107,145-159,187-208.
//
// Rows of masked particles: the loss gradient is 0 there, every per-particle op is row-wise and the
// pooling gradient is multiplied by the mask, so dZ is 0 on them in every layer -- masked and unmasked
// column sums coincide and one butterfly per layer serves both the time and the global-vector segment.
#ifndef MMB_FE_WAVES
#define MMB_FE_WAVES 4
#endif
#if MMB_FE_WAVES == 5     // second compilation (fused_epic_bwd_w5.hip): five waves per jet, N <= 160
#define mmb_fused_epic_train_backward mmb_fused_epic_train_backward_w5
#define mmb_fused_uni_train_backward mmb_fused_uni_train_backward_w5
#endif
#include "fused_common.h"

namespace mmb {

// per layer, forward order: per-particle layers {Wx^T in fragment order, Wt (O,Et), Wg (O,G)},
// per-jet layers {W[:, first segment] (O,K0), W[:, second segment] (O,K1), -}; -1 = absent
struct BDesc {
  int a, b, c;
};

struct alignas(16) LdsB {
  float dtemb[128];                      // gradient of the time embedding, accumulated over the jet
  float dg[64], dgskip[64], dgnew[64], dgcat[64];
  float dgh[128], dzg[128];
  float dpool[256];
  float cpool[128];
  float svec[128];
  float partial[FE_WAVES * 128];
  float cnt;
};

struct BwdArgs {
  const float* P;          // packed backward parameters
  const BDesc* D;
  int nb_branch, nb_fused_exec, use_skip;      // nb_branch: continuous branch; nb_branch1 (below): discrete branch
  int B, N;
  const float* maskf;
  const int32_t* counts;
  const SaveSlot* slots;
  int n_slots;
  float drop_p;
  uint64_t seed, offset0;
  int nb_branch1;
  int tiles;               // bit 0 idle tiles of padded slots, bit 1 tile-to-SIMD steering: as the forward ran (TilePlan);
                           // bit 3: idle tiles leave their dZ rows unwritten (nobody reads them: bwd_tiles),
                           // bit 4: so do the padded lanes of working tiles
  const int32_t* jet_order;  // (B) launch order of the jets (mmb_jet_schedule) or NULL
};

__device__ __forceinline__ float gelu_grad(float z) {
  const float cdf = 0.5f * (1.0f + erf_f(z * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);   // exp(-z^2/2)
  return cdf + z * pdf;
}
__device__ __forceinline__ float lrelu_grad(float s) { return s > 0.0f ? 1.0f : 0.01f; }
__device__ __forceinline__ float drop_grad(float g, uint32_t r, float p) { return u01(r) < p ? 0.0f : g / (1.0f - p); }

template <int HT>
__device__ __forceinline__ void zero_tiles(f32x16 (&x)[HT]) {
  ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; x[t][r] = 0.0f; }); });
}

// acc += W^T-fragments * tiles
template <int KT, int OT, typename TileFn>
__device__ __forceinline__ void mma_go(f32x16 (&acc)[OT], TileFn&& tile, const float* P, int woff, int lane) {
  const WRsrc R = weight_rsrc(P);
#ifdef MMB_BWD_MMA_X              // the round-2 stream (two fragments ahead, issue order left to the compiler)
  AFrag<OT> pre;
  mma_prefetch(pre, R, woff, lane);
  mma_x<KT, OT>(acc, tile, R, woff, lane, pre);
#else                             // the forward's pinned software pipeline (fused_common.h: mma_r), one ring per GEMM
  constexpr int RD = (KT * 2 * OT) % 4 == 0 ? 4 : 2;
  WRing<RD> ring;
  ring_prime(ring, R, woff, lane * 16);
  mma_r<KT, OT, RD, false>(acc, tile, R, woff, 0, lane * 16, ring);
#endif
}

// s = column sums of the dZ tiles over the jet's particles -> LB.svec and row b of the next slot
template <int HT>
__device__ __forceinline__ void colsum(const f32x16 (&dz)[HT], LdsB& LB, Slots& sl, int tid, bool act, int W = 32 * HT) {
  const int lane = tid & 63, wave = tid >> 6;
  if (act) {
    pool_partials<HT>(dz, 1.0f, LB.partial, wave, lane);
  } else {                                                 // an idle tile: zero rows
    LB.partial[wave * 128 + lane] = 0.0f;
    LB.partial[wave * 128 + 64 + lane] = 0.0f;
  }
  __syncthreads();
  float* dst = sl.row(W);                                // W: true width of the layer (the slot's row length)
  if (tid < 32 * HT) {
    float s = (LB.partial[tid] + LB.partial[128 + tid]) + (LB.partial[256 + tid] + LB.partial[384 + tid]);
    if constexpr (FE_WAVES > 4) s += LB.partial[512 + tid];
    LB.svec[tid] = s;
    if (dst != nullptr && tid < W) dst[tid] = s;
  }
  __syncthreads();
}

// x[t][r] += m * c[feature]
template <int HT>
__device__ __forceinline__ void add_mc(f32x16 (&x)[HT], float m, const float* c, int half) {
  ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 cv = *reinterpret_cast<const float4*>(c + 32 * t + 8 * q + 4 * half);
      x[t][4 * q + 0] += m * cv.x; x[t][4 * q + 1] += m * cv.y; x[t][4 * q + 2] += m * cv.z; x[t][4 * q + 3] += m * cv.w; }); });
}

template <int HT, int ET, int G>
struct EpicB {
  static constexpr int H = 32 * HT;

  // gradient of the masked mean/sum pooling (epic.py:100-107) added to dx; LB.dpool holds d pooled (2H)
  static __device__ __forceinline__ void pool_bwd(f32x16 (&dx)[HT], LdsB& LB, float m, int tid, int half, bool act) {
    if (tid < H) LB.cpool[tid] = LB.dpool[tid] / LB.cnt + 0.01f * LB.dpool[H + tid];
    __syncthreads();
    if (act) add_mc<HT>(dx, m, LB.cpool, half);
  }

  // output layer  out = W [m temb, x, g] + b.  Slots: S_out.  In: dout (its dZ, already in HBM).
  // Out: dx = Wx^T dout, LB.dg = Wg^T s (+ add), LB.dtemb += Wt^T s.
  template <int OTO>
  static __device__ __forceinline__ void output_bwd(f32x16 (&dx)[HT], const f32x16 (&dout)[OTO], const BDesc o,
                                                    const float* __restrict__ P, const float* add, LdsB& LB, Slots& sl,
                                                    int tid, bool act, int ow = 32 * OTO) {
    const int lane = tid & 63;
    colsum<OTO>(dout, LB, sl, tid, act, ow);
    auto sv = [&](int k) { return LB.svec[k]; };
    WaveMV<32 * OTO, ET>::template run<MMB_ACT_NONE>(P + o.b, nullptr, sv, LB.dtemb, LB.dtemb, tid);
    WaveMV<32 * OTO, G>::template run<MMB_ACT_NONE>(P + o.c, nullptr, sv, LB.dg, add, tid);
    zero_tiles<HT>(dx);
    if (act) mma_go<OTO, HT>(dx, [&](auto fc) -> const f32x16& { return dout[decltype(fc)::v]; }, P, o.a, lane);
  }

  // EPiCLayer + encoder-level skip.  In: dx = d l_out, LB.dg = d g_out.  Out: dx = d x_local, LB.dg = d x_global;
  // the skip gradients accumulate in S (tiles) and LB.dgskip.
  // Slots: sign bits of y1, dz_l2, sign bits of lh's pre-activation, dz_l1, S_l1, g_new, dz_g2, gh, dz_g1.
  static __device__ __forceinline__ void layer_bwd(f32x16 (&dx)[HT], const BDesc* __restrict__ D, const float* __restrict__ P,
                                                   bool use_skip, LdsB& LB, float4* S, Slots& sl, float m, int tid,
                                                   float drop_p, uint64_t seed, uint64_t off_g, uint64_t off_l, bool act) {
    const int lane = tid & 63, half = lane >> 5;
    const BDesc g1 = D[0], g2 = D[1], l1 = D[2], l2 = D[3];
    const bool dr = drop_p > 0.0f;
    // An idle tile (padded slots only: every gradient row is exactly zero) keeps the slot walk, the barriers and its
    // share of the per-jet matvecs, stores its zero dZ rows for the weight-gradient kernels and skips the rest.  When
    // the caller's weight-gradient launches reduce over the rows of real particles only (bit 3 of A.tiles, set from
    // bit 1 of the entry points' use_skip), nobody reads those rows: sl.dead puts the tile's row accesses outside
    // every buffer, the stores are dropped by the bounds check (reverse kernel 0.90 -> 0.77 ms at B = 1024).
    // ---- l_out = drop(lrelu(y1)) + skip ;  y1 = x + W2 lh + b2
    const uint64_t sgn_y1 = sl.bits(tid);
    const Rows r_dz2 = sl.rows();
    const uint64_t sgn_lh = sl.bits(tid);                  // (both sign words before any store of this layer)
    const Rows r_dz1 = sl.rows();
    uint64_t keep = 0;
    if (dr && act) keep = dropout_keep_bits<HT>(drop_p, seed, off_l, (uint64_t)sl.b * sl.N + sl.p, half);
    if (act) ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      if (use_skip) {
        f32x16 sk = fetch_tile(S, t, tid);
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; sk[r] += dx[t][r]; });
        park_tile(S, t, sk, tid);
      }
      const uint32_t kw = (uint32_t)(keep >> (32 * (t >> 1))), sw = (uint32_t)(sgn_y1 >> (32 * (t >> 1)));
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
        constexpr int bit = 16 * (t & 1) + r;
        float g = dx[t][r];
        if (dr) g = ((kw >> bit) & 1u) ? g / (1.0f - drop_p) : 0.0f;
        dx[t][r] = ((sw >> bit) & 1u) ? g : 0.01f * g; }); });
    // ---- d lh = W2^T d y1 ; lh = lrelu(z1).  The dZ tiles are stored AFTER the GEMM that consumes them: vmcnt counts
    // loads and stores in one order, so a store issued before a GEMM holds up every weight fragment of that GEMM
    f32x16 dlh[HT];
    zero_tiles<HT>(dlh);
    if (act) mma_go<HT, HT>(dlh, [&](auto fc) -> const f32x16& { return dx[decltype(fc)::v]; }, P, l2.a, lane);
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v; store_tile(r_dz2, t, dx[t]); });
    if (act) ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      const uint32_t sw = (uint32_t)(sgn_lh >> (32 * (t >> 1)));
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
        if (!((sw >> (16 * (t & 1) + r)) & 1u)) dlh[t][r] *= 0.01f; }); });
    colsum<HT>(dlh, LB, sl, tid, act);
    // d x = d y1 (residual) + Wx1^T d z1
    if (act) mma_go<HT, HT>(dx, [&](auto fc) -> const f32x16& { return dlh[decltype(fc)::v]; }, P, l1.a, lane);
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v; store_tile(r_dz1, t, dlh[t]); });
    auto sv = [&](int k) { return LB.svec[k]; };
    WaveMV<H, ET>::template run<MMB_ACT_NONE>(P + l1.b, nullptr, sv, LB.dtemb, LB.dtemb, tid);
    WaveMV<H, G>::template run<MMB_ACT_NONE>(P + l1.c, nullptr, sv, LB.dgnew, nullptr, tid);
    // ---- g_out = drop(lrelu(g')) + gskip ;  g' = g + W gh + b
    const float* gnew = sl.row(G);
    float* dzg2 = sl.row(G);
    if (tid < G) {
      float g = LB.dg[tid];
      if (use_skip) LB.dgskip[tid] += g;
      if (dr) {
        const uint64_t idx = (uint64_t)sl.b * G + tid;
        uint32_t rr[4];
        philox4(seed, off_g, idx >> 2, 0xD0D0u, rr);
        g = drop_grad(g, rr[idx & 3], drop_p);
      }
      g = g * lrelu_grad(gnew ? gnew[tid] : 0.0f) + LB.dgnew[tid];
      LB.dgnew[tid] = g;
      LB.dg[tid] = g;                                    // residual: d g_in starts as d g'
      if (dzg2) dzg2[tid] = g;
    }
    __syncthreads();
    matvec<MMB_ACT_NONE, G, H>(P + g2.a, nullptr, LB.dgnew, LB.dgh, nullptr, tid);
    const float* gh = sl.row(H);
    float* dzg1 = sl.row(H);
    if (tid < H) {
      const float v = LB.dgh[tid] * lrelu_grad(gh ? gh[tid] : 0.0f);
      LB.dzg[tid] = v;
      if (dzg1) dzg1[tid] = v;
    }
    __syncthreads();
    matvec<MMB_ACT_NONE, H, 2 * H>(P + g1.a, nullptr, LB.dzg, LB.dpool, nullptr, tid);
    matvec<MMB_ACT_NONE, H, G>(P + g1.b, nullptr, LB.dzg, LB.dg, LB.dg, tid);
    pool_bwd(dx, LB, m, tid, half, act);
    __syncthreads();
  }

  // EPiCProjection.  In: dx = d x (without the skip share), LB.dg = d g.  Slots: pre_q2, dz_q2, pre_q0, dz_q0,
  // pre_p2, dz_p2, pre_p0, dz_p0, S_p0, [d_in (EIN wide) when EIN > 0].
  template <int EIN>
  static __device__ __forceinline__ void project_bwd(f32x16 (&dx)[HT], const BDesc* __restrict__ D, const float* __restrict__ P,
                                                     bool use_skip, LdsB& LB, float4* S, Slots& sl, float m, int tid,
                                                     bool act) {
    const int lane = tid & 63, half = lane >> 5;
    const BDesc p0 = D[0], p2 = D[1], q0 = D[2], q2 = D[3];
    if (use_skip && act) {                               // the projection's outputs ARE the skip tensors
      ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
        const f32x16 sk = fetch_tile(S, t, tid);
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; dx[t][r] += sk[r]; }); });
    }
    const float* pre_q2 = sl.row(G);
    float* dz_q2 = sl.row(G);
    if (tid < G) {
      const float g = (LB.dg[tid] + (use_skip ? LB.dgskip[tid] : 0.0f)) * gelu_grad(pre_q2 ? pre_q2[tid] : 0.0f);
      LB.dgnew[tid] = g;
      if (dz_q2) dz_q2[tid] = g;
    }
    __syncthreads();
    matvec<MMB_ACT_NONE, G, H>(P + q2.a, nullptr, LB.dgnew, LB.dgh, nullptr, tid);
    const float* pre_q0 = sl.row(H);
    float* dz_q0 = sl.row(H);
    if (tid < H) {
      const float v = LB.dgh[tid] * gelu_grad(pre_q0 ? pre_q0[tid] : 0.0f);
      LB.dzg[tid] = v;
      if (dz_q0) dz_q0[tid] = v;
    }
    __syncthreads();
    matvec<MMB_ACT_NONE, H, 2 * H>(P + q0.a, nullptr, LB.dzg, LB.dpool, nullptr, tid);
    matvec<MMB_ACT_NONE, H, ET>(P + q0.b, nullptr, LB.dzg, LB.dtemb, LB.dtemb, tid);    // context = time embedding
    pool_bwd(dx, LB, m, tid, half, act);
    // ---- local MLP: x = gelu(z2), z2 = W2 h + b ; h = gelu(z0), z0 = W0 [m temb, x_in] + b
    const Rows r_pre2 = sl.rows(), r_dz2 = sl.rows();
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      if (act) {                                         // (an idle tile's saved rows were never written: 0 * stale)
        const f32x16 a = load_tile(r_pre2, t);
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; dx[t][r] *= gelu_grad(a[r]); });
      }
      store_tile(r_dz2, t, dx[t]); });
    f32x16 dh[HT];
    zero_tiles<HT>(dh);
    if (act) mma_go<HT, HT>(dh, [&](auto fc) -> const f32x16& { return dx[decltype(fc)::v]; }, P, p2.a, lane);
    const Rows r_pre0 = sl.rows(), r_dz0 = sl.rows();
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      if (act) {
        const f32x16 a = load_tile(r_pre0, t);
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; dh[t][r] *= gelu_grad(a[r]); });
      }
      store_tile(r_dz0, t, dh[t]); });
    colsum<HT>(dh, LB, sl, tid, act);
    auto sv = [&](int k) { return LB.svec[k]; };
    WaveMV<H, ET>::template run<MMB_ACT_NONE>(P + p0.b, nullptr, sv, LB.dtemb, LB.dtemb, tid);
    if constexpr (EIN > 0) {
      f32x16 din[1];
      zero_tiles<1>(din);
      if (act) mma_go<HT, 1>(din, [&](auto fc) -> const f32x16& { return dh[decltype(fc)::v]; }, P, p0.a, lane);
      const int dd = sl.dead;                            // d_in goes to the embedding's reverse pass, which reads every
      sl.dead = 0;                                       // row: an idle tile keeps writing its zero rows here
      const Rows r = sl.rows();                          // (rows, EIN): only the column groups below EIN exist
      sl.dead = dd;
      ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
        if (8 * q + 4 * half < EIN) {
          const float4 v = make_float4(din[0][4 * q], din[0][4 * q + 1], din[0][4 * q + 2], din[0][4 * q + 3]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r.R, r.vo, 8 * q * 4, 0);
        } });
    }
  }
};

// Slot table: [0] d f (rows, 2HB: dZ of the fused output layer), [1] d h from the heads (rows, 2HB), [2] d cat[h1,h2]
// (rows, 2HB: dZ of the branch output layers), [3] d temb (B, ET); then in the order the kernel walks:
// heads (continuous, discrete): d_head, pre, dz, S;  fused encoder: S_out, layers last..first;
// branches (continuous, discrete): S_out, layers last..first, projection (+ d_in for the continuous one).
template <class DM, int NWT = FE_WAVES>
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_train_backward_kernel(BwdArgs A) {
  __shared__ LdsB LB;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  __shared__ int red_s[8];
  __shared__ short order_s[32 * FE_WAVES];
  constexpr int HBT = DM::HBT, HFT = DM::HFT, HB = DM::HB, GB = DM::GB, GF = DM::GF, ET = DM::ET;
  // tile t of a jet holds the particles it held in the forward (jet_order: a function of the mask; the sign words are
  // indexed by tile and lane); the hardware wave that computes it is chosen here (steer_wave)
  const int lane = threadIdx.x & 63, wave = steer_wave((A.tiles & 2) != 0, threadIdx.x >> 6, lane, red_s);
  const int tid = 64 * wave + lane, half = lane >> 5;
  const int nb0 = A.nb_branch, nb1 = A.nb_branch1, nbf = A.nb_fused_exec;
  const bool use_skip = A.use_skip != 0;
  const int base_f = 2 * 5 + 4 * (nb0 + nb1), base_h = base_f + 4 * nbf + 1;
  for (int bi = blockIdx.x; bi < A.B; bi += gridDim.x) {
    const int b = A.jet_order ? A.jet_order[bi] : bi;
    __syncthreads();
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    const int p = jet_order((A.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, slot, wave, lane, red_s, order_s);
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    const bool act = plan_tiles((A.tiles & 1) != 0, false, m, valid, wave, lane, red_s).active;
    if (tid == 0) LB.cnt = (float)A.counts[b];
    if (tid < 128) LB.dtemb[tid] = 0.0f;
    __syncthreads();
    Slots sl{A.slots, A.n_slots, 4, b, A.N, p, half};
    sl.dead = (A.tiles & 8) != 0 && (!act || ((A.tiles & 16) != 0 && m == 0.0f));
    const SaveSlot s_df = sl.at(0), s_dhb = sl.at(1), s_dboth = sl.at(2), s_dtemb = sl.at(3);
    // ---- heads: o = W2 h + b, h = gelu(W0 [m temb, f_i, h_i] + b)
    ffor<2>([&](auto hc) {
      constexpr int hi = decltype(hc)::v;
      const BDesc h0 = A.D[base_h + 2 * hi], h2 = A.D[base_h + 2 * hi + 1];
      const SaveSlot sd = sl.next();
      f32x16 dzo[1];
      zero_tiles<1>(dzo);
      if constexpr (hi == 0) {
        if (act && valid && half == 0 && (gp + 1) * sd.ld <= sd.n) {
#pragma unroll
          for (int d = 0; d < DM::DC; ++d) dzo[0][d] = sd.p[gp * sd.ld + d];
        }
      } else {
        static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
        if (act && valid && (gp + 1) * sd.ld <= sd.n) {
          const float4 v = *reinterpret_cast<const float4*>(sd.p + gp * sd.ld + 4 * half);
          dzo[0][0] = v.x; dzo[0][1] = v.y; dzo[0][2] = v.z; dzo[0][3] = v.w;
        }
      }
      f32x16 dh[HBT];
      zero_tiles<HBT>(dh);
      if (act) mma_go<1, HBT>(dh, [&](auto) -> const f32x16& { return dzo[0]; }, A.P, h2.a, lane);
      const Rows r_pre = sl.rows(), r_dz = sl.rows();
      ffor<HBT>([&](auto tc) { constexpr int t = decltype(tc)::v;
        if (act) {                                       // (an idle tile's saved rows were never written: 0 * stale)
          const f32x16 a = load_tile(r_pre, t);
          ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; dh[t][r] *= gelu_grad(a[r]); });
        }
        store_tile(r_dz, t, dh[t]); });
      colsum<HBT>(dh, LB, sl, tid, act);
      WaveMV<HB, ET>::template run<MMB_ACT_NONE>(A.P + h0.b, nullptr, [&](int k) { return LB.svec[k]; }, LB.dtemb,
                                                 LB.dtemb, tid);
      f32x16 din[HFT];
      zero_tiles<HFT>(din);
      if (act) mma_go<HBT, HFT>(din, [&](auto fc) -> const f32x16& { return dh[decltype(fc)::v]; }, A.P, h0.a, lane);
      const Rows r_df = sl.rows_of(s_df, hi * HB), r_dhb = sl.rows_of(s_dhb, hi * HB);
      ffor<HBT>([&](auto tc) { constexpr int t = decltype(tc)::v;
        store_tile(r_df, t, din[t]);
        store_tile(r_dhb, t, din[HBT + t]); });
    });
    // ---- fused encoder (no projection; its input and skip tensor is cat[h1, h2])
    {
      f32x16 dx[HFT];
      {
        f32x16 dout[HFT];
        const Rows r = sl.rows_of(s_df);
        ffor<HFT>([&](auto tc) { constexpr int t = decltype(tc)::v; dout[t] = load_tile(r, t); });
        EpicB<HFT, ET, GF>::template output_bwd<HFT>(dx, dout, A.D[base_f + 4 * nbf], A.P, nullptr, LB, sl, tid, act);
      }
      if (use_skip) {
        f32x16 z;
        ffor<16>([&](auto rc) { z[decltype(rc)::v] = 0.0f; });
        ffor<HFT>([&](auto tc) { park_tile(S, decltype(tc)::v, z, tid); });
      }
      if (tid < GF) LB.dgskip[tid] = 0.0f;
      __syncthreads();
      for (int i = nbf - 1; i >= 0; --i) {
        const uint64_t j = (uint64_t)(nb0 + nb1 + i);
        EpicB<HFT, ET, GF>::layer_bwd(dx, A.D + base_f + 4 * i, A.P, use_skip, LB, S, sl, m, tid, A.drop_p, A.seed,
                                      A.offset0 + 2 * j + 1, A.offset0 + 2 * j + 2, act);
      }
      const Rows r_dhb = sl.rows_of(s_dhb), r_db = sl.rows_of(s_dboth);
      ffor<HFT>([&](auto tc) { constexpr int t = decltype(tc)::v;
        const f32x16 a = load_tile(r_dhb, t);
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; dx[t][r] += a[r]; });
        if (use_skip) {
          const f32x16 sk = fetch_tile(S, t, tid);
          ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; dx[t][r] += sk[r]; });
        }
        store_tile(r_db, t, dx[t]); });
      if (tid < GF) LB.dgcat[tid] = LB.dg[tid] + (use_skip ? LB.dgskip[tid] : 0.0f);
      __syncthreads();
    }
    // ---- the two branch encoders
    ffor<2>([&](auto bc) {
      constexpr int br = decltype(bc)::v;
      const int nb = br == 0 ? nb0 : nb1;
      const int base_b = br * (5 + 4 * nb0);
      f32x16 dx[HBT];
      {
        f32x16 dout[HBT];
        const Rows r = sl.rows_of(s_dboth, br * HB);
        ffor<HBT>([&](auto tc) { constexpr int t = decltype(tc)::v; dout[t] = load_tile(r, t); });
        EpicB<HBT, ET, GB>::template output_bwd<HBT>(dx, dout, A.D[base_b + 4 + 4 * nb], A.P, LB.dgcat + br * GB, LB, sl,
                                                     tid, act);
      }
      if (use_skip) {
        f32x16 z;
        ffor<16>([&](auto rc) { z[decltype(rc)::v] = 0.0f; });
        ffor<HBT>([&](auto tc) { park_tile(S, decltype(tc)::v, z, tid); });
      }
      if (tid < GB) LB.dgskip[tid] = 0.0f;
      __syncthreads();
      for (int i = nb - 1; i >= 0; --i) {
        const uint64_t j = (uint64_t)(br * nb0 + i);
        EpicB<HBT, ET, GB>::layer_bwd(dx, A.D + base_b + 4 + 4 * i, A.P, use_skip, LB, S, sl, m, tid, A.drop_p, A.seed,
                                      A.offset0 + 2 * j + 1, A.offset0 + 2 * j + 2, act);
      }
      EpicB<HBT, ET, GB>::template project_bwd<(br == 0 ? DM::EC : 0)>(dx, A.D + base_b, A.P, use_skip, LB, S, sl, m, tid, act);
      __syncthreads();
    });
    float* dt = sl.row_of(s_dtemb, ET);
    if (tid < ET && dt != nullptr) dt[tid] = LB.dtemb[tid];
  }
}

// ---- discrete-only model (UniModalEPiC): one encoder with projection whose output layer emits the logits.
// Slot table: [0] d temb (B,ET); then d logits (rows,8), column sums of the output layer (B,8), layers
// last..first, projection.
template <class DM, int NWT = FE_WAVES>
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_uni_train_backward_kernel(BwdArgs A) {
  static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
  __shared__ LdsB LB;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ int red_s[8];
  __shared__ short order_s[32 * FE_WAVES];
  constexpr int HT = DM::HT, ET = DM::ET, G = DM::G;
  const int lane = threadIdx.x & 63, wave = steer_wave((A.tiles & 2) != 0, threadIdx.x >> 6, lane, red_s);
  const int tid = 64 * wave + lane, half = lane >> 5;
  const int nb = A.nb_branch;
  const bool use_skip = A.use_skip != 0;
  for (int bi = blockIdx.x; bi < A.B; bi += gridDim.x) {
    const int b = A.jet_order ? A.jet_order[bi] : bi;
    __syncthreads();
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    const int p = jet_order((A.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, slot, wave, lane, red_s, order_s);
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    const bool act = plan_tiles((A.tiles & 1) != 0, false, m, valid, wave, lane, red_s).active;
    if (tid == 0) LB.cnt = (float)A.counts[b];
    if (tid < 128) LB.dtemb[tid] = 0.0f;
    __syncthreads();
    Slots sl{A.slots, A.n_slots, 1, b, A.N, p, half};
    sl.dead = (A.tiles & 8) != 0 && (!act || ((A.tiles & 16) != 0 && m == 0.0f));
    const SaveSlot s_dtemb = sl.at(0);
    const SaveSlot sd = sl.next();
    f32x16 dx[HT];
    {
      f32x16 dout[1];
      zero_tiles<1>(dout);
      if (act && valid && (gp + 1) * sd.ld <= sd.n) {
        const float4 v = *reinterpret_cast<const float4*>(sd.p + gp * sd.ld + 4 * half);
        dout[0][0] = v.x; dout[0][1] = v.y; dout[0][2] = v.z; dout[0][3] = v.w;
      }
      EpicB<HT, ET, G>::template output_bwd<1>(dx, dout, A.D[4 + 4 * nb], A.P, nullptr, LB, sl, tid, act, DM::V);
    }
    if (use_skip) {
      f32x16 z;
      ffor<16>([&](auto rc) { z[decltype(rc)::v] = 0.0f; });
      ffor<HT>([&](auto tc) { park_tile(S, decltype(tc)::v, z, tid); });
    }
    if (tid < G) LB.dgskip[tid] = 0.0f;
    __syncthreads();
    for (int i = nb - 1; i >= 0; --i)
      EpicB<HT, ET, G>::layer_bwd(dx, A.D + 4 + 4 * i, A.P, use_skip, LB, S, sl, m, tid, A.drop_p, A.seed,
                                  A.offset0 + 2 * (uint64_t)i + 1, A.offset0 + 2 * (uint64_t)i + 2, act);
    EpicB<HT, ET, G>::template project_bwd<0>(dx, A.D, A.P, use_skip, LB, S, sl, m, tid, act);
    __syncthreads();
    float* dt = sl.row_of(s_dtemb, ET);
    if (tid < ET && dt != nullptr) dt[tid] = LB.dtemb[tid];
  }
}

// MMB_TRAIN_TILES=<bits> (developer switch; default 3; csrc/fused_epic.hip reads the same variable for the forward)
static int train_tiles() {
  const char* e = getenv("MMB_TRAIN_TILES");
  return e ? atoi(e) : 3;
}
// use_skip of the two entry points: bit 0 the encoder-level skip connection, bit 1 "the dZ rows of padded slots are
// never read" (the caller reduces its weight gradients over mmb_mask_rows lists) -> bit 3 of the kernels' tile plan
static int bwd_tiles(int& use_skip) {
  int tiles = train_tiles();
  if ((use_skip & 2) && (tiles & 1)) {
    tiles |= 8;
    const char* e = getenv("MMB_BWD_DEAD_LANES");          // padded lanes of working tiles too (developer switch: 0 = off)
    if (!e || atoi(e) != 0) tiles |= 16;
  }
  use_skip &= 1;
  return tiles;
}

}  // namespace mmb

using namespace mmb;

#if MMB_FE_WAVES == 4
extern "C" int mmb_fused_uni_train_backward_w5(int, int, int, const float*, const int32_t*, int, int, const float*,
                                               const int32_t*, const void*, int, float, uint64_t, uint64_t, const int32_t*,
                                               void*);
extern "C" int mmb_fused_epic_train_backward_w5(int, int, int, const float*, const int32_t*, int, int, int, int, const float*,
                                                const int32_t*, const void*, int, float, uint64_t, uint64_t, const int32_t*,
                                                void*);
#endif

extern "C" int mmb_fused_uni_train_backward(int variant, int B, int N, const float* packed_bwd, const int32_t* desc_bwd,
                                            int num_blocks, int use_skip, const float* maskf, const int32_t* counts,
                                            const void* slots, int n_slots, float drop_p, uint64_t seed,
                                            uint64_t offset0, const int32_t* jet_order, void* stream) {
#if MMB_FE_WAVES == 4
  if (N > 128) return mmb_fused_uni_train_backward_w5(variant, B, N, packed_bwd, desc_bwd, num_blocks, use_skip, maskf, counts,
                                                      slots, n_slots, drop_p, seed, offset0, jet_order, stream);
#endif
  MMB_REQUIRE(B > 0 && N > 0 && N <= 32 * FE_WAVES, "mmb_fused_uni_train_backward: needs 0 < N <= %d (got B=%d N=%d)", 32 * FE_WAVES, B, N);
  MMB_REQUIRE(packed_bwd && desc_bwd && maskf && counts && slots, "mmb_fused_uni_train_backward: NULL argument");
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_train_backward: unknown variant %d", variant);
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_fused_uni_train_backward: dropout p=%f outside [0,1)", drop_p);
  const int expect = 1 + 2 + num_blocks * 9 + 9;
  MMB_REQUIRE(n_slots == expect, "mmb_fused_uni_train_backward: %d slots given, the architecture has %d", n_slots,
              expect);
  const int tiles = bwd_tiles(use_skip);
  BwdArgs A{packed_bwd, reinterpret_cast<const BDesc*>(desc_bwd), num_blocks, 0, use_skip, B, N, maskf,
            counts, reinterpret_cast<const SaveSlot*>(slots), n_slots, drop_p, seed, offset0, 0, tiles, jet_order};
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_uni_train_backward_kernel<UniS2>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
    case 1: hipLaunchKernelGGL(fused_uni_train_backward_kernel<UniMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
    case 2: hipLaunchKernelGGL(fused_uni_train_backward_kernel<UniH64>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
    default: hipLaunchKernelGGL(fused_uni_train_backward_kernel<UniH32>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
  }
  return check_launch("mmb_fused_uni_train_backward");
}

extern "C" int mmb_fused_epic_train_backward(int variant, int B, int N, const float* packed_bwd, const int32_t* desc_bwd,
                                             int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip,
                                             const float* maskf,
                                             const int32_t* counts, const void* slots, int n_slots, float drop_p,
                                             uint64_t seed, uint64_t offset0, const int32_t* jet_order, void* stream) {
#if MMB_FE_WAVES == 4
  if (N > 128) return mmb_fused_epic_train_backward_w5(variant, B, N, packed_bwd, desc_bwd, nb_branch, nb_branch1, nb_fused_exec,
                                                       use_skip, maskf, counts, slots, n_slots, drop_p, seed, offset0, jet_order,
                                                       stream);
#endif
  MMB_REQUIRE(B > 0 && N > 0 && N <= 32 * FE_WAVES, "mmb_fused_epic_train_backward: needs 0 < N <= %d (got B=%d N=%d)", 32 * FE_WAVES, B, N);
  MMB_REQUIRE(packed_bwd && desc_bwd && maskf && counts && slots, "mmb_fused_epic_train_backward: NULL argument");
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_train_backward: unknown variant %d", variant);
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_fused_epic_train_backward: dropout p=%f outside [0,1)", drop_p);
  const int per_layer = 9;
  const int expect = 4 + 8 + (1 + nb_fused_exec * per_layer) + 2 * (1 + 9) + (nb_branch + nb_branch1) * per_layer + 1;
  MMB_REQUIRE(n_slots == expect, "mmb_fused_epic_train_backward: %d slots given, the architecture has %d", n_slots,
              expect);
  const int tiles = bwd_tiles(use_skip);
  BwdArgs A{packed_bwd, reinterpret_cast<const BDesc*>(desc_bwd), nb_branch, nb_fused_exec, use_skip, B, N, maskf,
            counts, reinterpret_cast<const SaveSlot*>(slots), n_slots, drop_p, seed, offset0, nb_branch1, tiles, jet_order};
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_train_backward_kernel<DimsS3>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
    case 1: hipLaunchKernelGGL(fused_train_backward_kernel<DimsMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
    case 2: hipLaunchKernelGGL(fused_train_backward_kernel<DimsWide>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
    default: hipLaunchKernelGGL(fused_train_backward_kernel<DimsNarrow>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A); break;
  }
  return check_launch("mmb_fused_epic_train_backward");
}
