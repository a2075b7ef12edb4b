// The per-particle tail of one nn.TransformerEncoderLayer in ONE launch (reference: encoders/particle_transformer.py:29-50,
// post-LN, ReLU FFN; inference / dropout-free arithmetic):
//
//     h1  = LayerNorm1(h + out_proj(a))                       a = the attention output (mmb_attention_fwd)
//     h'  = LayerNorm2(h1 + linear2(relu(linear1(h1))))
//     qkv = in_proj_next(h')                                  (optional: the NEXT layer's in_proj)
//
// Layered, these are mmb_linear_cat_ln, mmb_linear_cat, mmb_linear_cat_ln, mmb_linear_cat: h1, the 4d-wide FFN hidden
// tensor and h' each make a round trip through HBM, and the generation loop of the S4 model is bound by exactly that
// traffic (22 d floats per particle and layer; this kernel: 2 d in, d + 3 d out).  Everything here is per particle, so a
// wave simply owns 32 rows: activations stay TRANSPOSED in the MFMA C/D layout of the fused EPiC kernels (lane & 31 = row,
// (register, lane half) = feature), every GEMM is the exact-fp32 six-product bf16 stream of fused_common.h (mma_r: one
// weight-fragment ring for the whole kernel, the next GEMM's first fragments requested from the tail of the current
// one), the FFN runs in chunks of 64 hidden units that never leave the registers, LayerNorm is a sum over the lane's
// registers plus one exchange with lane ^ 32.  No barriers after the prologue, no LDS traffic but the bias / gamma /
// beta vectors.
#include "common.h"
#include "fused_common.h"

namespace mmb {

struct TailArgs {
  const float* P;              // [vectors | weight fragments]  (mmb_hip/fused.py: pack_transformer_tail)
  int n_vec;                   // floats of the vector part (multiple of 4)
  int wx_o, wx_1, wx_2, wx_in; // float offsets of the fragment blocks (W1 / W2: one block per 64-unit chunk, consecutive)
  int64_t M;
  const float* a; int64_t a_ld;
  const float* h; int64_t h_ld;
  float* out; int64_t out_ld;
  float* qkv; int64_t qkv_ld;  // NULL: no in_proj
  float eps1, eps2;
  int n_chunks;                // ff / 64
  const int32_t* a_idx;        // NULL, or per row the row of `a` to read (< 0: a zero row) -- the attention output of the
                               // compacted inference path is one V row per jet, read in place from the qkv buffer
};

// vector part, float offsets (d = model width, ff = FFN width): b_o, gamma1, beta1, b2, gamma2, beta2 (d each), b1 (ff),
// b_in (3 d)
template <int DT>
struct TailVec {
  static constexpr int D = 32 * DT;
  static constexpr int BO = 0, G1 = D, BE1 = 2 * D, B2 = 3 * D, G2 = 4 * D, BE2 = 5 * D, B1 = 6 * D;
};

template <int DT>
__device__ __forceinline__ void load_rows(f32x16 (&x)[DT], const float* __restrict__ src, int64_t ld, int64_t row, int half) {
  const float* p = src + row * ld + 4 * half;
  ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 v = *reinterpret_cast<const float4*>(p + 32 * t + 8 * q);
      x[t][4 * q + 0] = v.x; x[t][4 * q + 1] = v.y; x[t][4 * q + 2] = v.z; x[t][4 * q + 3] = v.w; }); });
}

// Rows to global memory in WHOLE 128-byte lines.  In the C/D layout a lane pair holds 32 contiguous bytes of a row, so a
// direct b128 store writes 32 rows x 32 bytes per instruction -- four partial writes per cache line, and a kernel made
// of such stores runs at 1.0-1.5 TB/s.  The tile goes through a wave-private LDS square instead (pitch 36 floats: the
// b128 accesses of both passes are conflict-free) and comes back with 8 consecutive lanes on one row: every store
// instruction writes 8 rows x 128 contiguous bytes.  No barrier: the square belongs to one wave and the LDS pipe is in
// order per wave.
constexpr int TP = 36;
template <int DT>
__device__ __forceinline__ void store_rows(float* __restrict__ dst, int64_t ld, int64_t row0, int64_t M, int lane, float* sq,
                                           const f32x16 (&x)[DT]) {
  const int p = lane & 31, half = lane >> 5;
  const int rr = lane >> 3, c4 = (lane & 7) * 4;
  ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      *reinterpret_cast<float4*>(sq + p * TP + 8 * q + 4 * half) =
          make_float4(x[t][4 * q + 0], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]); });
    __builtin_amdgcn_wave_barrier();
    ffor<4>([&](auto ic) { constexpr int i = decltype(ic)::v;
      const int r = 8 * i + rr;
      const float4 v = *reinterpret_cast<const float4*>(sq + r * TP + c4);
      if (row0 + r < M) *reinterpret_cast<float4*>(dst + (row0 + r) * ld + 32 * t + c4) = v; });
    __builtin_amdgcn_wave_barrier(); });
}

// x[t][4q+e] += v[32t + 8q + 4half + e]   (v in LDS)
template <int NT>
__device__ __forceinline__ void add_vec(f32x16 (&x)[NT], const float* v, int half) {
  ffor<NT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 b = *reinterpret_cast<const float4*>(v + 32 * t + 8 * q + 4 * half);
      x[t][4 * q + 0] += b.x; x[t][4 * q + 1] += b.y; x[t][4 * q + 2] += b.z; x[t][4 * q + 3] += b.w; }); });
}
template <int NT>
__device__ __forceinline__ void set_vec(f32x16 (&x)[NT], const float* v, int half) {
  ffor<NT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 b = *reinterpret_cast<const float4*>(v + 32 * t + 8 * q + 4 * half);
      x[t][4 * q + 0] = b.x; x[t][4 * q + 1] = b.y; x[t][4 * q + 2] = b.z; x[t][4 * q + 3] = b.w; }); });
}

// nn.LayerNorm over the 32*DT features of this lane's row (biased variance, eps inside the root): the lane holds half of
// them, lane ^ 32 the other half
template <int DT>
__device__ __forceinline__ void layer_norm(f32x16 (&x)[DT], const float* gamma, const float* beta, float eps, int half) {
  constexpr float inv_d = 1.0f / (32.0f * DT);
  float s = 0.0f;
  ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; s += x[t][r]; }); });
  s += __shfl_xor(s, 32, 64);
  const float mean = s * inv_d;
  float v = 0.0f;
  ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; const float dlt = x[t][r] - mean; v = fmaf(dlt, dlt, v); }); });
  v += __shfl_xor(v, 32, 64);
  const float rstd = 1.0f / sqrtf(v * inv_d + eps);
  ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 g = *reinterpret_cast<const float4*>(gamma + 32 * t + 8 * q + 4 * half);
      const float4 b = *reinterpret_cast<const float4*>(beta + 32 * t + 8 * q + 4 * half);
      x[t][4 * q + 0] = fmaf((x[t][4 * q + 0] - mean) * rstd, g.x, b.x);
      x[t][4 * q + 1] = fmaf((x[t][4 * q + 1] - mean) * rstd, g.y, b.y);
      x[t][4 * q + 2] = fmaf((x[t][4 * q + 2] - mean) * rstd, g.z, b.z);
      x[t][4 * q + 3] = fmaf((x[t][4 * q + 3] - mean) * rstd, g.w, b.w); }); });
}

constexpr int TAIL_MAX_VEC = 6 * 128 + 1024 + 3 * 128;      // d <= 128, ff <= 1024

template <int DT>
#ifndef MMB_TAIL_WG128
#define MMB_TAIL_WG128 2       // workgroups per CU of the d = 128 variant: 2 = 256 registers + 92 B/lane of spills, 1 = 385
#endif                         // registers, no spills; measured 299 / 91 us (2) against 312 / 97 us (1) at 153.6 k / 50 k rows
__global__ void __launch_bounds__(256, DT == 4 ? MMB_TAIL_WG128 : 2)
transformer_tail_kernel(TailArgs A) {
  typedef TailVec<DT> V;
  constexpr int D = 32 * DT, RD = 4;
  constexpr int S1 = DT * 2 * 2 * 768, S2 = 2 * 2 * DT * 768, SI = DT * 2 * DT * 768;   // floats per W1 / W2 chunk, in_proj block
  __shared__ __attribute__((aligned(16))) float vec[TAIL_MAX_VEC];
  __shared__ __attribute__((aligned(16))) float squares[4][32 * TP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, lane16 = lane * 16;
  for (int i = tid; i < A.n_vec; i += 256) vec[i] = A.P[i];
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int64_t row_raw = row0 + (lane & 31);
  const bool live = row_raw < A.M;
  const int64_t row = live ? row_raw : A.M - 1;               // rows past the end read the last row and store nothing
  const WRsrc R = weight_rsrc(A.P + A.n_vec);
  WRing<RD> ring;
  ring_prime(ring, R, A.wx_o, lane16);
  f32x16 x[DT];
  {
    f32x16 a[DT];
    int64_t arow = row;
    if (A.a_idx != nullptr) arow = A.a_idx[row];
    load_rows<DT>(a, A.a, A.a_ld, arow < 0 ? 0 : arow, half);
    if (arow < 0) ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; a[t][r] = 0.0f; }); });
    load_rows<DT>(x, A.h, A.h_ld, row, half);
    __syncthreads();                                          // the vectors are in LDS
    add_vec<DT>(x, vec + V::BO, half);                        // x = h + b_o + W_o a
    mma_r<DT, DT, RD, true>(x, [&](auto fc) -> const f32x16& { return a[decltype(fc)::v]; }, R, A.wx_o, A.wx_1, lane16, ring);
  }
  layer_norm<DT>(x, vec + V::G1, vec + V::BE1, A.eps1, half);  // x = h1
  f32x16 y[DT];
  ffor<DT>([&](auto tc) { constexpr int t = decltype(tc)::v; y[t] = x[t]; });
  add_vec<DT>(y, vec + V::B2, half);                          // y = h1 + b2 + W2 relu(W1 h1 + b1)
  const int nc = A.n_chunks;
#pragma unroll 1
  for (int c = 0; c < nc; ++c) {
    const int w1 = A.wx_1 + c * S1, w2 = A.wx_2 + c * S2;
    const int nxt = c + 1 < nc ? w1 + S1 : (A.qkv ? A.wx_in : A.wx_o);   // (no in_proj: a harmless re-read of W_o's head)
    f32x16 f[2];
    set_vec<2>(f, vec + V::B1 + 64 * c, half);
    mma_r<DT, 2, RD, true>(f, [&](auto fc) -> const f32x16& { return x[decltype(fc)::v]; }, R, w1, w2, lane16, ring);
    ffor<2>([&](auto tc) { constexpr int t = decltype(tc)::v;
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; f[t][r] = fmaxf(f[t][r], 0.0f); }); });
    mma_r<2, DT, RD, true>(y, [&](auto fc) -> const f32x16& { return f[decltype(fc)::v]; }, R, w2, nxt, lane16, ring);
  }
  layer_norm<DT>(y, vec + V::G2, vec + V::BE2, A.eps2, half);  // y = h'
  store_rows<DT>(A.out, A.out_ld, row0, A.M, lane, squares[wave], y);
  if (A.qkv) {
    const float* bin = vec + V::B1 + 64 * nc;
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      f32x16 o[DT];
      set_vec<DT>(o, bin + j * D, half);
      const int w = A.wx_in + j * SI;
      mma_r<DT, DT, RD, true>(o, [&](auto fc) -> const f32x16& { return y[decltype(fc)::v]; }, R, w, j < 2 ? w + SI : A.wx_o,
                              lane16, ring);
      store_rows<DT>(A.qkv + j * D, A.qkv_ld, row0, A.M, lane, squares[wave], o);
    }
  }
}

}  // namespace mmb

using namespace mmb;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// d: model width (64 or 128), ff: FFN width (multiple of 64, <= 1024).  packed: n_vec floats [b_o | gamma1 | beta1 | b2 |
// gamma2 | beta2 | b1 | b_in] followed by the weight fragments; wx_*: float offsets of the fragment blocks RELATIVE TO THE
// FRAGMENT PART (after the vectors).  a_idx: NULL, or M row indices into `a` (< 0: a zero row).
extern "C" int mmb_transformer_tail(int64_t M, int d, int ff, const float* packed, int n_vec, int wx_o, int wx_1, int wx_2,
                                    int wx_in, const float* a, int64_t a_ld, const float* h, int64_t h_ld, float* out,
                                    int64_t out_ld, float* qkv, int64_t qkv_ld, float eps1, float eps2, const int32_t* a_idx,
                                    void* stream) {
  MMB_REQUIRE(M > 0, "mmb_transformer_tail: empty shape");
  MMB_REQUIRE(d == 64 || d == 128, "mmb_transformer_tail: model width %d (compiled: 64, 128)", d);
  MMB_REQUIRE(ff > 0 && ff % 64 == 0 && ff <= 1024, "mmb_transformer_tail: FFN width %d (a multiple of 64, <= 1024)", ff);
  MMB_REQUIRE(packed && a && h && out, "mmb_transformer_tail: NULL argument");
  MMB_REQUIRE(n_vec == 6 * d + ff + 3 * d && n_vec <= TAIL_MAX_VEC, "mmb_transformer_tail: vector part has %d floats, expected %d",
              n_vec, 9 * d + ff);
  MMB_REQUIRE(a_ld >= d && h_ld >= d && out_ld >= d && (!qkv || qkv_ld >= 3 * (int64_t)d), "mmb_transformer_tail: row stride too small");
  MMB_REQUIRE(a_ld % 4 == 0 && h_ld % 4 == 0 && out_ld % 4 == 0 && (!qkv || qkv_ld % 4 == 0) && al16(a) && al16(h) && al16(out) &&
              (!qkv || al16(qkv)) && al16(packed), "mmb_transformer_tail: operands must be 16-byte aligned with row strides in multiples of 4 floats");
  MMB_REQUIRE(wx_o >= 0 && wx_1 >= 0 && wx_2 >= 0 && (!qkv || wx_in >= 0) && wx_o % 4 == 0 && wx_1 % 4 == 0 && wx_2 % 4 == 0 &&
              (!qkv || wx_in % 4 == 0), "mmb_transformer_tail: bad fragment offsets");
  MMB_REQUIRE((M + 127) / 128 < (1ll << 31), "mmb_transformer_tail: too many rows");
  TailArgs A{packed, n_vec, wx_o, wx_1, wx_2, qkv ? wx_in : 0, M, a, a_ld, h, h_ld, out, out_ld, qkv, qkv_ld, eps1, eps2, ff / 64, a_idx};
  dim3 grid((unsigned)((M + 127) / 128)), block(256);
  hipStream_t st = as_stream(stream);
  if (d == 64) hipLaunchKernelGGL((transformer_tail_kernel<2>), grid, block, 0, st, A);
  else hipLaunchKernelGGL((transformer_tail_kernel<4>), grid, block, 0, st, A);
  return check_launch("mmb_transformer_tail");
}
