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

// out[q, h*dh + d] = sum_j softmax_j(scale * q.k_j) v_j over the valid keys j;  lse[(b*nh+h)*N + q] = log sum exp
template <int DHP>
__global__ void __launch_bounds__(256)
attn_fwd_kernel(int N, int H, int nh, int dh, const float* __restrict__ qkv, int64_t ld,
                const float* __restrict__ maskf, int valid_is_zero, float scale, float* __restrict__ out,
                int64_t out_ld, float* __restrict__ lse) {
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
  for (int q = threadIdx.x; q < N; q += blockDim.x) {
    float qv[DHP], acc[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) {
      qv[d] = d < dh ? base[(int64_t)q * ld + d] * scale : 0.0f;
      acc[d] = 0.0f;
    }
    float m = -INFINITY, l = 0.0f;
    for (int j = 0; j < nk; ++j) {
      const float* kr = Ks + j * DHP;
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) s = fmaf(qv[d], kr[d], s);
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn);       // exp(-inf) = 0 on the first key
      const float p = __expf(s - mn);
      l = fmaf(l, alpha, p);
      const float* vr = Vs + j * DHP;
#pragma unroll
      for (int d = 0; d < DHP; ++d) acc[d] = fmaf(acc[d], alpha, p * vr[d]);
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
template <int DHP>
__global__ void __launch_bounds__(256)
attn_bwd_kernel(int N, int H, int nh, int dh, const float* __restrict__ qkv, int64_t ld,
                const float* __restrict__ maskf, int valid_is_zero, float scale, const float* __restrict__ out,
                int64_t out_ld, const float* __restrict__ dout, int64_t dout_ld, const float* __restrict__ lse,
                float* __restrict__ dqkv, int64_t dld) {
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
    for (int j = 0; j < nk; ++j) {
      const float* kr = A0 + j * DHP;
      const float* vr = A1 + j * DHP;
      float s = 0.0f, dp = 0.0f;
#pragma unroll
      for (int d = 0; d < DHP; ++d) {
        s = fmaf(qv[d], kr[d], s);
        dp = fmaf(go[d], vr[d], dp);
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
        const float ds = p * (dp - Ds[q]);
#pragma unroll
        for (int d = 0; d < DHP; ++d) {
          dv[d] = fmaf(p, gr[d], dv[d]);
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

static size_t attn_lds_bytes(int N, int dhp, bool bwd) {
  size_t fl = ((N + 1 + 3) & ~3) + 2 * (size_t)N * dhp;
  if (bwd) fl += N + ((N + 3) & ~3);
  return fl * sizeof(float);
}

static int pad_dh(int dh) { return dh <= 4 ? 4 : dh <= 8 ? 8 : dh <= 16 ? 16 : dh <= 32 ? 32 : 64; }

}  // namespace mmb

using namespace mmb;

#define MMB_ATTN_DISPATCH(KERNEL, ...)                                                              \
  switch (dhp) {                                                                                    \
    case 4: hipLaunchKernelGGL((KERNEL<4>), grid, block, lds, st, __VA_ARGS__); break;              \
    case 8: hipLaunchKernelGGL((KERNEL<8>), grid, block, lds, st, __VA_ARGS__); break;              \
    case 16: hipLaunchKernelGGL((KERNEL<16>), grid, block, lds, st, __VA_ARGS__); break;            \
    case 32: hipLaunchKernelGGL((KERNEL<32>), grid, block, lds, st, __VA_ARGS__); break;            \
    default: hipLaunchKernelGGL((KERNEL<64>), grid, block, lds, st, __VA_ARGS__); break;            \
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

extern "C" int mmb_attention_fwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                                 const float* maskf, int valid_is_zero, float* out, int64_t out_ld, float* lse,
                                 void* stream) {
  MMB_REQUIRE(qkv && maskf && out, "mmb_attention_fwd: NULL argument");
  const int dh = nheads > 0 ? H / nheads : 0, dhp = pad_dh(dh);
  const size_t lds = attn_lds_bytes(N, dhp, false);
  if (int rc = attn_check("mmb_attention_fwd", B, N, H, nheads, qkv_ld, lds)) return rc;
  MMB_REQUIRE(out_ld >= H, "mmb_attention_fwd: out row stride < H");
  dim3 grid(B * nheads), block(N <= 64 ? 64 : N <= 128 ? 128 : N <= 192 ? 192 : 256);
  hipStream_t st = as_stream(stream);
  const float scale = 1.0f / sqrtf((float)dh);
  MMB_ATTN_DISPATCH(attn_fwd_kernel, N, H, nheads, dh, qkv, qkv_ld, maskf, valid_is_zero, scale, out, out_ld, lse);
  return check_launch("mmb_attention_fwd");
}

extern "C" int mmb_attention_bwd(int B, int N, int H, int nheads, const float* qkv, int64_t qkv_ld,
                                 const float* maskf, int valid_is_zero, const float* out, int64_t out_ld,
                                 const float* dout, int64_t dout_ld, const float* lse, float* dqkv, int64_t dqkv_ld,
                                 void* stream) {
  MMB_REQUIRE(qkv && maskf && out && dout && lse && dqkv, "mmb_attention_bwd: NULL argument");
  const int dh = nheads > 0 ? H / nheads : 0, dhp = pad_dh(dh);
  const size_t lds = attn_lds_bytes(N, dhp, true);
  if (int rc = attn_check("mmb_attention_bwd", B, N, H, nheads, qkv_ld, lds)) return rc;
  MMB_REQUIRE(out_ld >= H && dout_ld >= H && dqkv_ld >= 3 * (int64_t)H, "mmb_attention_bwd: row stride too small");
  dim3 grid(B * nheads), block(N <= 64 ? 64 : N <= 128 ? 128 : N <= 192 ? 192 : 256);
  hipStream_t st = as_stream(stream);
  const float scale = 1.0f / sqrtf((float)dh);
  MMB_ATTN_DISPATCH(attn_bwd_kernel, N, H, nheads, dh, qkv, qkv_ld, maskf, valid_is_zero, scale, out, out_ld, dout,
                    dout_ld, lse, dqkv, dqkv_ld);
  return check_launch("mmb_attention_bwd");
}

extern "C" int mmb_layernorm_fwd(int64_t M, int D, const float* x, int64_t x_ld, const float* gamma, const float* beta,
                                 float eps, float* y, int64_t y_ld, float* stats, void* stream) {
  MMB_REQUIRE(M > 0 && D > 0, "mmb_layernorm_fwd: empty shape");
  MMB_REQUIRE(x && gamma && beta && y, "mmb_layernorm_fwd: NULL argument");
  MMB_REQUIRE(x_ld >= D && y_ld >= D, "mmb_layernorm_fwd: row stride < D");
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
