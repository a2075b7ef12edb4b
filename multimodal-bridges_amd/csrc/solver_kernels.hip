// K-B / K-C: per-step solver updates of the generation loop -- HBM-bound.
// Reference: model/solvers.py:21-33 (continuous), :59-109 (discrete), model/bridges.py:137-163 (rate).
#include "common.h"

namespace mmb {

// ------------------------------------------------------------------ K-B ---------------
__global__ void __launch_bounds__(256)
euler_step_kernel(int64_t n4, int64_t n, float* __restrict__ x, const float* __restrict__ v, float dt) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4* x4 = reinterpret_cast<float4*>(x);
  const float4* v4 = reinterpret_cast<const float4*>(v);
  for (int64_t j = i; j < n4; j += stride) {
    float4 a = x4[j], d = v4[j];
    a.x += dt * d.x; a.y += dt * d.y; a.z += dt * d.z; a.w += dt * d.w;
    x4[j] = a;
  }
  for (int64_t j = n4 * 4 + i; j < n; j += stride) x[j] += dt * v[j];
}

__global__ void __launch_bounds__(256)
euler_maruyama_kernel(int64_t n, int ND, float* __restrict__ x, const float* __restrict__ v, float dt,
                      const float* __restrict__ t, float t_scalar, float sigma,
                      const float* __restrict__ dW, uint64_t seed, uint64_t offset) {
  // two elements per lane so one Philox call (4 words -> 2 Box-Muller pairs) is fully used
  int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
  for (; i < n; i += stride) {
    float w0, w1;
    if (dW != nullptr) {
      w0 = dW[i];
      w1 = (i + 1 < n) ? dW[i + 1] : 0.0f;
    } else {
      uint32_t r[4];
      philox4(seed, offset, (uint64_t)(i >> 1), 0xE001u, r);
      box_muller(r[0], r[1], w0, w1);
    }
    float t0 = t ? t[i / ND] : t_scalar;
    float s0 = sigma * sqrtf(t0 * (1.0f - t0));
    x[i] += dt * v[i] + s0 * w0;
    if (i + 1 < n) {
      float t1 = t ? t[(i + 1) / ND] : t_scalar;
      float s1 = sigma * sqrtf(t1 * (1.0f - t1));
      x[i + 1] += dt * v[i + 1] + s1 * w1;
    }
  }
}

// ------------------------------------------------------------------ K-C ---------------
constexpr int MAXV = 32;

// rate_v = 1 + (w V/(1-w)) q_v + w q_k with q = softmax(logits) (bridges.py:152-162)
template <int VT>
__device__ inline void load_rates(const float* __restrict__ logits, int64_t p, int V, int kcur, float Bc,
                                  float Cc, float* r) {
  const float* lp = logits + p * (int64_t)V;
  float mx = -INFINITY;
  if constexpr (VT == 8) {
    float4 a = reinterpret_cast<const float4*>(lp)[0];
    float4 b = reinterpret_cast<const float4*>(lp)[1];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  } else {
    for (int v = 0; v < V; ++v) r[v] = lp[v];
  }
  const int VV = VT > 0 ? VT : V;
#pragma unroll
  for (int v = 0; v < VV; ++v) mx = fmaxf(mx, r[v]);
  float s = 0.0f;
#pragma unroll
  for (int v = 0; v < VV; ++v) { r[v] = expf(r[v] - mx); s += r[v]; }
  float qk = 0.0f;
#pragma unroll
  for (int v = 0; v < VV; ++v) { r[v] = r[v] / s; qk = (v == kcur) ? r[v] : qk; }
#pragma unroll
  for (int v = 0; v < VV; ++v) r[v] = 1.0f + Bc * r[v] + Cc * qk;
}

__device__ inline int argmax_first(const float* r, int V) {
  int best = 0;
  float bv = r[0];
  for (int v = 1; v < V; ++v)
    if (r[v] > bv) { bv = r[v]; best = v; }
  return best;
}

__device__ inline int64_t pymod(int64_t a, int64_t m) {
  int64_t r = a % m;
  return r < 0 ? r + m : r;
}

// MODE 0: rates only; 1: tau-leap; 2: discrete Euler
template <int VT, int MODE>
__global__ void __launch_bounds__(256)
jump_kernel(int P, int N, int V, const float* __restrict__ logits, int64_t* __restrict__ k,
            const float* __restrict__ t, float t_scalar, float dt, int thermo_kind, float c,
            const float* __restrict__ noise, uint64_t seed, uint64_t offset, int overflow_clamp,
            int final_argmax, float* __restrict__ rates_out) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    float tt = t ? t[p / N] : t_scalar;
    float w = thermo_w(thermo_kind, c, tt, 1.0f);
    float Bc = (w * (float)V) / (1.0f - w);
    int kcur = (int)k[p];
    float r[VT > 0 ? VT : MAXV];
    load_rates<VT>(logits, p, V, kcur, Bc, w, r);
    const int VV = VT > 0 ? VT : V;
    if (rates_out != nullptr) {
      float* ro = rates_out + (int64_t)p * V;
      if constexpr (VT == 8) {
        reinterpret_cast<float4*>(ro)[0] = make_float4(r[0], r[1], r[2], r[3]);
        reinterpret_cast<float4*>(ro)[1] = make_float4(r[4], r[5], r[6], r[7]);
      } else {
        for (int v = 0; v < V; ++v) ro[v] = r[v];
      }
    }
    if constexpr (MODE == 0) continue;
    if (final_argmax) {                       // multimodal_bridge_matching.py:253-255
      k[p] = argmax_first(r, VV);
      continue;
    }
    int64_t knew = kcur;
    if constexpr (MODE == 1) {
      if (noise != nullptr) {                 // injected Poisson counts (solvers.py:64-79)
        const float* dn = noise + (int64_t)p * V;
        float tot = 0.0f, net = 0.0f;
#pragma unroll
        for (int v = 0; v < VV; ++v) {
          float d = dn[v];
          tot += d;
          net += d * (float)(v - kcur);
        }
        int64_t ok = ((int64_t)tot <= 1) ? 1 : 0;
        int64_t moved = (int64_t)kcur + (int64_t)net * ok;
        knew = overflow_clamp ? (moved < 0 ? 0 : (moved > V - 1 ? V - 1 : moved)) : pymod(moved, V);
      } else {
        // Philox mode.  With independent Poisson(lambda_v) counts only two outcomes change k:
        // P(total = 1, at v) = lambda_v * exp(-Lambda).  Everything else leaves k as it is.
        uint32_t q[4];
        philox4(seed, offset, (uint64_t)p, 0xC001u, q);
        float u = u01(q[0]);
        float Lam = 0.0f;
#pragma unroll
        for (int v = 0; v < VV; ++v) Lam += r[v] * dt;
        float e = expf(-Lam);
        float acc = 0.0f;
        int hit = -1;
#pragma unroll
        for (int v = 0; v < VV; ++v) {
          acc += r[v] * dt * e;
          if (hit < 0 && u < acc) hit = v;
        }
        if (hit >= 0) knew = hit;
      }
    } else {                                   // discrete Euler (solvers.py:95-107)
      float dp[VT > 0 ? VT : MAXV];
      float off = 0.0f;
#pragma unroll
      for (int v = 0; v < VV; ++v) {
        float d = fminf(r[v] * dt, 1.0f);
        d = (v == kcur) ? 0.0f : d;
        dp[v] = d;
        off += d;
      }
      float diag = fmaxf(1.0f - off, 0.0f);
      float u;
      if (noise != nullptr) {
        u = noise[p];
      } else {
        uint32_t q[4];
        philox4(seed, offset, (uint64_t)p, 0xC002u, q);
        u = u01(q[0]);
      }
      float S = 0.0f;
#pragma unroll
      for (int v = 0; v < VV; ++v) { dp[v] = (v == kcur) ? diag : dp[v]; S += dp[v]; }
      float thr = u * S, acc = 0.0f;
      int cnt = 0;
#pragma unroll
      for (int v = 0; v < VV; ++v) { acc += dp[v]; cnt += (acc <= thr) ? 1 : 0; }
      knew = cnt < V ? cnt : V - 1;
    }
    k[p] = knew;
  }
}

template <int MODE>
static int launch_jump(const char* name, int B, int N, int V, const float* logits, int64_t* k, const float* t,
                       float t_scalar, float dt, int thermo_kind, float gamma, const float* noise,
                       uint64_t seed, uint64_t offset, int overflow_clamp, int final_argmax,
                       float* rates_out, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "%s: empty shape B=%d N=%d", name, B, N);
  MMB_REQUIRE(V > 1 && V <= MAXV, "%s: vocab_size %d outside [2,%d]", name, V, MAXV);
  MMB_REQUIRE((int64_t)B * N < (1ll << 31), "%s: B*N too large", name);
  MMB_REQUIRE(logits && k, "%s: logits/k must not be NULL", name);
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "%s: bad thermostat %d", name, thermo_kind);
  if (MODE == 0) MMB_REQUIRE(rates_out, "%s: rates must not be NULL", name);
  int P = B * N;
  float c = (float)(-(double)V * (double)gamma);
  dim3 grid(stream_grid(P, 256)), block(256);
  bool v8 = (V == 8) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) &&
            (!rates_out || (reinterpret_cast<uintptr_t>(rates_out) & 15) == 0);
  if (v8)
    hipLaunchKernelGGL((jump_kernel<8, MODE>), grid, block, 0, as_stream(stream), P, N, V, logits, k, t, t_scalar,
                       dt, thermo_kind, c, noise, seed, offset, overflow_clamp, final_argmax, rates_out);
  else
    hipLaunchKernelGGL((jump_kernel<0, MODE>), grid, block, 0, as_stream(stream), P, N, V, logits, k, t, t_scalar,
                       dt, thermo_kind, c, noise, seed, offset, overflow_clamp, final_argmax, rates_out);
  return check_launch(name);
}

// ------------------------------------------------------------------ post-processing ----
// Generation post-processing (pipeline/callbacks_generator.py:99-128): inverse of the data
// transform -- x*scale + shift per feature, for "log_pt" additionally exp(.) - 1e-6 on what the reference's
// `paths.continuous[:, :, 0]` selects -- followed by TensorMultiModal.apply_mask (tensorclass.py:96-106):
// x *= mask, k = (k*mask).long().  One lane per particle, in place; 2*(4 Dc + 8) + 4 B/particle.
// log_pt: 0 off; 1 feature 0 of every particle (a (J,N,Dc) sample); 2 EVERY feature of slot 0 of every jet -- what the
// same index expression selects on the stacked (T,J,N,Dc) paths the reference's callback hands over (:69,:116).
__device__ __forceinline__ float post_value(float v, int d, bool slot0, const float* scale, const float* shift,
                                            int log_pt) {
  if (scale != nullptr) v = v * scale[d] + shift[d];
  if ((log_pt == 1 && d == 0) || (log_pt == 2 && slot0)) v = expf(v) - 1e-6f;
  return v;
}

__global__ void __launch_bounds__(256)
postprocess_kernel(int64_t P, int N, int Dc, float* __restrict__ x, int64_t* __restrict__ k,
                   const float* __restrict__ maskf, const float* __restrict__ scale,
                   const float* __restrict__ shift, int log_pt) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const float m = maskf[p];
    const bool slot0 = (p % N) == 0;
    if (x != nullptr) {
      float* xp = x + p * Dc;
      for (int d = 0; d < Dc; ++d) xp[d] = post_value(xp[d], d, slot0, scale, shift, log_pt) * m;
    }
    if (k != nullptr) k[p] = (int64_t)((float)k[p] * m);           // (discrete * mask).long(): truncation toward zero
  }
}

// "onehot" samples (callbacks_generator.py:118-125): the last V of the D = Dc + V continuous channels carry the token
// scores.  After the inverse transform of ALL D channels: k = argmax of the last V (first maximum, like
// torch.argmax), x = the first Dc channels (none when Dc = 0), then apply_mask.  Out of place: (P,D) -> (P,Dc), (P).
__global__ void __launch_bounds__(256)
postprocess_onehot_kernel(int64_t P, int N, int D, int V, const float* __restrict__ xin, float* __restrict__ xout,
                          int64_t* __restrict__ kout, const float* __restrict__ maskf,
                          const float* __restrict__ scale, const float* __restrict__ shift, int log_pt) {
  const int Dc = D - V;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const float m = maskf[p];
    const bool slot0 = (p % N) == 0;
    const float* xp = xin + p * D;
    for (int d = 0; d < Dc; ++d) xout[p * Dc + d] = post_value(xp[d], d, slot0, scale, shift, log_pt) * m;
    int best = 0;
    float bv = post_value(xp[Dc], Dc, slot0, scale, shift, log_pt);
    for (int v = 1; v < V; ++v) {
      const float c = post_value(xp[Dc + v], Dc + v, slot0, scale, shift, log_pt);
      if (c > bv || (c != c && bv == bv)) { bv = c; best = v; }     // torch.argmax treats NaN as the maximum
    }
    kout[p] = (int64_t)((float)best * m);
  }
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_postprocess(int64_t P, int N, int Dc, float* x, int64_t* k, const float* maskf, const float* scale,
                               const float* shift, int log_pt, void* stream) {
  MMB_REQUIRE(P > 0 && N > 0, "mmb_postprocess: empty batch (P=%lld N=%d)", (long long)P, N);
  MMB_REQUIRE(maskf && (x || k), "mmb_postprocess: NULL argument");
  if (x) MMB_REQUIRE(Dc > 0, "mmb_postprocess: Dc=%d", Dc);
  MMB_REQUIRE((scale == nullptr) == (shift == nullptr), "mmb_postprocess: scale and shift go together");
  MMB_REQUIRE(log_pt >= 0 && log_pt <= 2, "mmb_postprocess: log_pt=%d", log_pt);
  if (log_pt) MMB_REQUIRE(scale, "mmb_postprocess: log_pt needs the standardisation statistics");
  hipLaunchKernelGGL(postprocess_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, as_stream(stream), P, N, Dc, x, k,
                     maskf, scale, shift, log_pt);
  return check_launch("mmb_postprocess");
}

extern "C" int mmb_postprocess_onehot(int64_t P, int N, int D, int V, const float* xin, float* xout, int64_t* kout,
                                      const float* maskf, const float* scale, const float* shift, int log_pt,
                                      void* stream) {
  MMB_REQUIRE(P > 0 && N > 0, "mmb_postprocess_onehot: empty batch (P=%lld N=%d)", (long long)P, N);
  MMB_REQUIRE(V > 0 && D >= V, "mmb_postprocess_onehot: D=%d channels cannot hold V=%d token scores", D, V);
  MMB_REQUIRE(xin && kout && maskf, "mmb_postprocess_onehot: NULL argument");
  MMB_REQUIRE((D == V) || xout, "mmb_postprocess_onehot: xout is NULL but D - V = %d continuous channels remain", D - V);
  MMB_REQUIRE((scale == nullptr) == (shift == nullptr), "mmb_postprocess_onehot: scale and shift go together");
  MMB_REQUIRE(log_pt >= 0 && log_pt <= 2, "mmb_postprocess_onehot: log_pt=%d", log_pt);
  if (log_pt) MMB_REQUIRE(scale, "mmb_postprocess_onehot: log_pt needs the standardisation statistics");
  hipLaunchKernelGGL(postprocess_onehot_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, as_stream(stream), P, N, D, V,
                     xin, xout, kout, maskf, scale, shift, log_pt);
  return check_launch("mmb_postprocess_onehot");
}

extern "C" int mmb_euler_step(int64_t n, float* x, const float* v, float dt, void* stream) {
  MMB_REQUIRE(n > 0, "mmb_euler_step: n=%lld", (long long)n);
  MMB_REQUIRE(x && v, "mmb_euler_step: NULL argument");
  bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  int64_t n4 = al ? n / 4 : 0;
  hipLaunchKernelGGL(euler_step_kernel, dim3(stream_grid(al ? n4 + 1 : n, 256)), dim3(256), 0, as_stream(stream),
                     n4, n, x, v, dt);
  return check_launch("mmb_euler_step");
}

extern "C" int mmb_euler_maruyama_step(int B, int N, int Dc, float* x, const float* v, float dt, const float* t,
                                       float t_scalar, float sigma, const float* dW, uint64_t seed,
                                       uint64_t offset, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && Dc > 0, "mmb_euler_maruyama_step: bad shape");
  MMB_REQUIRE(x && v, "mmb_euler_maruyama_step: NULL argument");
  int64_t n = (int64_t)B * N * Dc;
  hipLaunchKernelGGL(euler_maruyama_kernel, dim3(stream_grid((n + 1) / 2, 256)), dim3(256), 0, as_stream(stream),
                     n, N * Dc, x, v, dt, t, t_scalar, sigma, dW, seed, offset);
  return check_launch("mmb_euler_maruyama_step");
}

extern "C" int mmb_telegraph_rate(int B, int N, int V, const float* logits, const int64_t* k, const float* t,
                                  float t_scalar, int thermo_kind, float gamma, float* rates, void* stream) {
  return launch_jump<0>("mmb_telegraph_rate", B, N, V, logits, const_cast<int64_t*>(k), t, t_scalar, 0.0f,
                        thermo_kind, gamma, nullptr, 0, 0, 0, 0, rates, stream);
}

extern "C" int mmb_rate_tauleap(int B, int N, int V, const float* logits, int64_t* k, const float* t,
                                float t_scalar, float dt, int thermo_kind, float gamma, const float* delta_n,
                                uint64_t seed, uint64_t offset, int overflow_clamp, int final_argmax,
                                float* rates_out, void* stream) {
  return launch_jump<1>("mmb_rate_tauleap", B, N, V, logits, k, t, t_scalar, dt, thermo_kind, gamma, delta_n, seed,
                        offset, overflow_clamp, final_argmax, rates_out, stream);
}

extern "C" int mmb_rate_euler_jump(int B, int N, int V, const float* logits, int64_t* k, const float* t,
                                   float t_scalar, float dt, int thermo_kind, float gamma, const float* u,
                                   uint64_t seed, uint64_t offset, int final_argmax, float* rates_out,
                                   void* stream) {
  return launch_jump<2>("mmb_rate_euler_jump", B, N, V, logits, k, t, t_scalar, dt, thermo_kind, gamma, u, seed,
                        offset, 0, final_argmax, rates_out, stream);
}
