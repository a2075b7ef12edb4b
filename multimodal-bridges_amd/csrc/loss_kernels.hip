// K-F: masked MSE + masked cross-entropy, forward and head gradients in one pass; mask
// canonicalisation; MultiModeLoss.  Reference: model/multimodal_bridge_matching.py:165-192,268-306.
#include "common.h"
#include <mutex>

// Scratch of mmb_loss_fwd_bwd_noalloc: the per-block partial sums (2 floats per block, at most
// MMB_LOSS_MAX_BLOCKS blocks = 16 KiB) that mmb_loss_fwd_bwd takes as its caller-provided `partials` workspace.
// Every GPU the library runs on holds its own copy of this array for the life of the process.  In host code the
// NAME below is only a stand-in for the symbol: kernels get the per-device address from hipGetSymbolAddress.
static __device__ __align__(256) float scratch[2 * MMB_LOSS_MAX_BLOCKS];

namespace mmb {

__device__ inline float drift_value2(int bridge_kind, float t, float xt, float x0, float x1) {
  if (bridge_kind == MMB_BRIDGE_SCHRODINGER) {
    float den = t * (1.0f - t);
    float A = (1.0f - 2.0f * t) / den;
    float Bc = (t * t) / den;
    float C = -1.0f * ((1.0f - t) * (1.0f - t)) / den;
    return A * xt + Bc * x1 + C * x0;
  }
  return 0.0f * xt + 1.0f * x1 + (-1.0f) * x0;
}

constexpr int LOSS_MAXV = 32;

template <bool V8>
__global__ void __launch_bounds__(256)
loss_kernel(int P, int N, int Dc, int V, const float* __restrict__ head_c, const float* __restrict__ head_d,
            const float* __restrict__ t, const float* __restrict__ xt, const float* __restrict__ x0,
            const float* __restrict__ x1, int bridge_kind, const int64_t* __restrict__ k1,
            const float* __restrict__ maskf, const int32_t* __restrict__ total,
            const float* __restrict__ vocab_freq, const float* __restrict__ coef,
            float* __restrict__ partials, float* __restrict__ d_head_c, float* __restrict__ d_head_d) {
  __shared__ float red[2][4];
  const float msum = (float)total[0];
  const float gc = coef ? coef[0] / msum : 0.0f;
  const float gd = coef ? coef[1] / msum : 0.0f;
  float acc_c = 0.0f, acc_d = 0.0f;
  const float invN = 1.0f / (float)N;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const float m = maskf[p];
    if (head_c != nullptr) {
      int jet = (int)((float)p * invN);                  // p / N without the integer-division sequence:
      jet += ((jet + 1) * N <= p) ? 1 : 0;               // float estimate, then exact correction by +-1
      jet -= (jet * N > p) ? 1 : 0;
      const float tt = t[jet];
      const int64_t base = (int64_t)p * Dc;
      for (int d = 0; d < Dc; ++d) {
        float target = drift_value2(bridge_kind, tt, xt[base + d], x0[base + d], x1[base + d]);
        float diff = head_c[base + d] - target;
        acc_c += (diff * diff) * m;
        if (d_head_c) d_head_c[base + d] = gc * 2.0f * diff * m;
      }
    }
    if (head_d != nullptr) {
      const float* lp = head_d + (int64_t)p * V;
      const int y = (int)k1[p];
      const float wy = vocab_freq ? vocab_freq[y] : 1.0f;
      if (V8) {                                   // vocab 8: two float4 loads / stores, one exp per class
        const float4 a = reinterpret_cast<const float4*>(lp)[0], bq = reinterpret_cast<const float4*>(lp)[1];
        float l[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        float mx = l[0];
#pragma unroll
        for (int v = 1; v < 8; ++v) mx = fmaxf(mx, l[v]);
        float e[8], s = 0.0f, ly = 0.0f;
#pragma unroll
        for (int v = 0; v < 8; ++v) { e[v] = __expf(l[v] - mx); s += e[v]; ly = (v == y) ? l[v] : ly; }
        const float lse = mx + __logf(s);                // hardware exp2/log2: ~2 ulp, far inside the 1e-4 bar
        acc_d += (wy * (lse - ly)) * m;
        if (d_head_d) {
          const float sc = gd * wy * m, inv = __builtin_amdgcn_rcpf(s);
          float gq[8];
#pragma unroll
          for (int v = 0; v < 8; ++v) gq[v] = sc * (e[v] * inv - (v == y ? 1.0f : 0.0f));
          float4* gp = reinterpret_cast<float4*>(d_head_d + (int64_t)p * 8);
          gp[0] = make_float4(gq[0], gq[1], gq[2], gq[3]);
          gp[1] = make_float4(gq[4], gq[5], gq[6], gq[7]);
        }
      } else {
        float mx = -INFINITY;
        for (int v = 0; v < V; ++v) mx = fmaxf(mx, lp[v]);
        float s = 0.0f;
        for (int v = 0; v < V; ++v) s += expf(lp[v] - mx);
        const float lse = mx + logf(s);
        acc_d += (wy * (lse - lp[y])) * m;
        if (d_head_d) {
          float* gp = d_head_d + (int64_t)p * V;
          const float sc = gd * wy * m;
          for (int v = 0; v < V; ++v) gp[v] = sc * (expf(lp[v] - lse) - (v == y ? 1.0f : 0.0f));
        }
      }
    }
  }
  acc_c = wave_sum(acc_c);
  acc_d = wave_sum(acc_d);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wid] = acc_c; red[1][wid] = acc_d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// The paper-size case (both heads, Dc = 3, V = 8, 16-byte aligned, P % 4 == 0) with every access 16 bytes per lane:
// the MSE term is elementwise over the FLAT (P*3) arrays, so thread i takes float4 number i of head_c / xt / x0 / x1
// (floats 4i..4i+3 belong to particles p0 = 4i/3 and p0+1) -- fully coalesced, no 12-byte-stride dword triplets --
// and the cross-entropy of particle i (two float4 of logits).  3P/4 float4 items against P particles: the last
// quarter of the threads of a sweep has no MSE item.
__global__ void __launch_bounds__(256)
loss_kernel_fast(int P, int N, const float4* __restrict__ head_c, const float4* __restrict__ head_d,
                 const float* __restrict__ t, const float4* __restrict__ xt, const float4* __restrict__ x0,
                 const float4* __restrict__ x1, int bridge_kind, const int64_t* __restrict__ k1,
                 const float* __restrict__ maskf, const int32_t* __restrict__ total,
                 const float* __restrict__ vocab_freq, const float* __restrict__ coef, float* __restrict__ partials,
                 float4* __restrict__ d_head_c, float4* __restrict__ d_head_d) {
  __shared__ float red[2][4];
  const float msum = (float)total[0];
  const float gc = coef ? coef[0] / msum : 0.0f;
  const float gd = coef ? coef[1] / msum : 0.0f;
  float acc_c = 0.0f, acc_d = 0.0f;
  const float invN = 1.0f / (float)N;
  const int n4 = (P / 4) * 3;
  auto jet_of = [&](int p) {
    int jet = (int)((float)p * invN);                    // p / N: float estimate, exact correction by +-1
    jet += ((jet + 1) * N <= p) ? 1 : 0;
    jet -= (jet * N > p) ? 1 : 0;
    return jet;
  };
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
    // ---- cross-entropy of particle i
    const float4 a = head_d[2 * (int64_t)i], bq = head_d[2 * (int64_t)i + 1];
    const int y = (int)k1[i];
    const float m = maskf[i];
    float4 hv, xtv, x0v, x1v;
    float ma = 0.0f, mb = 0.0f, ta = 0.0f, tb = 0.0f;
    int r = 0;
    const bool has_c = i < n4;
    if (has_c) {
      hv = head_c[i]; xtv = xt[i]; x0v = x0[i]; x1v = x1[i];
      const unsigned f0 = 4u * (unsigned)i;
      const int p0 = (int)(__umulhi(f0, 0xAAAAAAABu) >> 1);          // f0 / 3
      r = (int)(f0 - 3u * (unsigned)p0);
      ma = maskf[p0]; mb = maskf[p0 + 1];
      ta = t[jet_of(p0)]; tb = t[jet_of(p0 + 1)];
    }
    {
      const float wy = vocab_freq ? vocab_freq[y] : 1.0f;
      float l[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
      float mx = l[0];
#pragma unroll
      for (int v = 1; v < 8; ++v) mx = fmaxf(mx, l[v]);
      float e[8], sm = 0.0f, ly = 0.0f;
#pragma unroll
      for (int v = 0; v < 8; ++v) { e[v] = __expf(l[v] - mx); sm += e[v]; ly = (v == y) ? l[v] : ly; }
      const float lse = mx + __logf(sm);
      acc_d += (wy * (lse - ly)) * m;
      if (d_head_d) {
        const float sc = gd * wy * m, inv = __builtin_amdgcn_rcpf(sm);
        float gq[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) gq[v] = sc * (e[v] * inv - (v == y ? 1.0f : 0.0f));
        d_head_d[2 * (int64_t)i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
        d_head_d[2 * (int64_t)i + 1] = make_float4(gq[4], gq[5], gq[6], gq[7]);
      }
    }
    if (has_c) {
      const float h[4] = {hv.x, hv.y, hv.z, hv.w}, xa[4] = {xtv.x, xtv.y, xtv.z, xtv.w};
      const float xb[4] = {x0v.x, x0v.y, x0v.z, x0v.w}, xc[4] = {x1v.x, x1v.y, x1v.z, x1v.w};
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool second = r + j >= 3;                  // floats r+j >= 3 belong to particle p0 + 1
        const float mm = second ? mb : ma, tt = second ? tb : ta;
        const float diff = h[j] - drift_value2(bridge_kind, tt, xa[j], xb[j], xc[j]);
        acc_c += (diff * diff) * mm;
        g[j] = gc * 2.0f * diff * mm;
      }
      if (d_head_c) d_head_c[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
  acc_c = wave_sum(acc_c);
  acc_d = wave_sum(acc_d);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wid] = acc_c; red[1][wid] = acc_d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void loss_finalize_kernel(int nblocks, const float* __restrict__ partials,
                                     const int32_t* __restrict__ total, float* __restrict__ losses) {
  // one wave; deterministic order
  float a = 0.0f, b = 0.0f;
  for (int i = threadIdx.x; i < nblocks; i += 64) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  a = wave_sum(a);
  b = wave_sum(b);
  if (threadIdx.x == 0) {
    const float msum = (float)total[0];
    losses[0] = a / msum;
    losses[1] = b / msum;
  }
}

__global__ void multimode_kernel(const float* losses, const float* s, float w0, float w1, float* coef_out,
                                 float* loss_out, float* d_s) {
  if (threadIdx.x != 0) return;
  float c0, c1;
  if (s != nullptr) { c0 = 0.5f * expf(-s[0]); c1 = 0.5f * expf(-s[1]); } else { c0 = w0; c1 = w1; }
  if (coef_out) { coef_out[0] = c0; coef_out[1] = c1; }
  if (loss_out) {
    float l = c0 * losses[0] + c1 * losses[1];
    if (s != nullptr) l = (c0 * losses[0] + 0.5f * s[0]) + (c1 * losses[1] + 0.5f * s[1]);
    loss_out[0] = l;
  }
  if (d_s && s) { d_s[0] = -c0 * losses[0] + 0.5f; d_s[1] = -c1 * losses[1] + 0.5f; }
}

template <typename MT>
__global__ void __launch_bounds__(256)
mask_prepare_kernel(int B, int N, const MT* __restrict__ mask, float* __restrict__ maskf, int32_t* __restrict__ counts,
                    int32_t* __restrict__ total) {
  // one wave per jet, grid-stride over jets; the batch total costs one atomic per block
  __shared__ int blk[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int mine = 0;
  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    int cnt = 0;
    for (int n = lane; n < N; n += 64) {
      const bool on = mask[(int64_t)b * N + n] != (MT)0;
      maskf[(int64_t)b * N + n] = on ? 1.0f : 0.0f;
      cnt += on ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) counts[b] = cnt;
    mine += cnt;
  }
  if (lane == 0) blk[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && total) {
    const int s = blk[0] + blk[1] + blk[2] + blk[3];
    if (s) atomicAdd(total, s);
  }
}

// N even and <= 128: a lane takes two consecutive slots (one 16-byte load of int64 masks / 8 bytes of f32, one 8-byte
// store) and a wave takes MPJ jets per sweep with their loads in flight together -- the one-jet-at-a-time loop
// above is latency-bound (two dependent 8-byte loads per lane, then a reduction, per jet).  Few, long blocks: every
// block ends in one atomic on the SAME address (total), and those serialise at the L2 (4096 of them cost 40 of the
// first version's 54 us at 8.4 M particles).
constexpr int MPJ = 8;
template <typename MT>
__global__ void __launch_bounds__(1024)
mask_prepare_pair_kernel(int B, int N, const MT* __restrict__ mask, float* __restrict__ maskf,
                         int32_t* __restrict__ counts, int32_t* __restrict__ total) {
  __shared__ int blk[16];
  const int wpb = blockDim.x >> 6;                        // waves per block (16: few blocks, full occupancy)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool active = 2 * lane < N;
  int mine = 0;
  struct alignas(2 * sizeof(MT)) Pair { MT a, b; };
  for (int b0 = (blockIdx.x * wpb + wave) * MPJ; b0 < B; b0 += gridDim.x * wpb * MPJ) {
    Pair v[MPJ];
#pragma unroll
    for (int j = 0; j < MPJ; ++j) {
      v[j].a = (MT)0; v[j].b = (MT)0;
      if (active && b0 + j < B) v[j] = *reinterpret_cast<const Pair*>(mask + (int64_t)(b0 + j) * N + 2 * lane);
    }
    int cnt[MPJ];
#pragma unroll
    for (int j = 0; j < MPJ; ++j) {
      const bool oa = v[j].a != (MT)0, ob = v[j].b != (MT)0;
      if (active && b0 + j < B)
        *reinterpret_cast<float2*>(maskf + (int64_t)(b0 + j) * N + 2 * lane) = make_float2(oa ? 1.0f : 0.0f, ob ? 1.0f : 0.0f);
      cnt[j] = (oa ? 1 : 0) + (ob ? 1 : 0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int j = 0; j < MPJ; ++j) cnt[j] += __shfl_xor(cnt[j], o, 64);
    }
    int own = 0;                                          // lane j keeps jet b0 + j's count
#pragma unroll
    for (int j = 0; j < MPJ; ++j) { own = lane == j ? cnt[j] : own; mine += cnt[j]; }
    if (lane < MPJ && b0 + lane < B) counts[b0 + lane] = own;
  }
  if (lane == 0) blk[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && total) {
    int s = 0;
    for (int w = 0; w < wpb; ++w) s += blk[w];
    if (s) atomicAdd(total, s);
  }
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_mask_prepare(int B, int N, const void* mask, int mask_is_i64, float* maskf, int32_t* counts,
                                int32_t* total, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_mask_prepare: empty shape B=%d N=%d", B, N);
  MMB_REQUIRE(mask && maskf && counts, "mmb_mask_prepare: NULL argument");
  if (total) {
    hipError_t e = hipMemsetAsync(total, 0, sizeof(int32_t), as_stream(stream));
    if (e != hipSuccess) { set_error("mmb_mask_prepare: memset failed: %s", hipGetErrorString(e)); return MMB_ERR_LAUNCH; }
  }
  const bool pairs = N % 2 == 0 && N <= 128 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(maskf) & 7) == 0;
  if (pairs) {
    int grid = (B + 16 * MPJ - 1) / (16 * MPJ);           // one sweep would take this many blocks; at most 2 per CU
    if (grid > 512) grid = 512;
    if (mask_is_i64)
      hipLaunchKernelGGL(mask_prepare_pair_kernel<int64_t>, dim3(grid), dim3(1024), 0, as_stream(stream), B, N,
                         (const int64_t*)mask, maskf, counts, total);
    else
      hipLaunchKernelGGL(mask_prepare_pair_kernel<float>, dim3(grid), dim3(1024), 0, as_stream(stream), B, N,
                         (const float*)mask, maskf, counts, total);
    return check_launch("mmb_mask_prepare");
  }
  if (mask_is_i64)
    hipLaunchKernelGGL(mask_prepare_kernel<int64_t>, dim3(stream_grid((int64_t)B * 64, 256)), dim3(256), 0,
                       as_stream(stream), B, N, (const int64_t*)mask, maskf, counts, total);
  else
    hipLaunchKernelGGL(mask_prepare_kernel<float>, dim3(stream_grid((int64_t)B * 64, 256)), dim3(256), 0,
                       as_stream(stream), B, N, (const float*)mask, maskf, counts, total);
  return check_launch("mmb_mask_prepare");
}

extern "C" int mmb_loss_fwd_bwd(int B, int N, int Dc, int V, const float* head_c, const float* head_d,
                                const float* t, const float* xt, const float* x0, const float* x1, int bridge_kind,
                                const int64_t* k1, const float* maskf, const int32_t* total,
                                const float* vocab_freq, const float* coef, float* partials, float* losses,
                                float* d_head_c, float* d_head_d, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_loss_fwd_bwd: empty shape");
  MMB_REQUIRE((int64_t)B * N < (1ll << 31), "mmb_loss_fwd_bwd: B*N = %lld particles is beyond the maximum of 2^31 - 1",
              (long long)B * N);
  MMB_REQUIRE(head_c || head_d, "mmb_loss_fwd_bwd: no head given");
  MMB_REQUIRE(maskf && total && partials && losses, "mmb_loss_fwd_bwd: NULL argument");
  if (head_c) MMB_REQUIRE(t && xt && x0 && x1 && Dc > 0, "mmb_loss_fwd_bwd: continuous head needs t,xt,x0,x1");
  if (head_d) MMB_REQUIRE(k1 && V > 1 && V <= LOSS_MAXV, "mmb_loss_fwd_bwd: discrete head needs k1, 1<V<=32");
  if (d_head_c || d_head_d) MMB_REQUIRE(coef, "mmb_loss_fwd_bwd: gradients need coef");
  int P = B * N;
  int grid = (P + 255) / 256;
  if (grid > MMB_LOSS_MAX_BLOCKS) grid = MMB_LOSS_MAX_BLOCKS;
  const bool v8 = head_d && V == 8 && ((reinterpret_cast<uintptr_t>(head_d) & 15) == 0) &&
                  (!d_head_d || (reinterpret_cast<uintptr_t>(d_head_d) & 15) == 0);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // the flat-array kernel indexes float4 number i of the (P*3) arrays in 32 bits (4*i wraps at P >= 2^30) and finds the
  // jet of a particle by a float estimate +-1 (exact below 2^24 jets): beyond either, the per-particle kernel
  const bool fast = v8 && head_c && Dc == 3 && P % 4 == 0 && al16(head_c) && al16(xt) && al16(x0) && al16(x1) &&
                    (!d_head_c || al16(d_head_c)) && P < (1 << 30) && B < (1 << 24);
  if (fast)
    hipLaunchKernelGGL(loss_kernel_fast, dim3(grid), dim3(256), 0, as_stream(stream), P, N,
                       reinterpret_cast<const float4*>(head_c), reinterpret_cast<const float4*>(head_d), t,
                       reinterpret_cast<const float4*>(xt), reinterpret_cast<const float4*>(x0),
                       reinterpret_cast<const float4*>(x1), bridge_kind, k1, maskf, total, vocab_freq, coef, partials,
                       reinterpret_cast<float4*>(d_head_c), reinterpret_cast<float4*>(d_head_d));
  else if (v8)
    hipLaunchKernelGGL(loss_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), P, N, Dc, V, head_c, head_d, t,
                       xt, x0, x1, bridge_kind, k1, maskf, total, vocab_freq, coef, partials, d_head_c, d_head_d);
  else
    hipLaunchKernelGGL(loss_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), P, N, Dc, V, head_c, head_d,
                       t, xt, x0, x1, bridge_kind, k1, maskf, total, vocab_freq, coef, partials, d_head_c, d_head_d);
  int rc = check_launch("mmb_loss_fwd_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), grid, partials, total, losses);
  return check_launch("mmb_loss_fwd_bwd(finalize)");
}

// ---- the entry point that allocates nothing (no counterpart in the reference) ----------------------------------
namespace {
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices];          // per device: each GPU holds its own copy of `scratch` at its own address
std::mutex g_scratch_mutex;

// device address of `scratch` on the CURRENT device, resolved once per device (no allocation: the array is part
// of the code object)
int scratch_on_current_device(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    mmb::set_error("mmb_loss_fwd_bwd_noalloc: hipGetDevice failed or device %d outside [0,%d): %s", dev, kMaxDevices,
                   hipGetErrorString(e));
    return MMB_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    float* p = nullptr;
    e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      mmb::set_error("mmb_loss_fwd_bwd_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                     hipGetErrorString(e));
      return MMB_ERR_LAUNCH;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return MMB_OK;
}
}  // namespace

extern "C" int mmb_loss_fwd_bwd_noalloc(int B, int N, int Dc, int V, const float* head_c, const float* head_d,
                                        const float* t, const float* xt, const float* x0, const float* x1,
                                        int bridge_kind, const int64_t* k1, const float* maskf, const int32_t* total,
                                        const float* vocab_freq, const float* coef, float* losses, float* d_head_c,
                                        float* d_head_d, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_loss_fwd_bwd_noalloc: empty shape");
  MMB_REQUIRE((int64_t)B * N < (1ll << 31),
              "mmb_loss_fwd_bwd_noalloc: B*N = %lld particles is beyond the maximum of 2^31 - 1", (long long)B * N);
  float* ptr = nullptr;
  if (int rc = scratch_on_current_device(&ptr)) return rc;
  // every block overwrites its own partial pair and the finalize kernel reads exactly the pairs of this launch's
  // grid: nothing in the scratch has to start at zero, a previous (larger) call's leftovers are never read
  return mmb_loss_fwd_bwd(B, N, Dc, V, head_c, head_d, t, xt, x0, x1, bridge_kind, k1, maskf, total, vocab_freq, coef,
                          ptr, losses, d_head_c, d_head_d, stream);
}

extern "C" int mmb_multimode_coef(const float* s, const float* fixed_w_host2, float* coef_out, void* stream) {
  MMB_REQUIRE(coef_out, "mmb_multimode_coef: NULL output");
  MMB_REQUIRE(s || fixed_w_host2, "mmb_multimode_coef: need learnable s or fixed weights");
  float w0 = fixed_w_host2 ? fixed_w_host2[0] : 0.0f, w1 = fixed_w_host2 ? fixed_w_host2[1] : 0.0f;
  hipLaunchKernelGGL(multimode_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const float*)nullptr, s, w0, w1,
                     coef_out, (float*)nullptr, (float*)nullptr);
  return check_launch("mmb_multimode_coef");
}

extern "C" int mmb_multimode_loss(const float* losses, const float* s, const float* fixed_w_host2, float* loss_out,
                                  float* d_s, void* stream) {
  MMB_REQUIRE(losses && loss_out, "mmb_multimode_loss: NULL argument");
  MMB_REQUIRE(s || fixed_w_host2, "mmb_multimode_loss: need learnable s or fixed weights");
  float w0 = fixed_w_host2 ? fixed_w_host2[0] : 0.0f, w1 = fixed_w_host2 ? fixed_w_host2[1] : 0.0f;
  hipLaunchKernelGGL(multimode_kernel, dim3(1), dim3(64), 0, as_stream(stream), losses, s, w0, w1,
                     (float*)nullptr, loss_out, d_s);
  return check_launch("mmb_multimode_loss");
}
