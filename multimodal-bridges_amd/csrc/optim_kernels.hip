// Optimizer step of the training loop on ONE flat parameter / gradient buffer: global L2-norm clip (Lightning's
// gradient_clip_val, pipeline/experiment.py:217 upstream) + torch.optim.AdamW (model/multimodal_bridge_matching.py:
// 151-154 upstream) in two launches instead of ~85 small torch launches per step.  HBM-bound: 4 B/param for the
// norm, 28 B/param for the update (read p, g, m, v; write p, m, v).
#include "common.h"

namespace mmb {

__global__ void __launch_bounds__(256)
sqnorm_partials_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.0f;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = g4[i];
    s = fmaf(a.x, a.x, s); s = fmaf(a.y, a.y, s); s = fmaf(a.z, a.z, s); s = fmaf(a.w, a.w, s);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float a = g[(n4 << 2) + threadIdx.x]; s = fmaf(a, a, s); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
  float decay;        // 1 - lr * weight_decay
  float b1, b2;       // betas
  float step_size;    // lr / (1 - b1^t)
  float inv_bc2_sqrt; // 1 / sqrt(1 - b2^t)
  float eps;
  float max_norm;     // <= 0: no clipping
};

// every block folds the (<= 1024) norm partials itself: no third launch, no host sync
__global__ void __launch_bounds__(256)
adamw_clip_kernel(int64_t n_active, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                  float* __restrict__ v, const float* __restrict__ partials, int n_partials, AdamArgs a,
                  float* __restrict__ norm_out, int write_clipped) {
  __shared__ float red[4];
  float coef = 1.0f;
  if (a.max_norm > 0.0f) {
    float s = 0.0f;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);          // torch.nn.utils.clip_grad_norm_'s clamp
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_active; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * a.decay;                                  // p.mul_(1 - lr*wd)
    const float mi = m[i] + (gi - m[i]) * (1.0f - a.b1);        // exp_avg.lerp_(g, 1-b1)
    const float vi = v[i] * a.b2 + (1.0f - a.b2) * gi * gi;     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(vi) * a.inv_bc2_sqrt + a.eps;
    pi -= a.step_size * (mi / denom);                           // p.addcdiv_(exp_avg, denom, value=-step_size)
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (write_clipped) g[i] = gi;
  }
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_adamw_clip_step(int64_t n, int64_t n_active, float* params, float* grads, float* exp_avg,
                                   float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int64_t step, float max_norm, float* partials,
                                   float* norm_out, int write_clipped, void* stream) {
  MMB_REQUIRE(n > 0 && n_active >= 0 && n_active <= n, "mmb_adamw_clip_step: bad sizes n=%lld n_active=%lld",
              (long long)n, (long long)n_active);
  MMB_REQUIRE(params && grads && exp_avg && exp_avg_sq, "mmb_adamw_clip_step: NULL argument");
  MMB_REQUIRE(step >= 1, "mmb_adamw_clip_step: step counts from 1");
  MMB_REQUIRE(max_norm <= 0.0f || partials, "mmb_adamw_clip_step: clipping needs the partials workspace");
  MMB_REQUIRE((reinterpret_cast<uintptr_t>(grads) & 15) == 0, "mmb_adamw_clip_step: grads must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  int nb = (int)((n / 4 + 255) / 256);
  if (nb < 1) nb = 1;
  if (nb > MMB_ADAMW_MAX_PARTIALS) nb = MMB_ADAMW_MAX_PARTIALS;
  if (max_norm > 0.0f) {
    hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(nb), dim3(256), 0, st, n, grads, partials);
    if (int rc = check_launch("mmb_adamw_clip_step(norm)")) return rc;
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  AdamArgs a{(float)(1.0 - (double)lr * (double)weight_decay), beta1, beta2, (float)((double)lr / bc1),
             (float)(1.0 / sqrt(bc2)), eps, max_norm};
  if (n_active == 0) return MMB_OK;
  hipLaunchKernelGGL(adamw_clip_kernel, dim3(stream_grid(n_active, 256)), dim3(256), 0, st, n_active, params, grads,
                     exp_avg, exp_avg_sq, partials, nb, a, norm_out, write_clipped);
  return check_launch("mmb_adamw_clip_step");
}
