// Toy hybrid dataset (BASELINE configs[0]): the input rows of the notebook MLP and the two toy
// distributions on device.  Reference: notebooks/data_toys.ipynb cell 3 (MultiModalMLP.forward: cat[time, x,
// embed(k)]), distributions/toys.py:8-55 (NGaussians, TwoMoons).
#include "common.h"

namespace mmb {

// rows[p, :] = [ t(jet of p), x[p, 0:Dc], table[k[p], 0:E] ]      (data_toys.ipynb cell 3, forward: torch.cat)
__global__ void __launch_bounds__(256)
toy_rows_kernel(int64_t P, int N, int Dc, int E, int V, const float* __restrict__ t, int64_t t_stride,
                const float* __restrict__ x, const int64_t* __restrict__ k, const float* __restrict__ table,
                float* __restrict__ rows) {
  const int W = 1 + Dc + E;
  const int64_t total = P * (int64_t)W;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = e / W;
    const int c = (int)(e - p * W);
    float v;
    if (c == 0) v = t[(p / N) * t_stride];
    else if (c <= Dc) v = x[p * Dc + (c - 1)];
    else {
      int64_t tok = k[p];
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);          // nn.Embedding raises on such an index; never read out of range
      v = table[tok * E + (c - 1 - Dc)];
    }
    rows[e] = v;
  }
}

// d table[k[p], e] += d rows[p, col0 + e]   (autograd of nn.Embedding): per-block partial table in LDS, then one
// atomic per touched element and block
constexpr int TOY_MAX_TABLE = 2048;
__global__ void __launch_bounds__(256)
toy_embedding_bwd_kernel(int64_t P, int E, int V, const int64_t* __restrict__ k, const float* __restrict__ d_rows,
                         int64_t ld, int col0, float* __restrict__ d_table) {
  __shared__ float acc[TOY_MAX_TABLE];
  const int n = V * E;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc[i] = 0.0f;
  __syncthreads();
  const int64_t total = P * (int64_t)E;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = e / E;
    const int c = (int)(e - p * E);
    const int64_t tok = k[p];
    if (tok >= 0 && tok < V) atomicAdd(&acc[tok * E + c], d_rows[p * ld + col0 + c]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (acc[i] != 0.0f) atomicAdd(&d_table[i], acc[i]);
}

// NGaussians (toys.py:8-43): point i belongs to gaussian i % V (the reference builds V equal blocks and permutes
// them; the datamodule permutes), centre 5 (cos, sin)(2 pi g / V), covariance sqrt(0.1) I  (std 0.1^(1/4)).
// TwoMoons (toys.py:45-55): torchdyn.generate_moons(n, noise) * 3 - 1; torchdyn is absent here, the law below is the
// scikit-learn one (outer arc (cos a, sin a), inner arc (1 - cos a, 0.5 - sin a), a on an even grid over [0, pi],
// N(0, noise) per coordinate) -- parity unpinned, see DESIGN.md.
__global__ void __launch_bounds__(256)
toy_sample_kernel(int kind, int64_t n, int V, float noise, uint64_t seed, uint64_t offset, float* __restrict__ x,
                  int64_t* __restrict__ k) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(seed, offset, (uint64_t)i, 0x7001u + (uint32_t)kind, r);
    float g0, g1;
    box_muller(r[0], r[1], g0, g1);
    float px, py;
    int64_t label;
    if (kind == 0) {
      const int g = (int)(i % V);
      const float a = 6.28318530717958647692f * (float)g / (float)V;
      const float sd = 0.56234132519034908f;                 // sqrt(sqrt(0.1)): the COVARIANCE is sqrt(0.1) I
      px = 5.0f * cosf(a) + sd * g0;
      py = 5.0f * sinf(a) + sd * g1;
      label = g;
    } else {
      const int64_t n_out = n / 2, n_in = n - n_out;
      if (i < n_out) {
        const float a = n_out > 1 ? 3.14159265358979323846f * (float)i / (float)(n_out - 1) : 0.0f;
        px = cosf(a); py = sinf(a); label = 0;
      } else {
        const int64_t j = i - n_out;
        const float a = n_in > 1 ? 3.14159265358979323846f * (float)j / (float)(n_in - 1) : 0.0f;
        px = 1.0f - cosf(a); py = 1.0f - sinf(a) - 0.5f; label = 1;
      }
      px = (px + noise * g0) * 3.0f - 1.0f;
      py = (py + noise * g1) * 3.0f - 1.0f;
    }
    x[2 * i] = px;
    x[2 * i + 1] = py;
    k[i] = label;
  }
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_toy_rows(int64_t P, int N, int Dc, int E, int V, const float* t, int64_t t_stride, const float* x,
                            const int64_t* k, const float* table, float* rows, void* stream) {
  MMB_REQUIRE(P > 0 && N > 0 && Dc > 0 && E > 0 && V > 0, "mmb_toy_rows: empty shape");
  MMB_REQUIRE(t && x && k && table && rows, "mmb_toy_rows: NULL argument");
  MMB_REQUIRE(t_stride == 0 || t_stride == 1, "mmb_toy_rows: t is a scalar (stride 0) or one value per jet (stride 1)");
  hipLaunchKernelGGL(toy_rows_kernel, dim3(stream_grid(P * (1 + Dc + E), 256)), dim3(256), 0, as_stream(stream), P, N,
                     Dc, E, V, t, t_stride, x, k, table, rows);
  return check_launch("mmb_toy_rows");
}

extern "C" int mmb_toy_embedding_bwd(int64_t P, int E, int V, const int64_t* k, const float* d_rows, int64_t ld,
                                     int col0, float* d_table, void* stream) {
  MMB_REQUIRE(P > 0 && E > 0 && V > 0, "mmb_toy_embedding_bwd: empty shape");
  MMB_REQUIRE(V * E <= TOY_MAX_TABLE, "mmb_toy_embedding_bwd: table of %d x %d exceeds %d entries", V, E, TOY_MAX_TABLE);
  MMB_REQUIRE(k && d_rows && d_table && ld >= col0 + E && col0 >= 0, "mmb_toy_embedding_bwd: bad argument");
  int grid = stream_grid(P * E, 256);
  if (grid > 64) grid = 64;                                   // every block ends in V*E atomics on the same table
  hipLaunchKernelGGL(toy_embedding_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), P, E, V, k, d_rows, ld,
                     col0, d_table);
  return check_launch("mmb_toy_embedding_bwd");
}

extern "C" int mmb_toy_sample(int kind, int64_t n, int V, float noise, uint64_t seed, uint64_t offset, float* x,
                              int64_t* k, void* stream) {
  MMB_REQUIRE(kind == 0 || kind == 1, "mmb_toy_sample: kind %d (0 NGaussians, 1 TwoMoons)", kind);
  MMB_REQUIRE(n > 0 && x && k, "mmb_toy_sample: bad argument");
  if (kind == 0) MMB_REQUIRE(V > 0, "mmb_toy_sample: NGaussians needs V > 0");
  hipLaunchKernelGGL(toy_sample_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), kind, n, V, noise,
                     seed, offset, x, k);
  return check_launch("mmb_toy_sample");
}
