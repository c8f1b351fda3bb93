// isf_bn_train.hip -- SURVEY 8f #2 (round 5): BatchNorm1d with BATCH statistics (+ residual, + ReLU) on [N, C] feature rows,
// forward and backward -- the norm / activation of the reference's sparse blocks in TRAINING mode
// (make_sparse_convmodule / SparseBasicBlock: ops/sparse_block.py:117-134,137-199; naiveSyncBN1d: ops/norm.py:136-211; the
// DynamicVFE layers: voxel_encoders/utils.py:116-144).
//
// Stock torch runs each BN as batch_norm_collect_statistics_channels_last + transform_input (forward),
// batch_norm_backward_reduce + backward_elemt (backward), a ReLU and a threshold_backward around it, the residual add and
// its gradient copy: 80-90 us per statistics kernel on rows a streaming pass covers in 10 (the channels-last kernels are
// built for [N, C, H, W] images), ~10 launches per layer and direction -- 4.5 ms and ~350 launches of a training step
// (profiles/r05_train_step_mid.txt).  Here, per direction: one reduction pass (per-block partial sums -> an ordered
// second-level sum: deterministic, no atomics) and one elementwise pass that does everything else:
//   forward   stats  = (sum x, sum x^2) per channel                      [all-reduced across ranks by the caller for sync-BN]
//             y      = relu(x * gamma * invstd + (beta - mean * gamma * invstd) + residual); running stats, saved mean / invstd
//   backward  sums   = (sum g, sum g * xhat), g = dy * (y > 0)           [all-reduced likewise]
//             dx     = gamma * invstd * (g - sum_g / n - xhat * sum_gx / n); dresidual = g; dgamma = sum_gx; dbeta = sum_g
#include <algorithm>

#include "isf_common.h"

namespace isf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = 512;

// MODE 0: (x, x^2); MODE 1: (g, g * xhat) with g = dy masked by y > 0 when y is given
template <int MODE>
__global__ __launch_bounds__(kBnThreads) void bn_partial_kernel(const float* __restrict__ a /* x | dy */,
                                                                const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ mean_invstd, int n, int c,
                                                                float* __restrict__ partial /* [blocks][2c] */) {
  extern __shared__ float red[];                         // [stripes][2c]
  const int cq = c >> 2;                                 // channel quads
  const int stripes = kBnThreads / cq;                   // row stripes of a block (cq <= 256)
  const int q = threadIdx.x % cq, stripe = threadIdx.x / cq;
  f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 mu = f32x4{0.f, 0.f, 0.f, 0.f}, is = f32x4{1.f, 1.f, 1.f, 1.f};
  f32x4 pv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) {
    mu = *reinterpret_cast<const f32x4*>(mean_invstd + 4 * q);
    is = *reinterpret_cast<const f32x4*>(mean_invstd + c + 4 * q);
  } else if (x) {                                        // MODE 0: `x` carries the pivot row [c] (or nullptr)
    pv = *reinterpret_cast<const f32x4*>(x + 4 * q);
  }
  if (stripe < stripes) {
    for (long long r = (long long)blockIdx.x * stripes + stripe; r < n; r += (long long)gridDim.x * stripes) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a + r * c + 4 * q);
      if (MODE == 0) {
        const f32x4 d = v - pv;           // sums about the pivot row (zero when none): see isf_bn1d_stats_pivot
        s0 += d;
        s1 += d * d;
      } else {
        f32x4 g = v;
        if (y) {
          const f32x4 yy = *reinterpret_cast<const f32x4*>(y + r * c + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
        }
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + r * c + 4 * q) - mu) * is;
        s0 += g;
        s1 += g * xh;
      }
    }
    *reinterpret_cast<f32x4*>(red + stripe * 2 * c + 4 * q) = s0;
    *reinterpret_cast<f32x4*>(red + stripe * 2 * c + c + 4 * q) = s1;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * c; e += kBnThreads) {   // stripes in order: deterministic
    float t = 0.f;
    for (int st = 0; st < stripes; ++st) t += red[st * 2 * c + e];
    partial[(size_t)blockIdx.x * 2 * c + e] = t;
  }
}

// second level: one wave per output value, lane l adds partials l, l + 64, ... and the wave folds its 64 sums in a fixed
// butterfly -- ordered (deterministic) and 64-way parallel (one thread walking 512 partials serially took 100 us per call:
// 4.8 ms of a training step, profiles/r05_train_step_after.txt)
__global__ __launch_bounds__(64) void bn_final_sum_kernel(const float* __restrict__ partial, int blocks, int c2,
                                                          float* __restrict__ out) {
  const int e = blockIdx.x;
  float t = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 64) t += partial[(size_t)b * c2 + e];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
  if (threadIdx.x == 0) out[e] = t;
}

__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const float* __restrict__ x, long long n4 /* n * c / 4 */, int c,
                                                              const float* __restrict__ stats, float count,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float momentum, int unbiased,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              const float* __restrict__ residual, int relu,
                                                              float* __restrict__ y, float* __restrict__ mean_invstd,
                                                              const float* __restrict__ pivot /* [c] or nullptr */,
                                                              long long* __restrict__ num_batches_tracked /* or nullptr */) {
  const int cq = c >> 2;
  const float inv_n = 1.f / count;
  if (blockIdx.x == 0) {                                 // the layer's buffers: running statistics, saved mean / invstd
    if (threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;   // (nn.BatchNorm's counter: no launch of its own)
    for (int ch = threadIdx.x; ch < c; ch += kBnThreads) {
      const float d = stats[ch] * inv_n;                 // mean about the pivot
      const float var = fmaxf(stats[c + ch] * inv_n - d * d, 0.f);
      const float m = d + (pivot ? pivot[ch] : 0.f);
      mean_invstd[ch] = m;
      mean_invstd[c + ch] = rsqrtf(var + eps);
      if (running_mean) {
        const float vr = (unbiased && count > 1.f) ? var * count / (count - 1.f) : var;
        running_mean[ch] += momentum * (m - running_mean[ch]);
        running_var[ch] += momentum * (vr - running_var[ch]);
      }
    }
  }
  for (long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kBnThreads) {
    const int q = (int)(i % cq);
    const f32x4 su = *reinterpret_cast<const f32x4*>(stats + 4 * q) * inv_n;
    const f32x4 sq = *reinterpret_cast<const f32x4*>(stats + c + 4 * q) * inv_n;
    const f32x4 ga = gamma ? *reinterpret_cast<const f32x4*>(gamma + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 be = beta ? *reinterpret_cast<const f32x4*>(beta + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 pv = pivot ? *reinterpret_cast<const f32x4*>(pivot + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float var = fmaxf(sq[j] - su[j] * su[j], 0.f);
      const float sc = ga[j] * rsqrtf(var + eps);
      v[j] = fmaf(v[j], sc, be[j] - (su[j] + pv[j]) * sc);
    }
    if (residual) v += reinterpret_cast<const f32x4*>(residual)[i];
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

__global__ __launch_bounds__(kBnThreads) void bn_backward_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, long long n4, int c,
    const float* __restrict__ mean_invstd, const float* __restrict__ gamma, const float* __restrict__ sums, float count,
    float* __restrict__ dx, float* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int cq = c >> 2;
  const float inv_n = 1.f / count;
  if (blockIdx.x == 0) {
    for (int ch = threadIdx.x; ch < c; ch += kBnThreads) {
      if (dbeta) dbeta[ch] = sums[ch];
      if (dgamma) dgamma[ch] = sums[c + ch];
    }
  }
  for (long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kBnThreads) {
    const int q = (int)(i % cq);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean_invstd + 4 * q);
    const f32x4 is = *reinterpret_cast<const f32x4*>(mean_invstd + c + 4 * q);
    const f32x4 ga = gamma ? *reinterpret_cast<const f32x4*>(gamma + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + 4 * q) * inv_n;
    const f32x4 sgx = *reinterpret_cast<const f32x4*>(sums + c + 4 * q) * inv_n;
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    if (y) {
      const f32x4 yy = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
    }
    if (dres) reinterpret_cast<f32x4*>(dres)[i] = g;
    const f32x4 xh = (reinterpret_cast<const f32x4*>(x)[i] - mu) * is;
    reinterpret_cast<f32x4*>(dx)[i] = ga * is * (g - sg - xh * sgx);
  }
}

static int bn_blocks(int n, int c) {
  const int stripes = kBnThreads / (c >> 2);
  return std::max(1, std::min(kBnMaxBlocks, ceil_div(n, stripes * 8)));
}

template <int MODE>
static int bn_reduce(const float* a, const float* x, const float* y, const float* mean_invstd, int n, int c, float* out,
                     hipStream_t st) {
  Arena& ar = arena_for_stream(st);
  ISF_TRY(ar.reset());
  const int blocks = bn_blocks(n, c);
  float* partial = nullptr;
  ISF_TRY(ar.alloc_n(&partial, (size_t)blocks * 2 * c));
  const int stripes = kBnThreads / (c >> 2);
  hipLaunchKernelGGL(bn_partial_kernel<MODE>, dim3(blocks), dim3(kBnThreads), (size_t)stripes * 2 * c * sizeof(float), st, a,
                     x, y, mean_invstd, n, c, partial);
  hipLaunchKernelGGL(bn_final_sum_kernel, dim3(2 * c), dim3(64), 0, st, partial, blocks, 2 * c, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

static bool bn_channels_ok(int c) { return c >= 4 && c <= 1024 && c % 4 == 0 && kBnThreads % (c >> 2) == 0; }

}  // namespace isf

extern "C" {

int isf_bn1d_stats(const float* x, int num_rows, int channels, float* stats, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(x && stats && num_rows > 0, ISF_ERR_ARG, "bn1d_stats: bad arguments");
  ISF_REQUIRE(bn_channels_ok(channels), ISF_ERR_UNSUPPORTED, "bn1d_stats: %d channels (4 * a divisor of 256)", channels);
  return bn_reduce<0>(x, nullptr, nullptr, nullptr, num_rows, channels, stats, as_stream(stream));
}

int isf_bn1d_stats_pivot(const float* x, int num_rows, int channels, const float* pivot, float* stats, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(x && stats && num_rows > 0, ISF_ERR_ARG, "bn1d_stats_pivot: bad arguments");
  ISF_REQUIRE(bn_channels_ok(channels), ISF_ERR_UNSUPPORTED, "bn1d_stats_pivot: %d channels (4 * a divisor of 256)", channels);
  return bn_reduce<0>(x, pivot, nullptr, nullptr, num_rows, channels, stats, as_stream(stream));
}

int isf_bn1d_apply(const float* x, int num_rows, int channels, const float* stats, float count, const float* gamma,
                   const float* beta, float eps, float momentum, int unbiased_running_var, float* running_mean,
                   float* running_var, const float* residual, int relu, float* y, float* mean_invstd, isf_stream_t stream) {
  return isf_bn1d_apply_pivot(x, num_rows, channels, stats, nullptr, count, gamma, beta, eps, momentum, unbiased_running_var,
                              running_mean, running_var, residual, relu, y, mean_invstd, stream);
}

int isf_bn1d_apply_pivot(const float* x, int num_rows, int channels, const float* stats, const float* pivot, float count,
                         const float* gamma, const float* beta, float eps, float momentum, int unbiased_running_var,
                         float* running_mean, float* running_var, const float* residual, int relu, float* y,
                         float* mean_invstd, isf_stream_t stream) {
  return isf_bn1d_apply_pivot_counted(x, num_rows, channels, stats, pivot, count, gamma, beta, eps, momentum,
                                      unbiased_running_var, running_mean, running_var, nullptr, residual, relu, y, mean_invstd,
                                      stream);
}

int isf_bn1d_apply_pivot_counted(const float* x, int num_rows, int channels, const float* stats, const float* pivot, float count,
                                 const float* gamma, const float* beta, float eps, float momentum, int unbiased_running_var,
                                 float* running_mean, float* running_var, long long* num_batches_tracked,
                                 const float* residual, int relu, float* y, float* mean_invstd, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(x && stats && y && mean_invstd && num_rows > 0 && count >= 1.f && (running_mean == nullptr) == (running_var == nullptr),
              ISF_ERR_ARG, "bn1d_apply: bad arguments");
  ISF_REQUIRE(bn_channels_ok(channels), ISF_ERR_UNSUPPORTED, "bn1d_apply: %d channels", channels);
  const long long n4 = (long long)num_rows * channels / 4;
  const int blocks = (int)std::min<long long>(4096, (n4 + kBnThreads * 4 - 1) / (kBnThreads * 4));
  hipLaunchKernelGGL(bn_apply_kernel, dim3(std::max(1, blocks)), dim3(kBnThreads), 0, as_stream(stream), x, n4, channels, stats,
                     count, gamma, beta, eps, momentum, unbiased_running_var, running_mean, running_var, residual, relu, y,
                     mean_invstd, pivot, num_batches_tracked);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_bn1d_backward_sums(const float* grad_y, const float* x, const float* y_relu, int num_rows, int channels,
                           const float* mean_invstd, float* sums, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(grad_y && x && mean_invstd && sums && num_rows > 0, ISF_ERR_ARG, "bn1d_backward_sums: bad arguments");
  ISF_REQUIRE(bn_channels_ok(channels), ISF_ERR_UNSUPPORTED, "bn1d_backward_sums: %d channels", channels);
  return bn_reduce<1>(grad_y, x, y_relu, mean_invstd, num_rows, channels, sums, as_stream(stream));
}

int isf_bn1d_backward_apply(const float* grad_y, const float* x, const float* y_relu, int num_rows, int channels,
                            const float* mean_invstd, const float* gamma, const float* sums, float count, float* grad_x,
                            float* grad_residual, float* grad_gamma, float* grad_beta, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(grad_y && x && mean_invstd && sums && grad_x && num_rows > 0 && count >= 1.f, ISF_ERR_ARG,
              "bn1d_backward_apply: bad arguments");
  ISF_REQUIRE(bn_channels_ok(channels), ISF_ERR_UNSUPPORTED, "bn1d_backward_apply: %d channels", channels);
  const long long n4 = (long long)num_rows * channels / 4;
  const int blocks = (int)std::min<long long>(4096, (n4 + kBnThreads * 4 - 1) / (kBnThreads * 4));
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(std::max(1, blocks)), dim3(kBnThreads), 0, as_stream(stream), grad_y, x,
                     y_relu, n4, channels, mean_invstd, gamma, sums, count, grad_x, grad_residual, grad_gamma, grad_beta);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
