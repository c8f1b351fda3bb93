// isf_fusion.hip -- HSF Point-to-Grid sampling (A8), IGF instance mining (A12) and the multi-scale deformable
// attention gather (A13).
#include <stdlib.h>

#include "isf_common.h"

namespace isf {

// ----------------------------------------------------------------------------------------------------------------
// A8  img_fv_to_bev + img_point_sampling (fusion_encoder.py:965-1070).  The reference loops over samples, builds
// [cam, 3, N] projections with four batched matmuls and calls grid_sample once per camera on an NCHW map, then
// sums over cameras and the T pillar slots and scatters into a zero canvas.  Here: one wave per pillar; the
// T x num_cam projections are computed one per lane, the in-view ones (ballot) are then visited wave-uniformly with
// lanes owning channels of an NHWC image map (1 KB contiguous per bilinear tap); the (camera, slot) sum stays in
// registers and the canvas is written once.
// cam[b*num_cam + k] = 20 floats: M (3x3 row-major) = lidar2img[:3,:3] . inv(lidar_aug[:3,:3]),
//                      v (3) = lidar2img[:3,3] - M . lidar_aug[:3,3], A (2x3) = img_aug[:2,:3], a (2) = img_aug[:2,3]
template <int CPL /* channels per lane */, bool SPLIT = false /* out = one split-format token matrix [B*bev*bev, 256] */>
__global__ __launch_bounds__(256) void p2g_kernel(const float* __restrict__ pillars, int pillar_ld, int T,
                                                  const int32_t* __restrict__ coors, int M,
                                                  const float* __restrict__ img /* [B*cam, H, W, C] */, int num_cam,
                                                  int H, int W, int C, const float* __restrict__ cam, float in_h,
                                                  float in_w, int bev, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= M) return;
  const int b = coors[4 * p], y = coors[4 * p + 2], x = coors[4 * p + 3];
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
  const int c0 = lane * CPL;
  // projection: one (slot, camera) pair per lane; the in-view pairs are then visited wave-uniformly
  const int pairs = T * num_cam;
  for (int base0 = 0; base0 < pairs; base0 += 64) {
    const int pr = base0 + lane;
    bool ok = false;
    float ix = 0.f, iy = 0.f;
    int k = 0;
    if (pr < pairs) {
      const int t = pr / num_cam;
      k = pr - t * num_cam;
      const float* pt = pillars + ((size_t)p * T + t) * pillar_ld;
      const float px = pt[0], py = pt[1], pz = pt[2];
      const float* m = cam + (size_t)(b * num_cam + k) * 20;
      float cx = m[0] * px + m[1] * py + m[2] * pz + m[9];
      float cy = m[3] * px + m[4] * py + m[5] * pz + m[10];
      float cz = m[6] * px + m[7] * py + m[8] * pz + m[11];
      cz = fminf(fmaxf(cz, 1e-5f), 1e5f);
      cx /= cz;
      cy /= cz;
      const float u = m[12] * cx + m[13] * cy + m[14] * cz + m[18];
      const float v = m[15] * cx + m[16] * cy + m[17] * cz + m[19];
      // (u / in_w - 0.5) * 2 then grid_sample's align_corners=False un-normalisation
      const float gx = (u / in_w - 0.5f) * 2.f, gy = (v / in_h - 0.5f) * 2.f;
      ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
      iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
      ok = ix > -1.f && ix < (float)W && iy > -1.f && iy < (float)H;   // else all four taps are outside
    }
    unsigned long long live = __ballot(ok);
    while (live) {
      const int src_lane = __ffsll((long long)live) - 1;
      live &= live - 1;
      const float sx = __shfl(ix, src_lane, 64), sy = __shfl(iy, src_lane, 64);
      const int sk = __shfl(k, src_lane, 64);
      const float fx = floorf(sx), fy = floorf(sy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float lx = sx - fx, ly = sy - fy;
      const float* base = img + (size_t)(b * num_cam + sk) * H * W * C + c0;
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int xx = x0 + (tap & 1), yy = y0 + (tap >> 1);
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        const float wgt = ((tap & 1) ? lx : 1.f - lx) * ((tap >> 1) ? ly : 1.f - ly);
        const float* src = base + ((size_t)yy * W + xx) * C;
        if (CPL == 4) {
          const float4 f = *reinterpret_cast<const float4*>(src);
          acc[0] = fmaf(wgt, f.x, acc[0]); acc[1] = fmaf(wgt, f.y, acc[1]);
          acc[2] = fmaf(wgt, f.z, acc[2]); acc[3] = fmaf(wgt, f.w, acc[3]);
        } else {
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = fmaf(wgt, src[c], acc[c]);
        }
      }
    }
  }
  if constexpr (SPLIT) {   // C == 256, CPL == 4: the pillar's cell is ONE token row, 1 KiB contiguous for the wave
    static_assert(!SPLIT || CPL == 4, "split output: 256 channels, four per lane");
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {acc[0], acc[1], acc[2], acc[3]};
    const h4 hi = __builtin_convertvector(v, h4);                  // the split of isf_f32_to_split, element for element
    const h4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f4), h4);
    const size_t tok = ((size_t)b * bev + y) * bev + x;
    uint2* o2 = reinterpret_cast<uint2*>(out) + split_hi_index(tok, 32, lane >> 1) * 2 + (lane & 1);
    o2[0] = *reinterpret_cast<const uint2*>(&hi);
    o2[8] = *reinterpret_cast<const uint2*>(&lo);                  // the lo piece: 4 x 16 bytes further
    return;
  }
  float* o = out + ((size_t)b * C + c0) * bev * bev + (size_t)y * bev + x;
#pragma unroll
  for (int c = 0; c < CPL; ++c) o[(size_t)c * bev * bev] = acc[c];
}

// 8f #2  Point-to-Grid backward: d loss / d img.  Same walk as the forward (one wave per pillar, projection of every
// (slot, camera) pair, in-view pairs visited wave-uniformly); each bilinear tap scatters weight * grad_canvas[cell]
// into the NHWC gradient map with hardware fp32 atomics (the reference's F.grid_sample backward does the same).  The
// sample positions do not depend on trainable tensors (points and calibration), so there is no gradient towards them.
template <int CPL>
__global__ __launch_bounds__(256) void p2g_backward_kernel(const float* __restrict__ pillars, int pillar_ld, int T,
                                                           const int32_t* __restrict__ coors, int M, int num_cam,
                                                           int H, int W, int C, const float* __restrict__ cam, float in_h,
                                                           float in_w, int bev, const float* __restrict__ grad_out,
                                                           float* __restrict__ grad_img /* [B*cam, H, W, C], zeroed */) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= M) return;
  const int b = coors[4 * p], y = coors[4 * p + 2], x = coors[4 * p + 3];
  const int c0 = lane * CPL;
  float g[CPL];
  const float* go = grad_out + ((size_t)b * C + c0) * bev * bev + (size_t)y * bev + x;
#pragma unroll
  for (int c = 0; c < CPL; ++c) g[c] = go[(size_t)c * bev * bev];
  const int pairs = T * num_cam;
  for (int base0 = 0; base0 < pairs; base0 += 64) {
    const int pr = base0 + lane;
    bool ok = false;
    float ix = 0.f, iy = 0.f;
    int k = 0;
    if (pr < pairs) {
      const int t = pr / num_cam;
      k = pr - t * num_cam;
      const float* pt = pillars + ((size_t)p * T + t) * pillar_ld;
      const float px = pt[0], py = pt[1], pz = pt[2];
      const float* m = cam + (size_t)(b * num_cam + k) * 20;
      float cx = m[0] * px + m[1] * py + m[2] * pz + m[9];
      float cy = m[3] * px + m[4] * py + m[5] * pz + m[10];
      float cz = m[6] * px + m[7] * py + m[8] * pz + m[11];
      cz = fminf(fmaxf(cz, 1e-5f), 1e5f);
      cx /= cz;
      cy /= cz;
      const float u = m[12] * cx + m[13] * cy + m[14] * cz + m[18];
      const float v = m[15] * cx + m[16] * cy + m[17] * cz + m[19];
      const float gx = (u / in_w - 0.5f) * 2.f, gy = (v / in_h - 0.5f) * 2.f;
      ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
      iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
      ok = ix > -1.f && ix < (float)W && iy > -1.f && iy < (float)H;
    }
    unsigned long long live = __ballot(ok);
    while (live) {
      const int src_lane = __ffsll((long long)live) - 1;
      live &= live - 1;
      const float sx = __shfl(ix, src_lane, 64), sy = __shfl(iy, src_lane, 64);
      const int sk = __shfl(k, src_lane, 64);
      const float fx = floorf(sx), fy = floorf(sy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float lx = sx - fx, ly = sy - fy;
      float* base = grad_img + (size_t)(b * num_cam + sk) * H * W * C + c0;
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int xx = x0 + (tap & 1), yy = y0 + (tap >> 1);
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        const float wgt = ((tap & 1) ? lx : 1.f - lx) * ((tap >> 1) ? ly : 1.f - ly);
        float* dst = base + ((size_t)yy * W + xx) * C;
#pragma unroll
        for (int c = 0; c < CPL; ++c) atomicAdd(dst + c, wgt * g[c]);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// A12  sigmoid -> 3x3 local-maximum suppression (1x1 for the listed classes) -> top-k over all classes
// (fusion_encoder.py:1100-1131).  The reference materialises the suppressed map and argsorts all K*H*W
// values per sample.  Here the suppression kernel compacts the surviving local maxima (typically ~1/9 of the
// cells) into a candidate list of 50-bit keys (value bits | reversed index: unique, ties broken by ascending
// flat index) and one workgroup per sample radix-selects the k-th key, collects the k winners and bitonic-sorts
// them.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// One block = 1024 cells (4 per thread): the candidates of a block are appended with ONE atomicAdd on the sample's
// counter -- with 256 cells per block the 1266 blocks of a 10 x 180 x 180 map queued behind each other on that one
// address (32 us for a 1.3 MB read; now 317 atomics).
static constexpr int kNmsCellsPerThread = 4;
__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* __restrict__ hm, int K, int H, int W,
                                                             unsigned pool1_mask, unsigned long long* __restrict__ cand,
                                                             int* __restrict__ cand_count, float* __restrict__ masked) {
  const int b = blockIdx.y;
  const int n = K * H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int cnt_s[kNmsCellsPerThread][4], base_s[kNmsCellsPerThread][4];
  bool keep[kNmsCellsPerThread];
  float hv[kNmsCellsPerThread];
  unsigned long long bal[kNmsCellsPerThread];
#pragma unroll
  for (int q = 0; q < kNmsCellsPerThread; ++q) {
    const int i = (blockIdx.x * kNmsCellsPerThread + q) * 256 + threadIdx.x;
    keep[q] = false;
    hv[q] = 0.f;
    if (i < n) {
      const int x = i % W, y = (i / W) % H, c = i / (W * H);
      const float* plane = hm + ((size_t)b * K + c) * H * W;
      hv[q] = sigmoidf_(plane[y * W + x]);
      if ((pool1_mask >> c) & 1u) {
        keep[q] = true;
      } else if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
        float mx = hv[q];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) mx = fmaxf(mx, sigmoidf_(plane[(y + dy) * W + x + dx]));
        keep[q] = hv[q] == mx;
      }
      if (masked) masked[(size_t)b * n + i] = keep[q] ? hv[q] : 0.f;
      keep[q] = keep[q] && hv[q] > 0.f;
    }
    bal[q] = __ballot(keep[q]);
    if (lane == 0) cnt_s[q][wave] = __popcll(bal[q]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // block-aggregated append
    int tot = 0;
    for (int q = 0; q < kNmsCellsPerThread; ++q)
      for (int w = 0; w < 4; ++w) tot += cnt_s[q][w];
    int base = tot ? atomicAdd(cand_count + b, tot) : 0;
    for (int q = 0; q < kNmsCellsPerThread; ++q)
      for (int w = 0; w < 4; ++w) { base_s[q][w] = base; base += cnt_s[q][w]; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kNmsCellsPerThread; ++q) {
    if (keep[q]) {
      const int i = (blockIdx.x * kNmsCellsPerThread + q) * 256 + threadIdx.x;
      const int pos = base_s[q][wave] + __popcll(bal[q] & ((1ull << lane) - 1));
      cand[(size_t)b * n + pos] = ((unsigned long long)__float_as_uint(hv[q]) << 19) | (unsigned long long)(0x7FFFF - i);
    }
  }
}

// Exact top-k over unique 50-bit keys in two stages (k <= 1024): a single workgroup per sample scanning ~35 k candidates
// five times took 235 us (two to four workgroups on the whole chip, every pass latency-bound on one CU -- 4 % of the
// full fusion forward, profiles/r02_v1_cfg3_kernels.txt).  Stage 1: one workgroup per 4096-candidate chunk selects the
// chunk's top-k (a chunk's losers cannot be in the global top-k) and writes them, zero-padded; stage 2: one workgroup
// per sample selects the top-k of the chunks' winners and writes the indices.  Key 0 = empty slot (real keys are > 0:
// positive score bits).
static constexpr int kTopkChunk = 4096;

// the workgroup's keys c[0..slots) (zeros ignored) -> sel[0..1024) sorted descending (zeros last); returns the number of
// real keys selected (min(k, number of non-zero keys)).  1024 threads.  Barrier count matters here (one workgroup per
// chunk / sample, nothing else to overlap with): the suffix sums over the 1024 radix bins are taken inside each wave
// with shuffles plus one pass over the 16 wave totals (2 barriers per digit instead of 20), and the bitonic sort keeps
// one key per thread in registers -- strides below 64 exchange through shuffles, the ten strides of 64 and more through
// two alternating LDS arrays (10 barriers instead of 55).  The keys are read from memory ONCE (up to four per thread stay
// in registers over the count, the five digit passes and the collection; more slots than that fall back to re-reading).
template <typename Fn>
__device__ __forceinline__ void topk_for_each_key(const unsigned long long* __restrict__ c, int slots, bool in_regs,
                                                  const unsigned long long (&my)[4], Fn fn) {
  if (in_regs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) fn(my[q]);
  } else {
    for (int i = threadIdx.x; i < slots; i += 1024) fn(c[i]);
  }
}

__device__ int topk_block_select(const unsigned long long* __restrict__ c, int slots, int k, unsigned long long* sel,
                                 unsigned long long* sel2, int* hist, int* wave_tot, unsigned long long* s_prefix,
                                 int* s_need, int* s_cnt) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool in_regs = slots <= 4 * 1024;   // block-uniform
  unsigned long long my[4] = {0ull, 0ull, 0ull, 0ull};
  if (in_regs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = t + q * 1024;
      if (i < slots) my[q] = c[i];
    }
  }
  if (t == 0) *s_cnt = 0;
  __syncthreads();
  int mine = 0;
  topk_for_each_key(c, slots, in_regs, my, [&](unsigned long long key) { mine += key != 0ull; });
  if (mine) atomicAdd(s_cnt, mine);
  __syncthreads();
  const int nc = *s_cnt;
  __syncthreads();
  unsigned long long thr = 1;   // every real key
  if (nc > k) {
    if (t == 0) { *s_prefix = 0; *s_need = k; }
    // 50-bit keys, 5 digits of 10 bits from the top
    for (int shift = 40; shift >= 0; shift -= 10) {
      hist[t] = 0;
      __syncthreads();   // also publishes s_prefix / s_need of the previous digit
      const unsigned long long prefix = *s_prefix;
      const int need = *s_need;
      topk_for_each_key(c, slots, in_regs, my, [&](unsigned long long key) {
        if (key != 0ull && (shift == 40 || (key >> (shift + 10)) == prefix))
          atomicAdd(&hist[(int)((key >> shift) & 1023)], 1);
      });
      __syncthreads();
      // inclusive suffix sum over the bins (bin 1023 first): inside the wave by shuffles, then the higher waves' totals
      const int v = hist[t];
      int sfx = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(sfx, d, 64);
        if (lane + d < 64) sfx += o;
      }
      if (lane == 0) wave_tot[wave] = sfx;
      __syncthreads();
      int above = 0;
      for (int u = wave + 1; u < 16; ++u) above += wave_tot[u];
      const int incl = sfx + above, excl = incl - v;
      if (excl < need && need <= incl) {   // exactly one bin
        *s_prefix = (prefix << 10) | (unsigned long long)t;
        *s_need = need - excl;
      }
      __syncthreads();   // wave_tot / s_prefix may be rewritten by the next digit
    }
    thr = *s_prefix;   // the k-th largest key (keys are unique)
  }
  if (t == 0) *s_cnt = 0;
  sel[t] = 0;
  __syncthreads();
  topk_for_each_key(c, slots, in_regs, my, [&](unsigned long long key) {
    if (key >= thr) {
      const int pos = atomicAdd(s_cnt, 1);
      if (pos < 1024) sel[pos] = key;
    }
  });
  __syncthreads();
  // bitonic sort, descending (unused slots hold key 0 = smallest), one key per thread
  unsigned long long key = sel[t];
  unsigned long long* bufs[2] = {sel2, sel};
  int which = 0;
  for (int size = 2; size <= 1024; size <<= 1) {
    const bool desc = (t & size) == 0;
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      unsigned long long other;
      if (stride >= 64) {
        unsigned long long* buf = bufs[which];
        which ^= 1;
        buf[t] = key;
        __syncthreads();
        other = buf[t ^ stride];
      } else {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)key, stride, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), stride, 64);
        other = ((unsigned long long)hi << 32) | lo;
      }
      const bool lower = (t & stride) == 0;
      const bool keep_max = lower == desc;
      key = keep_max ? (key > other ? key : other) : (key < other ? key : other);
    }
  }
  __syncthreads();
  sel[t] = key;
  __syncthreads();
  return nc < k ? nc : k;
}

// stage 1: grid (chunks, B); partial [B][chunks][k]
__global__ __launch_bounds__(1024) void topk_partial_kernel(const unsigned long long* __restrict__ cand,
                                                            const int* __restrict__ cand_count, int n, int k,
                                                            unsigned long long* __restrict__ partial) {
  __shared__ int hist[1024];
  __shared__ int wave_tot[16];
  __shared__ unsigned long long sel[1024], sel2[1024];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_need, s_cnt;
  const int b = blockIdx.y, j = blockIdx.x, t = threadIdx.x;
  const int lo = j * kTopkChunk;
  int cnt = cand_count[b] - lo;
  cnt = cnt < 0 ? 0 : (cnt > kTopkChunk ? kTopkChunk : cnt);
  if (cnt == 0) {   // the candidates are compacted to the front: most chunks are empty
    if (t < k) partial[((size_t)b * gridDim.x + j) * k + t] = 0ull;
    return;
  }
  (void)topk_block_select(cand + (size_t)b * n + lo, cnt, k, sel, sel2, hist, wave_tot, &s_prefix, &s_need, &s_cnt);
  if (t < k) partial[((size_t)b * gridDim.x + j) * k + t] = sel[t];
}

// stage 2: one workgroup per sample over the chunks' winners
__global__ __launch_bounds__(1024) void topk_final_kernel(const unsigned long long* __restrict__ partial, int slots,
                                                          int* __restrict__ cand_count /* read, then zeroed again */, int n,
                                                          int HW, int k,
                                                          int32_t* __restrict__ top_mod,
                                                          int32_t* __restrict__ top_raw) {
  __shared__ int hist[1024];
  __shared__ int wave_tot[16];
  __shared__ unsigned long long sel[1024], sel2[1024];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_need, s_cnt;
  const int b = blockIdx.x, t = threadIdx.x;
  // the candidates are compacted to the front: only the first ceil(count / chunk) chunks have winners (typically 9 of 80)
  const int live = min(slots, ((cand_count[b] + kTopkChunk - 1) / kTopkChunk) * k);
  __syncthreads();
  if (t == 0) cand_count[b] = 0;   // the workspace's zero counters go back to zero (Arena::zero_pool): no memset next time
  const int kk = topk_block_select(partial + (size_t)b * slots, live, k, sel, sel2, hist, wave_tot, &s_prefix, &s_need, &s_cnt);
  if (t < kk) {
    const int idx = 0x7FFFF - (int)(sel[t] & 0x7FFFF);
    top_raw[(size_t)b * k + t] = idx;
    top_mod[(size_t)b * k + t] = idx % HW;
  }
  // fewer than k positive local maxima: the reference's argsort continues into the zeros (tie order
  // unspecified there); continue with the smallest flat indices that are not already selected.
  if (kk < k && t == 0) {
    int filled = kk, idx = 0;
    while (filled < k && idx < n) {
      bool used = false;
      for (int j = 0; j < kk; ++j) used |= (0x7FFFF - (int)(sel[j] & 0x7FFFF)) == idx;
      if (!used) {
        top_raw[(size_t)b * k + filled] = idx;
        top_mod[(size_t)b * k + filled] = idx % HW;
        ++filled;
      }
      ++idx;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// A13  multi-scale deformable attention forward, single level (ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84,
// 237-299 in the reference) with the softmax over the sampling points and the location arithmetic of
// MSDeformAttn.forward (fusion_encoder.py:585-596) folded in.
// value [B, H*W, heads*D]; offsets [B*Q, heads*P*2]; logits [B*Q, heads*P]; ref [B*Q, 2] (x, y in [0,1]);
// out [B*Q, heads*D].  One thread per (b, q, head, d); the 16 lanes of a head share the tap geometry.
template <int D, int P>
__global__ __launch_bounds__(256) void msda_kernel(const float* __restrict__ value, const float* __restrict__ offsets,
                                                   const float* __restrict__ logits, const float* __restrict__ ref,
                                                   int B, int Q, int heads, int H, int W, float* __restrict__ out) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Q * heads * D;
  if (tid >= total) return;
  const int dch = (int)(tid % D);
  const int h = (int)((tid / D) % heads);
  const long long bq = tid / ((long long)D * heads);
  const int b = (int)(bq / Q);
  const float* lg = logits + bq * heads * P + h * P;
  float mx = -INFINITY;
#pragma unroll
  for (int p = 0; p < P; ++p) mx = fmaxf(mx, lg[p]);
  float e[P], sum = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) { e[p] = expf(lg[p] - mx); sum += e[p]; }
  const float rx = ref[bq * 2], ry = ref[bq * 2 + 1];
  const float* off = offsets + bq * heads * P * 2 + h * P * 2;
  const float* vb = value + (size_t)b * H * W * heads * D + h * D + dch;
  const size_t vstride = (size_t)heads * D;
  float acc = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const float lx = rx + off[2 * p] / (float)W, ly = ry + off[2 * p + 1] / (float)H;
    const float w_im = lx * (float)W - 0.5f, h_im = ly * (float)H - 0.5f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float fh = floorf(h_im), fw = floorf(w_im);
      const int h0 = (int)fh, w0 = (int)fw;
      const float lh = h_im - fh, lw = w_im - fw;
      float s = 0.f;
      if (h0 >= 0 && w0 >= 0) s += (1.f - lh) * (1.f - lw) * vb[((size_t)h0 * W + w0) * vstride];
      if (h0 >= 0 && w0 + 1 <= W - 1) s += (1.f - lh) * lw * vb[((size_t)h0 * W + w0 + 1) * vstride];
      if (h0 + 1 <= H - 1 && w0 >= 0) s += lh * (1.f - lw) * vb[((size_t)(h0 + 1) * W + w0) * vstride];
      if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) s += lh * lw * vb[((size_t)(h0 + 1) * W + w0 + 1) * vstride];
      acc += (e[p] / sum) * s;
    }
  }
  out[tid] = acc;
}

// ----------------------------------------------------------------------------------------------------------------
// 8f #2  multi-scale deformable attention backward, single level (ms_deform_im2col_cuda.cuh:301-920 in the reference:
// ms_deformable_col2im_* with their shared-memory / block reductions), through the softmax over the sampling points
// and the location arithmetic the forward folds in.  A 16-lane group owns one (b, q, head): the lanes are the D = 16
// channels, so the tap geometry is computed once per group and the channel reductions for d(attention) and d(location)
// are four xor-shuffles inside the group.  grad_value is accumulated with hardware fp32 atomics (the reference does
// the same, atomicAdd at :128-150), everything else is written once.
template <int D, int P>
__global__ __launch_bounds__(256) void msda_backward_kernel(
    const float* __restrict__ value, const float* __restrict__ offsets, const float* __restrict__ logits,
    const float* __restrict__ ref, const float* __restrict__ grad_out, int B, int Q, int heads, int H, int W,
    float* __restrict__ grad_value, float* __restrict__ grad_offsets, float* __restrict__ grad_logits) {
  static_assert(D == 16 && P == 16, "one 16-lane group per head: lanes = channels, and lane p stores point p");
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Q * heads * D;
  const bool live = tid < total;          // dead lanes of the last group keep shuffling with zeros
  const long long t = live ? tid : total - 1;
  const int dch = (int)(t % D);
  const int h = (int)((t / D) % heads);
  const long long bq = t / ((long long)D * heads);
  const int b = (int)(bq / Q);
  const float* lg = logits + bq * heads * P + h * P;
  float mx = -INFINITY;
#pragma unroll
  for (int p = 0; p < P; ++p) mx = fmaxf(mx, lg[p]);
  float a[P], sum = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) { a[p] = expf(lg[p] - mx); sum += a[p]; }
  const float rx = ref[bq * 2], ry = ref[bq * 2 + 1];
  const float* off = offsets + bq * heads * P * 2 + h * P * 2;
  const size_t vstride = (size_t)heads * D;
  const size_t vbase = (size_t)b * H * W * vstride + (size_t)h * D + dch;
  const float go = live ? grad_out[t] : 0.f;
  float ga[P];                     // d loss / d a[p], reduced over the channels
  float my_gx = 0.f, my_gy = 0.f;  // lane p keeps d loss / d offset of point p
#pragma unroll
  for (int p = 0; p < P; ++p) {
    a[p] /= sum;
    const float lx = rx + off[2 * p] / (float)W, ly = ry + off[2 * p + 1] / (float)H;
    const float w_im = lx * (float)W - 0.5f, h_im = ly * (float)H - 0.5f;
    float s = 0.f, dsw = 0.f, dsh = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float fh = floorf(h_im), fw = floorf(w_im);
      const int h0 = (int)fh, w0 = (int)fw;
      const float lh = h_im - fh, lw = w_im - fw;
      const float gv = a[p] * go;
      float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
      if (h0 >= 0 && w0 >= 0) {
        const size_t at = vbase + ((size_t)h0 * W + w0) * vstride;
        v00 = value[at];
        if (live) atomicAdd(grad_value + at, (1.f - lh) * (1.f - lw) * gv);
      }
      if (h0 >= 0 && w0 + 1 <= W - 1) {
        const size_t at = vbase + ((size_t)h0 * W + w0 + 1) * vstride;
        v01 = value[at];
        if (live) atomicAdd(grad_value + at, (1.f - lh) * lw * gv);
      }
      if (h0 + 1 <= H - 1 && w0 >= 0) {
        const size_t at = vbase + ((size_t)(h0 + 1) * W + w0) * vstride;
        v10 = value[at];
        if (live) atomicAdd(grad_value + at, lh * (1.f - lw) * gv);
      }
      if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) {
        const size_t at = vbase + ((size_t)(h0 + 1) * W + w0 + 1) * vstride;
        v11 = value[at];
        if (live) atomicAdd(grad_value + at, lh * lw * gv);
      }
      s = (1.f - lh) * (1.f - lw) * v00 + (1.f - lh) * lw * v01 + lh * (1.f - lw) * v10 + lh * lw * v11;
      dsw = (1.f - lh) * (v01 - v00) + lh * (v11 - v10);      // d s / d w_im
      dsh = (1.f - lw) * (v10 - v00) + lw * (v11 - v01);      // d s / d h_im
    }
    float r0 = go * s, r1 = a[p] * go * dsw, r2 = a[p] * go * dsh;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      r0 += __shfl_xor(r0, m, 16);
      r1 += __shfl_xor(r1, m, 16);
      r2 += __shfl_xor(r2, m, 16);
    }
    ga[p] = r0;
    // w_im = (rx + off_x / W) * W - 0.5  =>  d w_im / d off_x = 1 (likewise y)
    if (p == dch) { my_gx = r1; my_gy = r2; }
  }
  // softmax backward: d logit[p] = a[p] * (ga[p] - sum_j a[j] ga[j])
  float dot = 0.f, mine_a = 0.f, mine_ga = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    dot += a[p] * ga[p];
    if (p == dch) { mine_a = a[p]; mine_ga = ga[p]; }
  }
  if (live) {
    grad_logits[bq * heads * P + h * P + dch] = mine_a * (mine_ga - dot);
    float* go_off = grad_offsets + bq * heads * P * 2 + h * P * 2 + 2 * dch;
    go_off[0] = my_gx;
    go_off[1] = my_gy;
  }
}

}  // namespace isf

extern "C" {

int isf_p2g_forward_split(const float* pillars, int pillar_ld, int slots, const int32_t* pillar_coors, int num_pillars,
                          const float* img_nhwc, int batch_size, int num_cam, int feat_h, int feat_w, int channels,
                          const float* cam_params, int input_h, int input_w, int bev_size, void* out_split,
                          isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_pillars >= 0 && batch_size > 0 && slots > 0 && bev_size > 0, ISF_ERR_ARG, "p2g_split: bad sizes");
  ISF_REQUIRE(out_split, ISF_ERR_ARG, "p2g_split: null output");
  ISF_REQUIRE(channels == 256, ISF_ERR_UNSUPPORTED, "p2g_split: %d channels (one 256-channel token matrix)", channels);
  hipStream_t st = as_stream(stream);
  ISF_HIP_TRY(hipMemsetAsync(out_split, 0, (size_t)batch_size * channels * bev_size * bev_size * sizeof(float), st));
  if (num_pillars == 0) return ISF_OK;
  ISF_REQUIRE(pillars && pillar_coors && img_nhwc && cam_params && pillar_ld >= 3, ISF_ERR_ARG, "p2g_split: null pointer");
  hipLaunchKernelGGL((p2g_kernel<4, true>), dim3(ceil_div(num_pillars, 4)), dim3(256), 0, st, pillars, pillar_ld, slots,
                     pillar_coors, num_pillars, img_nhwc, num_cam, feat_h, feat_w, channels, cam_params, (float)input_h,
                     (float)input_w, bev_size, reinterpret_cast<float*>(out_split));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_p2g_forward(const float* pillars, int pillar_ld, int slots, const int32_t* pillar_coors, int num_pillars,
                    const float* img_nhwc, int batch_size, int num_cam, int feat_h, int feat_w, int channels,
                    const float* cam_params, int input_h, int input_w, int bev_size, float* out,
                    isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_pillars >= 0 && batch_size > 0 && slots > 0 && bev_size > 0, ISF_ERR_ARG, "p2g: bad sizes");
  ISF_REQUIRE(out, ISF_ERR_ARG, "p2g: null output");
  hipStream_t st = as_stream(stream);
  ISF_HIP_TRY(hipMemsetAsync(out, 0, (size_t)batch_size * channels * bev_size * bev_size * sizeof(float), st));
  if (num_pillars == 0) return ISF_OK;
  ISF_REQUIRE(pillars && pillar_coors && img_nhwc && cam_params, ISF_ERR_ARG, "p2g: null pointer");
  ISF_REQUIRE(channels % 64 == 0 && channels <= 512 && pillar_ld >= 3, ISF_ERR_UNSUPPORTED,
              "p2g: channels %d (need %%64 == 0, <= 512)", channels);
  const dim3 grid(ceil_div(num_pillars, 4)), block(256);
#define ISF_P2G(CPL)                                                                                              \
  hipLaunchKernelGGL((p2g_kernel<CPL>), grid, block, 0, st, pillars, pillar_ld, slots, pillar_coors, num_pillars, \
                     img_nhwc, num_cam, feat_h, feat_w, channels, cam_params, (float)input_h, (float)input_w,      \
                     bev_size, out)
  switch (channels / 64) {
    case 1: ISF_P2G(1); break;
    case 2: ISF_P2G(2); break;
    case 4: ISF_P2G(4); break;
    case 8: ISF_P2G(8); break;
    default: set_error("p2g: channels %d not built (64, 128, 256, 512)", channels); return ISF_ERR_UNSUPPORTED;
  }
#undef ISF_P2G
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_p2g_backward(const float* pillars, int pillar_ld, int slots, const int32_t* pillar_coors, int num_pillars,
                     int batch_size, int num_cam, int feat_h, int feat_w, int channels, const float* cam_params,
                     int input_h, int input_w, int bev_size, const float* grad_out, float* grad_img_nhwc,
                     isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_pillars >= 0 && batch_size > 0 && num_cam > 0 && feat_h > 0 && feat_w > 0 && bev_size > 0 &&
                  grad_img_nhwc, ISF_ERR_ARG, "p2g_backward: bad arguments");
  hipStream_t st = as_stream(stream);
  ISF_HIP_TRY(hipMemsetAsync(grad_img_nhwc, 0, sizeof(float) * (size_t)batch_size * num_cam * feat_h * feat_w * channels, st));
  if (num_pillars == 0) return ISF_OK;
  ISF_REQUIRE(pillars && pillar_coors && cam_params && grad_out, ISF_ERR_ARG, "p2g_backward: null pointer");
  ISF_REQUIRE(channels % 64 == 0 && channels <= 512 && pillar_ld >= 3, ISF_ERR_UNSUPPORTED,
              "p2g_backward: channels %d (need %%64 == 0, <= 512)", channels);
  const dim3 grid(ceil_div(num_pillars, 4)), block(256);
#define ISF_P2GB(CPL)                                                                                                     \
  hipLaunchKernelGGL((p2g_backward_kernel<CPL>), grid, block, 0, st, pillars, pillar_ld, slots, pillar_coors, num_pillars, \
                     num_cam, feat_h, feat_w, channels, cam_params, (float)input_h, (float)input_w, bev_size, grad_out,    \
                     grad_img_nhwc)
  switch (channels / 64) {
    case 1: ISF_P2GB(1); break;
    case 2: ISF_P2GB(2); break;
    case 4: ISF_P2GB(4); break;
    case 8: ISF_P2GB(8); break;
    default: set_error("p2g_backward: channels %d not built (64, 128, 256, 512)", channels); return ISF_ERR_UNSUPPORTED;
  }
#undef ISF_P2GB
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_instance_topk(const float* heatmap, int batch_size, int num_classes, int height, int width, int k,
                      unsigned pool1_class_mask, int32_t* top_index, int32_t* top_index_raw, float* masked_heatmap,
                      isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_classes > 0 && num_classes <= 32 && height >= 3 && width >= 3, ISF_ERR_ARG,
              "instance_topk: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(heatmap && top_index && top_index_raw, ISF_ERR_ARG, "instance_topk: null pointer");
  const long long n = (long long)num_classes * height * width;
  ISF_REQUIRE(k > 0 && k <= 1024 && k <= n && n <= 0x7FFFF, ISF_ERR_UNSUPPORTED,
              "instance_topk: k %d (<= 1024) over %lld cells (<= 524287)", k, n);
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  unsigned long long* cand = nullptr;
  int* count = nullptr;
  ISF_TRY(a.alloc_n(&cand, (size_t)batch_size * n));
  if (batch_size <= kZeroInts) {
    ISF_TRY(zero_ints(a, &count));     // zero on entry, zeroed again by topk_final_kernel: no memset launch
  } else {
    ISF_TRY(a.alloc_n(&count, (size_t)batch_size + 16));
    ISF_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * batch_size, st));
  }
  hipLaunchKernelGGL(nms_candidates_kernel, dim3(ceil_div(n, 256 * kNmsCellsPerThread), batch_size), dim3(256), 0, st, heatmap,
                     num_classes, height, width, pool1_class_mask, cand, count, masked_heatmap);
  ISF_LAUNCH_CHECK();
  const int chunks = ceil_div(n, kTopkChunk);   // candidates are compacted to the front: the tail chunks are empty
  unsigned long long* partial = nullptr;
  ISF_TRY(a.alloc_n(&partial, (size_t)batch_size * chunks * k));
  hipLaunchKernelGGL(topk_partial_kernel, dim3(chunks, batch_size), dim3(1024), 0, st, cand, count, (int)n, k, partial);
  hipLaunchKernelGGL(topk_final_kernel, dim3(batch_size), dim3(1024), 0, st, partial, chunks * k, count, (int)n,
                     height * width, k, top_index, top_index_raw);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_msda_forward(const float* value, const float* sampling_offsets, const float* attention_logits,
                     const float* reference_points, int batch_size, int num_queries, int num_heads, int head_dim,
                     int num_points, int height, int width, float* out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_queries >= 0, ISF_ERR_ARG, "msda: bad sizes");
  if (batch_size == 0 || num_queries == 0) return ISF_OK;
  ISF_REQUIRE(value && sampling_offsets && attention_logits && reference_points && out, ISF_ERR_ARG,
              "msda: null pointer");
  ISF_REQUIRE(head_dim == 16 && num_points == 16, ISF_ERR_UNSUPPORTED,
              "msda: built for head_dim 16, 16 points, one level (got %d, %d)", head_dim, num_points);
  const long long total = (long long)batch_size * num_queries * num_heads * head_dim;
  hipLaunchKernelGGL((msda_kernel<16, 16>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), value,
                     sampling_offsets, attention_logits, reference_points, batch_size, num_queries, num_heads, height,
                     width, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_msda_backward(const float* value, const float* sampling_offsets, const float* attention_logits,
                      const float* reference_points, const float* grad_out, int batch_size, int num_queries,
                      int num_heads, int head_dim, int num_points, int height, int width, float* grad_value,
                      float* grad_offsets, float* grad_logits, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_queries >= 0 && height > 0 && width > 0, ISF_ERR_ARG, "msda_backward: bad sizes");
  ISF_REQUIRE(grad_value, ISF_ERR_ARG, "msda_backward: null grad_value");
  hipStream_t st = as_stream(stream);
  ISF_HIP_TRY(hipMemsetAsync(grad_value, 0,
                             sizeof(float) * (size_t)batch_size * height * width * num_heads * head_dim, st));
  if (batch_size == 0 || num_queries == 0) return ISF_OK;
  ISF_REQUIRE(value && sampling_offsets && attention_logits && reference_points && grad_out && grad_offsets &&
                  grad_logits, ISF_ERR_ARG, "msda_backward: null pointer");
  ISF_REQUIRE(head_dim == 16 && num_points == 16, ISF_ERR_UNSUPPORTED,
              "msda_backward: built for head_dim 16, 16 points, one level (got %d, %d)", head_dim, num_points);
  const long long total = (long long)batch_size * num_queries * num_heads * head_dim;
  hipLaunchKernelGGL((msda_backward_kernel<16, 16>), dim3(ceil_div(total, 256)), dim3(256), 0, st, value,
                     sampling_offsets, attention_logits, reference_points, grad_out, batch_size, num_queries, num_heads,
                     height, width, grad_value, grad_offsets, grad_logits);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
