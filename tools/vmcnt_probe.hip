// vmcnt_probe.hip -- does `s_waitcnt vmcnt(N)` order LDS-DMA loads (global_load_lds_dwordx4) and ordinary register
// loads (global_load_dwordx4) of ONE wave strictly in issue order on gfx950?
//
// Why: the sparse-conv kernel (is-fusion_amd/csrc/isf_spconv16.hip) prefetches weights by LDS-DMA and activation rows
// into registers ONE step ahead and waits with vmcnt(0).  One step of MFMAs (~800-2300 cycles) barely covers an L2
// hit and not an HBM miss; a 3-stage ring with counted waits (vmcnt(N) = "everything but the youngest N has landed")
// would double the latency budget.  Round 1's attempt at that produced NaNs at full size only.
// Before re-trying, this probe settles the hardware question in isolation: an OLD, slow LDS-DMA (HBM miss) followed
// by a YOUNG, fast register load (cache hit); wait vmcnt(1); read the LDS words the DMA should have written.
//   mode 0: DMA(cold) ; REG(hot) ; vmcnt(1) ; check LDS
//   mode 1: DMA(cold) ; REG(hot) ; DMA(cold') ; REG(hot) ; vmcnt(2) ; check the first DMA's LDS
//   mode 2: REG(cold) ; DMA(hot) ; vmcnt(1) ; check the register value          (the reverse pairing)
// Zero mismatches in all modes over ~10^8 checks => in-order completion holds and the NaNs were a kernel bug.
//
//   hipcc --offload-arch=gfx950 -O2 tools/vmcnt_probe.hip -o gpurun_out/vmcnt_probe && gpurun_out/vmcnt_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base_bytes) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base_bytes)
      : "memory");
}

__device__ __forceinline__ uint4 reg_load16(const void* gsrc) {
  uint4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(gsrc) : "memory");
  return r;
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

__host__ __device__ inline uint4 pattern(uint64_t i) {
  const uint32_t a = (uint32_t)i, b = (uint32_t)(i >> 32);
  return make_uint4(a * 2654435761u + 1u, a ^ 0x5a5a5a5au, ~a + b, a * 40503u + b);
}

__global__ void fill(uint4* p, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = pattern(i);
}

__device__ __forceinline__ bool same(uint4 a, uint4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ cold, uint64_t n_cold,
                                             const uint4* __restrict__ hot, int iters, int mode,
                                             unsigned long long* __restrict__ bad, unsigned* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) uint4 buf[2][4][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base0 = __builtin_amdgcn_readfirstlane(lds_addr(&buf[0][wave][0]));
  const unsigned base1 = __builtin_amdgcn_readfirstlane(lds_addr(&buf[1][wave][0]));
  unsigned acc = 0;
  unsigned long long mism = 0;
  uint64_t pos = ((uint64_t)blockIdx.x * 4 + wave) * 7919u;
  for (int it = 0; it < iters; ++it) {
    // a different, far-away 1 KiB line group every iteration: misses every cache level
    pos = (pos * 6364136223846793005ull + 1442695040888963407ull);
    const uint64_t i0 = (pos % (n_cold / 64)) * 64 + lane;
    const uint64_t i1 = ((pos >> 17) % (n_cold / 64)) * 64 + lane;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (mode == 0) {
      glds16(cold + i0, base0);
      const uint4 r = reg_load16(hot + lane);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      const uint4 got = buf[0][wave][lane];
      if (!same(got, pattern(i0))) ++mism;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += r.x;
    } else if (mode == 1) {
      glds16(cold + i0, base0);
      const uint4 r0 = reg_load16(hot + lane);
      glds16(cold + i1, base1);
      const uint4 r1 = reg_load16(hot + 64 + lane);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      const uint4 got = buf[0][wave][lane];
      if (!same(got, pattern(i0))) ++mism;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint4 got1 = buf[1][wave][lane];
      if (!same(got1, pattern(i1))) ++mism;
      acc += r0.x + r1.y;
    } else {
      const uint4 r = reg_load16(cold + i0);
      glds16(hot + lane, base0);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      uint4 rr = r;   // use the register only after the counted wait
      asm volatile("" : "+v"(rr.x), "+v"(rr.y), "+v"(rr.z), "+v"(rr.w));
      if (!same(rr, pattern(i0))) ++mism;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += buf[0][wave][lane].x;
    }
  }
  if (mism) atomicAdd(bad, mism);
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const uint64_t n_cold = (1ull << 30) / sizeof(uint4) * 2;   // 2 GiB
  uint4 *cold, *hot;
  unsigned long long* bad;
  unsigned* sink;
  CHECK(hipMalloc(&cold, n_cold * sizeof(uint4)));
  CHECK(hipMalloc(&hot, 128 * sizeof(uint4)));
  CHECK(hipMalloc(&bad, 3 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, sizeof(unsigned)));
  CHECK(hipMemset(bad, 0, 3 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, n_cold);
  std::vector<uint4> h(128);
  for (int i = 0; i < 128; ++i) h[i] = pattern(1000 + i);
  CHECK(hipMemcpy(hot, h.data(), 128 * sizeof(uint4), hipMemcpyHostToDevice));
  CHECK(hipDeviceSynchronize());
  const int iters = 2000, grid = 2048;
  unsigned long long res[3];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, cold, n_cold, hot, iters, mode, bad + mode, sink);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipMemcpy(res, bad, sizeof(res), hipMemcpyDeviceToHost));
  const double checks = (double)grid * 256 * iters;
  printf("{\"probe\": \"vmcnt order of LDS-DMA vs register loads\", \"checks_per_mode\": %.3g, "
         "\"mismatch_mode0_dma_then_reg\": %llu, \"mismatch_mode1_two_pairs\": %llu, "
         "\"mismatch_mode2_reg_then_dma\": %llu}\n", checks, res[0], res[1], res[2]);
  return (res[0] | res[1] | res[2]) ? 1 : 0;
}
