// mfma_step.hip -- what bounds the multiply section of spconv_f16x3_kernel when NO global memory traffic is left?
// (round-2 knock-out: with gathers and weight DMA removed the 256 -> 256 layer still needs 1.10 of 1.41 ms for 0.53 ms of MFMA
// work.)  A workgroup = 4 waves; per step a wave reads 16 B fragments (ds_read_b128, 16 KiB shared stage) and issues
// 48 v_mfma_f32_16x16x32_f16 on 16 accumulators (2 row groups x 8 column tiles, 3 products each), one barrier per step.
// Variants (all with the same MFMA count):
//   chain     a0 a0 a0 a1 a1 a1 ...    three dependent products back to back per accumulator (what hipcc emits)
//   inter2    (rg0 nt) (rg1 nt) x 3    the two row groups' accumulators alternate (dependent distance 2)
//   inter16   product-major sweeps     every accumulator once per sweep (dependent distance 16)
//   nobar     chain without the per-step barrier;   nolds  chain with the B fragments kept in registers
//   front     chain, all 16 fragments read before the first MFMA
// Prints cycles per step per wave slot at 1, 2 and 3 workgroups per CU (dynamic LDS pads the occupancy).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_step.hip -o /tmp/mfma_step && /tmp/mfma_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { CHAIN, INTER2, INTER16, NOBAR, NOLDS, FRONT, NV };
static const char* kName[NV] = {"chain", "inter2", "inter16", "nobar", "nolds", "front"};

__device__ __forceinline__ void mfma(f32x4& c, const h8& a, const h8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int V>
__global__ __launch_bounds__(256) void probe(int steps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const u32x4* bst = reinterpret_cast<const u32x4*>(smem);   // [16 fragments][64 lanes] = 16 KiB
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 64; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  f32x4 acc[2][8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  h8 ah[2], al[2];
  for (int r = 0; r < 2; ++r) {
    for (int j = 0; j < 8; ++j) { ah[r][j] = (_Float16)(0.001f * (lane + j + r)); al[r][j] = (_Float16)(1e-6f * lane); }
  }
  h8 breg[16];
  if (V == NOLDS) {
#pragma unroll
    for (int f = 0; f < 16; ++f) { const u32x4 u = bst[f * 64 + lane]; breg[f] = *reinterpret_cast<const h8*>(&u); }
  }
  for (int s = 0; s < steps; ++s) {
    if (V != NOBAR) __syncthreads();
    if (V == CHAIN || V == NOBAR || V == NOLDS) {
      u32x4 bh_n, bl_n;
      if (V != NOLDS) { bh_n = bst[lane]; bl_n = bst[64 + lane]; }
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        h8 bh, bl;
        if (V == NOLDS) { bh = breg[2 * n]; bl = breg[2 * n + 1]; }
        else {
          bh = *reinterpret_cast<const h8*>(&bh_n); bl = *reinterpret_cast<const h8*>(&bl_n);
          if (n + 1 < 8) { bh_n = bst[(2 * n + 2) * 64 + lane]; bl_n = bst[(2 * n + 3) * 64 + lane]; }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          mfma(acc[r][n], al[r], bh);
          mfma(acc[r][n], ah[r], bl);
          mfma(acc[r][n], ah[r], bh);
        }
      }
    } else if (V == INTER2) {
      u32x4 bh_n = bst[lane], bl_n = bst[64 + lane];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const h8 bh = *reinterpret_cast<const h8*>(&bh_n), bl = *reinterpret_cast<const h8*>(&bl_n);
        if (n + 1 < 8) { bh_n = bst[(2 * n + 2) * 64 + lane]; bl_n = bst[(2 * n + 3) * 64 + lane]; }
        mfma(acc[0][n], al[0], bh); mfma(acc[1][n], al[1], bh);
        mfma(acc[0][n], ah[0], bl); mfma(acc[1][n], ah[1], bl);
        mfma(acc[0][n], ah[0], bh); mfma(acc[1][n], ah[1], bh);
      }
    } else {   // INTER16 / FRONT: all fragments in registers first
      h8 b[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) { const u32x4 u = bst[f * 64 + lane]; b[f] = *reinterpret_cast<const h8*>(&u); }
      if (V == FRONT) {
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            mfma(acc[r][n], al[r], b[2 * n]);
            mfma(acc[r][n], ah[r], b[2 * n + 1]);
            mfma(acc[r][n], ah[r], b[2 * n]);
          }
      } else {
#pragma unroll
        for (int n = 0; n < 8; ++n) { mfma(acc[0][n], al[0], b[2 * n]); mfma(acc[1][n], al[1], b[2 * n]); }
#pragma unroll
        for (int n = 0; n < 8; ++n) { mfma(acc[0][n], ah[0], b[2 * n + 1]); mfma(acc[1][n], ah[1], b[2 * n + 1]); }
#pragma unroll
        for (int n = 0; n < 8; ++n) { mfma(acc[0][n], ah[0], b[2 * n]); mfma(acc[1][n], ah[1], b[2 * n]); }
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int n = 0; n < 8; ++n) t += acc[r][n][0] + acc[r][n][3];
  if (t == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int V>
static void run(int wgs_per_cu, int cus) {
  const int steps = 2000;
  const size_t lds = wgs_per_cu == 1 ? 96 * 1024 : wgs_per_cu == 2 ? 64 * 1024 : 48 * 1024;   // 160 KiB / lds = occupancy
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float* out;
  hipMalloc(&out, (size_t)cus * wgs_per_cu * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<V><<<cus * wgs_per_cu, 256, lds>>>(50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<V><<<cus * wgs_per_cu, 256, lds>>>(steps, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: wgs_per_cu waves, each 48 MFMAs per step
  const double ns_per_step = ms * 1e6 / steps;
  const double mfma_ns = ns_per_step / (48.0 * wgs_per_cu);
  printf("  %-8s %d WG/CU: %8.1f ns per step (all resident waves advance one step) = %6.2f ns per MFMA per SIMD\n", kName[V], wgs_per_cu,
         ns_per_step, mfma_ns);
  hipFree(out);
}

int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
  printf("CUs %d, clock %d kHz: 16 cycles per 16x16x32 f16 MFMA = %.2f ns at that clock\n", cus, clk, 16.0 * 1e6 / clk);
  for (int w = 1; w <= 3; ++w) {
    run<CHAIN>(w, cus); run<INTER2>(w, cus); run<INTER16>(w, cus); run<NOBAR>(w, cus); run<NOLDS>(w, cus); run<FRONT>(w, cus);
  }
  return 0;
}
