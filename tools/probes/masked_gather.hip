// masked_gather.hip -- what does a 16-byte-per-lane gather cost the CU's vector-memory path?
// The sparse-conv A gathers read, per wave instruction, 16 rows x 64 B (lane = (row = lane & 15, k-group = lane >> 4)
// reads 16 B at row * 128 + kg * 16); at the sparse levels only ~24 % of the rows are live.  Questions:
//   (1) does a masked-off lane cost anything?            full / q25 (random lanes) / r25 (4 of 16 rows) / h50
//   (2) what is the rate of the gather pattern against a fully coalesced 1-KiB load?       full vs seq vs adj
//       adj = 16 consecutive 128-byte rows, 64 B of each (the split format today); adj64 = the same rows if hi and lo
//       lived in separate planes of 64-byte records (one contiguous KiB)
//   (3) is the cost per 64-byte piece or per 128-byte line?   pair: both halves of a row's line in ONE instruction
//       (8 rows x 128 B) vs the conv's two instructions of 16 rows x 64 B
//   (4) LDS-DMA (global_load_lds_dwordx4) of a contiguous KiB, as the weight stages use
// Every wave issues ITER loads from a table that stays in the XCD's L2; prints time per wave instruction per CU
// (12 waves per CU resident).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/masked_gather.hip -o /tmp/masked_gather && /tmp/masked_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 1024;
enum { FULL, Q25, R25, H50, NONE, SEQ, ADJ, ADJ64, PAIR, DMA, NMODES };
static const char* kName[NMODES] = {"full", "q25", "r25", "h50", "none", "seq", "adj", "adj64", "pair", "dma"};

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ buf, int row_mask, uint4* __restrict__ out) {
  __shared__ uint4 stage[4 * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wave = blockIdx.x * 4 + wid;
  const int col = lane & 15, kg = lane >> 4;
  bool live = true;
  if (MODE == Q25) live = (((unsigned)(lane * 2654435761u + wave * 40503u) >> 13) & 3u) == 0u;
  if (MODE == R25) live = (col & 3) == (wave & 3);
  if (MODE == H50) live = (col & 1) == (wave & 1);
  if (MODE == NONE) live = false;
  uint4 acc = make_uint4(0, 0, 0, 0);
  int r = (wave * 977 + col * 131) & row_mask;
  if (MODE == SEQ || MODE == DMA || MODE == ADJ || MODE == ADJ64) r = (wave * 977) & row_mask;   // wave-uniform walk
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)(stage + wid * 64));
#pragma unroll 8
  for (int it = 0; it < ITER; ++it) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (MODE == SEQ) {            // one contiguous KiB: 8 full lines
      v = buf[(size_t)((r & ~7) & row_mask) * 8 + lane];
    } else if (MODE == ADJ) {     // 16 consecutive rows (consecutive lines), 64 B of each
      v = buf[(size_t)(((r & ~15) + col) & row_mask) * 8 + kg];
    } else if (MODE == ADJ64) {   // 16 consecutive 64-byte records (hi and lo in separate planes): one contiguous KiB
      v = buf[(size_t)(((r & ~7) & row_mask) * 8) + col * 4 + kg];
    } else if (MODE == PAIR) {    // 8 random rows, the whole 128-byte line of each (lane = (row = lane & 7, piece = lane >> 3))
      v = buf[(size_t)((r * 7 + (lane & 7) * 61) & row_mask) * 8 + (lane >> 3)];
    } else if (MODE == DMA) {     // contiguous KiB straight into LDS
      const uint4* src = buf + (size_t)((r & ~7) & row_mask) * 8 + lane;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory");
    } else if (live) {
      v = buf[(size_t)r * 8 + kg];
    }
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    r = (r * 5 + 1 + (MODE == SEQ || MODE == DMA || MODE == ADJ || MODE == ADJ64 ? 8 : col)) & row_mask;   // pseudo-random walk
  }
  if (MODE == DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = stage[wid * 64 + lane];
  }
  if (acc.x == 0x12345678u) out[wave * 64 + lane] = acc;   // keeps the loads alive; false for the zero-filled table
}

template <int MODE>
static float run(const uint4* buf, int row_mask, uint4* out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, buf, row_mask, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, buf, row_mask, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  const int grid = 256 * 3;                       // 12 waves per CU
  uint4 *out = nullptr;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(uint4));
  for (int rows_log2 : {8, 15, 19}) {             // 32 KiB (L1), 4 MiB (L2), 64 MiB (MALL / HBM)
    const int rows = 1 << rows_log2, m = rows - 1;
    uint4* buf = nullptr;
    hipMalloc(&buf, (size_t)rows * 128);
    hipMemset(buf, 0, (size_t)rows * 128);
    const float t[NMODES] = {run<FULL>(buf, m, out, grid), run<Q25>(buf, m, out, grid), run<R25>(buf, m, out, grid),
                             run<H50>(buf, m, out, grid),  run<NONE>(buf, m, out, grid), run<SEQ>(buf, m, out, grid),
                             run<ADJ>(buf, m, out, grid),  run<ADJ64>(buf, m, out, grid), run<PAIR>(buf, m, out, grid),
                             run<DMA>(buf, m, out, grid)};
    printf("table %6d KiB (ns per wave instruction per CU):", rows / 8);
    for (int k = 0; k < NMODES; ++k) printf("  %s %.1f", kName[k], t[k] * 1e6 / (12.0 * ITER));
    printf("\n");
    hipFree(buf);
  }
  return 0;
}
