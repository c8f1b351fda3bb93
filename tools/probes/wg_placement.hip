// wg_placement.hip -- where does the dispatcher put the workgroups of a launch that fits the chip in one round?
// The conv tile plan (isf_spconv16.h, conv16_plan) deals f full + (k - f) half tiles "per CU" by ORDER only and relies
// on the round-robin placement this probe measures: workgroups with the conv kernel's footprint (256 threads, 46848 B of
// LDS => 3 per CU) record HW_ID / XCC_ID and spin for ~40 us.  Prints, per grid size, the histogram of workgroups per
// CU and -- for a 768-block launch ordered like a plan of 64 full + 32 half tiles per XCD -- how many CUs got exactly
// two "full" and one "half" block.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wg_placement.hip -o /tmp/wg_placement && /tmp/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

__global__ __launch_bounds__(256) void probe(Rec* rec, unsigned long long spin_ticks) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    smem[0] = 1;
    rec[blockIdx.x] = Rec{hw, xcc, t0, (unsigned long long)wall_clock64()};
  }
  __syncthreads();
}

int main() {
  const int lds = 46848;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, lds);
  printf("# occupancy %d workgroups per CU (256 threads, %d B LDS)\n", occ, lds);
  for (int grid : {636, 768, 512, 700}) {
    Rec* d = nullptr;
    hipMalloc(&d, grid * sizeof(Rec));
    for (int rep = 0; rep < 2; ++rep) {   // second run: warm
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 4000ull /* 100 MHz ticks = 40 us */);
      hipDeviceSynchronize();
    }
    std::vector<Rec> h(grid);
    hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;   // key: xcc | se | sh | cu
    unsigned long long tmin = ~0ull, tmax = 0, late = 0;
    for (int b = 0; b < grid; ++b) {
      const unsigned hw = h[b].hw_id;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[b].xcc_id & 0xf;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
      if (h[b].t0 < tmin) tmin = h[b].t0;
      if (h[b].t1 > tmax) tmax = h[b].t1;
    }
    for (int b = 0; b < grid; ++b) late += (h[b].t0 - tmin > 2000);   // started > 20 us after the first: a second round
    std::map<int, int> hist;
    int two_one = 0, xcd_ok = 0;
    for (auto& kv : per_cu) {
      hist[(int)kv.second.size()]++;
      int full = 0, half = 0;
      bool same_xcd = true;
      for (int b : kv.second) {
        ((b >> 3) < 64 ? full : half)++;
        same_xcd &= ((b & 7) == (kv.second[0] & 7));
      }
      two_one += (full == 2 && half == 1);
      xcd_ok += same_xcd;
    }
    printf("grid %4d: %zu CUs used, span %.1f us, %llu blocks started late; blocks per CU:", grid, per_cu.size(),
           (tmax - tmin) / 100.0, late);
    for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("; CUs whose blocks all have the same (blockIdx & 7): %d", xcd_ok);
    if (grid == 768) printf("; CUs with exactly 2 of blocks j<64 and 1 of j>=64 (j = blockIdx>>3): %d", two_one);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
