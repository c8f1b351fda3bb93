// isf_runtime.hip -- error state, per-device workspace arena, occupancy-index (rank bitmap) kernels.
#include <atomic>
#include <thread>

#include "isf_common.h"

#include <map>
#include <memory>
#include <mutex>

namespace isf {

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------ arena
static const size_t kMinBlock = (size_t)64 << 20;

int Arena::reset() {
  if (blocks_.size() > 1) {  // coalesce into one block so the steady state never calls hipMalloc
    size_t total = 0;
    for (auto& b : blocks_) total += b.cap;
    ISF_HIP_TRY(hipDeviceSynchronize());
    for (auto& b : blocks_) ISF_HIP_TRY(hipFree(b.base));
    blocks_.clear();
    char* p = nullptr;
    size_t want = total + total / 4;
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      set_error("arena: hipMalloc(%zu) failed", want);
      return ISF_ERR_NOMEM;
    }
    blocks_.push_back({p, want, 0});
  }
  for (auto& b : blocks_) b.off = 0;
  return ISF_OK;
}

int Arena::alloc(void** out, size_t bytes) {
  bytes = round_up(bytes ? bytes : 1, 256);
  for (auto& b : blocks_) {
    if (b.off + bytes <= b.cap) {
      *out = b.base + b.off;
      b.off += bytes;
      return ISF_OK;
    }
  }
  size_t cap = bytes > kMinBlock ? bytes : kMinBlock;
  char* p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) {
    (void)hipGetLastError();
    set_error("arena: hipMalloc(%zu) failed", cap);
    return ISF_ERR_NOMEM;
  }
  blocks_.push_back({p, cap, bytes});
  *out = p;
  return ISF_OK;
}

int Arena::release() {
  if (!blocks_.empty() || bytemaps.fine || side || mailbox_host || zero_pool || ks_scratch || ks_count)
    ISF_HIP_TRY(hipDeviceSynchronize());
  for (auto& b : blocks_) ISF_HIP_TRY(hipFree(b.base));
  blocks_.clear();
  if (bytemaps.fine) {
    (void)hipFree(bytemaps.fine);
    (void)hipFree(bytemaps.coarse);
    bytemaps = ByteMaps();
  }
  for (auto e : events) (void)hipEventDestroy(e);
  events.clear();
  next_event = 0;
  if (side) {
    (void)hipStreamDestroy(side);
    side = nullptr;
  }
  if (mailbox_host) {
    (void)hipHostFree(mailbox_host);
    mailbox_host = mailbox_dev = nullptr;
  }
  if (zero_pool) {
    (void)hipFree(zero_pool);
    zero_pool = nullptr;
  }
  if (ks_scratch) (void)hipFree(ks_scratch);
  if (ks_count) (void)hipFree(ks_count);
  ks_scratch = nullptr;
  ks_count = nullptr;
  ks_scratch_bytes = 0;
  ks_count_cap = 0;
  return ISF_OK;
}

int Arena::list_blocks(unsigned long long* base_cap, int max_pairs, int* num_pairs) const {
  int n = 0;
  for (auto& b : blocks_) {
    if (n < max_pairs) { base_cap[2 * n] = (unsigned long long)(size_t)b.base; base_cap[2 * n + 1] = b.cap; }
    ++n;
  }
  if (bytemaps.fine) {
    if (n < max_pairs) { base_cap[2 * n] = (unsigned long long)(size_t)bytemaps.fine; base_cap[2 * n + 1] = bytemaps.words * 64; }
    ++n;
    if (n < max_pairs) { base_cap[2 * n] = (unsigned long long)(size_t)bytemaps.coarse; base_cap[2 * n + 1] = bytemaps.words; }
    ++n;
  }
  *num_pairs = n;
  return ISF_OK;
}

size_t Arena::capacity() const {
  size_t t = 0;
  for (auto& b : blocks_) t += b.cap;
  return t;
}

// registry: (device, stream) -> workspace.  Entries live until isf_release_workspace().
static std::mutex g_mu;
static std::map<std::pair<int, hipStream_t>, std::unique_ptr<Arena>> g_arenas;

Arena& arena_for_stream(hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  auto& slot = g_arenas[std::make_pair(dev, st)];
  if (!slot) slot.reset(new Arena());
  return *slot;
}

int zero_ints(Arena& a, int** out) {
  if (!a.zero_pool) {
    ISF_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&a.zero_pool), kZeroInts * sizeof(int)));
    ISF_HIP_TRY(hipMemset(a.zero_pool, 0, kZeroInts * sizeof(int)));   // once per workspace (synchronous)
  }
  *out = a.zero_pool;
  return ISF_OK;
}

int ksplit_buffers(Arena& a, size_t scratch_bytes, int counters, float** scratch, unsigned** count, hipStream_t st) {
  if (scratch_bytes > a.ks_scratch_bytes || counters > a.ks_count_cap) {
    ISF_HIP_TRY(hipStreamSynchronize(st));     // launches that still use the old buffers are behind us on this stream
    if (scratch_bytes > a.ks_scratch_bytes) {
      if (a.ks_scratch) ISF_HIP_TRY(hipFree(a.ks_scratch));
      a.ks_scratch = nullptr;
      a.ks_scratch_bytes = 0;
      const size_t want = round_up(scratch_bytes + scratch_bytes / 4, (size_t)1 << 20);
      ISF_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&a.ks_scratch), want));
      a.ks_scratch_bytes = want;
    }
    if (counters > a.ks_count_cap) {
      if (a.ks_count) ISF_HIP_TRY(hipFree(a.ks_count));
      a.ks_count = nullptr;
      a.ks_count_cap = 0;
      const int want = (int)round_up((size_t)counters * 2, 4096);
      ISF_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&a.ks_count), (size_t)want * sizeof(unsigned)));
      ISF_HIP_TRY(hipMemset(a.ks_count, 0, (size_t)want * sizeof(unsigned)));   // once: the kernels restore the zeros
      a.ks_count_cap = want;
    }
  }
  *scratch = a.ks_scratch;
  *count = a.ks_count;
  return ISF_OK;
}

int side_stream(Arena& a, hipStream_t* out) {
  if (!a.side) ISF_HIP_TRY(hipStreamCreateWithFlags(&a.side, hipStreamNonBlocking));
  *out = a.side;
  return ISF_OK;
}

// events are recycled round-robin; 256 is far more than one forward records, so an event is never re-recorded
// while a wait on its previous recording is still pending in the same call
int pooled_event(Arena& a, hipEvent_t* out) {
  if (a.events.size() < 256) {
    hipEvent_t e;
    ISF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    a.events.push_back(e);
    *out = e;
    return ISF_OK;
  }
  *out = a.events[a.next_event];
  a.next_event = (a.next_event + 1) % a.events.size();
  return ISF_OK;
}

// make stream `waiter` wait for everything enqueued on `producer` so far
int stream_wait_stream(Arena& a, hipStream_t waiter, hipStream_t producer) {
  hipEvent_t e;
  ISF_TRY(pooled_event(a, &e));
  ISF_HIP_TRY(hipEventRecord(e, producer));
  ISF_HIP_TRY(hipStreamWaitEvent(waiter, e, 0));
  return ISF_OK;
}

static constexpr unsigned kMailboxSlots = 64;

__global__ void publish_int_kernel(const int* __restrict__ src, volatile int* box, int ticket) {
  box[0] = *src;
  __threadfence_system();
  box[1] = ticket;
  __threadfence_system();
}

int post_int(Arena& a, const int* dev, hipStream_t st, unsigned* ticket) {
  if (!a.mailbox_host) {
    void* h = nullptr;
    // coherent (fine-grained) explicitly: the host polls this memory while the kernel that writes it may still be
    // running; the default follows HIP_HOST_COHERENT
    ISF_HIP_TRY(hipHostMalloc(&h, kMailboxSlots * 2 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h, 0, kMailboxSlots * 2 * sizeof(int));
    void* d = nullptr;
    ISF_HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
    a.mailbox_host = reinterpret_cast<int*>(h);
    a.mailbox_dev = reinterpret_cast<int*>(d);
    a.mailbox_seq = 0;
  }
  const unsigned t = ++a.mailbox_seq;          // tickets start at 1: a zeroed slot never matches
  const unsigned slot = t % kMailboxSlots;
  hipLaunchKernelGGL(publish_int_kernel, dim3(1), dim3(1), 0, st, dev, a.mailbox_dev + 2 * slot, (int)t);
  ISF_LAUNCH_CHECK();
  *ticket = t;
  return ISF_OK;
}

int wait_int(Arena& a, unsigned ticket, hipStream_t st, int* value) {
  ISF_REQUIRE(a.mailbox_host && ticket != 0 && a.mailbox_seq - ticket < kMailboxSlots, ISF_ERR_ARG,
              "wait_int: ticket %u is not in flight", ticket);
  volatile int* box = a.mailbox_host + 2 * (ticket % kMailboxSlots);
  unsigned spins = 0;
  while ((unsigned)box[1] != ticket) {
    if ((++spins & 0x3fff) == 0) {             // every 16 k polls: has the stream died or drained without our kernel?
      const hipError_t q = hipStreamQuery(st);
      if (q != hipSuccess && q != hipErrorNotReady) {
        set_error("wait_int: stream error while waiting for a device count: %s", hipGetErrorString(q));
        return ISF_ERR_HIP;
      }
      if (q == hipSuccess && (unsigned)box[1] != ticket) {   // drained: the write must be visible by now
        ISF_HIP_TRY(hipStreamSynchronize(st));
        if ((unsigned)box[1] != ticket) {
          set_error("wait_int: ticket %u never arrived", ticket);
          return ISF_ERR_HIP;
        }
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if (spins > (1u << 22)) std::this_thread::yield();   // a count that takes this long: stop burning the core
  }
  std::atomic_thread_fence(std::memory_order_acquire);   // the value was written before the ticket (device-side fence)
  *value = box[0];
  return ISF_OK;
}

int read_int(const int* dev, int* host, hipStream_t st) {
  ISF_HIP_TRY(hipMemcpyAsync(host, dev, sizeof(int), hipMemcpyDeviceToHost, st));
  ISF_HIP_TRY(hipStreamSynchronize(st));
  return ISF_OK;
}

// ------------------------------------------------------------------------------------ occupancy index
// Three streaming kernels over the bitmap words: per-block popcount totals -> scan of the block totals
// -> per-word exclusive prefix.  HBM-bound integer work: reads the bitmap twice, writes nwords*4 B.
static constexpr int kScanThreads = 256;
static constexpr int kWordsPerThread = 4;
static constexpr int kWordsPerBlock = kScanThreads * kWordsPerThread;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
  // wave-level inclusive scan with DPP-free shuffles (64-wide), then scan of the 4 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    uint32_t s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void occ_block_count_kernel(
    const unsigned long long* __restrict__ bits, size_t nwords, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t lds[kScanThreads / 64];
  const size_t w0 = (size_t)blockIdx.x * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
  uint32_t c = 0;
  if (w0 + kWordsPerThread <= nwords) {
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(bits + w0);
    ulonglong2 a = p[0], b = p[1];
    c = __popcll(a.x) + __popcll(a.y) + __popcll(b.x) + __popcll(b.y);
  } else {
    for (int i = 0; i < kWordsPerThread; ++i)
      if (w0 + i < nwords) c += __popcll(bits[w0 + i]);
  }
  uint32_t tot;
  (void)block_exclusive_scan(c, lds, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void occ_scan_sums_kernel(uint32_t* __restrict__ block_sums,
                                                             int nblocks, int* __restrict__ total) {
  // single workgroup; nblocks is at most a few hundred thousand -> chunked scan with a running carry
  __shared__ uint32_t lds[1024];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    uint32_t v = i < nblocks ? block_sums[i] : 0;
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      uint32_t t = threadIdx.x >= d ? lds[threadIdx.x - d] : 0;
      __syncthreads();
      lds[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t carry = carry_s;
    const uint32_t inc = lds[threadIdx.x];
    if (i < nblocks) block_sums[i] = carry + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = (int)carry_s;
}

// Offset of a block = sum of the totals of the blocks before it.  Up to kLookBehindBlocks blocks every block adds them up
// itself (SUMS = true: `block_sums` holds raw totals; the last block also writes the grand total) -- the single-workgroup scan
// in between is one launch more than the few KiB of L2 reads it saves; larger grids (level 0: 5 000 blocks) keep it.
static constexpr int kLookBehindBlocks = 2048;
__device__ __forceinline__ uint32_t scan_look_behind(const uint32_t* __restrict__ block_sums, uint32_t* lds) {
  uint32_t s = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kScanThreads) s += block_sums[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  uint32_t t = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) t += lds[w];
  __syncthreads();
  return t;
}

template <bool SUMS>
__global__ __launch_bounds__(kScanThreads) void occ_write_prefix_kernel(
    const unsigned long long* __restrict__ bits, size_t nwords,
    const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ prefix, int* __restrict__ total) {
  __shared__ uint32_t lds[kScanThreads / 64];
  const size_t w0 = (size_t)blockIdx.x * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
  uint32_t pc[kWordsPerThread];
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) pc[i] = (w0 + i < nwords) ? __popcll(bits[w0 + i]) : 0;
  uint32_t c = pc[0] + pc[1] + pc[2] + pc[3];
  uint32_t tot;
  const uint32_t base = SUMS ? scan_look_behind(block_offsets, lds) : block_offsets[blockIdx.x];
  uint32_t ex = block_exclusive_scan(c, lds, &tot) + base;
  if (SUMS && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = (int)(base + tot);
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    if (w0 + i < nwords) prefix[w0 + i] = ex;
    ex += pc[i];
  }
}

size_t occ_bits_bytes(const OccIndex& occ) {   // what occ_create's zero fill covers (for callers that batch their fills)
  return round_up(occ.nwords, kWordsPerBlock) * sizeof(unsigned long long);
}

int occ_create(Arena& a, OccIndex* occ, int B, int D, int H, int W, hipStream_t st, bool zero) {
  occ->B = B; occ->D = D; occ->H = H; occ->W = W;
  occ->ncells = (unsigned long long)B * D * H * W;
  occ->nwords = (size_t)((occ->ncells + 63) / 64);
  // pad the word count to a multiple of kWordsPerThread so vector loads stay in bounds
  const size_t alloc_words = round_up(occ->nwords, kWordsPerBlock);
  ISF_TRY(a.alloc_n(&occ->bits, alloc_words));
  ISF_TRY(a.alloc_n(&occ->prefix, alloc_words));
  ISF_TRY(a.alloc_n(&occ->total, 64));
  if (zero) ISF_HIP_TRY(hipMemsetAsync(occ->bits, 0, alloc_words * sizeof(unsigned long long), st));
  return ISF_OK;
}

// ---- atomic-free marking for the big level-0 grid ------------------------------------------------------
// Device-scope atomicOr runs at the memory side on this chip (the XCD L2s are not coherent), ~3 G/s: 0.37 ms
// for 1.2 M points.  Instead: two persistent all-zero byte maps, `fine` (1 byte per cell) and `coarse`
// (1 byte per 64-cell word).  Pass A: every point stores 1 into both (idempotent plain stores: racing
// writers write the same value).  Pass B: one thread per word; touched words gather their 64 fine bytes into
// the bitmap word and zero what they read, so both maps are clean again for the next frame; untouched words
// write 0 (which also replaces the bitmap memset).  HBM traffic per frame: nwords (coarse) + 64 B per touched
// word, instead of ~1.2 M fabric atomics.

__global__ void occ_bytemap_mark_kernel(const int32_t* __restrict__ coors4, int n, int B, int D, int H, int W,
                                        unsigned char* __restrict__ fine, unsigned char* __restrict__ coarse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coors4)[i];
  if (c.x < 0 || c.y < 0 || c.z < 0 || c.w < 0 || c.x >= B || c.y >= D || c.z >= H || c.w >= W) return;
  const unsigned long long cell = (((unsigned long long)c.x * D + c.y) * H + c.z) * W + c.w;
  fine[cell] = 1;
  coarse[cell >> 6] = 1;
}

__global__ __launch_bounds__(256) void occ_bytemap_pack_kernel(unsigned char* __restrict__ fine,
                                                               unsigned char* __restrict__ coarse,
                                                               size_t nwords_alloc,
                                                               unsigned long long* __restrict__ bits) {
  // thread -> 4 consecutive words (one 4-byte load of the coarse map)
  const size_t w0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (w0 >= nwords_alloc) return;
  const unsigned int c4 = *reinterpret_cast<const unsigned int*>(coarse + w0);
  unsigned long long out[4] = {0ull, 0ull, 0ull, 0ull};
  if (c4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!((c4 >> (8 * j)) & 0xffu)) continue;
      uint4* f = reinterpret_cast<uint4*>(fine + (w0 + j) * 64);
      unsigned long long m = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 v = f[q];
        const unsigned int d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if ((d[e] >> (8 * b)) & 0xffu) m |= 1ull << (q * 16 + e * 4 + b);
        f[q] = make_uint4(0, 0, 0, 0);
      }
      out[j] = m;
    }
    *reinterpret_cast<unsigned int*>(coarse + w0) = 0u;
  }
  reinterpret_cast<ulonglong2*>(bits + w0)[0] = make_ulonglong2(out[0], out[1]);
  reinterpret_cast<ulonglong2*>(bits + w0)[1] = make_ulonglong2(out[2], out[3]);
}

// voxelize + mark: thread -> point; the frame of a point by a walk over <= 8 offsets; coordinates exactly as
// dynamic_voxelize_kernel computes them (voxel_of_point), (b, z, y, x) or (b, -1, -1, -1)
__global__ __launch_bounds__(256) void occ_voxelize_mark_kernel(const float* __restrict__ points, int P, int C, VoxGeom g,
                                                                VoxBatch vb, int D, int32_t* __restrict__ coors4,
                                                                unsigned char* __restrict__ fine,
                                                                unsigned char* __restrict__ coarse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int b = 0;
#pragma unroll
  for (int k = 1; k < kVoxMaxBatch; ++k) b += (k < vb.B && (long long)i >= vb.off[k]) ? 1 : 0;
  int cx, cy, cz;
  const bool ok = voxel_of_point(points + (size_t)i * C, g.vx, g.vy, g.vz, g.x0, g.y0, g.z0, g.gx, g.gy, g.gz, cx, cy, cz);
  reinterpret_cast<int4*>(coors4)[i] = ok ? make_int4(b, cz, cy, cx) : make_int4(b, -1, -1, -1);
  if (ok && cz < D) {
    const unsigned long long cell = (((unsigned long long)b * D + cz) * g.gy + cy) * g.gx + cx;
    fine[cell] = 1;
    coarse[cell >> 6] = 1;
  }
}

static int occ_bytemaps_ensure(Arena& a, const OccIndex& occ, hipStream_t st, size_t* alloc_words_out) {
  ByteMaps& bm = a.bytemaps;
  const size_t alloc_words = round_up(occ.nwords, kWordsPerBlock);
  *alloc_words_out = alloc_words;
  if (bm.words < alloc_words) {  // (re)allocate the persistent maps; zeroed once, kept zero by the pack pass
    ISF_HIP_TRY(hipStreamSynchronize(st));
    if (bm.fine) { ISF_HIP_TRY(hipFree(bm.fine)); ISF_HIP_TRY(hipFree(bm.coarse)); bm = ByteMaps(); }
    if (hipMalloc(&bm.fine, alloc_words * 64) != hipSuccess || hipMalloc(&bm.coarse, alloc_words) != hipSuccess) {
      (void)hipGetLastError();
      set_error("occupancy byte maps: hipMalloc(%zu) failed", alloc_words * 65);
      return ISF_ERR_NOMEM;
    }
    ISF_HIP_TRY(hipMemsetAsync(bm.fine, 0, alloc_words * 64, st));
    ISF_HIP_TRY(hipMemsetAsync(bm.coarse, 0, alloc_words, st));
    bm.words = alloc_words;
  }
  return ISF_OK;
}

int occ_voxelize_mark_bytemap(Arena& a, const OccIndex& occ, const float* points, int P, int C, const VoxGeom& g,
                              const VoxBatch& vb, int32_t* coors4, hipStream_t st) {
  ISF_REQUIRE(vb.B >= 1 && vb.B <= kVoxMaxBatch && vb.B == occ.B && g.gy == occ.H && g.gx == occ.W && g.gz <= occ.D,
              ISF_ERR_ARG, "voxelize + mark: %d frames, grid %d x %d x %d vs index %d x %d x %d", vb.B, g.gz, g.gy, g.gx,
              occ.D, occ.H, occ.W);
  size_t alloc_words = 0;
  ISF_TRY(occ_bytemaps_ensure(a, occ, st, &alloc_words));
  ByteMaps& bm = a.bytemaps;
  if (P > 0)
    hipLaunchKernelGGL(occ_voxelize_mark_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, P, C, g, vb, occ.D, coors4,
                       bm.fine, bm.coarse);
  hipLaunchKernelGGL(occ_bytemap_pack_kernel, dim3(ceil_div((long long)(alloc_words / 4), 256)), dim3(256), 0, st,
                     bm.fine, bm.coarse, alloc_words, occ.bits);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int occ_mark_coords4_bytemap(Arena& a, const OccIndex& occ, const int32_t* coors4, int n, hipStream_t st) {
  ByteMaps& bm = a.bytemaps;
  const size_t alloc_words = round_up(occ.nwords, kWordsPerBlock);
  if (bm.words < alloc_words) {  // (re)allocate the persistent maps; zeroed once, kept zero by the pack pass
    ISF_HIP_TRY(hipStreamSynchronize(st));
    if (bm.fine) { ISF_HIP_TRY(hipFree(bm.fine)); ISF_HIP_TRY(hipFree(bm.coarse)); bm = ByteMaps(); }
    if (hipMalloc(&bm.fine, alloc_words * 64) != hipSuccess || hipMalloc(&bm.coarse, alloc_words) != hipSuccess) {
      (void)hipGetLastError();
      set_error("occupancy byte maps: hipMalloc(%zu) failed", alloc_words * 65);
      return ISF_ERR_NOMEM;
    }
    ISF_HIP_TRY(hipMemsetAsync(bm.fine, 0, alloc_words * 64, st));
    ISF_HIP_TRY(hipMemsetAsync(bm.coarse, 0, alloc_words, st));
    bm.words = alloc_words;
  }
  if (n > 0)
    hipLaunchKernelGGL(occ_bytemap_mark_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, coors4, n, occ.B, occ.D,
                       occ.H, occ.W, bm.fine, bm.coarse);
  hipLaunchKernelGGL(occ_bytemap_pack_kernel, dim3(ceil_div((long long)(alloc_words / 4), 256)), dim3(256), 0, st,
                     bm.fine, bm.coarse, alloc_words, occ.bits);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// up to four byte fills in ONE launch (the memsets in front of a chain of small kernels are launches like any other)
struct FillJobs {
  unsigned long long* ptr[4];
  unsigned long long words[4];   // 8-byte words (arena allocations are 256-byte aligned and padded)
  unsigned long long value[4];
  unsigned long long first[5];
};
__global__ __launch_bounds__(256) void fill_many_kernel(FillJobs j) {
  const unsigned long long total = j.first[4];
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) k += i >= j.first[q] ? 1 : 0;
    j.ptr[k][i - j.first[k]] = j.value[k];
  }
}

int fill_many(hipStream_t st, int n, void* const* ptrs, const size_t* bytes, const unsigned char* byte_values) {
  ISF_REQUIRE(n >= 1 && n <= 4, ISF_ERR_ARG, "fill_many: %d jobs (1..4)", n);
  FillJobs j;
  unsigned long long at = 0;
  for (int k = 0; k < 4; ++k) {
    const bool on = k < n && bytes[k] > 0;
    j.ptr[k] = on ? reinterpret_cast<unsigned long long*>(ptrs[k]) : nullptr;
    j.words[k] = on ? (bytes[k] + 7) / 8 : 0;   // rounds up inside the allocation's 256-byte padding
    j.value[k] = on ? 0x0101010101010101ull * byte_values[k] : 0;
    j.first[k] = at;
    at += j.words[k];
  }
  j.first[4] = at;
  if (at == 0) return ISF_OK;
  const unsigned long long blocks = (at + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(fill_many_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, j);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int occ_scan(Arena& a, const OccIndex& occ, hipStream_t st) {
  const int nblocks = ceil_div((long long)occ.nwords, kWordsPerBlock);
  uint32_t* sums = nullptr;
  ISF_TRY(a.alloc_n(&sums, (size_t)nblocks + 1));
  hipLaunchKernelGGL(occ_block_count_kernel, dim3(nblocks), dim3(kScanThreads), 0, st, occ.bits,
                     occ.nwords, sums);
  if (nblocks <= kLookBehindBlocks) {
    hipLaunchKernelGGL(occ_write_prefix_kernel<true>, dim3(nblocks), dim3(kScanThreads), 0, st, occ.bits, occ.nwords, sums,
                       occ.prefix, occ.total);
  } else {
    hipLaunchKernelGGL(occ_scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nblocks, occ.total);
    hipLaunchKernelGGL(occ_write_prefix_kernel<false>, dim3(nblocks), dim3(kScanThreads), 0, st, occ.bits, occ.nwords, sums,
                       occ.prefix, occ.total);
  }
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ---- generic exclusive scan of uint32 (same three-kernel scheme); out has n + 1 entries, out[n] = total
__global__ __launch_bounds__(kScanThreads) void scan_u32_block_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                      uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t lds[kScanThreads / 64];
  const size_t i0 = (size_t)blockIdx.x * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i)
    if (i0 + i < n) c += in[i0 + i];
  uint32_t tot;
  (void)block_exclusive_scan(c, lds, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

template <bool SUMS>
__global__ __launch_bounds__(kScanThreads) void scan_u32_write_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                      const uint32_t* __restrict__ block_offsets,
                                                                      uint32_t* __restrict__ out) {
  __shared__ uint32_t lds[kScanThreads / 64];
  const size_t i0 = (size_t)blockIdx.x * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
  uint32_t v[kWordsPerThread];
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) v[i] = (i0 + i < n) ? in[i0 + i] : 0;
  uint32_t tot;
  const uint32_t base = SUMS ? scan_look_behind(block_offsets, lds) : block_offsets[blockIdx.x];
  uint32_t ex = block_exclusive_scan(v[0] + v[1] + v[2] + v[3], lds, &tot) + base;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    if (i0 + i < n) out[i0 + i] = ex;
    ex += v[i];
    if (i0 + i + 1 == n) out[n] = ex;
  }
}

int scan_u32_exclusive(Arena& a, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
  if (n == 0) {
    ISF_HIP_TRY(hipMemsetAsync(out, 0, sizeof(uint32_t), st));
    return ISF_OK;
  }
  const int nblocks = ceil_div((long long)n, kWordsPerBlock);
  uint32_t* sums = nullptr;
  int* total = nullptr;
  ISF_TRY(a.alloc_n(&sums, (size_t)nblocks + 1));
  ISF_TRY(a.alloc_n(&total, 64));
  hipLaunchKernelGGL(scan_u32_block_kernel, dim3(nblocks), dim3(kScanThreads), 0, st, in, n, sums);
  if (nblocks <= kLookBehindBlocks) {
    hipLaunchKernelGGL(scan_u32_write_kernel<true>, dim3(nblocks), dim3(kScanThreads), 0, st, in, n, sums, out);
  } else {
    hipLaunchKernelGGL(occ_scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nblocks, total);
    hipLaunchKernelGGL(scan_u32_write_kernel<false>, dim3(nblocks), dim3(kScanThreads), 0, st, in, n, sums, out);
  }
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

__global__ void occ_mark_coords4_kernel(unsigned long long* __restrict__ bits,
                                        const int32_t* __restrict__ coors4, int n, int B, int D, int H,
                                        int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coors4)[i];
  if (c.x < 0 || c.y < 0 || c.z < 0 || c.w < 0 || c.x >= B || c.y >= D || c.z >= H || c.w >= W) return;
  const unsigned long long cell = (((unsigned long long)c.x * D + c.y) * H + c.z) * W + c.w;
  const unsigned long long bit = 1ull << (cell & 63);
  unsigned long long* p = bits + (cell >> 6);
  if (!(__builtin_nontemporal_load(p) & bit)) atomicOr(p, bit);
}

int occ_mark_coords4(const OccIndex& occ, const int32_t* coors4, int n, hipStream_t st) {
  if (n <= 0) return ISF_OK;
  hipLaunchKernelGGL(occ_mark_coords4_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, occ.bits,
                     coors4, n, occ.B, occ.D, occ.H, occ.W);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

__global__ void occ_compact_coords4_kernel(const unsigned long long* __restrict__ bits,
                                           const uint32_t* __restrict__ prefix, size_t nwords, int D,
                                           int H, int W, int32_t* __restrict__ out) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned long long word = bits[w];
  if (!word) return;
  uint32_t r = prefix[w];
  // decode the word's first cell once (the only divisions), then walk the set bits with carries
  unsigned long long cell = (unsigned long long)w << 6;
  int x0 = (int)(cell % W); cell /= W;
  int y0 = (int)(cell % H); cell /= H;
  int z0 = (int)(cell % D);
  int b0 = (int)(cell / D);
  while (word) {
    const int bit = __ffsll((long long)word) - 1;
    word &= word - 1;
    int x = x0 + bit, y = y0, z = z0, b = b0;
    while (x >= W) {  // at most ceil(64/W) iterations
      x -= W;
      if (++y == H) { y = 0; if (++z == D) { z = 0; ++b; } }
    }
    reinterpret_cast<int4*>(out)[r] = make_int4(b, z, y, x);
    ++r;
  }
}

int occ_compact_coords4(const OccIndex& occ, int32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(occ_compact_coords4_kernel, dim3(ceil_div((long long)occ.nwords, 256)), dim3(256),
                     0, st, occ.bits, occ.prefix, occ.nwords, occ.D, occ.H, occ.W, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // namespace isf

// ------------------------------------------------------------------------------------ C ABI: runtime
extern "C" {

int isf_version(void) { return (0 << 16) | (1 << 8) | 0; }

const char* isf_last_error(void) { return isf::g_err; }

int isf_device_count(int* count_host) {
  if (!count_host) return ISF_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count_host = n;
  return ISF_OK;
}

int isf_release_workspace(void) {
  std::lock_guard<std::mutex> lock(isf::g_mu);
  int cur = 0;
  ISF_HIP_TRY(hipGetDevice(&cur));
  int rc = ISF_OK;
  for (auto& kv : isf::g_arenas) {
    ISF_HIP_TRY(hipSetDevice(kv.first.first));
    const int r = kv.second->release();
    if (r != ISF_OK) rc = r;
  }
  isf::g_arenas.clear();
  ISF_HIP_TRY(hipSetDevice(cur));
  return rc;
}

// DIAGNOSTIC: every device allocation behind the workspaces of the current device: (stream handle, base, bytes) triples,
// per workspace its blocks, then its two byte maps -- tools/graph_fault.py maps a GPU fault address onto them
int isf_debug_workspace_blocks(unsigned long long* stream_base_bytes, int max_triples, int* num_triples) {
  if (!stream_base_bytes || !num_triples || max_triples < 0) return ISF_ERR_ARG;
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::lock_guard<std::mutex> lock(isf::g_mu);
  int n = 0;
  for (auto& kv : isf::g_arenas) {
    if (kv.first.first != cur) continue;
    unsigned long long tmp[64];
    int k = 0;
    (void)kv.second->list_blocks(tmp, 32, &k);
    for (int i = 0; i < k && i < 32; ++i) {
      if (n < max_triples) {
        stream_base_bytes[3 * n] = (unsigned long long)(size_t)kv.first.second;
        stream_base_bytes[3 * n + 1] = tmp[2 * i];
        stream_base_bytes[3 * n + 2] = tmp[2 * i + 1];
      }
      ++n;
    }
  }
  *num_triples = n;
  return ISF_OK;
}

int isf_workspace_bytes(size_t* bytes_host) {
  if (!bytes_host) return ISF_ERR_ARG;
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::lock_guard<std::mutex> lock(isf::g_mu);
  size_t t = 0;
  for (auto& kv : isf::g_arenas)
    if (kv.first.first == cur) t += kv.second->capacity();
  *bytes_host = t;
  return ISF_OK;
}

}  // extern "C"
