// Shared by the translation units of libntsynt_hip.so (ntsynt_hip.hip: context, genomes, sketch, Bloom filter; nts_comm_fasta.hip: the
// two exchanges and the FASTA parse; nts_graph.hip: graph build and the graph stage in HBM): the allocation cache (one per process:
// inline variables), the opaque types of the C ABI, the per-call helpers (error text, scratch buffers, timing events) and the few
// kernels and primitives more than one unit launches.  Helpers in the anonymous namespace are compiled into each unit that uses them.
#pragma once
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/ntsynt_hip.h"
#include "nts_device.h"
#include "nts_knobs.h"

using namespace nts;

// ---- device memory accounting ---------------------------------------------------------------------------------------
// Every device allocation of the library goes through dev_malloc / dev_free: live bytes and their high-water mark per
// process (all contexts), read by nts_mem_stats.  The reference publishes exactly two figures per run, wall clock and peak
// memory (README.md:156-158; `--benchmark` records the RSS of every rule, bin/ntsynt_run_pipeline.smk:26-35); its HBM
// counterpart is this mark.
namespace nts_mem {
inline std::mutex mu;
struct Slab;
struct Block
{
  size_t bytes;
  int device;     // -1: allocated with flags (never cached)
  Slab* slab;     // the cached allocation the block was cut from; nullptr: a hipMalloc of its own
  size_t off;
};
inline std::map<void*, Block> sizes;
inline std::atomic<uint64_t> live{0}, peak{0};
// calls of hipMalloc / hipFree made through here and the host time they took (nts_alloc_stats: what a cold call spends allocating)
inline std::atomic<uint64_t> alloc_calls{0}, alloc_ns{0};
// what else a leg of a run wants to know about its allocations (nts_mem_events): allocations that failed and were tried again after the
// cache was emptied, bytes asked of / given back to the driver, host time spent waiting for the device before a block was kept
inline std::atomic<uint64_t> oom_retries{0}, driver_bytes_in{0}, driver_bytes_out{0}, free_sync_ns{0}, reserve_calls{0};
struct AllocClock
{
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~AllocClock()
  {
    alloc_calls.fetch_add(1);
    alloc_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
  }
};

// Freed blocks are kept (up to CACHE_LIMIT bytes in all) and handed out again, whole or IN PIECES: a freed hipMalloc allocation becomes a
// "slab" whose free ranges are indexed by size; a request takes the smallest free range that holds it and leaves the rest of the range
// in the index; a piece that comes back is joined with its free neighbours.  Why: on some boxes of the pool this build runs on a
// hipMalloc takes 20-100 ms now and then (the driver's round-4 line: 94.8 ms for the first sketch of a fresh genome; builder boxes in
// round 5: 45.9 and 98.6 ms in the 41 allocations of a process's first sketch, 111.9 ms in the FIVE allocations of a later genome's
// 2-bit image and tables, where the box next to it takes 0.04 ms: bench.py `cold`), and every genome, filter and context of a run
// allocates and frees: 2-bit images, tables, 14.8 GB filters, the FASTA ingest's 3 GB image given back before the first sketch.  With
// exact-size reuse only (round 5's first version) the first sketch of a process still went to the driver 41 times -- nothing it asks
// for has the size of anything freed before it; cut from what the ingest or an earlier filter left, it does not.  A slab goes back to
// the driver when it is wholly free and the cache is over its limit, when an allocation fails (every wholly free slab is released
// and the allocation tried again) and on nts_mem_trim.  `live` / `peak` count blocks in use, not cached ranges.
constexpr uint64_t CACHE_LIMIT = 96ull << 30;
constexpr size_t CACHE_MIN_BLOCK = 64u << 10; // requests below this are "small": served from slabs of their own (SMALL_SLAB bytes each), so
                                               // that a long-lived 4 KB workspace never holds a multi-GB allocation in the cache
constexpr size_t SMALL_SLAB = 8u << 20;
constexpr size_t SMALL_GRAIN = 256;
constexpr size_t GRAIN = 4096;                 // cached requests are rounded up to this; pieces are cut at multiples of it
struct Slab
{
  char* base;
  size_t bytes;
  int device;
  std::map<size_t, size_t> free; // offset -> length of the free ranges, none adjacent to another
  size_t in_use = 0;
  bool small = false;  // serves requests below CACHE_MIN_BLOCK only
  bool pinned = false; // reserved ahead of a run (nts_mem_reserve): stays when the cache is over its limit; leaves on trim / when an allocation fails
};
// (device * 2 + small, length, address): lower_bound = the smallest range of that kind that holds a request
typedef std::tuple<int, size_t, char*> FreeKey;
inline int kind_of(const Slab* sl) { return sl->device * 2 + (sl->small ? 1 : 0); }
inline std::map<FreeKey, Slab*> free_index;
inline std::set<Slab*> slabs;
inline uint64_t cached_bytes = 0; // sum of the free ranges
inline uint64_t cached_pinned = 0; // ... of which in reserved slabs (not counted against CACHE_LIMIT: they were asked for)
inline uint64_t cached_small = 0;  // ... of which in the slabs of the small requests (nts_mem_cache_stats leaves them out)
inline std::atomic<uint64_t> cache_hits{0};

inline void count_live(size_t n)
{
  const uint64_t now = live.fetch_add(n) + n;
  uint64_t seen = peak.load();
  while (now > seen && !peak.compare_exchange_weak(seen, now)) {
  }
}

// (callers hold `mu`)
inline void range_add(Slab* sl, size_t off, size_t len)
{
  sl->free[off] = len;
  free_index[FreeKey(kind_of(sl), len, sl->base + off)] = sl;
  cached_bytes += len;
  if (sl->pinned) cached_pinned += len;
  if (sl->small) cached_small += len;
}
inline void range_del(Slab* sl, size_t off, size_t len)
{
  sl->free.erase(off);
  free_index.erase(FreeKey(kind_of(sl), len, sl->base + off));
  cached_bytes -= len;
  if (sl->pinned) cached_pinned -= len;
  if (sl->small) cached_small -= len;
}
// [off, off + len) of the slab is free again: joined with the free ranges that touch it
inline void range_release(Slab* sl, size_t off, size_t len)
{
  auto next = sl->free.find(off + len);
  if (next != sl->free.end()) {
    const size_t nl = next->second;
    range_del(sl, off + len, nl);
    len += nl;
  }
  auto prev = sl->free.lower_bound(off);
  if (prev != sl->free.begin()) {
    --prev;
    if (prev->first + prev->second == off) {
      const size_t po = prev->first, pl = prev->second;
      range_del(sl, po, pl);
      off = po;
      len += pl;
    }
  }
  range_add(sl, off, len);
}
// wholly free slabs leave the cache while it holds more than `limit` bytes (the largest first); the caller frees what `gone` collects
inline void shed(uint64_t limit, std::vector<void*>& gone, bool pinned_too = false)
{
  while (cached_bytes - (pinned_too ? 0 : cached_pinned) > limit) {
    Slab* pick = nullptr;
    for (Slab* sl : slabs)
      if (sl->in_use == 0 && (pinned_too || !sl->pinned) && (!pick || sl->bytes > pick->bytes)) pick = sl;
    if (!pick) break;
    range_del(pick, 0, pick->bytes); // (wholly free: one range)
    driver_bytes_out.fetch_add(pick->bytes);
    gone.push_back(pick->base);
    slabs.erase(pick);
    delete pick;
  }
}

// every wholly free slab back to the driver; returns the bytes released
inline uint64_t trim()
{
  std::vector<void*> gone;
  uint64_t before = 0, after = 0;
  {
    std::lock_guard<std::mutex> g(mu);
    before = cached_bytes - cached_small; // (the small requests' slabs go too when wholly free; the figure reported is the large blocks')
    shed(0, gone, true);
    after = cached_bytes - cached_small;
  }
  for (void* q : gone) {
    AllocClock clk;
    ::hipFree(q);
  }
  return before - after;
}

// one hipMalloc of `bytes` that goes straight into the cache as a free slab (callers do not hold `mu`)
inline hipError_t slab_from_driver(int dev, size_t bytes, bool small, bool pinned)
{
  void* q = nullptr;
  hipError_t e;
  {
    AllocClock clk;
    e = ::hipMalloc(&q, bytes);
  }
  if (e != hipSuccess || !q) return e == hipSuccess ? hipErrorOutOfMemory : e;
  driver_bytes_in.fetch_add(bytes);
  std::lock_guard<std::mutex> g(mu);
  Slab* sl = new Slab();
  sl->base = (char*)q;
  sl->bytes = bytes;
  sl->device = dev;
  sl->small = small;
  sl->pinned = pinned;
  slabs.insert(sl);
  range_add(sl, 0, bytes);
  return hipSuccess;
}

// (callers hold `mu`) a piece of `need` bytes from the smallest free range of the kind that holds it
inline bool cut_from_cache(int dev, bool small, size_t need, size_t min_rest, void** p)
{
  auto it = free_index.lower_bound(FreeKey(dev * 2 + (small ? 1 : 0), need, nullptr));
  if (it == free_index.end() || std::get<0>(it->first) != dev * 2 + (small ? 1 : 0)) return false;
  Slab* sl = it->second;
  const size_t len = std::get<1>(it->first), off = (size_t)(std::get<2>(it->first) - sl->base);
  range_del(sl, off, len);
  size_t take = need;
  if (len - need >= min_rest)
    range_add(sl, off + need, len - need); // (what is left stays in the index; a sliver that no request could use goes along)
  else
    take = len;
  sl->in_use += take;
  *p = sl->base + off;
  sizes[*p] = { take, dev, sl, off };
  cache_hits.fetch_add(1);
  count_live(take);
  return true;
}

inline hipError_t dev_malloc(void** p, size_t n)
{
  int dev = 0;
  ::hipGetDevice(&dev);
  const bool small = n < CACHE_MIN_BLOCK;
  if (small && n) {
    // small requests live in slabs of their own: the first one of a process (or the one that finds them full) asks the driver for a slab
    const size_t need = (n + SMALL_GRAIN - 1) / SMALL_GRAIN * SMALL_GRAIN;
    for (int attempt = 0; attempt < 2; ++attempt) {
      {
        std::lock_guard<std::mutex> g(mu);
        if (cut_from_cache(dev, true, need, SMALL_GRAIN, p)) return hipSuccess;
      }
      if (attempt == 0 && slab_from_driver(dev, SMALL_SLAB, true, true) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
    }
  }
  const size_t need = (n + GRAIN - 1) / GRAIN * GRAIN;
  if (n && !small) {
    std::lock_guard<std::mutex> g(mu);
    if (cut_from_cache(dev, false, need, CACHE_MIN_BLOCK, p)) return hipSuccess;
  }
  hipError_t e;
  {
    AllocClock clk;
    e = ::hipMalloc(p, small ? n : need);
  }
  if (e == hipErrorOutOfMemory && trim() > 0) { // (what the cache held may be what was missing)
    (void)hipGetLastError();
    oom_retries.fetch_add(1);
    AllocClock clk;
    e = ::hipMalloc(p, small ? n : need);
  }
  if (e == hipSuccess && *p) {
    driver_bytes_in.fetch_add(small ? n : need);
    {
      std::lock_guard<std::mutex> g(mu);
      sizes[*p] = { small ? n : need, dev, nullptr, 0 };
    }
    count_live(small ? n : need);
  }
  return e;
}

// `bytes` of device memory taken from the driver in ONE call and kept in the cache for the requests to come (nts_mem_reserve).  When
// the device cannot give that much, what it can (less a margin) is taken instead; *got = the bytes reserved.
inline hipError_t reserve(int dev, uint64_t bytes, uint64_t* got)
{
  if (got) *got = 0;
  int cur = 0;
  ::hipGetDevice(&cur);
  if (cur != dev) ::hipSetDevice(dev);
  size_t fr = 0, tot = 0;
  hipError_t e = ::hipMemGetInfo(&fr, &tot);
  if (e == hipSuccess) {
    const size_t margin = 2ull << 30;
    if (bytes + margin > fr) bytes = fr > margin ? fr - margin : 0;
    bytes = bytes / GRAIN * GRAIN;
    if (bytes >= CACHE_MIN_BLOCK) {
      e = slab_from_driver(dev, bytes, false, true);
      if (e == hipSuccess) {
        reserve_calls.fetch_add(1);
        if (got) *got = bytes;
      }
    }
    bool have_small = false;
    {
      std::lock_guard<std::mutex> g(mu);
      for (Slab* sl : slabs) have_small |= sl->small && sl->device == dev;
    }
    if (e == hipSuccess && !have_small) (void)slab_from_driver(dev, SMALL_SLAB, true, true);
  }
  if (e != hipSuccess) (void)hipGetLastError();
  if (cur != dev) ::hipSetDevice(cur);
  return e;
}

template <class T>
inline hipError_t dev_malloc(T** p, size_t n)
{
  return dev_malloc((void**)p, n);
}

// the same with allocation flags (hipDeviceMallocUncached / hipDeviceMallocFinegrained: how the L2 treats the memory); never cached
inline hipError_t dev_malloc_flags(void** p, size_t n, unsigned flags)
{
  AllocClock clk;
  const hipError_t e = ::hipExtMallocWithFlags(p, n, flags);
  if (e == hipSuccess && *p) {
    {
      std::lock_guard<std::mutex> g(mu);
      sizes[*p] = { n, -1, nullptr, 0 };
    }
    count_live(n);
  }
  return e;
}

inline hipError_t dev_free(void* p)
{
  if (!p) return hipSuccess;
  Block blk = { 0, -1, nullptr, 0 };
  bool known = false;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = sizes.find(p);
    if (it != sizes.end()) {
      blk = it->second;
      known = true;
    }
  }
  if (known && blk.device >= 0 && (blk.slab || blk.bytes >= CACHE_MIN_BLOCK)) {
    // what hipFree does before it gives memory back: nothing queued on the device still uses the block (it may be handed to another
    // stream or context next).  The block stays in `sizes` until then: nobody else can be given its range.
    int cur = 0;
    ::hipGetDevice(&cur);
    if (cur != blk.device) ::hipSetDevice(blk.device);
    const auto ts = std::chrono::steady_clock::now();
    const hipError_t es = ::hipDeviceSynchronize();
    free_sync_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ts).count());
    if (cur != blk.device) ::hipSetDevice(cur);
    std::vector<void*> gone;
    bool kept = false;
    {
      std::lock_guard<std::mutex> g(mu);
      sizes.erase(p);
      live.fetch_sub(blk.bytes);
      if (blk.slab) { // a piece of a cached allocation goes back to it whatever happened (the slab is freed as a whole, or not at all)
        blk.slab->in_use -= blk.bytes;
        range_release(blk.slab, blk.off, blk.bytes);
        kept = true;
      } else if (es == hipSuccess) {
        Slab* sl = new Slab();
        sl->base = (char*)p;
        sl->bytes = blk.bytes;
        sl->device = blk.device;
        slabs.insert(sl);
        range_add(sl, 0, blk.bytes);
        kept = true;
      }
      if (kept) shed(CACHE_LIMIT, gone);
    }
    for (void* q : gone) {
      AllocClock clk;
      ::hipFree(q);
    }
    if (kept) return hipSuccess;
    driver_bytes_out.fetch_add(blk.bytes);
    AllocClock clk;
    return ::hipFree(p);
  }
  if (known) {
    driver_bytes_out.fetch_add(blk.bytes);
    std::lock_guard<std::mutex> g(mu);
    sizes.erase(p);
    live.fetch_sub(blk.bytes);
  } else {
    // not a block in use.  Inside a cached allocation it is a second free of a piece (or of the allocation itself): giving the address
    // to hipFree would take the whole allocation away from under the cache and the pieces in use -- refused, and said once
    std::lock_guard<std::mutex> g(mu);
    for (Slab* sl : slabs)
      if ((char*)p >= sl->base && (char*)p < sl->base + sl->bytes) {
        static bool said = false;
        if (!said) {
          said = true;
          fprintf(stderr, "ntsynt_hip: device block %p freed twice (ignored)\n", p);
          void* bt[24];
          backtrace_symbols_fd(bt, backtrace(bt, 24), 2);
        }
        return hipErrorInvalidValue;
      }
  }
  AllocClock clk;
  return ::hipFree(p);
}
} // namespace nts_mem
using nts_mem::dev_free;
using nts_mem::dev_malloc;
using nts_mem::dev_malloc_flags;

inline std::atomic<int> g_live_contexts{0};
// Sketches of several genomes at once (one context each, device.SketchPool / NTS_SKETCH_POOL): the select kernels of the contexts of a
// device run one after the other -- each waits for the one launched before it -- while a genome's latency-bound tail (compaction,
// window decisions, gather, uncovered ranges, finalize) floats next to the following genome's select kernel.  Started together the
// select kernels would share the chip and finish together, and the tails would again find nothing to hide behind.
namespace nts_chain {
inline std::mutex mu;
inline hipEvent_t ev[32] = {};
inline bool live[32] = {};
} // namespace nts_chain

namespace {

constexpr uint64_t PAD = 256;          // invalid bytes before and after the sequence
constexpr int HASH_THREADS = 256;
constexpr int HASH_PER_THREAD = 32;    // consecutive k-mers rolled by one lane
constexpr int WIN_THREADS = 512;     // 8 waves share one tile: shorter phases, twice the waves per CU for the same LDS
constexpr uint32_t WIN_TILE = 4096;    // windows per workgroup
constexpr uint32_t WIN_CHUNK = 16;     // elements scanned sequentially by one lane
constexpr uint32_t WIN_MAX_W = 12000;  // LDS bound: (WIN_TILE + w) * 8 B + tables <= 160 KiB
constexpr uint32_t MAIL_WORDS = 32768; // 64-bit words of the pinned result mailbox (nts_ctx::mail)

std::string g_init_error;

struct Timing
{
  double ms = 0;
  uint64_t launches = 0;
};

} // namespace

struct nts_ctx
{
  int device = 0;
  bool counted = false; // among the process's live contexts (nts_init got through)
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr; // bulk device -> host copies that may run behind later kernels (nts_bf_download)
  // pinned host page the device writes small results into (counters, the first uncovered ranges): one stream
  // synchronisation reads them, instead of a chain of tiny device -> host copies
  uint64_t* mail = nullptr;     // host address
  uint64_t* d_mail = nullptr;   // the same memory as the device sees it
  uint64_t mail_seq = 0;        // last arrival flag posted
  uint8_t* stage = nullptr;     // pinned staging area for small host -> device tables (grow-only)
  size_t stage_bytes = 0;
  std::string err;
  int profiling = 0; // 0 off, 1 every kernel group, 2 only the dominant kernels (an event pair costs ~10 us of stream bubble)
  std::map<std::string, Timing> timings;
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::map<uint32_t, uint64_t*> init_tabs; // per k: device table for the first k-mer of a lane (HashParams::init)
  std::vector<hipEvent_t> spare_events; // recycled timing events (creating one costs microseconds of host time)
  // grow-only device scratch, reused across calls (a ctx serves one call at a time)
  std::map<std::string, std::pair<void*, size_t>> ws;
  // sketch policy: 0 auto (pruned when w >= 200), 1 dense, 2 pruned; prune_c/w = fraction of hashes kept as candidates
  std::vector<std::pair<void*, uint64_t>> mx_pool; // recycled result allocations
  size_t win_lds_set = 0;
  bool bin_lds_set = false;
  uint32_t n_cus = 0; // compute units of the device (asked once)
  bool small_gap_path = true; // uncovered ranges: device-side sort + merge when they are few (nts_pruned.inc)
  bool sel_ctl_clean = false; // the pruned pass's control block was cleared by the previous call's last kernel
  int bf_build_mode = 0; // 0 auto (binned build for large genomes), 1 one atomic per k-mer, 2 binned whenever it applies
  int sketch_mode = 0;
  uint32_t prune_c = 0; // 0 = adaptive (from the filter's occupancy), else fixed
  uint32_t last_c = 0;
  uint64_t last_candidates = 0, last_gaps = 0, last_gap_kmers = 0;
  uint64_t last_many_listed = 0; // candidates of k_hash_select_hi tiles that listed more than their slots hold (repeats, pieces)
  uint64_t last_bf_direct = 0;   // indices of the last partitioned Bloom build that bypassed the buckets (full bucket, lanes in pieces)
  size_t win_fused_lds_set = 0;  // dynamic LDS k_window_min<true> was last allowed
  uint32_t last_comm_sparse = 0; // the last all-reduce of a filter gathered set-bit indices instead of chunks
  uint64_t last_x2_packed_bytes = 0, last_x2_unpacked_bytes = 0, last_x2_sent_bytes = 0; // the last exchange 2 (nts_comm_last_exchange2)
  uint32_t last_bf_fallback = 0; // 1: its late list ran full (store-only build fell back to read-and-OR / fused AND build was redone unfused)
  uint32_t last_bf_sparse_level = 0;     // the last nts_bf_insert_and went the literal way over a sparse running filter (bf_level_sparse)
  uint64_t last_bf_sparse_accepted = 0;  // and accepted this many k-mers
  // dense sketch over a sparse filter: summary consulted before the filter, key tiles without an accepted k-mer skipped
  const uint32_t* cur_summary = nullptr;
  const uint32_t* cur_fold = nullptr; // folded copy of the filter for the LDS first look (k_hash_accept4), or null
  int fold_mode = 0;                  // 0 auto, 1 never (tests)
  bool acc4_lds_set = false;
  bool acc4r_lds_set = false;
  uint32_t cur_summary_shift = 0;
  uint32_t* cur_tile_any = nullptr;
  // pinned staging buffers + streams of the bulk transfers done by host threads (FASTA bytes up: nts_genome_from_fasta; filter
  // bits down: nts_bf_save), allocated on first use and kept: allocating pinned memory per call cost more than a small transfer
  struct IoLane
  {
    hipStream_t stream = nullptr;
    uint8_t* stage[2] = { nullptr, nullptr };
  };
  std::vector<IoLane> io_up, io_down;
  const nts_bf* cur_rep = nullptr; // filter-out filter of the running nts_sketch_ex call (indexlr -r), or null
  double dense_seg_per_window = 3.0; // minimizers per w k-mers the every-k-mer path sizes its output segments for; raised by the call that needed more
                                     // (accepted k-mers in clusters: 4 % divergence at w = 48 gives 3.4), so that only that one call runs twice
  double fused_seg_per_window = 2.5; // the same for the window tiles that hash their own k-mers (per w + 1 k-mers)
  bool elim_needs_full_cap = false; // a call's candidate lists did not fit half the capacity sized for the accepted k-mers (run_pruned)
  int select_impl = 0;  // candidate selection of the pruned sketch: 0 auto, 1 full-width kernel, 2 upper-halves kernel also for assemblies in pieces
  int summary_mode = 0; // 0 auto, 1 never (tests)
  uint32_t last_summary = 0;
  // tiered selection (nts_tiers.inc): 0 auto, 1 never, 2 wherever it applies; figures of the last call that went that way
  uint64_t comm_piece = 0;        // bytes per piece of exchange 1's reduce-scatter (0: 256 MiB; NTS_COMM_PIECE at nts_init)
  int comm_sparse_mode = 0;       // 1: never gather set-bit indices (experiments build: NTS_COMM_SPARSE=0)
  uint64_t comm_sparse_below = 0; // gather indices when the fullest chunk holds at most this many bits (0: chunk bytes / 128)
  unsigned io_threads = 8;        // host threads of a FASTA upload (NTS_IO_THREADS at nts_init)
  int gap_tiers_off = 0;          // 1: the uncovered ranges of the one-threshold selection go to the dense kernels (nts_sketch_tiers mode 1)
  int tier_mode = 0;
  double tier_x0 = 0;       // accepted k-mers per window the first tier aims at (0: the default)
  uint32_t tier_half = 0;   // 1: tiers in steps of 1.5 / 1.33 instead of 2
  uint64_t last_tier_probes = 0, last_tier_rounds = 0, last_tiers = 0;
};

struct nts_genome
{
  uint64_t n = 0; // bytes of concatenated sequence
  uint32_t n_rec = 0;
  uint8_t* d_code = nullptr; // PAD + n + PAD bytes; base i at d_code[PAD + i]
  // the same bases, 2 bits each, 16 per word (base i in word i/16 at bit 2*(i%16); invalid bases read as 0): the
  // register-resident base streams of k_hash_select.  Built on first use.
  mutable uint32_t* d_pack = nullptr;
  std::vector<uint64_t> rec_off, rec_len;
  uint64_t total_bases = 0;
  std::vector<uint64_t> part_bases; // nts_genome_concat: bases of each part (empty for an uploaded genome)
  // maximal stretches [a,b) of valid bases, clipped to records, ascending
  std::vector<uint64_t> st_a, st_b;
  uint64_t* d_rec_off = nullptr; // [n_rec] record offsets on the device
  // per-k run table + record tables, built on first use and kept on the device (unmasked sketches)
  mutable std::map<uint32_t, struct GenomeTables*> tables;
};

struct nts_bf
{
  uint64_t bytes = 0;
  uint64_t alloc_bytes = 0; // bytes behind d_words (>= bytes rounded up to 16; nts_bf_create_sharded: world x chunk)
  uint32_t* d_words = nullptr;
  bool owned = true;
  mutable int64_t popcnt = -1; // cached number of set bits, -1 = unknown (any write invalidates it)
  uint64_t version = 0;        // bumped by every write through the library
  // summary of a sparse filter (built on demand by the dense sketch): bit g = "some bit of filter bits [g << shift, (g+1) << shift)
  // is set"; small enough to stay in the L2, so that a probe of an all-but-empty filter ends there (nts_sketch)
  mutable uint32_t* d_summary = nullptr;
  mutable uint64_t summary_words = 0;
  mutable uint32_t summary_shift = 0;
  mutable uint64_t summary_version = ~0ULL;
  mutable double summary_density = 1.0;
  mutable uint32_t* d_fold = nullptr; // the filter folded onto 2^19 bits (bit i mod 2^19), built with the summary: LDS-resident first look
  mutable std::mutex mu;              // sketches of several genomes may run on contexts of their own at once (SketchPool): the summary is built once
};

struct nts_mx
{
  uint64_t n = 0;
  uint64_t cap_bytes = 0; // size of the single allocation behind d_h1 | d_pos | d_rec
  uint64_t* d_h1 = nullptr;
  uint32_t* d_rec = nullptr;
  uint64_t* d_pos = nullptr;
};

namespace {

// A launch's work-items are counted in 32 bits per dimension (the dispatch packet's grid size): gridDim.x * blockDim.x beyond 2^32 - 1
// runs TRUNCATED and reports nothing (seen: 8.9 M window tiles of 512 lanes -- every tile past the 2^23rd dropped, hipGetLastError
// silent).  Every launch of the library goes through NTS_LAUNCH: one that does not fit is handed an empty block, which HIP refuses, so
// that the hipGetLastError() that follows every launch sequence (the result mailbox at the latest) ends the call with an error.
inline std::atomic<uint64_t> g_refused_launches{0};
inline dim3 nts_checked_block(const dim3& grid, const dim3& block)
{
  if ((uint64_t)grid.x * (uint64_t)block.x > 0xFFFFFFFFull) {
    g_refused_launches.fetch_add(1);
    return dim3(0, 1, 1);
  }
  return block;
}
#define NTS_LAUNCH_(kern, grid, block, ...) hipLaunchKernelGGL(kern, grid, nts_checked_block(grid, block), __VA_ARGS__)
#define NTS_LAUNCH(...) NTS_LAUNCH_(__VA_ARGS__) // (arguments expanded first: E_GRID(n) stands for grid, block, LDS bytes and stream)

#define HIP_TRY(ctx, expr)                                                                          \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_) +                              \
                   (g_refused_launches.exchange(0) ? " -- a launch of more than 2^32 - 1 work-items was refused instead of run truncated (NTS_LAUNCH)" : ""); \
      return e_ == hipErrorOutOfMemory ? NTS_ENOMEM : NTS_EHIP;                                     \
    }                                                                                               \
  } while (0)

int fail(nts_ctx* ctx, int code, const std::string& msg)
{
  if (ctx) ctx->err = msg;
  return code;
}

// device scratch buffer `name` of at least `bytes` bytes (nullptr + ctx->err on failure)
void* ws_get(nts_ctx* ctx, const char* name, size_t bytes)
{
  auto& b = ctx->ws[name];
  if (b.second >= bytes && b.first) return b.first;
  if (b.first) {
    hipStreamSynchronize(ctx->stream);
    dev_free(b.first);
    b.first = nullptr;
    b.second = 0;
  }
  const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
  void* p = nullptr;
  hipError_t e = dev_malloc(&p, want);
  if (e != hipSuccess) {
    e = dev_malloc(&p, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) {
      ctx->err = std::string("hipMalloc scratch '") + name + "': " + hipGetErrorString(e);
      return nullptr;
    }
    b.second = std::max<size_t>(bytes, 256);
  } else {
    b.second = want;
  }
  b.first = p;
  return p;
}

void ws_release(nts_ctx* ctx)
{
  for (auto& kv : ctx->ws)
    if (kv.second.first) dev_free(kv.second.first);
  ctx->ws.clear();
}

// `want` lanes (stream + two pinned buffers of `chunk` bytes) of a transfer pool, created on first use; fewer if memory is short
constexpr uint64_t IO_CHUNK = (uint64_t)8 << 20;
unsigned io_lanes(nts_ctx* ctx, std::vector<nts_ctx::IoLane>& pool, unsigned want)
{
  while (pool.size() < want) {
    nts_ctx::IoLane l;
    if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess) break;
    if (hipHostMalloc((void**)&l.stage[0], IO_CHUNK) != hipSuccess || hipHostMalloc((void**)&l.stage[1], IO_CHUNK) != hipSuccess) {
      if (l.stage[0]) hipHostFree(l.stage[0]);
      hipStreamDestroy(l.stream);
      break;
    }
    pool.push_back(l);
  }
  return (unsigned)std::min<size_t>(pool.size(), want);
}

void io_release(std::vector<nts_ctx::IoLane>& pool)
{
  for (auto& l : pool) {
    hipStreamDestroy(l.stream);
    hipHostFree(l.stage[0]);
    hipHostFree(l.stage[1]);
  }
  pool.clear();
}

// ---- timing: HIP events on the context's stream around each kernel ---------------------------
struct ScopedTimer
{
  nts_ctx* ctx;
  const char* name;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  ScopedTimer(nts_ctx* c, const char* n, bool major = false)
    : ctx(c)
    , name(n)
    , on(c->profiling == 1 || (c->profiling == 2 && major))
  {
    if (on) {
      a = take();
      b = take();
      hipEventRecord(a, ctx->stream);
    }
  }
  hipEvent_t take()
  {
    hipEvent_t e = nullptr;
    if (!ctx->spare_events.empty()) {
      e = ctx->spare_events.back();
      ctx->spare_events.pop_back();
    } else {
      hipEventCreate(&e);
    }
    return e;
  }
  ~ScopedTimer()
  {
    if (on) {
      hipEventRecord(b, ctx->stream);
      ctx->pending.push_back({ name, { a, b } });
    }
  }
};

void drain_timings(nts_ctx* ctx)
{
  for (auto& p : ctx->pending) {
    hipEventSynchronize(p.second.second);
    float ms = 0;
    hipEventElapsedTime(&ms, p.second.first, p.second.second);
    auto& t = ctx->timings[p.first];
    t.ms += ms;
    t.launches += 1;
    ctx->spare_events.push_back(p.second.first);
    ctx->spare_events.push_back(p.second.second);
  }
  ctx->pending.clear();
}


__global__ __launch_bounds__(256) void k_bf_and(uint4* __restrict__ acc, const uint4* __restrict__ other, uint64_t n16)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) {
    uint4 a = acc[i];
    const uint4 o = other[i];
    a.x &= o.x;
    a.y &= o.y;
    a.z &= o.z;
    a.w &= o.w;
    acc[i] = a;
  }
}

__global__ __launch_bounds__(256) void k_bf_popcount(const uint4* __restrict__ words, uint64_t n16, unsigned long long* __restrict__ total)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (; i < n16; i += stride) {
    const uint4 v = words[i];
    acc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(total, acc);
}

// Exclusive prefix sum of n counts into 64-bit offsets by ONE workgroup (n is the number of tiles or workgroups of
// the previous kernel: thousands, at human scale ~2*10^5): one launch instead of the two a library scan takes, which
// is what counts when the whole sketch of a small genome is a few hundred microseconds.
constexpr uint32_t SCAN1_THREADS = 1024;
constexpr uint32_t SCAN1_ITEMS = 4;
template <typename T>
__global__ __launch_bounds__(SCAN1_THREADS) void k_scan_excl(const T* __restrict__ in, uint64_t n, uint64_t* __restrict__ out)
{
  __shared__ uint64_t s_wave[SCAN1_THREADS / 64];
  __shared__ uint64_t s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n; base += (uint64_t)SCAN1_THREADS * SCAN1_ITEMS) {
    const uint64_t i0 = base + (uint64_t)tid * SCAN1_ITEMS;
    uint64_t v[SCAN1_ITEMS], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < SCAN1_ITEMS; ++q) {
      v[q] = (i0 + q < n) ? (uint64_t)in[i0 + q] : 0ULL;
      sum += v[q];
    }
    uint64_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t up = __shfl_up(inc, d, 64);
      if ((int)lane >= d) inc += up;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint64_t before = s_carry + inc - sum;
    for (uint32_t q = 0; q < wv; ++q) before += s_wave[q];
#pragma unroll
    for (uint32_t q = 0; q < SCAN1_ITEMS; ++q) {
      if (i0 + q < n) out[i0 + q] = before;
      before += v[q];
    }
    __syncthreads();
    if (tid == SCAN1_THREADS - 1) s_carry = before;
    __syncthreads();
  }
}

// exclusive scan of per-tile / per-workgroup counts: one single-workgroup kernel while the list is short (one launch,
// ~4 us), the library's two-kernel scan beyond (a single workgroup would take ~0.1 ms over 2*10^5 counts)
constexpr uint64_t SCAN1_MAX = 8192;

template <typename T>
struct WidenU64
{
  __host__ __device__ uint64_t operator()(T x) const { return (uint64_t)x; }
};

template <typename T>
int scan_counts(nts_ctx* ctx, const T* d_in, uint64_t n, uint64_t* d_out)
{
  if (n <= SCAN1_MAX) {
    hipLaunchKernelGGL(k_scan_excl<T>, dim3(1), dim3(SCAN1_THREADS), 0, ctx->stream, d_in, n, d_out);
    return NTS_OK;
  }
  auto src = rocprim::make_transform_iterator(d_in, WidenU64<T>()); // (32-bit counts are widened on the way in)
  size_t bytes = 0;
  HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, bytes, src, d_out, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
  void* tmp = ws_get(ctx, "sel_scan_tmp", std::max<size_t>(bytes, 16));
  if (!tmp) return NTS_ENOMEM;
  HIP_TRY(ctx, rocprim::exclusive_scan(tmp, bytes, src, d_out, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
  return NTS_OK;
}

// device memory of a result list: recycled through a small per-context pool (hipMalloc/hipFree synchronise)
int alloc_result(nts_ctx* ctx, nts_mx* mx, uint64_t count)
{
  const uint64_t need = count * 20;
  for (size_t i = 0; i < ctx->mx_pool.size(); ++i) {
    if (ctx->mx_pool[i].second >= need && ctx->mx_pool[i].second <= 4 * need + (1u << 20)) {
      mx->d_h1 = (uint64_t*)ctx->mx_pool[i].first;
      mx->cap_bytes = ctx->mx_pool[i].second;
      ctx->mx_pool.erase(ctx->mx_pool.begin() + i);
      break;
    }
  }
  if (!mx->d_h1) {
    const uint64_t cap = need + need / 8 + 4096;
    HIP_TRY(ctx, dev_malloc((void**)&mx->d_h1, cap));
    mx->cap_bytes = cap;
  }
  mx->d_pos = mx->d_h1 + count;
  mx->d_rec = (uint32_t*)(mx->d_pos + count);
  return NTS_OK;
}

inline HashParams make_hash_params(uint32_t k)
{
  HashParams hp;
  const uint64_t seed[4] = { SEED_A, SEED_C, SEED_G, SEED_T };
  uint64_t rotk[4];
  for (int c = 0; c < 4; ++c) {
    uint64_t x = seed[c];
    for (uint32_t i = 0; i < k; ++i) x = srol1(x);
    rotk[c] = x;
    hp.seed[c] = seed[c];
  }
  for (int cin = 0; cin < 4; ++cin)
    for (int cout = 0; cout < 4; ++cout) {
      hp.roll_f[cin * 4 + cout] = seed[cin] ^ rotk[cout];
      hp.roll_r[cin * 4 + cout] = rotk[3 - cin] ^ seed[3 - cout];
    }
  hp.k = k;
  hp.init = nullptr;
  hp.init4 = nullptr;
  return hp;
}

inline FastMod make_fastmod(uint64_t m)
{
  FastMod fm;
  fm.m = m;
  // floor(2^64 / m) for m >= 2, not a power of two or otherwise: (2^64-1)/m differs only when m | 2^64
  unsigned __int128 one = ((unsigned __int128)1) << 64;
  fm.inv = (uint64_t)(one / m);
  fm.inv32 = (uint32_t)fm.inv;
  fm.m_lo = (uint32_t)m;
  fm.m_hi = (uint32_t)(m >> 32);
  fm.form = (fm.inv >> 32) ? 0u : ((m >> 38) ? 1u : 2u);
  if (const char* e = NTS_KNOB("NTS_FASTMOD_FORM")) fm.form = std::min<uint32_t>(fm.form, (uint32_t)atoi(e)); // (tests: the longer forms)
  return fm;
}

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_w, uint32_t* total)
{
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += y;
  }
  __syncthreads();
  if (lane == 63) s_w[wv] = x;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t q = 0; q < wv; ++q) base += s_w[q];
  *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + x - v;
}

} // namespace

extern "C" int nts_genome_finish_impl(nts_ctx* ctx, nts_genome* g); // ntsynt_hip.hip: stretches of valid bases + record table of a genome whose codes are in HBM
