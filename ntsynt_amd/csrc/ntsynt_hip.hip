// libntsynt_hip.so -- MI355X (gfx950) implementation of ntSynt's hot path behind the C ABI of
// include/ntsynt_hip.h.  HIP kernels for: sequence encoding + valid-stretch detection, canonical
// ntHash2 rolling hash with Bloom insert / probe, window-of-w rightmost-argmin over valid k-mers,
// Bloom AND / popcount, and the minimizer-graph join.  No MFMA: integer ALU + LDS + HBM.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   genome   : one byte per base, code 0..3 = A,C,G,T(U) (case-insensitive), 4 = anything else;
//              records concatenated, PAD invalid bytes in front and behind.
//   runs     : maximal intervals of consecutive valid k-mer starts ("run table"): run_pos[i] =
//              offset of the first k-mer, run_vstart[i] = its index in the compact (valid-k-mer)
//              numbering.  Windows are defined over that compact numbering.
//   keys     : one u64 per valid k-mer: h0, or KEY_MAX when the common Bloom filter rejects it.
//   bloom    : bit idx = h0 % bits at byte idx/8, bit idx%8 (LSB first).
#include "nts_internal.h"

namespace {

// =================================================================================================
// Kernels
// =================================================================================================

// ASCII -> code, in place.  16 bytes per lane, coalesced.
__global__ __launch_bounds__(256) void k_encode(uint8_t* __restrict__ buf, uint64_t n)
{
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= n) return;
  if (i + 16 <= n) {
    uint4 v = *reinterpret_cast<const uint4*>(buf + i);
    uint32_t wds[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t out = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t c = (wds[q] >> (8 * b)) & 0xFFu;
        c &= 0xDFu; // fold case
        uint32_t code = (c == 'A') ? 0u : (c == 'C') ? 1u : (c == 'G') ? 2u : (c == 'T' || c == 'U') ? 3u : 4u;
        out |= code << (8 * b);
      }
      wds[q] = out;
    }
    *reinterpret_cast<uint4*>(buf + i) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  } else {
    for (uint64_t j = i; j < n; ++j) {
      uint32_t c = buf[j] & 0xDFu;
      buf[j] = (c == 'A') ? 0 : (c == 'C') ? 1 : (c == 'G') ? 2 : (c == 'T' || c == 'U') ? 3 : 4;
    }
  }
}

// codes (one byte per base) -> 2 bits per base, 16 bases per word; a lane packs 16 bases
__global__ __launch_bounds__(256) void k_pack2(const uint8_t* __restrict__ code, uint64_t n_words, uint32_t* __restrict__ pack)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words) return;
  const uint4 v = *reinterpret_cast<const uint4*>(code + 16 * i);
  auto pack4 = [](uint32_t x) -> uint32_t { // bytes b0..b3 (2 significant bits each) -> b0 | b1<<2 | b2<<4 | b3<<6
    x &= 0x03030303u;
    x |= x >> 6;
    x |= x >> 12;
    return x & 0xFFu;
  };
  pack[i] = pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24);
}

// Boundaries of maximal valid stretches.  code points at base 0 (PAD bytes before it are invalid).
// pass 0: count starts; pass 1: append starts / ends (unordered; sorted afterwards).
template <int PASS>
__global__ __launch_bounds__(256) void k_stretch(const uint8_t* __restrict__ code,
                                                 uint64_t n,
                                                 unsigned long long* __restrict__ counter,
                                                 uint64_t* __restrict__ starts,
                                                 uint64_t* __restrict__ ends,
                                                 unsigned long long* __restrict__ cursors)
{
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (base >= n + 1) return;
  // bytes base-1 .. base+15 ; position n is the (invalid) pad byte so a trailing stretch gets its end.  The lane's sixteen bytes
  // in one 16-byte load (code + base is 16-byte aligned, PAD bytes follow the sequence), validity of all of them as a 16-bit
  // mask by word arithmetic -- byte by byte, lanes 16 bytes apart, this was 2.8 ms per pass over a 3 Gbp genome
  const uint4 v = *reinterpret_cast<const uint4*>(code + base);
  const bool prev = code[(int64_t)base - 1] < CODE_INVALID;
  auto invalid4 = [](uint32_t x) -> uint32_t { // bit b: byte b of x is an invalid code (>= 4)
    const uint32_t t = x & 0xFCFCFCFCu;
    const uint32_t nz = (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
    return (((nz >> 7) * 0x01020408u) >> 24) & 0xFu;
  };
  const uint32_t inv = invalid4(v.x) | (invalid4(v.y) << 4) | (invalid4(v.z) << 8) | (invalid4(v.w) << 12);
  const uint32_t valid = ~inv & 0xFFFFu;
  const uint64_t left = n + 1 - base; // positions base .. n count
  const uint32_t lim = left >= 16 ? 0xFFFFu : ((1u << (uint32_t)left) - 1u);
  const uint32_t diff = (valid ^ (((valid << 1) | (prev ? 1u : 0u)) & 0xFFFFu)) & lim;
  uint32_t st = diff & valid, en = diff & ~valid;
  if (PASS == 0) {
    if (st) atomicAdd(counter, (unsigned long long)__popc(st));
  } else {
    while (st) {
      const uint32_t j = (uint32_t)__ffs((int)st) - 1u;
      st &= st - 1u;
      starts[atomicAdd(&cursors[0], 1ULL)] = base + j;
    }
    while (en) {
      const uint32_t j = (uint32_t)__ffs((int)en) - 1u;
      en &= en - 1u;
      ends[atomicAdd(&cursors[1], 1ULL)] = base + j;
    }
  }
}

// ---- canonical ntHash over the compact k-mer numbering ------------------------------------------
// A workgroup owns KEY_TILE = 256 lanes x 32 consecutive compact indices.  Fast path (the whole tile
// lies in one run, k <= FAST_K_MAX): the tile's bases are staged into LDS with coalesced 16-byte
// loads, in a layout padded by 4 bytes per 32 (lane stride 36 B => conflict-free byte reads); each
// lane hashes its first k-mer directly and rolls 31 times.  Bloom probes are issued in batches of 8
// independent loads per lane to keep many HBM requests in flight.  Generic path (tile spans runs, as
// in the masked refinement rounds, or huge k): every lane walks the run table itself.
//   MODE_KEYS   : keys[phys(j)] = h0, or KEY_MAX if a filter is given and rejects h0   (rows B1+B2)
//   MODE_INSERT : bf_out |= bit(h0)                                                   (row A2)
//   MODE_CASCADE: if bf_in has bit(h0): bf_out |= bit(h0)                             (row A3, literal)
//   MODE_REPEAT : if bit(h0) was set in bf_in already: bf_out |= bit(h0); bf_in |= bit(h0)   (bin/ntsynt_make_repeat_bfs.py:56-67;
//                 the returning atomic makes "the second hit of a bit" well defined whatever the order of the lanes)
// Key layout in HBM is tile-transposed so that both this kernel's stores and the window kernel's loads
// coalesce: phys(j) = tile*8192 + (j%32)*256 + (j%8192)/32.
enum { MODE_KEYS = 0, MODE_INSERT = 1, MODE_CASCADE = 2, MODE_REPEAT = 3 };
constexpr uint32_t KEY_TILE = HASH_THREADS * HASH_PER_THREAD; // 8192
constexpr uint32_t FAST_K_MAX = 128;
constexpr uint32_t SEQ_LDS_DWORDS = 2400; // (15 + 8192 + 127) bytes in the padded layout, rounded up
static_assert(HASH_PER_THREAD == 32 && HASH_THREADS == 256, "key layout assumes 256 x 32 tiles");

__host__ __device__ __forceinline__ uint64_t key_phys(uint64_t j)
{
  const uint64_t tile = j / KEY_TILE;
  const uint32_t r = (uint32_t)(j % KEY_TILE);
  return tile * KEY_TILE + (uint64_t)(r & 31u) * 256u + (r >> 5);
}

template <int MODE>
__device__ __forceinline__ void hash_emit(uint64_t h0, bool live, uint64_t phys, const uint32_t* __restrict__ bf_in,
                                          uint32_t* __restrict__ bf_out, const FastMod& fm, uint64_t* __restrict__ keys,
                                          const uint32_t* __restrict__ bf_rep = nullptr, const FastMod* fm_rep = nullptr)
{
  if (MODE == MODE_KEYS) {
    uint64_t key = h0;
    if (bf_in != nullptr && !bf_test(bf_in, fm(h0))) key = KEY_MAX;
    if (bf_rep != nullptr && key != KEY_MAX && bf_test(bf_rep, (*fm_rep)(h0))) key = KEY_MAX; // filter-out (indexlr -r)
    if (live) keys[phys] = key;
  } else if (MODE == MODE_INSERT) {
    if (live) bf_set(bf_out, fm(h0));
  } else if (MODE == MODE_REPEAT) {
    if (live) {
      const uint64_t idx = fm(h0);
      const uint32_t bit = 1u << (idx & 31);
      const uint32_t old = atomicOr(const_cast<uint32_t*>(bf_in) + (idx >> 5), bit);
      if (old & bit) bf_set_unless_set(bf_out, idx);
    }
  } else {
    const uint64_t idx = fm(h0);
    if (live && bf_test(bf_in, idx)) bf_set(bf_out, idx);
  }
}

template <int MODE>
__global__ __launch_bounds__(HASH_THREADS) void k_hash(const uint8_t* __restrict__ code,
                                                       const uint64_t* __restrict__ run_pos,
                                                       const uint64_t* __restrict__ run_vstart,
                                                       uint32_t n_runs,
                                                       uint64_t n_valid,
                                                       HashParams hp,
                                                       const uint32_t* __restrict__ bf_in,
                                                       uint32_t* __restrict__ bf_out,
                                                       FastMod fm,
                                                       uint64_t* __restrict__ keys,
                                                       const uint32_t* __restrict__ tile_ids,
                                                       const uint2* __restrict__ tile_span,
                                                       const uint32_t* __restrict__ bf_rep,
                                                       FastMod fm_rep)
{
  // bf_rep (MODE_KEYS): filter-out Bloom filter, indexlr -r -- a k-mer present in it is rejected like one absent from bf_in
  // tile_span (with tile_ids): per listed tile the first and last in-tile index whose key anybody reads (the uncovered ranges
  // inside it); only those are probed, the rest are written as rejected -- a listed tile holds one or two ranges of ~w..3w
  // k-mers, and probing all 8192 was nine tenths of this pass
  __shared__ uint64_t s_tab[36];
  __shared__ uint32_t s_seq[SEQ_LDS_DWORDS];
  const uint32_t tid = threadIdx.x;
  if (tid < 16) {
    s_tab[tid] = hp.roll_f[tid];
    s_tab[16 + tid] = hp.roll_r[tid];
  }
  if (tid < 4) s_tab[32 + tid] = hp.seed[tid];
  const uint32_t k = hp.k;
  // tile_ids (optional): the key tiles to compute, for the dense fallback on uncovered ranges only
  const uint64_t J0 = (uint64_t)(tile_ids ? tile_ids[blockIdx.x] : blockIdx.x) * KEY_TILE;
  // with a tile list, the keys of the i-th listed tile go to slot i of a compact key buffer
  const uint64_t KB = (uint64_t)blockIdx.x * KEY_TILE;
  const uint32_t tile_len = (uint32_t)min((uint64_t)KEY_TILE, n_valid - J0);
  const uint2 span = (tile_ids && tile_span) ? tile_span[blockIdx.x] : make_uint2(0u, KEY_TILE - 1u);
  // run holding J0 (same for every lane: broadcast loads)
  uint32_t lo = 0, hi = n_runs;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (run_vstart[mid] <= J0)
      lo = mid;
    else
      hi = mid;
  }
  const uint64_t v0 = run_vstart[lo], v1 = run_vstart[lo + 1];
  const bool single = (J0 + tile_len <= v1) && (k <= FAST_K_MAX);
  if (single) {
    // ---- fast path --------------------------------------------------------------------------------
    const uint64_t P0 = run_pos[lo] + (J0 - v0);
    const uint32_t a = (uint32_t)(P0 & 15u);
    const uint8_t* src = code + (P0 - a);
    const uint32_t n_bytes = a + tile_len + k - 1;
    const uint32_t n16 = (n_bytes + 15u) >> 4;
    for (uint32_t c = tid; c < n16; c += HASH_THREADS) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + 16u * c);
      const uint32_t d = 4u * c + (c >> 1);
      s_seq[d] = v.x;
      s_seq[d + 1] = v.y;
      s_seq[d + 2] = v.z;
      s_seq[d + 3] = v.w;
    }
    __syncthreads();
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(s_seq);
    auto base_at = [&](uint32_t s) -> uint32_t { return sb[s + 4u * (s >> 5)] & 3u; };
    const uint32_t first = 32u * tid;
    const uint32_t n_mine = first < tile_len ? min(32u, tile_len - first) : 0u;
    uint32_t s = a + first;
    uint64_t f = 0, r = 0;
    if (n_mine) hash_init(hp, [&](uint32_t i) { return base_at(s + i); }, f, r);
    const uint64_t out_base = KB + tid;
#pragma unroll 1
    for (uint32_t b0 = 0; b0 < 32; b0 += 8) {
      uint64_t h[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        h[u] = f + r;
        const uint32_t cout = base_at(s), cin = base_at(s + k);
        f = srol1(f) ^ s_tab[cin * 4 + cout];
        r = sror1(r ^ s_tab[16 + cin * 4 + cout]);
        ++s;
      }
      if (MODE == MODE_KEYS) {
        if (bf_in != nullptr) {
          uint32_t wd[8], bit[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint64_t idx = fm(h[u]);
            const uint32_t e = first + b0 + (uint32_t)u;
            const bool wanted = e >= span.x && e <= span.y;
            wd[u] = bf_in[wanted ? idx >> 5 : 0ULL]; // (word 0 is cached; the result is not used)
            bit[u] = wanted ? (uint32_t)idx & 31u : 32u;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (bit[u] == 32u || !((wd[u] >> bit[u]) & 1u)) h[u] = KEY_MAX;
        }
        if (bf_rep != nullptr) {
          uint32_t wd[8], bit[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint64_t idx = fm_rep(h[u]);
            const bool wanted = h[u] != KEY_MAX; // (a k-mer rejected already reads word 0)
            wd[u] = bf_rep[wanted ? idx >> 5 : 0ULL];
            bit[u] = wanted ? (uint32_t)idx & 31u : 32u;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (bit[u] != 32u && ((wd[u] >> bit[u]) & 1u)) h[u] = KEY_MAX;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (b0 + u < n_mine) keys[out_base + (uint64_t)(b0 + u) * 256u] = h[u];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) hash_emit<MODE>(h[u], b0 + u < n_mine, 0, bf_in, bf_out, fm, keys);
      }
    }
    return;
  }
  // ---- generic path ---------------------------------------------------------------------------------
  __syncthreads();
  uint64_t j = J0 + 32ull * tid;
  if (j >= n_valid) return;
  const uint64_t j_end = min(j + (uint64_t)HASH_PER_THREAD, n_valid);
  lo = 0;
  hi = n_runs;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (run_vstart[mid] <= j)
      lo = mid;
    else
      hi = mid;
  }
  uint32_t ri = lo;
  while (j < j_end) {
    const uint64_t rv0 = run_vstart[ri], rv1 = run_vstart[ri + 1];
    const uint64_t seg_end = min(j_end, rv1);
    uint64_t p = run_pos[ri] + (j - rv0);
    uint64_t f = 0, r = 0;
    for (uint32_t i = 0; i < k; ++i) {
      f = srol1(f) ^ s_tab[32 + code[p + i]];
      r = srol1(r) ^ s_tab[32 + 3 - code[p + k - 1 - i]];
    }
    for (;;) {
      const uint32_t e = (uint32_t)(j - J0);
      if (MODE == MODE_KEYS && (e < span.x || e > span.y))
        keys[KB + (key_phys(j) - J0)] = KEY_MAX;
      else
        hash_emit<MODE>(f + r, true, KB + (key_phys(j) - J0), bf_in, bf_out, fm, keys, bf_rep, &fm_rep);
      ++j;
      if (j >= seg_end) break;
      const uint32_t cout = code[p], cin = code[p + k];
      f = srol1(f) ^ s_tab[cin * 4 + cout];
      r = sror1(r ^ s_tab[16 + cin * 4 + cout]);
      ++p;
    }
    ++ri;
  }
}

// ---- dense sketch over a sparse filter ------------------------------------------------------------------------------
// When the common filter is all but empty (many divergent genomes: BASELINE's 8 x 3 Gbp at 10 % leaves 3e-6 of the bits),
// almost every probe of the every-k-mer-probed path fetches a 128-byte line of zeros from HBM.  A summary with one bit per
// 2^shift filter bits (<= 4 MiB, mostly resident in the L2) answers those probes; only where the summary bit is set is the filter
// itself read.  Same keys as k_hash<MODE_KEYS> bit for bit; key tiles without a single accepted k-mer are not written at all
// and flagged in tile_any, so that the window kernel skips them.
__global__ __launch_bounds__(256) void k_bf_summary(const uint4* __restrict__ words, uint64_t n16, uint32_t shift, uint32_t* __restrict__ summary,
                                                    uint32_t* __restrict__ fold, uint32_t fold_words)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = words[i];
    if (v.x | v.y | v.z | v.w) {
      const uint64_t g = (i * 128u) >> shift; // shift >= 7: the 128 bits of a word share one granule
      atomicOr(&summary[g >> 5], 1u << (g & 31u));
      // the folded copy: bit (index mod 2^19) -- whole 32-bit words fold onto whole words (fold_words is a power of two)
      const uint32_t w0 = (uint32_t)((4 * i) & (fold_words - 1));
      if (v.x) atomicOr(&fold[w0], v.x);
      if (v.y) atomicOr(&fold[(w0 + 1) & (fold_words - 1)], v.y);
      if (v.z) atomicOr(&fold[(w0 + 2) & (fold_words - 1)], v.z);
      if (v.w) atomicOr(&fold[(w0 + 3) & (fold_words - 1)], v.w);
      // a second table behind the first, addressed by a multiplicative hash of the index (fold2_slot): another look in LDS that is
      // independent of the first and of the summary (k_hash_accept4r)
      const uint32_t vv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t m = vv[q];
        while (m) {
          const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
          m &= m - 1u;
          const uint32_t s2 = fold2_slot(i * 128u + 32u * q + b);
          atomicOr(&fold[fold_words + (s2 >> 5)], 1u << (s2 & 31u));
        }
      }
    }
  }
}

__global__ __launch_bounds__(HASH_THREADS) void k_hash_keys_sparse(const uint8_t* __restrict__ code, const uint64_t* __restrict__ run_pos,
                                                                   const uint64_t* __restrict__ run_vstart, uint32_t n_runs, uint64_t n_valid,
                                                                   HashParams hp, const uint32_t* __restrict__ bf_in, FastMod fm,
                                                                   const uint32_t* __restrict__ summary, uint32_t shift,
                                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ tile_any)
{
  __shared__ uint64_t s_tab[36];
  __shared__ uint32_t s_seq[SEQ_LDS_DWORDS];
  const uint32_t tid = threadIdx.x;
  if (tid < 16) {
    s_tab[tid] = hp.roll_f[tid];
    s_tab[16 + tid] = hp.roll_r[tid];
  }
  if (tid < 4) s_tab[32 + tid] = hp.seed[tid];
  const uint32_t k = hp.k;
  const uint64_t J0 = (uint64_t)blockIdx.x * KEY_TILE;
  const uint32_t tile_len = (uint32_t)min((uint64_t)KEY_TILE, n_valid - J0);
  uint32_t lo = 0, hi = n_runs;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (run_vstart[mid] <= J0)
      lo = mid;
    else
      hi = mid;
  }
  const uint64_t v0 = run_vstart[lo], v1 = run_vstart[lo + 1];
  const bool single = (J0 + tile_len <= v1) && (k <= FAST_K_MAX);
  auto accepted = [&](uint64_t h) -> bool {
    const uint64_t idx = fm(h);
    const uint64_t g = idx >> shift;
    if (!((summary[g >> 5] >> (g & 31u)) & 1u)) return false;
    return (bf_in[idx >> 5] >> ((uint32_t)idx & 31u)) & 1u;
  };
  if (single) {
    const uint64_t P0 = run_pos[lo] + (J0 - v0);
    const uint32_t a = (uint32_t)(P0 & 15u);
    const uint8_t* src = code + (P0 - a);
    const uint32_t n_bytes = a + tile_len + k - 1;
    const uint32_t n16 = (n_bytes + 15u) >> 4;
    for (uint32_t c = tid; c < n16; c += HASH_THREADS) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + 16u * c);
      const uint32_t d = 4u * c + (c >> 1);
      s_seq[d] = v.x;
      s_seq[d + 1] = v.y;
      s_seq[d + 2] = v.z;
      s_seq[d + 3] = v.w;
    }
    __syncthreads();
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(s_seq);
    auto base_at = [&](uint32_t s) -> uint32_t { return sb[s + 4u * (s >> 5)] & 3u; };
    const uint32_t first = 32u * tid;
    const uint32_t n_mine = first < tile_len ? min(32u, tile_len - first) : 0u;
    uint32_t s = a + first;
    uint64_t f = 0, r = 0;
    if (n_mine) hash_init(hp, [&](uint32_t i) { return base_at(s + i); }, f, r);
    uint32_t acc_mask = 0;
    uint64_t acc_h[4]; // the lane's first accepted hashes (accepted k-mers are rare here); more: recomputed below
    uint32_t n_acc = 0;
#pragma unroll 1
    for (uint32_t b0 = 0; b0 < 32; b0 += 8) {
      uint64_t h[8], idx[8];
      uint32_t sw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        h[u] = f + r;
        const uint32_t cout = base_at(s), cin = base_at(s + k);
        f = srol1(f) ^ s_tab[cin * 4 + cout];
        r = sror1(r ^ s_tab[16 + cin * 4 + cout]);
        ++s;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        idx[u] = fm(h[u]);
        sw[u] = summary[idx[u] >> (shift + 5)];
      }
      uint32_t pass = 0, fw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + u < n_mine && ((sw[u] >> ((idx[u] >> shift) & 31u)) & 1u)) pass |= 1u << u;
      // the filter reads of the few k-mers that passed, as one round of independent loads (the others re-read word 0)
#pragma unroll
      for (int u = 0; u < 8; ++u) fw[u] = bf_in[((pass >> u) & 1u) ? (idx[u] >> 5) : 0];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (((pass >> u) & 1u) && ((fw[u] >> ((uint32_t)idx[u] & 31u)) & 1u)) {
          acc_mask |= 1u << (b0 + u);
          if (n_acc < 4) acc_h[n_acc] = h[u];
          ++n_acc;
        }
      }
    }
    const int any = __syncthreads_or(acc_mask != 0);
    if (tid == 0) tile_any[blockIdx.x] = any ? 1u : 0u;
    if (!any) return;
    // (rare) the tile holds accepted k-mers: all of its keys are written
    uint64_t* out = keys + J0 + tid;
    if (n_acc <= 4) {
      uint32_t q = 0;
      for (uint32_t u = 0; u < n_mine; ++u) {
        uint64_t key = KEY_MAX;
        if ((acc_mask >> u) & 1u) key = acc_h[q++];
        out[(uint64_t)u * 256u] = key;
      }
    } else {
      // more accepted k-mers in one lane than were kept: hash the lane's k-mers once more
      uint32_t s2 = a + first;
      uint64_t f2 = 0, r2 = 0;
      hash_init(hp, [&](uint32_t i) { return base_at(s2 + i); }, f2, r2);
      for (uint32_t u = 0; u < n_mine; ++u) {
        out[(uint64_t)u * 256u] = ((acc_mask >> u) & 1u) ? f2 + r2 : KEY_MAX;
        const uint32_t cout = base_at(s2), cin = base_at(s2 + k);
        f2 = srol1(f2) ^ s_tab[cin * 4 + cout];
        r2 = sror1(r2 ^ s_tab[16 + cin * 4 + cout]);
        ++s2;
      }
    }
    return;
  }
  // ---- generic path (tile crosses runs): every key written, the tile flagged ---------------------------------------
  __syncthreads();
  if (tid == 0) tile_any[blockIdx.x] = 1u;
  uint64_t j = J0 + 32ull * tid;
  if (j >= n_valid) return;
  const uint64_t j_end = min(j + (uint64_t)HASH_PER_THREAD, n_valid);
  lo = 0;
  hi = n_runs;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (run_vstart[mid] <= j)
      lo = mid;
    else
      hi = mid;
  }
  uint32_t ri = lo;
  while (j < j_end) {
    const uint64_t rv0 = run_vstart[ri], rv1 = run_vstart[ri + 1];
    const uint64_t seg_end = min(j_end, rv1);
    uint64_t p = run_pos[ri] + (j - rv0);
    uint64_t f = 0, r = 0;
    for (uint32_t i = 0; i < k; ++i) {
      f = srol1(f) ^ s_tab[32 + code[p + i]];
      r = srol1(r) ^ s_tab[32 + 3 - code[p + k - 1 - i]];
    }
    for (;;) {
      const uint64_t h = f + r;
      keys[key_phys(j)] = accepted(h) ? h : KEY_MAX;
      ++j;
      if (j >= seg_end) break;
      const uint32_t cout = code[p], cin = code[p + k];
      f = srol1(f) ^ s_tab[cin * 4 + cout];
      r = sror1(r ^ s_tab[16 + cin * 4 + cout]);
      ++p;
    }
    ++ri;
  }
}

// plain (untransposed) copy of the keys, for the nts_hash_all test hook
__global__ __launch_bounds__(256) void k_keys_linear(const uint64_t* __restrict__ keys, uint64_t n, uint64_t* __restrict__ out)
{
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = keys[key_phys(j)];
}

// ---- window-of-w rightmost argmin over the compact keys of one record ------------------------------
// A workgroup owns WIN_TILE consecutive windows of one record; it loads the E = windows + w - 1 keys it
// needs into LDS (un-transposing the key layout on the way; LDS index e + e/32 keeps both the
// transposing stores and the later strided reads conflict-free).  Per chunk of c = min(16, w) keys a
// lane finds the chunk's argmin; a sparse table over the chunk argmins answers "whole chunks" ranges.
// Each lane then walks its contiguous slice of windows the way a sequential scan would: one full
// range query for its first window, then O(1) per window (the new key replaces the winner if it is
// <=, btllib's rightmost-minimum rule); only when the winner slides out of the window is the range
// query repeated.  A winner is emitted when it differs from the previous window's winner and its key
// is not KEY_MAX.
struct WinParams
{
  const uint64_t* keys;       // compact keys, tile-transposed (key_phys)
  const uint32_t* tile_ids;   // if not null: `keys` holds only these key tiles (ascending), tile_ids[i] in slot i
  uint32_t n_tile_ids;
  const uint64_t* rec_vstart; // [n_rec] compact index of the record's first valid k-mer
  const uint64_t* rec_nv;     // [n_rec] valid k-mers in the record
  const uint64_t* tile_start; // [n_rec+1] prefix sum of tiles per record
  uint32_t n_rec;
  uint32_t w;
  uint32_t chunk;
  uint32_t levels;            // sparse-table levels
  uint64_t* out_j;            // compact index of each emitted minimizer: N_SEG segments of seg_cap slots,
  uint64_t* out_key;          // its key (= h0)                          pre-filled with ~0 (sorts last)
  unsigned long long* seg_count; // [N_SEG] slots reserved per segment (one returning atomic per workgroup)
  uint64_t seg_cap;
  // per-tile mode (few tiles: the uncovered ranges of a pruned sketch): tile b writes its winners in index order to
  // out_j/out_key[b * tile_cap ...] and their number to tile_cnt[b] (more than tile_cap: nothing written, the
  // collector sees the count) -- tiles are in index order, so no sort is needed afterwards
  uint32_t* tile_cnt;
  uint32_t tile_cap;
  // if not null (whole-genome dense pass over a sparse filter): tile_any[t] == 0 <=> key tile t holds no accepted k-mer and
  // was not written; its keys count as KEY_MAX
  const uint32_t* tile_any;
  // k_window_min<true> (short windows, whole genome): no key array -- the tile hashes and probes its own k-mers
  const uint8_t* code;
  const uint64_t* run_pos;
  const uint64_t* run_vstart;
  uint32_t n_runs;
  HashParams hp;
  const uint32_t* bf;
  FastMod fm;
  // k_window_min<true>, whole genome: a tile writes its winners IN INDEX ORDER into the room it reserves in one of the N_SEG
  // segments and leaves (offset, count) in a directory; a scan of the counts and one gather (k_cand_compact_slots) then give the
  // ordered list -- no fill of the segments, no radix sort.  (A look-back chain for the tile's final offset was tried first: tiles
  // that wait for their predecessors' counts hold their LDS and wave slots, and the probes in flight halve: 166 instead of 84 ms.)
  uint64_t* dir_off;  // [n_tiles] first slot of the tile's winners in out_j / out_key (~0: its reservation did not fit)
  uint32_t* dir_cnt;  // [n_tiles]
  uint32_t tile_base = 0; // first tile of this launch (a launch holds fewer than 2^32 / WIN_THREADS tiles: uncovered ranges by the million)
};
constexpr uint32_t N_SEG = 64;
// Short windows (w < WIN_FUSE_W: the last refinement round's w = 10, -d < 1's defaults bin/ntSynt:89-91, and anything below 64): the
// pruned selection does not apply (its tiles list c/w of their k-mers: everything), so every k-mer is hashed and probed.  Through
// the key array that is 8 bytes written and read back per k-mer (2 x 24 GB per 3 Gbp genome: k_hash<MODE_KEYS> 68 ms + k_window_min
// 21 ms); the halo a window tile shares with its neighbour is (w - 1) / 4096 < 2 % here, so the tile computes the keys it needs
// itself, in LDS, and the array is never made.  (At w = 1000 the halo is 24 %: there the array stays.)
constexpr uint32_t WIN_FUSE_W = 64;
constexpr uint32_t WIN_FUSE_PER = (WIN_TILE + 1 + WIN_FUSE_W - 2 + WIN_THREADS - 1) / WIN_THREADS; // elements a lane hashes: 9

__device__ __forceinline__ uint32_t pe(uint32_t e)
{
  return e + (e >> 5);
}

// slot of a key tile in the key buffer: the tile itself, or its rank in the list of tiles that were computed
__device__ __forceinline__ uint64_t win_key_slot(const WinParams& P, uint64_t tile)
{
  if (P.tile_ids == nullptr) return tile;
  uint32_t lo = 0, hi = P.n_tile_ids;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (P.tile_ids[mid] <= tile)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// smaller key wins, ties to the larger index
__device__ __forceinline__ uint32_t better_idx(const uint64_t* s_key, uint32_t a, uint32_t b)
{
  const uint64_t ka = s_key[pe(a)], kb = s_key[pe(b)];
  if (ka < kb) return a;
  if (kb < ka) return b;
  return a > b ? a : b;
}

template <bool FUSED>
__global__ __launch_bounds__(WIN_THREADS) void k_window_min(WinParams P)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t lo = 0, hi = P.n_rec;
  const uint32_t blk = blockIdx.x + P.tile_base;
  const uint64_t b = blk;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (P.tile_start[mid] <= b)
      lo = mid;
    else
      hi = mid;
  }
  const uint32_t rec = lo;
  const uint64_t nv = P.rec_nv[rec];
  const uint32_t w = P.w;
  const uint64_t n_win_rec = nv - w + 1;
  const uint64_t t0 = (b - P.tile_start[rec]) * (uint64_t)WIN_TILE;
  const uint32_t cnt = (uint32_t)min((uint64_t)WIN_TILE, n_win_rec - t0);
  const uint64_t tf = t0 > 0 ? t0 - 1 : 0;      // first window evaluated (one of overlap)
  const uint32_t n_win = (uint32_t)(t0 + cnt - tf);
  const uint32_t E = n_win + w - 1;             // elements tf .. tf+E-1 of the record
  const uint32_t c = P.chunk;
  const uint32_t n_chunks = (E + c - 1) / c;

  const uint64_t ja = P.rec_vstart[rec] + tf, jb = ja + E;
  if (P.tile_any != nullptr) {
    // no accepted k-mer anywhere in the span: no window of this tile has a minimizer
    bool any = false;
    for (uint64_t tile = ja / KEY_TILE; tile * KEY_TILE < jb; ++tile) any |= P.tile_any[tile] != 0;
    if (!any) return;
  }
  uint64_t* s_key = reinterpret_cast<uint64_t*>(smem);
  uint32_t* s_ctl = reinterpret_cast<uint32_t*>(s_key + pe(E) + 1); // [0] emitted count, [1..2] reserved base
  uint16_t* s_list = reinterpret_cast<uint16_t*>(s_ctl + 4);         // winners of this tile (<= n_win)
  uint16_t* s_st = s_list + ((n_win + 8) & ~7u);                     // levels x n_chunks
  if (threadIdx.x == 0) s_ctl[0] = 0;

  if (FUSED) {
    // ---- compute: the tile's own keys (h0 of every k-mer of [ja, jb), KEY_MAX where the filter rejects it) -------------------
    uint64_t* s_tab = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(s_st + (size_t)P.levels * n_chunks) + 15u) & ~(uintptr_t)15u); // [36] roll tables, seeds
    uint8_t* s_seq = reinterpret_cast<uint8_t*>(s_tab + 36); // the tile's bases, one byte each (16-byte aligned: 36 words = 288 bytes)
    const uint32_t tid = threadIdx.x, k = P.hp.k;
    if (tid < 16) {
      s_tab[tid] = P.hp.roll_f[tid];
      s_tab[16 + tid] = P.hp.roll_r[tid];
    }
    if (tid < 4) s_tab[32 + tid] = P.hp.seed[tid];
    // run holding ja (same for every lane: broadcast loads)
    uint32_t rl = 0, rh = P.n_runs;
    while (rh - rl > 1) {
      const uint32_t mid = rl + ((rh - rl) >> 1);
      if (P.run_vstart[mid] <= ja)
        rl = mid;
      else
        rh = mid;
    }
    const uint64_t v0 = P.run_vstart[rl], v1 = P.run_vstart[rl + 1];
    const bool single = jb <= v1 && k <= FAST_K_MAX;
    const uint32_t e_lo = tid * WIN_FUSE_PER;
    const uint32_t n_mine = e_lo < E ? min(WIN_FUSE_PER, E - e_lo) : 0u;
    if (single) {
      const uint64_t P0 = P.run_pos[rl] + (ja - v0);
      const uint32_t a = (uint32_t)(P0 & 15u);
      const uint8_t* src = P.code + (P0 - a);
      const uint32_t n16 = (a + E + k - 1 + 15u) >> 4;
      for (uint32_t cidx = tid; cidx < n16; cidx += WIN_THREADS)
        reinterpret_cast<uint4*>(s_seq)[cidx] = *reinterpret_cast<const uint4*>(src + 16u * cidx);
    }
    __syncthreads(); // (uniform: the tables, and the staged bases of a tile inside one run)
    if (single) {
      const uint64_t P0 = P.run_pos[rl] + (ja - v0);
      const uint32_t a = (uint32_t)(P0 & 15u);
      auto base_at = [&](uint32_t x) -> uint32_t { return s_seq[x] & 3u; };
      uint32_t sx = a + e_lo;
      uint64_t f = 0, r = 0;
      if (n_mine) hash_init(P.hp, [&](uint32_t i) { return base_at(sx + i); }, f, r);
      uint64_t h[WIN_FUSE_PER];
#pragma unroll
      for (int u = 0; u < (int)WIN_FUSE_PER; ++u) {
        h[u] = f + r;
        if ((uint32_t)u + 1u < n_mine) { // (the bases past the lane's last k-mer may lie outside what was staged)
          const uint32_t cout = base_at(sx), cin = base_at(sx + k);
          f = srol1(f) ^ s_tab[cin * 4 + cout];
          r = sror1(r ^ s_tab[16 + cin * 4 + cout]);
          ++sx;
        }
      }
      if (P.bf != nullptr) {
        uint32_t wd[WIN_FUSE_PER], bit[WIN_FUSE_PER];
#pragma unroll
        for (int u = 0; u < (int)WIN_FUSE_PER; ++u) {
          const bool wanted = (uint32_t)u < n_mine;
          const uint64_t idx = P.fm(h[u]);
          wd[u] = P.bf[wanted ? idx >> 5 : 0ULL]; // (word 0 is cached; the result is not used)
          bit[u] = wanted ? (uint32_t)idx & 31u : 32u;
        }
#pragma unroll
        for (int u = 0; u < (int)WIN_FUSE_PER; ++u)
          if (bit[u] == 32u || !((wd[u] >> bit[u]) & 1u)) h[u] = KEY_MAX;
      }
#pragma unroll
      for (int u = 0; u < (int)WIN_FUSE_PER; ++u)
        if ((uint32_t)u < n_mine) s_key[pe(e_lo + (uint32_t)u)] = h[u];
    } else if (n_mine) {
      // a tile over several runs (N stretches inside the record) or a very long k: every lane walks the run table itself
      uint64_t j = ja + e_lo;
      const uint64_t j_end = j + n_mine;
      uint32_t ri = rl;
      rh = P.n_runs;
      while (rh - ri > 1) {
        const uint32_t mid = ri + ((rh - ri) >> 1);
        if (P.run_vstart[mid] <= j)
          ri = mid;
        else
          rh = mid;
      }
      while (j < j_end) {
        const uint64_t rv0 = P.run_vstart[ri], rv1 = P.run_vstart[ri + 1];
        const uint64_t seg_end = min(j_end, rv1);
        uint64_t p = P.run_pos[ri] + (j - rv0);
        uint64_t f = 0, r = 0;
        for (uint32_t i = 0; i < k; ++i) {
          f = srol1(f) ^ s_tab[32 + P.code[p + i]];
          r = srol1(r) ^ s_tab[32 + 3 - P.code[p + k - 1 - i]];
        }
        for (;;) {
          uint64_t key = f + r;
          if (P.bf != nullptr && !bf_test(P.bf, P.fm(key))) key = KEY_MAX;
          s_key[pe((uint32_t)(j - ja))] = key;
          ++j;
          if (j >= seg_end) break;
          const uint32_t cout = P.code[p], cin = P.code[p + k];
          f = srol1(f) ^ s_tab[cin * 4 + cout];
          r = sror1(r ^ s_tab[16 + cin * 4 + cout]);
          ++p;
        }
        ++ri;
      }
    }
  } else
  // ---- load: walk the key tiles the range [ja, jb) touches, one transposed row at a time -----------
  for (uint64_t tile = ja / KEY_TILE; tile * KEY_TILE < jb; ++tile) {
    const uint64_t tb = tile * KEY_TILE;
    const bool written = P.tile_any == nullptr || P.tile_any[tile] != 0;
    const uint32_t r_lo = (uint32_t)(max(ja, tb) - tb);
    const uint32_t r_hi = (uint32_t)(min(jb, tb + KEY_TILE) - 1 - tb);
    const uint32_t col_lo = r_lo >> 5, col_hi = r_hi >> 5;
    // 256 columns x 32 rows per key tile; the workgroup's WIN_THREADS / 256 thread groups split the rows
    constexpr int ROWS = 32 / (WIN_THREADS / 256);
    const uint32_t col = col_lo + (threadIdx.x & 255u);
    const uint32_t row0 = (threadIdx.x >> 8) * ROWS;
    if (col <= col_hi) {
      const uint64_t* g = P.keys + win_key_slot(P, tile) * KEY_TILE + col;
      const int64_t e0 = (int64_t)(tb + 32ull * col) - (int64_t)ja; // element index of row 0
      uint64_t v[ROWS];
#pragma unroll
      for (int i = 0; i < ROWS; ++i) v[i] = written ? g[(uint64_t)(row0 + i) * 256u] : KEY_MAX;
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        const int64_t e = e0 + row0 + i;
        if (e >= 0 && e < (int64_t)E) s_key[pe((uint32_t)e)] = v[i];
      }
    }
  }
  __syncthreads();

  // rightmost argmin of the span [a, z) (at most 16 keys): the keys are fetched with independent LDS reads
  // first, so the compare chain does not wait for one LDS round trip per element
  auto span_argmin = [&](uint32_t a, uint32_t z, uint32_t best, uint64_t kb, bool have) -> uint32_t {
    uint64_t kv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) kv[q] = (a + q < z) ? s_key[pe(a + q)] : KEY_MAX;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (a + q < z && (!have || kv[q] <= kb)) {
        kb = kv[q];
        best = a + q;
        have = true;
      }
    }
    return best;
  };
  // ---- chunk argmins + sparse table ------------------------------------------------------------------
  for (uint32_t ch = threadIdx.x; ch < n_chunks; ch += WIN_THREADS) {
    const uint32_t a = ch * c, z = min(a + c, E);
    s_st[ch] = (uint16_t)span_argmin(a, z, a, 0, false);
  }
  __syncthreads();
  for (uint32_t L = 1; L < P.levels; ++L) {
    const uint32_t half = 1u << (L - 1);
    const uint16_t* prev = s_st + (size_t)(L - 1) * n_chunks;
    uint16_t* cur = s_st + (size_t)L * n_chunks;
    for (uint32_t i = threadIdx.x; i + 2 * half <= n_chunks; i += WIN_THREADS)
      cur[i] = (uint16_t)better_idx(s_key, prev[i], prev[i + half]);
    __syncthreads();
  }

  // rightmost argmin of elements [e, e+w-1]
  auto range_query = [&](uint32_t e) -> uint32_t {
    const uint32_t last = e + w - 1;
    const uint32_t ca = e / c, cb = last / c;
    const uint32_t head_end = (cb > ca) ? (ca + 1) * c : last + 1;
    uint32_t best = span_argmin(e, head_end, e, 0, false);
    if (cb > ca) {
      if (cb - ca >= 2) {
        const uint32_t len = cb - ca - 1;
        const uint32_t L = 31 - __clz(len);
        const uint16_t* lvl = s_st + (size_t)L * n_chunks;
        best = better_idx(s_key, best, lvl[ca + 1]);
        best = better_idx(s_key, best, lvl[cb - (1u << L)]);
      }
      best = span_argmin(cb * c, last + 1, best, s_key[pe(best)], true);
    }
    return best;
  };

  const uint32_t per = (n_win + WIN_THREADS - 1) / WIN_THREADS;
  const uint32_t w_lo = threadIdx.x * per;
  const uint32_t w_hi = min(w_lo + per, n_win);
  if (w_lo < w_hi) {
    // keys entering the lane's windows, fetched up front (independent LDS reads)
    constexpr int MAX_PER = (WIN_TILE + 1 + WIN_THREADS - 1) / WIN_THREADS; // 17
    uint64_t kin[MAX_PER];
#pragma unroll
    for (int i = 0; i < MAX_PER; ++i) {
      const uint32_t x = w_lo + i;
      kin[i] = (x < w_hi) ? s_key[pe(x + w - 1)] : KEY_MAX;
    }
    uint32_t cur = range_query(w_lo > 0 ? w_lo - 1 : 0);
    uint64_t kcur = s_key[pe(cur)];
    if (w_lo == 0 && t0 == 0 && kcur != KEY_MAX) s_list[atomicAdd(&s_ctl[0], 1u)] = (uint16_t)cur; // very first window
#pragma unroll
    for (int i = 0; i < MAX_PER; ++i) {
      const uint32_t e = w_lo + i;
      if (e == 0 || e >= w_hi) continue; // window 0 of the tile is the first window or the overlap: handled above
      uint32_t nxt = cur;
      uint64_t knx = kcur;
      if (cur < e) {
        nxt = range_query(e);
        knx = s_key[pe(nxt)];
      } else if (kin[i] <= kcur) {
        nxt = e + w - 1;
        knx = kin[i];
      }
      if (nxt != cur && knx != KEY_MAX) s_list[atomicAdd(&s_ctl[0], 1u)] = (uint16_t)nxt;
      cur = nxt;
      kcur = knx;
    }
  }
  __syncthreads();
  const uint32_t n_emit = s_ctl[0];
  if (P.tile_cnt != nullptr) {
    // ---- per-tile mode: rank the tile's winners (distinct indices, a handful) by counting, write them in order ----
    if (threadIdx.x == 0) P.tile_cnt[blk] = n_emit;
    if (n_emit == 0 || n_emit > P.tile_cap) return;
    const uint64_t jb = P.rec_vstart[rec] + tf;
    for (uint32_t i = threadIdx.x; i < n_emit; i += WIN_THREADS) {
      const uint32_t idx = s_list[i];
      uint32_t rank = 0;
      for (uint32_t x = 0; x < n_emit; ++x) rank += s_list[x] < idx ? 1u : 0u;
      P.out_j[(uint64_t)blk * P.tile_cap + rank] = jb + idx;
      P.out_key[(uint64_t)blk * P.tile_cap + rank] = s_key[pe(idx)];
    }
    return;
  }
  if (FUSED && P.dir_cnt != nullptr) {
    // ---- ordered flush: room in a segment, the winners in index order inside it, (offset, count) into the directory ------------
    uint64_t* s_tab = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(s_st + (size_t)P.levels * n_chunks) + 15u) & ~(uintptr_t)15u);
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_tab + 36); // (the tile's staged bases are not needed any more)
    const uint32_t words = (E + 31u) >> 5;                      // <= 131
    uint32_t* s_pref = s_bits + words;
    const uint32_t seg = blk % N_SEG;
    if (threadIdx.x == 0) {
      unsigned long long base = n_emit ? atomicAdd(&P.seg_count[seg], (unsigned long long)n_emit) : 0ull;
      const bool fits = base + n_emit <= P.seg_cap;
      P.dir_cnt[blk] = n_emit;
      P.dir_off[blk] = fits ? (uint64_t)seg * P.seg_cap + base : ~0ULL;
      if (!fits) base = ~0ull;
      s_ctl[1] = (uint32_t)base;
      s_ctl[2] = (uint32_t)(base >> 32);
    }
    // rank of a winner inside the tile: winners are distinct element indices below E -- a bitmap and the popcounts in front of each word
    for (uint32_t q = threadIdx.x; q < words; q += WIN_THREADS) s_bits[q] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_emit; i += WIN_THREADS) atomicOr(&s_bits[s_list[i] >> 5], 1u << (s_list[i] & 31u));
    __syncthreads();
    if (threadIdx.x < words) {
      uint32_t acc = 0;
      for (uint32_t q = 0; q < threadIdx.x; ++q) acc += __popc(s_bits[q]);
      s_pref[threadIdx.x] = acc;
    }
    __syncthreads();
    const uint64_t base = ((uint64_t)s_ctl[2] << 32) | s_ctl[1];
    if (base == ~0ULL) return; // (the segment is full: the host sees its counter and runs the pass again with larger segments)
    const uint64_t jbase = P.rec_vstart[rec] + tf;
    for (uint32_t i = threadIdx.x; i < n_emit; i += WIN_THREADS) {
      const uint32_t idx = s_list[i];
      const uint64_t slot = (uint64_t)seg * P.seg_cap + base + s_pref[idx >> 5] + __popc(s_bits[idx >> 5] & ((1u << (idx & 31u)) - 1u));
      P.out_j[slot] = jbase + idx;
      P.out_key[slot] = s_key[pe(idx)];
    }
    return;
  }
  // ---- flush: one returning atomic per workgroup, on one of N_SEG counters ---------------------------
  if (n_emit == 0) return;
  const uint32_t seg = blk % N_SEG;
  if (threadIdx.x == 0) {
    const unsigned long long base = atomicAdd(&P.seg_count[seg], (unsigned long long)n_emit);
    s_ctl[1] = (uint32_t)base;
    s_ctl[2] = (uint32_t)(base >> 32);
  }
  __syncthreads();
  const uint64_t base = ((uint64_t)s_ctl[2] << 32) | s_ctl[1];
  const uint64_t jbase = P.rec_vstart[rec] + tf;
  for (uint32_t i = threadIdx.x; i < n_emit; i += WIN_THREADS) {
    const uint64_t slot = base + i;
    if (slot < P.seg_cap) {
      const uint32_t idx = s_list[i];
      P.out_j[(uint64_t)seg * P.seg_cap + slot] = jbase + idx;
      P.out_key[(uint64_t)seg * P.seg_cap + slot] = s_key[pe(idx)];
    }
  }
}

// ---- bench/test utility: synthetic genome generated in HBM, and read-back of a slice ----------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// base i of the ancestor = mix64(seed_anc, i) & 3; genome = ancestor with substitutions at rate thr / 2^32
__global__ __launch_bounds__(256) void k_synth(uint8_t* __restrict__ code, uint64_t n, uint64_t seed_anc, uint64_t seed_gen, uint32_t thr)
{
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i0 >= n) return;
  uint32_t wds[4] = { 0, 0, 0, 0 };
  for (int b = 0; b < 16; ++b) {
    const uint64_t i = i0 + b;
    uint32_t base = (uint32_t)(mix64(seed_anc + i * 0x9E3779B97F4A7C15ULL) & 3u);
    const uint64_t y = mix64(seed_gen ^ (i * 0xD1B54A32D192ED03ULL));
    if ((uint32_t)y < thr) base = (base + 1u + (uint32_t)((y >> 32) % 3u)) & 3u;
    if (i >= n) base = CODE_INVALID;
    wds[b >> 2] |= base << (8 * (b & 3));
  }
  if (i0 + 16 <= n) {
    *reinterpret_cast<uint4*>(code + i0) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  } else {
    for (uint64_t i = i0; i < n; ++i) code[i] = (wds[(i - i0) >> 2] >> (8 * ((i - i0) & 3))) & 0xFF;
  }
}

// The assembly-like ancestor (nts_synth_repeats, include/ntsynt_hip.h): interspersed repeat copies as a function of the ancestor
// coordinate.  ntsynt_amd/synth.py `_anc_bases` is the numpy statement of the same arithmetic.
struct SynthRep
{
  uint32_t on;
  nts_synth_repeats r;
};

__device__ __forceinline__ bool synth_layer(uint64_t s, uint64_t seed, uint64_t salt, uint32_t cell_log2, uint32_t len_full, uint32_t len_min,
                                            uint32_t prob, uint32_t nfam, uint32_t div_min, uint32_t div_max, uint32_t& base)
{
  if (prob == 0) return false;
  const uint64_t cell = s >> cell_log2;
  const uint32_t o = (uint32_t)(s & ((1ull << cell_log2) - 1ull));
  const uint64_t h = mix64((seed ^ salt) + cell * 0xA24BAED4963EE407ULL);
  if ((uint32_t)(h & 255u) >= prob) return false;
  const uint32_t len = len_min + (uint32_t)((h >> 8) & 0xFFFFu) % (len_full - len_min + 1u);
  const uint32_t off = (uint32_t)((h >> 24) & 0xFFFFFu) % ((1u << cell_log2) - len + 1u);
  if (o < off || o >= off + len) return false;
  const uint32_t idx = o - off;
  const uint32_t fam = (uint32_t)((h >> 44) & 0xFFu) % nfam;
  const bool rev = (h >> 52) & 1u;
  const uint32_t lvl = (uint32_t)(h >> 53) & 15u;
  const uint32_t div = div_min + lvl * (div_max - div_min) / 15u;
  const uint32_t ci = rev ? len_full - 1u - idx : len_full - len + idx;
  uint32_t cb = (uint32_t)(mix64((seed ^ salt ^ 0x5555555555555555ULL) + (((uint64_t)fam << 32) + ci) * 0x9E3779B97F4A7C15ULL) & 3u);
  if (rev) cb = 3u - cb;
  const uint64_t y = mix64((seed ^ salt ^ 0xAAAAAAAAAAAAAAAAULL) + s * 0xD1B54A32D192ED03ULL);
  if ((uint32_t)(y & 1023u) < div) cb = (cb + 1u + (uint32_t)((y >> 32) % 3u)) & 3u;
  base = cb;
  return true;
}

constexpr uint64_t SYNTH_SALT_LINE = 0x4C494E454C494E45ULL, SYNTH_SALT_SINE = 0x53494E4553494E45ULL, SYNTH_SALT_SAT = 0x5341544553415445ULL;

__device__ __forceinline__ uint32_t synth_ancestor_base(uint64_t s, uint64_t seed_anc, const SynthRep& R)
{
  uint32_t base = (uint32_t)(mix64(seed_anc + s * 0x9E3779B97F4A7C15ULL) & 3u);
  if (R.on) {
    synth_layer(s, seed_anc, SYNTH_SALT_LINE, R.r.line_cell_log2, R.r.line_len, R.r.line_min_len, R.r.line_prob_256, R.r.line_families, R.r.div_min_1024,
                R.r.div_max_1024, base);
    synth_layer(s, seed_anc, SYNTH_SALT_SINE, R.r.sine_cell_log2, R.r.sine_len, R.r.sine_len, R.r.sine_prob_256, R.r.sine_families, R.r.div_min_1024,
                R.r.div_max_1024, base);
  }
  return base;
}

__device__ __forceinline__ uint32_t synth_tandem_base(uint64_t s, uint32_t fam, uint64_t seed_anc, const SynthRep& R)
{
  const uint32_t u = (uint32_t)(s % R.r.sat_unit);
  uint32_t b = (uint32_t)(mix64((seed_anc ^ SYNTH_SALT_SAT) + (((uint64_t)fam << 32) + u) * 0x9E3779B97F4A7C15ULL) & 3u);
  const uint64_t y = mix64((seed_anc ^ SYNTH_SALT_SAT ^ 0xAAAAAAAAAAAAAAAAULL) + (((uint64_t)fam << 40) ^ s) * 0xD1B54A32D192ED03ULL);
  if ((uint32_t)(y & 1023u) < R.r.sat_div_1024) b = (b + 1u + (uint32_t)((y >> 32) % 3u)) & 3u;
  return b;
}

// The same family with structural events: the genome is a tiling of pieces (ascending dst), each a stretch of the ancestor
// read forwards or as its reverse complement, a stretch of sequence of the genome's own (an insertion), a satellite array, or a
// run of N.  Substitutions are keyed by the genome coordinate, as in k_synth.
__global__ __launch_bounds__(256) void k_synth_plan(uint8_t* __restrict__ code, uint64_t n, const nts_synth_piece* __restrict__ pieces, uint32_t n_pieces,
                                                    uint64_t seed_anc, uint64_t seed_gen, uint32_t thr, SynthRep R)
{
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i0 >= n) return;
  uint32_t lo = 0, hi = n_pieces; // the piece that holds i0
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (pieces[mid].dst <= i0)
      lo = mid;
    else
      hi = mid;
  }
  nts_synth_piece pc = pieces[lo];
  for (int b = 0; b < 16; ++b) {
    const uint64_t i = i0 + b;
    if (i >= n) break;
    while (i >= pc.dst + pc.len && lo + 1 < n_pieces) pc = pieces[++lo];
    const uint64_t off = i - pc.dst;
    const bool rev = pc.flags & NTS_SYNTH_REVCOMP;
    const uint64_t sidx = rev ? pc.src + pc.len - 1 - off : pc.src + off;
    uint32_t base;
    if (pc.flags & NTS_SYNTH_NRUN)
      base = CODE_INVALID;
    else {
      if (pc.flags & NTS_SYNTH_NOVEL)
        base = (uint32_t)(mix64((seed_gen ^ 0x5bd1e995a7c3f1d7ULL) + sidx * 0x9E3779B97F4A7C15ULL) & 3u);
      else if (pc.flags & NTS_SYNTH_TANDEM)
        base = synth_tandem_base(sidx, pc.reserved, seed_anc, R);
      else
        base = synth_ancestor_base(sidx, seed_anc, R);
      if (rev) base = 3u - base;
      const uint64_t y = mix64(seed_gen ^ (i * 0xD1B54A32D192ED03ULL));
      if ((uint32_t)y < thr) base = (base + 1u + (uint32_t)((y >> 32) % 3u)) & 3u;
    }
    code[i] = (uint8_t)base;
  }
}

__global__ __launch_bounds__(256) void k_decode(const uint8_t* __restrict__ code, uint64_t n, uint8_t* __restrict__ ascii)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t c = code[i];
  ascii[i] = c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : c == 3 ? 'T' : 'N';
}

// ---- microbenchmark: random Bloom probes without any hashing, same access shape as the probe batches of
// k_hash (8 independent 4-byte loads per lane in flight).  Gives the empirical ceiling for sector-granular
// random reads that the dense sketch is measured against (DESIGN.md "Rooflines").
__global__ __launch_bounds__(256) void k_bench_probe(const uint32_t* __restrict__ words, FastMod fm, uint64_t n, uint64_t seed,
                                                     unsigned long long* __restrict__ sink)
{
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t b = 0; b < 32; b += 8) {
    uint32_t wd[8], bit[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint64_t q = t * 32 + b + u;
      const uint64_t idx = fm(mix64(q * 0x9E3779B97F4A7C15ULL + seed));
      wd[u] = q < n ? words[idx >> 5] : 0u;
      bit[u] = (uint32_t)idx & 31u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (wd[u] >> bit[u]) & 1u;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sink, (unsigned long long)acc);
}

// compact index -> (record, position in record), and the printed hash h1.  With a second list (b_j / b_k, its length at
// nb_dev; disjoint indices) the two ordered lists are merged on the way: every element finds its slot by a search in the
// other list -- the few winners of the uncovered ranges join the sparse winners without a pass of their own.
//
// The searches (slot in the other list, run, record) of the 6 M elements of the first list were chains of 5-13 dependent loads
// each -- 0.12 ms per 3 Gbp genome, most of the kernel.  The list is ordered: k_fin_chunks makes the three searches once per
// FIN_CHUNK consecutive elements (for the chunk's first index), and an element walks on from its chunk's answers -- nearly always
// zero steps; after FIN_WALK steps it searches the rest (an assembly in thousands of pieces).
constexpr uint32_t FIN_CHUNK = 1024, FIN_WALK = 6;

__device__ __forceinline__ uint32_t fin_last_le(const uint64_t* __restrict__ a, uint32_t lo, uint32_t hi, uint64_t x) // last i in [lo, hi): a[i] <= x
{
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] <= x)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

__device__ __forceinline__ uint64_t fin_lower_bound(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t x) // first i in [lo, hi): a[i] >= x
{
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_fin_chunks(const uint64_t* __restrict__ j_sorted, uint64_t na_host, const uint64_t* __restrict__ n_out_dev,
                                                    const uint64_t* __restrict__ b_j, const uint64_t* __restrict__ nb_dev,
                                                    const uint64_t* __restrict__ run_pos, const uint64_t* __restrict__ run_vstart, uint32_t n_runs,
                                                    const uint64_t* __restrict__ rec_off, uint32_t n_rec, uint64_t* __restrict__ c_b,
                                                    uint32_t* __restrict__ c_run, uint32_t* __restrict__ c_rec)
{
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // elements of the first list: all of them without a second list (their number is then n_out, or *n_out_dev)
  const uint64_t na = b_j ? na_host : (n_out_dev ? *n_out_dev : na_host);
  if (c * FIN_CHUNK >= na) return;
  const uint64_t j = j_sorted[c * FIN_CHUNK];
  c_b[c] = b_j ? fin_lower_bound(b_j, 0, *nb_dev, j) : 0;
  const uint32_t run = fin_last_le(run_vstart, 0, n_runs, j);
  c_run[c] = run;
  c_rec[c] = fin_last_le(rec_off, 0, n_rec, run_pos[run] + (j - run_vstart[run]));
}

__global__ __launch_bounds__(256) void k_finalize(const uint64_t* __restrict__ j_sorted,
                                                  const uint64_t* __restrict__ key_sorted,
                                                  uint64_t n_out,
                                                  const uint64_t* __restrict__ n_out_dev, // if not null: the count lives here
                                                  const uint64_t* __restrict__ b_j,
                                                  const uint64_t* __restrict__ b_k,
                                                  const uint64_t* __restrict__ nb_dev,
                                                  uint64_t na, // (merging) length of the first list; n_out_dev = na + *nb_dev
                                                  const uint64_t* __restrict__ run_pos,
                                                  const uint64_t* __restrict__ run_vstart,
                                                  uint32_t n_runs,
                                                  const uint64_t* __restrict__ rec_off,
                                                  uint32_t n_rec,
                                                  uint32_t k,
                                                  const uint64_t* __restrict__ c_b, const uint32_t* __restrict__ c_run, const uint32_t* __restrict__ c_rec,
                                                  uint64_t* __restrict__ h1,
                                                  uint32_t* __restrict__ rec,
                                                  uint64_t* __restrict__ pos)
{
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (n_out_dev ? *n_out_dev : n_out)) return;
  const bool merging = b_j != nullptr;
  const bool from_a = !merging || t < na;
  const uint64_t self = from_a ? t : t - na;
  const uint64_t j = from_a ? j_sorted[self] : b_j[self];
  const uint64_t key = from_a ? key_sorted[self] : b_k[self];
  uint64_t i = self;
  uint32_t run, a;
  if (from_a) {
    const uint64_t c = self / FIN_CHUNK;
    if (merging) { // slot among the second list's elements: from the chunk's answer on
      const uint64_t nb = *nb_dev;
      uint64_t lo = c_b[c];
      uint32_t steps = 0;
      while (lo < nb && b_j[lo] < j) {
        ++lo;
        if (++steps == FIN_WALK) {
          lo = fin_lower_bound(b_j, lo, nb, j);
          break;
        }
      }
      i = self + lo;
    }
    run = c_run[c];
    for (uint32_t steps = 0; run + 1 < n_runs && run_vstart[run + 1] <= j;) {
      ++run;
      if (++steps == FIN_WALK) {
        run = fin_last_le(run_vstart, run, n_runs, j);
        break;
      }
    }
  } else { // (the few elements of the second list: whole searches)
    i = self + fin_lower_bound(j_sorted, 0, na, j);
    run = fin_last_le(run_vstart, 0, n_runs, j);
  }
  const uint64_t gp = run_pos[run] + (j - run_vstart[run]);
  if (from_a) {
    a = c_rec[self / FIN_CHUNK];
    for (uint32_t steps = 0; a + 1 < n_rec && rec_off[a + 1] <= gp;) {
      ++a;
      if (++steps == FIN_WALK) {
        a = fin_last_le(rec_off, a, n_rec, gp);
        break;
      }
    }
  } else {
    a = fin_last_le(rec_off, 0, n_rec, gp);
  }
  h1[i] = extend_h1(key, k);
  rec[i] = a;
  pos[i] = gp - rec_off[a];
}


// =================================================================================================
// Host helpers
// =================================================================================================


// hash parameters with the device-resident init table (cached per k in the context)
int hash_params_for(nts_ctx* ctx, uint32_t k, HashParams* out)
{
  *out = make_hash_params(k);
  auto it = ctx->init_tabs.find(k);
  if (it == ctx->init_tabs.end()) {
    const uint32_t ng = (k + 3) / 4;
    std::vector<uint64_t> tab((size_t)k * 8 + (size_t)ng * 512);
    const uint64_t seed[4] = { SEED_A, SEED_C, SEED_G, SEED_T };
    for (int b = 0; b < 4; ++b) {
      uint64_t x = seed[3 - b]; // srol^i(seed[3-b]), i ascending
      for (uint32_t i = 0; i < k; ++i) {
        tab[((size_t)i * 4 + b) * 2 + 1] = x;
        x = srol1(x);
      }
      uint64_t y = seed[b]; // srol^(k-1-i)(seed[b]): i descending
      for (uint32_t i = k; i-- > 0;) {
        tab[((size_t)i * 4 + b) * 2] = y;
        y = srol1(y);
      }
    }
    uint64_t* tab4 = tab.data() + (size_t)k * 8; // four bases at a time (bases past k do not count)
    for (uint32_t g = 0; g < ng; ++g)
      for (uint32_t v = 0; v < 256; ++v) {
        uint64_t f = 0, r = 0;
        for (uint32_t j = 0; j < 4 && 4 * g + j < k; ++j) {
          const uint32_t b = (v >> (2 * j)) & 3u;
          f ^= tab[((size_t)(4 * g + j) * 4 + b) * 2];
          r ^= tab[((size_t)(4 * g + j) * 4 + b) * 2 + 1];
        }
        tab4[((size_t)g * 256 + v) * 2] = f;
        tab4[((size_t)g * 256 + v) * 2 + 1] = r;
      }
    uint64_t* d = nullptr;
    HIP_TRY(ctx, dev_malloc((void**)&d, tab.size() * 8));
    hipError_t e = hipMemcpy(d, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      dev_free(d);
      HIP_TRY(ctx, e);
    }
    it = ctx->init_tabs.emplace(k, d).first;
  }
  out->init = it->second;
  out->init4 = it->second + (size_t)k * 8;
  return NTS_OK;
}

__global__ __launch_bounds__(256) void k_mod_indices(uint64_t* __restrict__ h, uint64_t n, FastMod fm)
{
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) h[i] = fm(h[i]);
}


inline uint64_t key_buffer_elems(uint64_t n_valid)
{
  return std::max<uint64_t>((n_valid + KEY_TILE - 1) / KEY_TILE, 1) * KEY_TILE;
}

struct RunTable
{
  std::vector<uint64_t> pos, vstart; // vstart has n_runs+1 entries
  std::vector<uint64_t> rec_vstart, rec_nv;
  uint64_t n_valid = 0;
};

// stretches minus mask intervals, keep pieces of at least k bases -> runs of k-mer starts
int build_runs(nts_ctx* ctx, const nts_genome* g, uint32_t k, const nts_interval* mask, uint64_t n_mask, RunTable& rt)
{
  std::vector<std::pair<uint64_t, uint64_t>> mk;
  mk.reserve(n_mask);
  for (uint64_t i = 0; i < n_mask; ++i) {
    if (mask[i].rec >= g->n_rec) return fail(ctx, NTS_EINVAL, "mask interval: record index out of range");
    const uint64_t len = g->rec_len[mask[i].rec];
    const uint64_t s = std::min(mask[i].start, len), e = std::min(mask[i].end, len);
    if (e > s) mk.push_back({ g->rec_off[mask[i].rec] + s, g->rec_off[mask[i].rec] + e });
  }
  std::sort(mk.begin(), mk.end());
  size_t mi = 0;
  const size_t ns = g->st_a.size();
  rt.pos.clear();
  rt.vstart.clear();
  uint64_t v = 0;
  auto add_piece = [&](uint64_t a, uint64_t b) {
    if (b > a && b - a >= k) {
      rt.pos.push_back(a);
      rt.vstart.push_back(v);
      v += (b - a) - k + 1;
    }
  };
  for (size_t s = 0; s < ns; ++s) {
    uint64_t a = g->st_a[s];
    const uint64_t b = g->st_b[s];
    while (mi < mk.size() && mk[mi].second <= a) ++mi;
    size_t q = mi;
    while (q < mk.size() && mk[q].first < b) {
      if (mk[q].first > a) add_piece(a, mk[q].first);
      a = std::max(a, mk[q].second);
      if (a >= b) break;
      ++q;
    }
    if (a < b) add_piece(a, b);
  }
  rt.vstart.push_back(v);
  rt.n_valid = v;
  // per-record compact ranges
  rt.rec_vstart.assign(g->n_rec, 0);
  rt.rec_nv.assign(g->n_rec, 0);
  size_t ri = 0;
  const size_t nr = rt.pos.size();
  for (uint32_t r = 0; r < g->n_rec; ++r) {
    const uint64_t end = g->rec_off[r] + g->rec_len[r];
    while (ri < nr && rt.pos[ri] < g->rec_off[r]) ++ri; // (cannot happen: stretches are clipped)
    rt.rec_vstart[r] = ri < nr ? rt.vstart[ri] : v;
    size_t q = ri;
    while (q < nr && rt.pos[q] < end) ++q;
    rt.rec_nv[r] = (q < nr ? rt.vstart[q] : v) - rt.rec_vstart[r];
    ri = q;
  }
  return NTS_OK;
}

// host vector -> named scratch buffer (the vector must outlive the stream work; callers sync)
template <typename T>
int ws_upload(nts_ctx* ctx, const char* name, const std::vector<T>& h, T** d)
{
  *d = (T*)ws_get(ctx, name, std::max<size_t>(h.size(), 1) * sizeof(T));
  if (!*d) return NTS_ENOMEM;
  if (!h.empty()) HIP_TRY(ctx, hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return NTS_OK;
}

// Several small host arrays -> one scratch buffer, through the pinned staging area and a single asynchronous copy
// (a pageable hipMemcpyAsync per table costs ~20 us each on the host).  dev[i] receives the device address of
// part i; parts are 16-byte aligned.  The staging area is reused by the next call: callers synchronise the
// stream before they return, which every user of the tables does.
struct HostPart
{
  const void* p;
  size_t bytes;
};

int upload_packed(nts_ctx* ctx, const char* name, std::initializer_list<HostPart> parts, void** dev)
{
  size_t total = 0;
  for (const HostPart& h : parts) total += (h.bytes + 15) & ~(size_t)15;
  total = std::max<size_t>(total, 16);
  if (ctx->stage_bytes < total) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stage) hipHostFree(ctx->stage);
    ctx->stage = nullptr;
    ctx->stage_bytes = 0;
    const size_t want = std::max<size_t>(total + total / 4, (size_t)1 << 20);
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->stage, want, hipHostMallocDefault));
    ctx->stage_bytes = want;
  }
  uint8_t* d = (uint8_t*)ws_get(ctx, name, total);
  if (!d) return NTS_ENOMEM;
  size_t at = 0, i = 0;
  for (const HostPart& h : parts) {
    if (h.bytes) memcpy(ctx->stage + at, h.p, h.bytes);
    dev[i++] = d + at;
    at += (h.bytes + 15) & ~(size_t)15;
  }
  HIP_TRY(ctx, hipMemcpyAsync(d, ctx->stage, total, hipMemcpyHostToDevice, ctx->stream));
  return NTS_OK;
}

} // namespace

// Run table and per-record tables of one (genome, k): host copy + device arrays.  Cached in the genome
// handle for unmasked calls (`owned`), built into context scratch for masked refinement rounds.
struct GenomeTables
{
  RunTable rt;
  uint64_t *d_run_pos = nullptr, *d_run_vstart = nullptr, *d_rec_vstart = nullptr, *d_rec_nv = nullptr;
  uint32_t n_runs = 0;
  bool owned = false;
  mutable std::map<uint32_t, std::pair<uint64_t, uint64_t*>> win_tiles; // w -> (tiles, device prefix array)
  uint64_t n_win_tiles(uint32_t w) const
  {
    auto it = win_tiles.find(w);
    if (it != win_tiles.end()) return it->second.first;
    uint64_t n = 0;
    for (uint64_t nv : rt.rec_nv) n += ((nv >= w ? nv - w + 1 : 0) + WIN_TILE - 1) / WIN_TILE;
    win_tiles[w] = { n, nullptr };
    return n;
  }
  int win_tiles_device(nts_ctx* ctx, uint32_t w, const uint64_t** out) const
  {
    n_win_tiles(w);
    auto& e = win_tiles[w];
    if (!e.second) {
      std::vector<uint64_t> ts(rt.rec_nv.size() + 1, 0);
      for (size_t r = 0; r < rt.rec_nv.size(); ++r) {
        const uint64_t nv = rt.rec_nv[r];
        ts[r + 1] = ts[r] + ((nv >= w ? nv - w + 1 : 0) + WIN_TILE - 1) / WIN_TILE;
      }
      uint64_t* d = nullptr;
      if (owned) {
        HIP_TRY(ctx, dev_malloc((void**)&d, ts.size() * 8));
        HIP_TRY(ctx, hipMemcpy(d, ts.data(), ts.size() * 8, hipMemcpyHostToDevice));
        e.second = d;
      } else {
        int rc = ws_upload(ctx, "win_tiles_masked", ts, &d);
        if (rc) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // `ts` dies here
        *out = d;
        return NTS_OK;
      }
    }
    *out = e.second;
    return NTS_OK;
  }
  void release()
  {
    if (!owned) return;
    dev_free(d_run_pos);
    dev_free(d_run_vstart);
    dev_free(d_rec_vstart);
    dev_free(d_rec_nv);
    for (auto& kv : win_tiles)
      if (kv.second.second) dev_free(kv.second.second);
  }
};

namespace {

int get_tables(nts_ctx* ctx, const nts_genome* g, uint32_t k, const nts_interval* mask, uint64_t n_mask, GenomeTables& scratch,
               const GenomeTables** out)
{
  if (n_mask == 0) {
    auto it = g->tables.find(k);
    if (it != g->tables.end()) {
      *out = it->second;
      return NTS_OK;
    }
    GenomeTables* T = new GenomeTables();
    int rc = build_runs(ctx, g, k, nullptr, 0, T->rt);
    if (rc) {
      delete T;
      return rc;
    }
    if (T->rt.pos.size() > 0xFFFFFFF0ULL) {
      delete T;
      return fail(ctx, NTS_ERANGE, "too many valid runs");
    }
    T->owned = true;
    T->n_runs = (uint32_t)T->rt.pos.size();
    auto up = [&](const std::vector<uint64_t>& h, uint64_t** d) -> bool {
      if (dev_malloc((void**)d, std::max<size_t>(h.size(), 1) * 8) != hipSuccess) return false;
      return h.empty() || hipMemcpy(*d, h.data(), h.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(T->rt.pos, &T->d_run_pos) || !up(T->rt.vstart, &T->d_run_vstart) || !up(T->rt.rec_vstart, &T->d_rec_vstart) ||
        !up(T->rt.rec_nv, &T->d_rec_nv)) {
      T->release();
      delete T;
      return fail(ctx, NTS_ENOMEM, "genome tables: device allocation failed");
    }
    g->tables[k] = T;
    *out = T;
    return NTS_OK;
  }
  int rc = build_runs(ctx, g, k, mask, n_mask, scratch.rt);
  if (rc) return rc;
  if (scratch.rt.pos.size() > 0xFFFFFFF0ULL) return fail(ctx, NTS_ERANGE, "too many valid runs");
  scratch.n_runs = (uint32_t)scratch.rt.pos.size();
  if ((rc = ws_upload(ctx, "run_pos", scratch.rt.pos, &scratch.d_run_pos))) return rc;
  if ((rc = ws_upload(ctx, "run_vstart", scratch.rt.vstart, &scratch.d_run_vstart))) return rc;
  if ((rc = ws_upload(ctx, "rec_vstart", scratch.rt.rec_vstart, &scratch.d_rec_vstart))) return rc;
  if ((rc = ws_upload(ctx, "rec_nv", scratch.rt.rec_nv, &scratch.d_rec_nv))) return rc;
  *out = &scratch;
  return NTS_OK;
}

template <int MODE>
int launch_hash(nts_ctx* ctx, const char* name, const nts_genome* g, const GenomeTables& T, uint32_t k,
                const nts_bf* bf_in, nts_bf* bf_out, uint64_t* keys, const uint32_t* d_tile_ids = nullptr, uint64_t n_tile_ids = 0,
                const uint2* d_tile_span = nullptr)
{
  const RunTable& rt = T.rt;
  if (rt.n_valid == 0) return NTS_OK;
  if (d_tile_ids && n_tile_ids == 0) return NTS_OK;
  HashParams hp;
  {
    const int rc_hp = hash_params_for(ctx, k, &hp);
    if (rc_hp) return rc_hp;
  }
  const uint64_t bits = (bf_in ? bf_in->bytes : (bf_out ? bf_out->bytes : 8)) * 8;
  const FastMod fm = make_fastmod(bits);
  const uint64_t per_block = (uint64_t)HASH_THREADS * HASH_PER_THREAD;
  const uint64_t blocks = d_tile_ids ? n_tile_ids : (rt.n_valid + per_block - 1) / per_block;
  if (blocks > 0x7FFFFFFFULL) return fail(ctx, NTS_ERANGE, "genome too large for one launch");
  ScopedTimer t(ctx, name, true);
  if (MODE == MODE_KEYS && bf_in && !d_tile_ids && ctx->cur_summary && ctx->cur_tile_any) {
    NTS_LAUNCH(k_hash_keys_sparse, dim3((uint32_t)blocks), dim3(HASH_THREADS), 0, ctx->stream, g->d_code + PAD, T.d_run_pos,
                       T.d_run_vstart, T.n_runs, rt.n_valid, hp, bf_in->d_words, fm, ctx->cur_summary, ctx->cur_summary_shift, keys,
                       ctx->cur_tile_any);
    HIP_TRY(ctx, hipGetLastError());
    return NTS_OK;
  }
  NTS_LAUNCH(k_hash<MODE>, dim3((uint32_t)blocks), dim3(HASH_THREADS), 0, ctx->stream, g->d_code + PAD, T.d_run_pos,
                     T.d_run_vstart, T.n_runs, rt.n_valid, hp, bf_in ? bf_in->d_words : nullptr, bf_out ? bf_out->d_words : nullptr,
                     fm, keys, d_tile_ids, d_tile_span, (MODE == MODE_KEYS && ctx->cur_rep) ? ctx->cur_rep->d_words : nullptr,
                     make_fastmod((MODE == MODE_KEYS && ctx->cur_rep) ? ctx->cur_rep->bytes * 8 : 64));
  HIP_TRY(ctx, hipGetLastError());
  return NTS_OK;
}

#include "nts_pruned.inc"
#include "nts_tiers.inc"
#include "nts_bloom_bin.inc"
#include "nts_bf_sparse.inc"
#include "nts_microbench.inc"

// acc &= the filter of genome g the literal way, for a running filter that holds few bits (defined behind the sketch's host code,
// whose accept kernels and summary it uses): 0 = done, 1 = does not apply or did not fit (acc is untouched), < 0 = error
int bf_level_sparse(nts_ctx* ctx, nts_bf* acc, const nts_genome* g, const GenomeTables& T, uint32_t k, int64_t pop_before);

} // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int nts_init(int device, nts_ctx** out)
{
  if (!out) return NTS_EINVAL;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_init_error = std::string("no HIP device: ") + hipGetErrorString(e);
    return NTS_EHIP;
  }
  if (device < 0 || device >= n) {
    g_init_error = "device index out of range";
    return NTS_EINVAL;
  }
  nts_ctx* ctx = new nts_ctx();
  ctx->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess) {
    g_init_error = std::string("hipSetDevice/hipStreamCreate: ") + hipGetErrorString(e);
    delete ctx;
    return NTS_EHIP;
  }
  if ((e = hipHostMalloc((void**)&ctx->mail, MAIL_WORDS * sizeof(uint64_t), hipHostMallocMapped)) != hipSuccess ||
      (e = hipHostGetDevicePointer((void**)&ctx->d_mail, ctx->mail, 0)) != hipSuccess) {
    g_init_error = std::string("hipHostMalloc (result mailbox): ") + hipGetErrorString(e);
    nts_destroy(ctx);
    return NTS_EHIP;
  }
  // what a context reads from the environment, once (nts_knobs.h)
  if (const char* v = getenv("NTS_COMM_PIECE")) ctx->comm_piece = strtoull(v, nullptr, 10);
  if (const char* v = getenv("NTS_IO_THREADS")) ctx->io_threads = (unsigned)std::max(1, std::min(32, atoi(v)));
  if (const char* v = NTS_KNOB("NTS_COMM_SPARSE")) ctx->comm_sparse_mode = atoi(v) == 0 ? 1 : 0;
  if (const char* v = NTS_KNOB("NTS_COMM_SPARSE_BELOW")) ctx->comm_sparse_below = strtoull(v, nullptr, 10);
  ctx->counted = true;
  g_live_contexts.fetch_add(1);
  *out = ctx;
  return NTS_OK;
}

void nts_destroy(nts_ctx* ctx)
{
  if (!ctx) return;
  hipSetDevice(ctx->device);
  drain_timings(ctx);
  for (hipEvent_t e : ctx->spare_events) hipEventDestroy(e);
  hipStreamSynchronize(ctx->stream);
  ws_release(ctx);
  for (auto& p : ctx->mx_pool) dev_free(p.first);
  io_release(ctx->io_up);
  io_release(ctx->io_down);
  for (auto& kv : ctx->init_tabs) dev_free(kv.second);
  if (ctx->mail) hipHostFree(ctx->mail);
  if (ctx->stage) hipHostFree(ctx->stage);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  const bool last = ctx->counted && g_live_contexts.fetch_sub(1) == 1;
  delete ctx;
  // the process's last context is gone: what the allocation cache holds goes back to the driver (another process on this GPU that runs
  // out of memory cannot ask this one to let go)
  if (last) nts_mem::trim();
}

const char* nts_last_error(nts_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_init_error.c_str();
}

int nts_sync(nts_ctx* ctx)
{
  if (!ctx) return NTS_EINVAL;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

void* nts_stream(nts_ctx* ctx)
{
  return ctx ? (void*)ctx->stream : nullptr;
}

int nts_profile(nts_ctx* ctx, int enable)
{
  if (!ctx) return NTS_EINVAL;
  drain_timings(ctx);
  ctx->timings.clear();
  ctx->profiling = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
  return NTS_OK;
}

int nts_mem_stats(nts_ctx* ctx, uint64_t* live_bytes, uint64_t* peak_bytes, uint64_t* device_used_bytes, uint64_t* device_total_bytes)
{
  if (live_bytes) *live_bytes = nts_mem::live.load();
  if (peak_bytes) *peak_bytes = nts_mem::peak.load();
  size_t fr = 0, tot = 0;
  if (ctx && (device_used_bytes || device_total_bytes)) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemGetInfo(&fr, &tot));
  }
  if (device_used_bytes) *device_used_bytes = tot - fr;
  if (device_total_bytes) *device_total_bytes = tot;
  return NTS_OK;
}

// hipMalloc / hipFree calls the library has made in this process and the host time spent in them (a cold call's allocation share)
int nts_alloc_stats(uint64_t* calls, double* ms)
{
  if (calls) *calls = nts_mem::alloc_calls.load();
  if (ms) *ms = (double)nts_mem::alloc_ns.load() * 1e-6;
  return NTS_OK;
}

// Every block the library's allocation cache holds goes back to the driver; bytes released.  (A process that shares the GPU with other
// users of its memory calls this when a run is over; an allocation that fails does it by itself.)
uint64_t nts_mem_trim(void)
{
  return nts_mem::trim();
}

// bytes the cache holds now, and how many allocations it has served
int nts_mem_cache_stats(uint64_t* cached_bytes, uint64_t* hits)
{
  std::lock_guard<std::mutex> g(nts_mem::mu);
  if (cached_bytes) *cached_bytes = nts_mem::cached_bytes - nts_mem::cached_small;
  if (hits) *hits = nts_mem::cache_hits.load();
  return NTS_OK;
}

void nts_mem_reset_peak(void)
{
  nts_mem::peak.store(nts_mem::live.load());
}

// One driver allocation of `bytes` on `device`, kept for the allocations of the run to come (they are cut from it; nothing goes back
// to the driver before nts_mem_trim, an allocation failure, or the end of the process).
int nts_mem_reserve(int device, uint64_t bytes, uint64_t* reserved_bytes)
{
  return nts_mem::reserve(device, bytes, reserved_bytes) == hipSuccess ? NTS_OK : NTS_EHIP;
}

// out[0..7]: hipMalloc+hipFree calls, ns spent in them, allocations served from the cache, allocations retried after the cache was
// emptied, bytes taken from the driver, bytes given back, ns spent in hipDeviceSynchronize before keeping a freed block, reserve calls
int nts_mem_events(uint64_t out[8])
{
  if (!out) return NTS_EINVAL;
  out[0] = nts_mem::alloc_calls.load();
  out[1] = nts_mem::alloc_ns.load();
  out[2] = nts_mem::cache_hits.load();
  out[3] = nts_mem::oom_retries.load();
  out[4] = nts_mem::driver_bytes_in.load();
  out[5] = nts_mem::driver_bytes_out.load();
  out[6] = nts_mem::free_sync_ns.load();
  out[7] = nts_mem::reserve_calls.load();
  return NTS_OK;
}

int nts_timing(nts_ctx* ctx, const char* name, double* total_ms, uint64_t* launches)
{
  if (!ctx || !name) return NTS_EINVAL;
  drain_timings(ctx);
  auto it = ctx->timings.find(name);
  if (total_ms) *total_ms = it == ctx->timings.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == ctx->timings.end() ? 0 : it->second.launches;
  return NTS_OK;
}

int nts_bf_size_bytes_ex(uint64_t genome_bp, double fpr, int rounding, uint64_t* approx_bytes, uint64_t* ctor_bytes)
{
  if (!(fpr > 0.0 && fpr < 1.0) || rounding < NTS_BF_ROUND_UP || rounding > NTS_BF_ROUND_NONE) return NTS_EINVAL;
  // src/ntsynt_make_common_bf.cpp:38-39
  const long long genome_size = (long long)genome_bp;
  const long long size_bits = (long long)std::ceil(((double)(-1 * genome_size)) / std::log(1 - fpr));
  const uint64_t approx = (uint64_t)(size_bits / 8);
  if (approx_bytes) *approx_bytes = approx;
  uint64_t ctor = approx;
  if (rounding == NTS_BF_ROUND_UP) ctor = (uint64_t)(std::ceil((double)approx / 8.0) * 8.0);
  if (rounding == NTS_BF_ROUND_DOWN) ctor = approx / 8 * 8;
  if (ctor_bytes) *ctor_bytes = ctor;
  return NTS_OK;
}

int nts_bf_size_bytes(uint64_t genome_bp, double fpr, uint64_t* approx_bytes, uint64_t* ctor_bytes)
{
  return nts_bf_size_bytes_ex(genome_bp, fpr, NTS_BF_ROUND_UP, approx_bytes, ctor_bytes);
}

// The part of a genome's set-up that needs its codes in HBM: maximal stretches of valid bases (clipped to records) and the
// record table on the device.  g->n, g->n_rec, g->rec_off, g->rec_len and g->d_code are in place.
int nts_genome_finish_impl(nts_ctx* ctx, nts_genome* g) // (also called by the FASTA parse, nts_comm_fasta.hip)
{
  const uint64_t n = g->n;
  const uint32_t n_rec = g->n_rec;
  // valid stretches: count, append, sort.  (Scratch from the context's workspaces: every hipFree waits for the whole device, and
  // the pipeline builds the filter of the previous genome on another context while this one comes up.)
  unsigned long long* d_cnt = (unsigned long long*)ws_get(ctx, "gf_cnt", 3 * sizeof(unsigned long long));
  if (!d_cnt) return NTS_ENOMEM;
  hipMemsetAsync(d_cnt, 0, 3 * sizeof(unsigned long long), ctx->stream);
  const uint64_t sblocks = (n + 1 + 4095) / 4096;
  NTS_LAUNCH(k_stretch<0>, dim3((uint32_t)sblocks), dim3(256), 0, ctx->stream, g->d_code + PAD, n, d_cnt, nullptr, nullptr,
                     nullptr);
  unsigned long long n_st = 0;
  hipMemcpyAsync(&n_st, d_cnt, sizeof(n_st), hipMemcpyDeviceToHost, ctx->stream);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(ctx, NTS_EHIP, std::string("encode/stretch count: ") + hipGetErrorString(hipGetLastError()));
  std::vector<uint64_t> hs(n_st), he(n_st);
  if (n_st) {
    uint64_t* d_s = (uint64_t*)ws_get(ctx, "gf_s", n_st * 8);
    uint64_t* d_e = (uint64_t*)ws_get(ctx, "gf_e", n_st * 8);
    uint64_t* d_s2 = (uint64_t*)ws_get(ctx, "gf_s2", n_st * 8);
    uint64_t* d_e2 = (uint64_t*)ws_get(ctx, "gf_e2", n_st * 8);
    if (!d_s || !d_e || !d_s2 || !d_e2) return NTS_ENOMEM;
    size_t tmp_bytes = 0;
    NTS_LAUNCH(k_stretch<1>, dim3((uint32_t)sblocks), dim3(256), 0, ctx->stream, g->d_code + PAD, n, d_cnt, d_s, d_e, d_cnt + 1);
    rocprim::radix_sort_keys(nullptr, tmp_bytes, d_s, d_s2, n_st, 0, 64, ctx->stream);
    void* d_tmp = ws_get(ctx, "gf_tmp", std::max<size_t>(tmp_bytes, 16));
    if (!d_tmp) return NTS_ENOMEM;
    rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_s, d_s2, n_st, 0, 64, ctx->stream);
    rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_e, d_e2, n_st, 0, 64, ctx->stream);
    hipMemcpyAsync(hs.data(), d_s2, n_st * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(he.data(), d_e2, n_st * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ctx, NTS_EHIP, "stretch detection failed");
  }
  // clip stretches to records (k-mers never span two records)
  uint32_t r = 0;
  for (size_t s = 0; s < hs.size(); ++s) {
    uint64_t a = hs[s];
    const uint64_t b = he[s];
    while (a < b) {
      while (r < n_rec && g->rec_off[r] + g->rec_len[r] <= a) ++r;
      if (r >= n_rec) break;
      const uint64_t ra = g->rec_off[r], rb = ra + g->rec_len[r];
      const uint64_t x = std::max(a, ra), y = std::min(b, rb);
      if (y > x) {
        g->st_a.push_back(x);
        g->st_b.push_back(y);
      }
      if (b <= rb) break;
      a = rb;
    }
  }
  if (dev_malloc((void**)&g->d_rec_off, std::max<uint32_t>(n_rec, 1) * 8) != hipSuccess ||
      (n_rec && hipMemcpy(g->d_rec_off, g->rec_off.data(), n_rec * 8, hipMemcpyHostToDevice) != hipSuccess))
    return fail(ctx, NTS_ENOMEM, "hipMalloc record offsets");
  return NTS_OK;
}

int nts_genome_upload(nts_ctx* ctx, const uint8_t* seq, uint64_t n, const uint64_t* rec_off, const uint64_t* rec_len,
                      uint32_t n_rec, nts_genome** out)
{
  if (!ctx || !out || (n && !seq) || (n_rec && (!rec_off || !rec_len))) return fail(ctx, NTS_EINVAL, "nts_genome_upload: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t prev_end = 0;
  for (uint32_t r = 0; r < n_rec; ++r) {
    if (rec_off[r] < prev_end || rec_off[r] + rec_len[r] > n)
      return fail(ctx, NTS_EINVAL, "nts_genome_upload: records must be ascending, disjoint and inside seq");
    prev_end = rec_off[r] + rec_len[r];
  }
  nts_genome* g = new nts_genome();
  g->n = n;
  g->n_rec = n_rec;
  g->rec_off.assign(rec_off, rec_off + n_rec);
  g->rec_len.assign(rec_len, rec_len + n_rec);
  for (uint32_t r = 0; r < n_rec; ++r) g->total_bases += rec_len[r];
  const uint64_t dev_bytes = PAD + n + PAD;
  hipError_t e = dev_malloc((void**)&g->d_code, dev_bytes);
  if (e != hipSuccess) {
    delete g;
    return fail(ctx, NTS_ENOMEM, std::string("hipMalloc genome: ") + hipGetErrorString(e));
  }
  auto bail = [&](int code, const std::string& msg) {
    dev_free(g->d_code);
    delete g;
    return fail(ctx, code, msg);
  };
  if (hipMemsetAsync(g->d_code, CODE_INVALID, PAD, ctx->stream) != hipSuccess ||
      hipMemsetAsync(g->d_code + PAD + n, CODE_INVALID, PAD, ctx->stream) != hipSuccess)
    return bail(NTS_EHIP, "hipMemset pads");
  if (n) {
    if (hipMemcpyAsync(g->d_code + PAD, seq, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return bail(NTS_EHIP, "hipMemcpy genome");
    const uint64_t blocks = (n + 4095) / 4096;
    NTS_LAUNCH(k_encode, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, g->d_code + PAD, n);
  }
  if (int rc = nts_genome_finish_impl(ctx, g)) {
    dev_free(g->d_code);
    dev_free(g->d_rec_off);
    delete g;
    return rc;
  }
  *out = g;
  return NTS_OK;
}

// A batch of genomes as one device genome: the parts' records one after the other (record ids of part p start at the
// number of records before it), bases copied device to device.  Records never share k-mers, so sketching the batch
// gives every part's minimizers -- with one sequence of launches for all of them, which is what counts while a
// genome is small enough for the fixed cost of a launch sequence to show.
int nts_genome_concat(nts_ctx* ctx, uint32_t n_parts, const nts_genome* const* parts, nts_genome** out)
{
  if (!ctx || !out || !n_parts || !parts) return fail(ctx, NTS_EINVAL, "nts_genome_concat: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t n = 0, n_rec = 0;
  for (uint32_t p = 0; p < n_parts; ++p) {
    if (!parts[p]) return fail(ctx, NTS_EINVAL, "nts_genome_concat: null part");
    n += parts[p]->n;
    n_rec += parts[p]->n_rec;
  }
  if (n_rec > 0xFFFFFFF0ULL) return fail(ctx, NTS_ERANGE, "nts_genome_concat: too many records");
  nts_genome* g = new nts_genome();
  g->n = n;
  g->n_rec = (uint32_t)n_rec;
  if (dev_malloc((void**)&g->d_code, PAD + n + PAD) != hipSuccess ||
      dev_malloc((void**)&g->d_rec_off, std::max<uint64_t>(n_rec, 1) * 8) != hipSuccess) {
    dev_free(g->d_code);
    delete g;
    return fail(ctx, NTS_ENOMEM, "nts_genome_concat: hipMalloc");
  }
  bool ok = hipMemsetAsync(g->d_code, CODE_INVALID, PAD, ctx->stream) == hipSuccess &&
            hipMemsetAsync(g->d_code + PAD + n, CODE_INVALID, PAD, ctx->stream) == hipSuccess;
  uint64_t base = 0;
  for (uint32_t p = 0; p < n_parts && ok; ++p) {
    const nts_genome* q = parts[p];
    if (q->n) ok = hipMemcpyAsync(g->d_code + PAD + base, q->d_code + PAD, q->n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
    for (uint32_t r = 0; r < q->n_rec; ++r) {
      g->rec_off.push_back(base + q->rec_off[r]);
      g->rec_len.push_back(q->rec_len[r]);
    }
    for (size_t i = 0; i < q->st_a.size(); ++i) {
      g->st_a.push_back(base + q->st_a[i]);
      g->st_b.push_back(base + q->st_b[i]);
    }
    g->total_bases += q->total_bases;
    g->part_bases.push_back(q->total_bases);
    base += q->n;
  }
  if (ok && n_rec) ok = hipMemcpyAsync(g->d_rec_off, g->rec_off.data(), n_rec * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
  if (ok) ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {
    const std::string msg = std::string("nts_genome_concat: ") + hipGetErrorString(hipGetLastError());
    dev_free(g->d_code);
    dev_free(g->d_rec_off);
    delete g;
    return fail(ctx, NTS_EHIP, msg);
  }
  *out = g;
  return NTS_OK;
}

// Records [rec0, rec1) of a resident genome as a resident genome of their own (a device-to-device copy): the shard of a genome
// that one rank of its group works on when there are fewer genomes than GPUs (SURVEY.md 8(e) last paragraph: "shard by contig --
// windows never cross records --, OR the partial filters of a genome, then AND across genomes"; the reference parallelises over
// records the same way, src/ntsynt_make_common_bf.cpp:128-131,145-153).  Record r of the slice is record rec0 + r of `g`.
int nts_genome_slice(nts_ctx* ctx, const nts_genome* g, uint32_t rec0, uint32_t rec1, nts_genome** out)
{
  if (!ctx || !g || !out || rec0 > rec1 || rec1 > g->n_rec) return fail(ctx, NTS_EINVAL, "nts_genome_slice: record range outside the genome");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t a = rec0 < g->n_rec ? g->rec_off[rec0] : g->n;
  const uint64_t b = rec1 > rec0 ? g->rec_off[rec1 - 1] + g->rec_len[rec1 - 1] : a;
  const uint64_t n = b - a;
  const uint32_t n_rec = rec1 - rec0;
  nts_genome* q = new nts_genome();
  q->n = n;
  q->n_rec = n_rec;
  if (dev_malloc((void**)&q->d_code, PAD + n + PAD) != hipSuccess || dev_malloc((void**)&q->d_rec_off, std::max<uint64_t>(n_rec, 1) * 8) != hipSuccess) {
    dev_free(q->d_code);
    delete q;
    return fail(ctx, NTS_ENOMEM, "nts_genome_slice: hipMalloc");
  }
  for (uint32_t r = rec0; r < rec1; ++r) {
    q->rec_off.push_back(g->rec_off[r] - a);
    q->rec_len.push_back(g->rec_len[r]);
    q->total_bases += g->rec_len[r];
  }
  const size_t i0 = (size_t)(std::lower_bound(g->st_a.begin(), g->st_a.end(), a) - g->st_a.begin()); // (stretches are clipped to records)
  for (size_t i = i0; i < g->st_a.size() && g->st_a[i] < b; ++i) {
    q->st_a.push_back(g->st_a[i] - a);
    q->st_b.push_back(g->st_b[i] - a);
  }
  bool ok = hipMemsetAsync(q->d_code, CODE_INVALID, PAD, ctx->stream) == hipSuccess &&
            hipMemsetAsync(q->d_code + PAD + n, CODE_INVALID, PAD, ctx->stream) == hipSuccess;
  if (ok && n) ok = hipMemcpyAsync(q->d_code + PAD, g->d_code + PAD + a, n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
  if (ok && n_rec) ok = hipMemcpyAsync(q->d_rec_off, q->rec_off.data(), (size_t)n_rec * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
  if (ok) ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {
    const std::string msg = std::string("nts_genome_slice: ") + hipGetErrorString(hipGetLastError());
    dev_free(q->d_code);
    dev_free(q->d_rec_off);
    delete q;
    return fail(ctx, NTS_EHIP, msg);
  }
  *out = q;
  return NTS_OK;
}

void nts_genome_free(nts_ctx* ctx, nts_genome* g)
{
  if (!g) return;
  if (ctx) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
  }
  for (auto& kv : g->tables) {
    kv.second->release();
    delete kv.second;
  }
  if (g->d_rec_off) dev_free(g->d_rec_off);
  if (g->d_code) dev_free(g->d_code);
  if (g->d_pack) dev_free(g->d_pack);
  delete g;
}

uint64_t nts_genome_bases(const nts_genome* g)
{
  return g ? g->total_bases : 0;
}

int nts_genome_synth(nts_ctx* ctx, uint64_t total_bp, uint32_t n_contigs, uint64_t seed_ancestor, uint64_t seed_genome,
                     double substitution_rate, nts_genome** out)
{
  if (!ctx || !out || total_bp == 0 || n_contigs == 0 || substitution_rate < 0 || substitution_rate >= 1)
    return fail(ctx, NTS_EINVAL, "nts_genome_synth: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t per = total_bp / n_contigs;
  if (per == 0) return fail(ctx, NTS_EINVAL, "nts_genome_synth: contigs would be empty");
  const uint64_t n = per * n_contigs;
  nts_genome* g = new nts_genome();
  g->n = n;
  g->n_rec = n_contigs;
  for (uint32_t r = 0; r < n_contigs; ++r) {
    g->rec_off.push_back(per * r);
    g->rec_len.push_back(per);
    g->st_a.push_back(per * r);
    g->st_b.push_back(per * (r + 1));
  }
  g->total_bases = n;
  if (dev_malloc((void**)&g->d_code, PAD + n + PAD) != hipSuccess || dev_malloc((void**)&g->d_rec_off, n_contigs * 8) != hipSuccess) {
    dev_free(g->d_code);
    delete g;
    return fail(ctx, NTS_ENOMEM, "nts_genome_synth: hipMalloc");
  }
  hipMemsetAsync(g->d_code, CODE_INVALID, PAD, ctx->stream);
  hipMemsetAsync(g->d_code + PAD + n, CODE_INVALID, PAD, ctx->stream);
  hipMemcpyAsync(g->d_rec_off, g->rec_off.data(), n_contigs * 8, hipMemcpyHostToDevice, ctx->stream);
  const uint32_t thr = (uint32_t)(substitution_rate * 4294967296.0);
  NTS_LAUNCH(k_synth, dim3((uint32_t)((n + 4095) / 4096)), dim3(256), 0, ctx->stream, g->d_code + PAD, n, seed_ancestor, seed_genome, thr);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    dev_free(g->d_code);
    dev_free(g->d_rec_off);
    delete g;
    return fail(ctx, NTS_EHIP, std::string("nts_genome_synth: ") + hipGetErrorString(e));
  }
  *out = g;
  return NTS_OK;
}

int nts_genome_synth_plan(nts_ctx* ctx, uint32_t n_rec, const uint64_t* rec_len, uint32_t n_pieces, const nts_synth_piece* pieces,
                          uint64_t seed_ancestor, uint64_t seed_genome, double substitution_rate, nts_genome** out)
{
  return nts_genome_synth_plan_ex(ctx, n_rec, rec_len, n_pieces, pieces, seed_ancestor, seed_genome, substitution_rate, nullptr, out);
}

int nts_genome_synth_plan_ex(nts_ctx* ctx, uint32_t n_rec, const uint64_t* rec_len, uint32_t n_pieces, const nts_synth_piece* pieces,
                             uint64_t seed_ancestor, uint64_t seed_genome, double substitution_rate, const nts_synth_repeats* rep, nts_genome** out)
{
  if (!ctx || !out || !rec_len || !pieces || n_rec == 0 || n_pieces == 0 || substitution_rate < 0 || substitution_rate >= 1)
    return fail(ctx, NTS_EINVAL, "nts_genome_synth_plan: bad arguments");
  SynthRep R;
  memset(&R, 0, sizeof(R));
  if (rep) {
    R.on = 1;
    R.r = *rep;
    const nts_synth_repeats& r = R.r;
    const bool sine_ok = r.sine_prob_256 == 0 || (r.sine_cell_log2 >= 4 && r.sine_cell_log2 <= 20 && r.sine_len >= 1 && r.sine_len <= (1u << r.sine_cell_log2) &&
                                                  r.sine_families >= 1 && r.sine_prob_256 <= 256);
    const bool line_ok = r.line_prob_256 == 0 || (r.line_cell_log2 >= 4 && r.line_cell_log2 <= 20 && r.line_min_len >= 1 && r.line_min_len <= r.line_len &&
                                                  r.line_len <= (1u << r.line_cell_log2) && r.line_families >= 1 && r.line_prob_256 <= 256);
    if (!sine_ok || !line_ok || r.div_min_1024 > r.div_max_1024 || r.div_max_1024 > 1024 || r.sat_div_1024 > 1024)
      return fail(ctx, NTS_EINVAL, "nts_genome_synth_plan_ex: repeat parameters out of range");
  }
  for (uint32_t p = 0; p < n_pieces; ++p)
    if ((pieces[p].flags & NTS_SYNTH_TANDEM) && (!rep || rep->sat_unit == 0))
      return fail(ctx, NTS_EINVAL, "nts_genome_synth_plan_ex: a tandem piece needs rep->sat_unit");
  uint64_t n = 0;
  for (uint32_t r = 0; r < n_rec; ++r) {
    if (rec_len[r] == 0) return fail(ctx, NTS_EINVAL, "nts_genome_synth_plan: empty record");
    n += rec_len[r];
  }
  uint64_t at = 0;
  for (uint32_t p = 0; p < n_pieces; ++p) { // the pieces tile [0, n)
    if (pieces[p].dst != at || pieces[p].len == 0) return fail(ctx, NTS_EINVAL, "nts_genome_synth_plan: the pieces do not tile the genome");
    at += pieces[p].len;
  }
  if (at != n) return fail(ctx, NTS_EINVAL, "nts_genome_synth_plan: the pieces do not add up to the records");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  nts_genome* g = new nts_genome();
  g->n = n;
  g->n_rec = n_rec;
  uint64_t off = 0;
  for (uint32_t r = 0; r < n_rec; ++r) {
    g->rec_off.push_back(off);
    g->rec_len.push_back(rec_len[r]);
    off += rec_len[r];
  }
  g->total_bases = n;
  // maximal stretches of valid bases, clipped to records: the pieces that are not N runs, merged where they touch
  {
    uint32_t r = 0;
    for (uint32_t p = 0; p < n_pieces; ++p) {
      if (pieces[p].flags & NTS_SYNTH_NRUN) continue;
      uint64_t a = pieces[p].dst;
      const uint64_t b = a + pieces[p].len;
      while (a < b) {
        while (g->rec_off[r] + g->rec_len[r] <= a) ++r;
        const uint64_t e = std::min(b, g->rec_off[r] + g->rec_len[r]);
        if (!g->st_b.empty() && g->st_b.back() == a && a != g->rec_off[r])
          g->st_b.back() = e;
        else {
          g->st_a.push_back(a);
          g->st_b.push_back(e);
        }
        a = e;
      }
    }
  }
  nts_synth_piece* d_pieces = nullptr;
  if (dev_malloc((void**)&g->d_code, PAD + n + PAD) != hipSuccess || dev_malloc((void**)&g->d_rec_off, (size_t)n_rec * 8) != hipSuccess ||
      dev_malloc((void**)&d_pieces, (size_t)n_pieces * sizeof(nts_synth_piece)) != hipSuccess) {
    dev_free(g->d_code);
    dev_free(g->d_rec_off);
    dev_free(d_pieces);
    delete g;
    return fail(ctx, NTS_ENOMEM, "nts_genome_synth_plan: hipMalloc");
  }
  hipMemsetAsync(g->d_code, CODE_INVALID, PAD, ctx->stream);
  hipMemsetAsync(g->d_code + PAD + n, CODE_INVALID, PAD, ctx->stream);
  hipMemcpyAsync(g->d_rec_off, g->rec_off.data(), (size_t)n_rec * 8, hipMemcpyHostToDevice, ctx->stream);
  hipMemcpyAsync(d_pieces, pieces, (size_t)n_pieces * sizeof(nts_synth_piece), hipMemcpyHostToDevice, ctx->stream);
  const uint32_t thr = (uint32_t)(substitution_rate * 4294967296.0);
  NTS_LAUNCH(k_synth_plan, dim3((uint32_t)((n + 4095) / 4096)), dim3(256), 0, ctx->stream, g->d_code + PAD, n, d_pieces, n_pieces, seed_ancestor,
                     seed_genome, thr, R);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  dev_free(d_pieces);
  if (e != hipSuccess) {
    dev_free(g->d_code);
    dev_free(g->d_rec_off);
    delete g;
    return fail(ctx, NTS_EHIP, std::string("nts_genome_synth_plan: ") + hipGetErrorString(e));
  }
  *out = g;
  return NTS_OK;
}

int nts_genome_download(nts_ctx* ctx, const nts_genome* g, uint64_t offset, uint64_t len, uint8_t* ascii)
{
  if (!ctx || !g || !ascii || offset > g->n || len > g->n - offset) return fail(ctx, NTS_EINVAL, "nts_genome_download: range outside the genome");
  if (len == 0) return NTS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t* d = (uint8_t*)ws_get(ctx, "decode", len);
  if (!d) return NTS_ENOMEM;
  NTS_LAUNCH(k_decode, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, ctx->stream, g->d_code + PAD + offset, len, d);
  HIP_TRY(ctx, hipMemcpyAsync(ascii, d, len, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

int nts_genome_valid_kmers(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint64_t* n_valid)
{
  if (!ctx || !g || !n_valid || k == 0) return fail(ctx, NTS_EINVAL, "nts_genome_valid_kmers: bad arguments");
  uint64_t v = 0;
  for (size_t s = 0; s < g->st_a.size(); ++s) {
    const uint64_t len = g->st_b[s] - g->st_a[s];
    if (len >= k) v += len - k + 1;
  }
  *n_valid = v;
  return NTS_OK;
}

// ---- Bloom filter ------------------------------------------------------------------------------------
static int bf_create_alloc(nts_ctx* ctx, uint64_t bytes, uint64_t alloc, nts_bf** out)
{
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  nts_bf* bf = new nts_bf();
  bf->bytes = bytes;
  bf->alloc_bytes = alloc;
  // NTS_BF_MEM=uncached | finegrained: the filter in memory the L2 does not cache / keeps coherent (an experiment: does a random
  // 4-byte probe then move less than a 128-byte line?  DESIGN.md 4.2)
  hipError_t e;
  const char* kind = NTS_KNOB("NTS_BF_MEM");
  if (kind && !strcmp(kind, "uncached"))
    e = dev_malloc_flags((void**)&bf->d_words, alloc, hipDeviceMallocUncached);
  else if (kind && !strcmp(kind, "finegrained"))
    e = dev_malloc_flags((void**)&bf->d_words, alloc, hipDeviceMallocFinegrained);
  else
    e = dev_malloc((void**)&bf->d_words, alloc);
  if (e != hipSuccess) {
    delete bf;
    return fail(ctx, NTS_ENOMEM, std::string("hipMalloc bloom: ") + hipGetErrorString(e));
  }
  e = hipMemsetAsync(bf->d_words, 0, alloc, ctx->stream);
  if (e != hipSuccess) {
    dev_free(bf->d_words);
    delete bf;
    return fail(ctx, NTS_EHIP, "hipMemset bloom");
  }
  bf->popcnt = 0;
  *out = bf;
  return NTS_OK;
}

int nts_bf_create(nts_ctx* ctx, uint64_t bytes, nts_bf** out)
{
  if (!ctx || !out || bytes == 0) return fail(ctx, NTS_EINVAL, "nts_bf_create: bytes must be positive");
  return bf_create_alloc(ctx, bytes, (bytes + 15) / 16 * 16, out); // AND / popcount run on 16-byte lanes; tail stays zero
}

int nts_bf_create_sharded(nts_ctx* ctx, uint64_t bytes, int world, nts_bf** out)
{
  if (!ctx || !out || bytes == 0 || world < 1) return fail(ctx, NTS_EINVAL, "nts_bf_create_sharded: bad arguments");
  const uint64_t chunk = ((bytes + world - 1) / world + 15) / 16 * 16; // the layout nts_bf_allreduce_and exchanges
  return bf_create_alloc(ctx, bytes, chunk * world, out);
}

int nts_bf_fill_ones(nts_ctx* ctx, nts_bf* bf)
{
  if (!ctx || !bf) return fail(ctx, NTS_EINVAL, "nts_bf_fill_ones: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  bf->popcnt = -1;
  ++bf->version;
  HIP_TRY(ctx, hipMemsetAsync(bf->d_words, 0xFF, bf->bytes, ctx->stream)); // the pad behind `bytes` stays zero
  return NTS_OK;
}

void nts_bf_free(nts_ctx* ctx, nts_bf* bf)
{
  if (!bf) return;
  if (ctx) hipSetDevice(ctx->device);
  if (bf->d_words && bf->owned) dev_free(bf->d_words);
  if (bf->d_summary) dev_free(bf->d_summary);
  if (bf->d_fold) dev_free(bf->d_fold);
  delete bf;
}

int nts_bf_wrap(nts_ctx* ctx, void* device_ptr, uint64_t bytes, nts_bf** out)
{
  if (!ctx || !out || !device_ptr || bytes == 0 || ((uintptr_t)device_ptr % 16) != 0)
    return fail(ctx, NTS_EINVAL, "nts_bf_wrap: need a 16-byte aligned buffer and a positive byte count");
  nts_bf* bf = new nts_bf();
  bf->bytes = bytes;
  bf->alloc_bytes = (bytes + 15) / 16 * 16;
  bf->d_words = (uint32_t*)device_ptr;
  bf->owned = false;
  *out = bf;
  return NTS_OK;
}

int nts_and_raw(nts_ctx* ctx, void* acc_dev, const void* other_dev, uint64_t bytes)
{
  if (!ctx || !acc_dev || !other_dev || (bytes % 16) != 0 || ((uintptr_t)acc_dev % 16) != 0 || ((uintptr_t)other_dev % 16) != 0)
    return fail(ctx, NTS_EINVAL, "nts_and_raw: buffers must be 16-byte aligned, size a multiple of 16");
  if (bytes == 0) return NTS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t n16 = bytes / 16;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, 256 * 8);
  ScopedTimer t(ctx, "bf_and");
  NTS_LAUNCH(k_bf_and, dim3(blocks), dim3(256), 0, ctx->stream, (uint4*)acc_dev, (const uint4*)other_dev, n16);
  HIP_TRY(ctx, hipGetLastError());
  return NTS_OK;
}

uint64_t nts_bf_bytes(const nts_bf* bf)
{
  return bf ? bf->bytes : 0;
}

void* nts_bf_device_ptr(nts_bf* bf)
{
  // the caller may write through this pointer (collectives on the filter's own memory): what the library remembers about the
  // filter's contents -- "holds no bit" (the store-only finish of the partitioned build), the cached popcount, the summary -- is void
  if (bf) {
    bf->popcnt = -1;
    ++bf->version;
  }
  return bf ? (void*)bf->d_words : nullptr;
}

int nts_bf_clear(nts_ctx* ctx, nts_bf* bf)
{
  if (!ctx || !bf) return fail(ctx, NTS_EINVAL, "nts_bf_clear: bad arguments");
  bf->popcnt = 0;
  ++bf->version;
  HIP_TRY(ctx, hipMemsetAsync(bf->d_words, 0, (bf->bytes + 15) / 16 * 16, ctx->stream));
  return NTS_OK;
}

static int bf_hash_pass(nts_ctx* ctx, const nts_bf* prev, nts_bf* next, const nts_genome* g, uint32_t k)
{
  if (!ctx || !next || !g || k == 0) return fail(ctx, NTS_EINVAL, "bloom pass: bad arguments");
  if (prev && prev->bytes != next->bytes) return fail(ctx, NTS_EINVAL, "bloom pass: filters differ in size");
  const bool was_empty = next->popcnt == 0; // (a new or cleared filter)
  next->popcnt = -1;
  ++next->version;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  GenomeTables scratch;
  const GenomeTables* T = nullptr;
  int rc = get_tables(ctx, g, k, nullptr, 0, scratch, &T);
  if (rc) return rc;
  uint32_t* late_ctl = (uint32_t*)(ctx->mail + MAIL_WORDS - 8); // (pinned; the mailbox proper ends below: its users wait for their own flag)
  for (int i = 0; i < 8; ++i) late_ctl[i] = 0;
  if (prev) {
    rc = launch_hash<MODE_CASCADE>(ctx, "bf_cascade", g, *T, k, prev, next, nullptr);
  } else {
    rc = ctx->bf_build_mode == 1 ? 1 : bf_insert_binned(ctx, next, g, *T, k, ctx->bf_build_mode == 2, was_empty, 0, late_ctl);
    if (rc == 1) rc = launch_hash<MODE_INSERT>(ctx, "bf_insert", g, *T, k, nullptr, next, nullptr);
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // run table buffers are released on return
  if (!prev) {
    ctx->last_bf_direct = (uint64_t)late_ctl[2] | ((uint64_t)late_ctl[3] << 32);
    ctx->last_bf_fallback = late_ctl[0];
    // the store-only finish counted the bits it left (k_bin3 + k_bin_late): the filter was empty, so that is its popcount
    if (was_empty && late_ctl[6] == 1u && late_ctl[0] == 0u && next->owned) next->popcnt = (int64_t)((uint64_t)late_ctl[4] | ((uint64_t)late_ctl[5] << 32));
  }
  return NTS_OK;
}

// acc &= the filter of genome g: one level of the reference's cascade (src/ntsynt_make_common_bf.cpp:134-160; with one hash
// function `new_bf[h] |= bf.contains(h)` over a genome's k-mers is the AND of the running filter with the genome's own filter,
// SURVEY.md F8) inside the partitioned build's last pass: no second filter, no clearing of it, no separate AND pass -- per level
// 14.8 GB read + <= 14.8 GB written at 3 Gbp instead of 14.8 (memset) + 14.8 (store) + 44.4 (k_bf_and).  Where the partitioned
// build does not apply (small inputs, k > 128, forced atomic mode) or its list of bucket-bypassing indices ran full, the level is
// done the plain way: the genome's filter in a temporary allocation, then k_bf_and.
int nts_bf_insert_and(nts_ctx* ctx, nts_bf* acc, const nts_genome* g, uint32_t k)
{
  if (!ctx || !acc || !g || k == 0) return fail(ctx, NTS_EINVAL, "nts_bf_insert_and: bad arguments");
  const int64_t pop_before = acc->owned ? acc->popcnt : -1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  GenomeTables scratch;
  const GenomeTables* T = nullptr;
  int rc = get_tables(ctx, g, k, nullptr, 0, scratch, &T);
  if (rc) return rc;
  uint32_t* late_ctl = (uint32_t*)(ctx->mail + MAIL_WORDS - 8);
  for (int i = 0; i < 8; ++i) late_ctl[i] = 0;
  // a running filter that is all but empty: the literal level through the sparse-filter lookups (bf_level_sparse) instead of a build
  rc = bf_level_sparse(ctx, acc, g, *T, k, pop_before);
  if (rc < 0) return rc;
  if (rc == 0) {
    ctx->last_bf_direct = 0;
    ctx->last_bf_fallback = 0;
    return NTS_OK;
  }
  acc->popcnt = -1;
  ++acc->version;
  // (a running filter known to hold fewer than one bit per 2^12: most 64 KiB slices are empty, k_bin3 looks before it reads residues)
  const bool sparse = pop_before >= 0 && (uint64_t)pop_before < ((acc->bytes * 8) >> 12);
  const bool fused_ok = ctx->bf_build_mode != 1 && !(NTS_KNOB("NTS_BIN_FUSED_AND") && atoi(NTS_KNOB("NTS_BIN_FUSED_AND")) == 0);
  rc = fused_ok ? bf_insert_binned(ctx, acc, g, *T, k, ctx->bf_build_mode == 2, false, sparse ? 2 : 1, late_ctl) : 1;
  if (rc != 0 && rc != 1) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->last_bf_direct = (uint64_t)late_ctl[2] | ((uint64_t)late_ctl[3] << 32);
  ctx->last_bf_fallback = 0;
  if (rc == 0 && late_ctl[0] == 0) {
    if (late_ctl[6] == 1u && acc->owned) acc->popcnt = (int64_t)((uint64_t)late_ctl[4] | ((uint64_t)late_ctl[5] << 32)); // counted by the AND finish
    return NTS_OK;
  }
  // the plain way (acc is untouched: the fused build did not apply, or it gave up and put its parked bits back)
  ctx->last_bf_fallback = rc == 0 ? 1u : 0u;
  nts_bf* own = nullptr;
  if (int rc2 = bf_create_alloc(ctx, acc->bytes, (acc->bytes + 15) / 16 * 16, &own)) return rc2;
  rc = bf_hash_pass(ctx, nullptr, own, g, k);
  if (rc == NTS_OK) {
    const uint64_t n16 = (acc->bytes + 15) / 16;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, 256 * 8);
    ScopedTimer t(ctx, "bf_and");
    NTS_LAUNCH(k_bf_and, dim3(blocks), dim3(256), 0, ctx->stream, (uint4*)acc->d_words, (const uint4*)own->d_words, n16);
  }
  const uint32_t fb = ctx->last_bf_fallback;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  nts_bf_free(ctx, own);
  ctx->last_bf_fallback = fb;
  if (rc) return rc;
  if (e != hipSuccess) return fail(ctx, NTS_EHIP, std::string("nts_bf_insert_and: ") + hipGetErrorString(e));
  return NTS_OK;
}

// bin/ntsynt_make_repeat_bfs.py:56-67 for one genome: every k-mer whose bit is already set in `genome_bf` (a k-mer seen before in
// this genome, or a collision) sets its bit in `repeat_bf`, otherwise in `genome_bf`.  The outcome -- the bits hit at least
// twice -- does not depend on the order of the k-mers.
int nts_bf_insert_repeats(nts_ctx* ctx, nts_bf* genome_bf, nts_bf* repeat_bf, const nts_genome* g, uint32_t k)
{
  if (!ctx || !genome_bf || !repeat_bf || !g || k == 0) return fail(ctx, NTS_EINVAL, "nts_bf_insert_repeats: bad arguments");
  if (genome_bf->bytes != repeat_bf->bytes) return fail(ctx, NTS_EINVAL, "nts_bf_insert_repeats: filters differ in size");
  genome_bf->popcnt = -1;
  ++genome_bf->version;
  repeat_bf->popcnt = -1;
  ++repeat_bf->version;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  GenomeTables scratch;
  const GenomeTables* T = nullptr;
  int rc = get_tables(ctx, g, k, nullptr, 0, scratch, &T);
  if (rc) return rc;
  rc = launch_hash<MODE_REPEAT>(ctx, "bf_repeats", g, *T, k, genome_bf, repeat_bf, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

int nts_bf_build_mode(nts_ctx* ctx, int mode)
{
  if (!ctx || mode < 0 || mode > 2) return fail(ctx, NTS_EINVAL, "nts_bf_build_mode: mode must be 0 (auto), 1 (atomic) or 2 (binned)");
  ctx->bf_build_mode = mode;
  return NTS_OK;
}

int nts_bf_insert(nts_ctx* ctx, nts_bf* bf, const nts_genome* g, uint32_t k)
{
  return bf_hash_pass(ctx, nullptr, bf, g, k);
}

int nts_bf_cascade(nts_ctx* ctx, const nts_bf* prev, nts_bf* next, const nts_genome* g, uint32_t k)
{
  if (!prev) return fail(ctx, NTS_EINVAL, "nts_bf_cascade: prev is NULL");
  return bf_hash_pass(ctx, prev, next, g, k);
}

int nts_bf_and(nts_ctx* ctx, nts_bf* acc, const nts_bf* other)
{
  if (!ctx || !acc || !other || acc->bytes != other->bytes) return fail(ctx, NTS_EINVAL, "nts_bf_and: filters differ in size");
  acc->popcnt = -1;
  ++acc->version;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t n16 = (acc->bytes + 15) / 16;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, 256 * 8);
  ScopedTimer t(ctx, "bf_and");
  NTS_LAUNCH(k_bf_and, dim3(blocks), dim3(256), 0, ctx->stream, (uint4*)acc->d_words, (const uint4*)other->d_words, n16);
  HIP_TRY(ctx, hipGetLastError());
  return NTS_OK;
}

int nts_bf_popcount(nts_ctx* ctx, const nts_bf* bf, uint64_t* bits_set)
{
  if (!ctx || !bf || !bits_set) return fail(ctx, NTS_EINVAL, "nts_bf_popcount: bad arguments");
  if (bf->popcnt >= 0 && bf->owned) {
    *bits_set = (uint64_t)bf->popcnt;
    return NTS_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  unsigned long long* d = (unsigned long long*)ws_get(ctx, "popcnt", 64); // (not hipMalloc + hipFree: a hipFree waits for the whole device)
  if (!d) return NTS_ENOMEM;
  hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
  const uint64_t n16 = (bf->bytes + 15) / 16;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, 256 * 8);
  {
    ScopedTimer t(ctx, "bf_popcount");
    NTS_LAUNCH(k_bf_popcount, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4*)bf->d_words, n16, d);
  }
  unsigned long long h = 0;
  hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(ctx, NTS_EHIP, std::string("popcount: ") + hipGetErrorString(e));
  bf->popcnt = (int64_t)h;
  *bits_set = h;
  return NTS_OK;
}

int nts_bench_random_probe(nts_ctx* ctx, const nts_bf* bf, uint64_t n_probes, uint32_t repeats, double* avg_ms, uint64_t* hits)
{
  if (!ctx || !bf || !avg_ms || n_probes == 0 || repeats == 0) return fail(ctx, NTS_EINVAL, "nts_bench_random_probe: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  unsigned long long* d = (unsigned long long*)ws_get(ctx, "bench_sink", 8);
  if (!d) return NTS_ENOMEM;
  HIP_TRY(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
  const FastMod fm = make_fastmod(bf->bytes * 8);
  const uint64_t blocks = (n_probes + 8191) / 8192;
  if (blocks > 0x7FFFFFFFULL) return fail(ctx, NTS_ERANGE, "nts_bench_random_probe: too many probes");
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  NTS_LAUNCH(k_bench_probe, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, bf->d_words, fm, n_probes, (uint64_t)1, d); // warm-up
  hipEventRecord(a, ctx->stream);
  for (uint32_t r = 0; r < repeats; ++r)
    NTS_LAUNCH(k_bench_probe, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, bf->d_words, fm, n_probes, (uint64_t)(r + 2), d);
  hipEventRecord(b, ctx->stream);
  unsigned long long h = 0;
  hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  if (e != hipSuccess) return fail(ctx, NTS_EHIP, std::string("bench: ") + hipGetErrorString(e));
  *avg_ms = ms / repeats;
  if (hits) *hits = h;
  return NTS_OK;
}

int nts_mod_indices(nts_ctx* ctx, uint64_t bits, int form, const uint64_t* h, uint64_t n, uint64_t* out)
{
  if (!ctx || bits < 2 || !h || !out || form < -1 || form > 2) return fail(ctx, NTS_EINVAL, "nts_mod_indices: bad arguments");
  if (n == 0) return NTS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t* d = (uint64_t*)ws_get(ctx, "mod_idx", n * 8);
  if (!d) return NTS_ENOMEM;
  FastMod fm = make_fastmod(bits);
  if (form >= 0) fm.form = std::min<uint32_t>(fm.form, (uint32_t)form);
  HIP_TRY(ctx, hipMemcpyAsync(d, h, n * 8, hipMemcpyHostToDevice, ctx->stream));
  NTS_LAUNCH(k_mod_indices, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65535)), dim3(256), 0, ctx->stream, d, n, fm);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

int nts_bench_valu(nts_ctx* ctx, int kind, uint32_t waves_per_simd, uint32_t iters, double* wall_ms, double* cycles_per_instr,
                   double* instr_per_wave)
{
  if (!ctx || kind < 0 || kind >= VK_COUNT || waves_per_simd == 0 || waves_per_simd > 8 || iters == 0)
    return fail(ctx, NTS_EINVAL, "nts_bench_valu: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
  const uint32_t cus = (uint32_t)prop.multiProcessorCount;
  // a 256-lane workgroup = four waves, one per SIMD of its CU; `waves_per_simd` workgroups per CU
  const uint32_t blocks = cus * waves_per_simd;
  const uint64_t n_waves = (uint64_t)blocks * 4;
  unsigned long long* d_cyc = (unsigned long long*)ws_get(ctx, "valu_cycles", n_waves * 8);
  uint32_t* d_sink = (uint32_t*)ws_get(ctx, "valu_sink", 64);
  if (!d_cyc || !d_sink) return NTS_ENOMEM;
  ValuBenchFn fn = valu_bench_fn(kind);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  NTS_LAUNCH(fn, dim3(blocks), dim3(256), 0, ctx->stream, iters / 8 + 1, d_cyc, d_sink); // warm-up (clocks, code)
  hipEventRecord(a, ctx->stream);
  NTS_LAUNCH(fn, dim3(blocks), dim3(256), 0, ctx->stream, iters, d_cyc, d_sink);
  hipEventRecord(b, ctx->stream);
  std::vector<unsigned long long> cyc(n_waves);
  hipMemcpyAsync(cyc.data(), d_cyc, n_waves * 8, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  if (e != hipSuccess) return fail(ctx, NTS_EHIP, std::string("nts_bench_valu: ") + hipGetErrorString(e));
  const double per_iter = (double)VB_CHAINS * VB_UNROLL * (kind == VK_ADD_CO_PAIR || kind == VK_CMP_ADDC ? 2 : kind == VK_ROLL31_STEP ? 9 : 1);
  const double n_instr = per_iter * iters;
  std::sort(cyc.begin(), cyc.end());
  const double median = (double)cyc[n_waves / 2];
  if (wall_ms) *wall_ms = ms;
  // all waves of a SIMD run their loops at the same time: the SIMD issued waves_per_simd x n_instr in `median` cycles
  if (cycles_per_instr) *cycles_per_instr = median / (n_instr * waves_per_simd);
  if (instr_per_wave) *instr_per_wave = n_instr;
  return NTS_OK;
}

int nts_bf_download(nts_ctx* ctx, const nts_bf* bf, uint8_t* host, uint64_t bytes)
{
  if (!ctx || !bf || !host || bytes != bf->bytes) return fail(ctx, NTS_EINVAL, "nts_bf_download: size mismatch");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // The copy runs on its own stream, after everything queued on the compute stream so far: a caller may run it
  // on a second host thread (the pipeline writes the filter file behind the sketches) without stalling later kernels.
  hipEvent_t ready;
  HIP_TRY(ctx, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ready, ctx->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ready, 0);
  if (e == hipSuccess) e = hipMemcpyAsync(host, bf->d_words, bytes, hipMemcpyDeviceToHost, ctx->copy_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
  hipEventDestroy(ready);
  HIP_TRY(ctx, e);
  return NTS_OK;
}

int nts_bf_upload(nts_ctx* ctx, nts_bf* bf, const uint8_t* host, uint64_t bytes)
{
  if (!ctx || !bf || !host || bytes != bf->bytes) return fail(ctx, NTS_EINVAL, "nts_bf_upload: size mismatch");
  bf->popcnt = -1;
  ++bf->version;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(bf->d_words, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

// bf->save(path) (src/ntsynt_make_common_bf.cpp:164) without a host copy of the filter: `header` first, then the bit array
// streamed out of HBM by a few host threads, each with its own pinned staging buffer and HIP stream -- a device -> host copy
// of the next chunk runs while the previous one is written to the file.  Sees everything queued on the context's
// stream before the call; safe to run on a second host thread while the first keeps sketching.
int nts_bf_save(nts_ctx* ctx, const nts_bf* bf, const char* path, const void* header, uint64_t header_bytes, uint32_t n_threads)
{
  if (!ctx || !bf || !path || (header_bytes && !header)) return fail(ctx, NTS_EINVAL, "nts_bf_save: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipEvent_t ready;
  HIP_TRY(ctx, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(ready, ctx->stream));
  const int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0644);
  if (fd < 0) {
    hipEventDestroy(ready);
    return fail(ctx, NTS_EINVAL, std::string("nts_bf_save: cannot create ") + path);
  }
  // (positional writes from a few threads; a shared file mapping was tried and took twice as long -- 3.6 M first-touch
  // faults on the mapping for a 14.8 GB filter)
  auto write_at = [fd](const void* src, uint64_t len, uint64_t at) {
    const uint8_t* p = (const uint8_t*)src;
    while (len) {
      const ssize_t got = pwrite(fd, p, len, (off_t)at);
      if (got <= 0) return false;
      p += got;
      at += (uint64_t)got;
      len -= (uint64_t)got;
    }
    return true;
  };
  bool ok = header_bytes == 0 || write_at(header, header_bytes, 0);
  std::atomic<uint64_t> next(0);
  std::atomic<bool> failed(false);
  const uint64_t CHUNK = IO_CHUNK;
  const uint64_t n_chunks = (bf->bytes + CHUNK - 1) / CHUNK;
  const uint8_t* src = (const uint8_t*)bf->d_words;
  const int device = ctx->device;
  // a small filter does not pay for six threads: one per 64 MiB, at most n_threads (default 6)
  const unsigned T = ok ? io_lanes(ctx, ctx->io_down, (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_threads ? n_threads : 6, (n_chunks + 7) / 8))) : 0;
  if (ok && T == 0) ok = false;
  auto worker = [&](unsigned lane) {
    if (hipSetDevice(device) != hipSuccess) {
      failed.store(true);
      return;
    }
    nts_ctx::IoLane& io = ctx->io_down[lane];
    hipStream_t st = io.stream;
    uint8_t* const* stage = io.stage;
    if (hipStreamWaitEvent(st, ready, 0) != hipSuccess) {
      failed.store(true);
      return;
    }
    // two chunks in flight per thread: while chunk c is written to the file, chunk c' is on its way from HBM
    uint64_t cur = next.fetch_add(1), cur_len = 0;
    int slot = 0;
    if (cur < n_chunks) {
      cur_len = std::min(CHUNK, bf->bytes - cur * CHUNK);
      if (hipMemcpyAsync(stage[slot], src + cur * CHUNK, cur_len, hipMemcpyDeviceToHost, st) != hipSuccess) failed.store(true);
    }
    while (cur < n_chunks && !failed.load()) {
      if (hipStreamSynchronize(st) != hipSuccess) {
        failed.store(true);
        break;
      }
      const uint64_t nxt = next.fetch_add(1);
      uint64_t nxt_len = 0;
      if (nxt < n_chunks) {
        nxt_len = std::min(CHUNK, bf->bytes - nxt * CHUNK);
        if (hipMemcpyAsync(stage[slot ^ 1], src + nxt * CHUNK, nxt_len, hipMemcpyDeviceToHost, st) != hipSuccess) failed.store(true);
      }
      if (!write_at(stage[slot], cur_len, header_bytes + cur * CHUNK)) failed.store(true);
      cur = nxt;
      cur_len = nxt_len;
      slot ^= 1;
    }
    hipStreamSynchronize(st);
  };
  if (ok) {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
  }
  close(fd);
  hipEventDestroy(ready);
  if (!ok || failed.load()) return fail(ctx, NTS_EHIP, std::string("nts_bf_save: writing ") + path + " failed");
  return NTS_OK;
}

// ---- sketch ---------------------------------------------------------------------------------------------
int nts_hash_all(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint64_t** h0, uint64_t* n_out)
{
  if (!ctx || !g || !h0 || !n_out || k == 0) return fail(ctx, NTS_EINVAL, "nts_hash_all: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  GenomeTables scratch;
  const GenomeTables* T = nullptr;
  int rc = get_tables(ctx, g, k, nullptr, 0, scratch, &T);
  if (rc) return rc;
  const RunTable& rt = T->rt;
  uint64_t* d_keys = nullptr;
  uint64_t* d_lin = nullptr;
  HIP_TRY(ctx, dev_malloc((void**)&d_keys, key_buffer_elems(rt.n_valid) * 8));
  HIP_TRY(ctx, dev_malloc((void**)&d_lin, std::max<uint64_t>(rt.n_valid, 1) * 8));
  rc = launch_hash<MODE_KEYS>(ctx, "hash_only", g, *T, k, nullptr, nullptr, d_keys);
  uint64_t* host = (uint64_t*)malloc(std::max<uint64_t>(rt.n_valid, 1) * 8);
  if (rc == NTS_OK && rt.n_valid) {
    NTS_LAUNCH(k_keys_linear, dim3((uint32_t)((rt.n_valid + 255) / 256)), dim3(256), 0, ctx->stream, d_keys, rt.n_valid, d_lin);
    hipMemcpyAsync(host, d_lin, rt.n_valid * 8, hipMemcpyDeviceToHost, ctx->stream);
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  dev_free(d_keys);
  dev_free(d_lin);
  if (rc != NTS_OK || e != hipSuccess) {
    free(host);
    return rc != NTS_OK ? rc : fail(ctx, NTS_EHIP, std::string("hash_all: ") + hipGetErrorString(e));
  }
  *h0 = host;
  *n_out = rt.n_valid;
  return NTS_OK;
}

} // extern "C"

namespace {

struct OutSegs
{
  uint64_t* d_j = nullptr;
  uint64_t* d_key = nullptr;
  unsigned long long* d_count = nullptr;
  uint64_t seg_cap = 0;
  uint32_t* d_tile_cnt = nullptr; // per-tile mode (see WinParams): d_j/d_key are [n_tiles][tile_cap]
  uint32_t tile_cap = 0;
};

// window tiles over a table of (pseudo-)records; returns the number of tiles
uint64_t tiles_of(const std::vector<uint64_t>& nv, uint32_t w, std::vector<uint64_t>& tile_start)
{
  tile_start.assign(nv.size() + 1, 0);
  for (size_t r = 0; r < nv.size(); ++r) {
    const uint64_t n_win = nv[r] >= w ? nv[r] - w + 1 : 0;
    tile_start[r + 1] = tile_start[r] + (n_win + WIN_TILE - 1) / WIN_TILE;
  }
  return tile_start.back();
}

// dense window kernel over (pseudo-)records already resident on the device; keys must be present for them
struct WinFuse // what k_window_min<true> hashes and probes with (launch_window_dense: d_keys == nullptr)
{
  const uint8_t* code;
  const uint64_t* run_pos;
  const uint64_t* run_vstart;
  uint32_t n_runs;
  HashParams hp;
  const uint32_t* bf;
  FastMod fm;
  uint64_t* dir_off = nullptr; // ordered output (whole genome): the tiles' directory, see WinParams
  uint32_t* dir_cnt = nullptr;
};

int launch_window_dense(nts_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_vs, const uint64_t* d_nv, const uint64_t* d_ts,
                        uint32_t n_rec, uint64_t n_tiles, uint32_t w, const OutSegs& out, const char* tag,
                        const uint32_t* d_tile_ids = nullptr, uint64_t n_tile_ids = 0, const WinFuse* fuse = nullptr)
{
  if (n_tiles == 0) return NTS_OK;
  if (n_tiles > 0x7FFFFFFFULL) return fail(ctx, NTS_ERANGE, "too many window tiles for one launch");
  WinParams P;
  P.keys = d_keys;
  P.tile_ids = d_tile_ids;
  P.n_tile_ids = (uint32_t)n_tile_ids;
  P.rec_vstart = d_vs;
  P.rec_nv = d_nv;
  P.tile_start = d_ts;
  P.n_rec = n_rec;
  P.w = w;
  P.chunk = std::min<uint32_t>(WIN_CHUNK, w);
  const uint32_t max_full = w / P.chunk;
  P.levels = 1;
  while ((1u << P.levels) <= max_full) ++P.levels;
  const uint32_t E_max = WIN_TILE + 1 + w - 1;
  const uint32_t chunks_max = (E_max + P.chunk - 1) / P.chunk;
  size_t lds = (size_t)(E_max + E_max / 32 + 2) * 8 + 16 + (size_t)(WIN_TILE + 16) * 2 + (size_t)P.levels * chunks_max * 2 + 64;
  if (fuse) lds += 16 + 36 * 8 + (size_t)E_max + FAST_K_MAX + 48; // roll tables and seeds, the tile's bases
  if (lds > 160 * 1024) return fail(ctx, NTS_ERANGE, "window tile does not fit LDS");
  if (fuse) {
    if (lds > ctx->win_fused_lds_set) {
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_window_min<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      ctx->win_fused_lds_set = lds;
    }
  } else if (lds > ctx->win_lds_set) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_window_min<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->win_lds_set = lds;
  }
  P.out_j = out.d_j;
  P.out_key = out.d_key;
  P.seg_count = out.d_count;
  P.seg_cap = out.seg_cap;
  P.tile_cnt = out.d_tile_cnt;
  P.tile_cap = out.tile_cap;
  P.tile_any = (d_tile_ids == nullptr && out.d_tile_cnt == nullptr && !fuse) ? ctx->cur_tile_any : nullptr;
  P.code = nullptr;
  P.run_pos = P.run_vstart = nullptr;
  P.n_runs = 0;
  P.bf = nullptr;
  P.dir_off = nullptr;
  P.dir_cnt = nullptr;
  if (fuse && !out.d_tile_cnt) {
    P.dir_off = fuse->dir_off;
    P.dir_cnt = fuse->dir_cnt;
  }
  if (fuse) {
    P.code = fuse->code;
    P.run_pos = fuse->run_pos;
    P.run_vstart = fuse->run_vstart;
    P.n_runs = fuse->n_runs;
    P.hp = fuse->hp;
    P.bf = fuse->bf;
    P.fm = fuse->fm;
  }
  ScopedTimer t(ctx, tag, fuse != nullptr);
  // (a launch counts its work-items in 32 bits: tiles by the million -- one per uncovered range of a selection forced to list next to
  //  nothing -- go out in as many launches as it takes)
  uint64_t per_launch = 0xFFFFFFFFull / WIN_THREADS;
  if (NTS_KNOB("NTS_WIN_TILES_PER_LAUNCH")) per_launch = strtoull(NTS_KNOB("NTS_WIN_TILES_PER_LAUNCH"), nullptr, 10);
  for (uint64_t t0 = 0; t0 < n_tiles; t0 += per_launch) {
    const uint32_t n_now = (uint32_t)std::min<uint64_t>(per_launch, n_tiles - t0);
    P.tile_base = (uint32_t)t0;
    if (fuse)
      NTS_LAUNCH(k_window_min<true>, dim3(n_now), dim3(WIN_THREADS), lds, ctx->stream, P);
    else
      NTS_LAUNCH(k_window_min<false>, dim3(n_now), dim3(WIN_THREADS), lds, ctx->stream, P);
  }
  HIP_TRY(ctx, hipGetLastError());
  return NTS_OK;
}

constexpr uint32_t GAP_PEEK = 8192; // uncovered ranges fetched together with the counters (a second round trip for more: 0.1 ms per 3 Gbp sketch when a family with insertions had 1500)

// small device results -> the context's pinned mailbox (up to 6 ranges of 64-bit words)
struct MailParams
{
  const uint64_t* src[8];
  uint32_t n[8];
  uint32_t off[8];
  const unsigned long long* limit[8]; // if not null: only the first min(n, *limit) words of the range are wanted (a list and its device-side length)
  uint32_t count;
  uint64_t seq; // arrival flag value, written to the last word of the mailbox
  uint64_t* mail;
  uint64_t* zero; // after the copies: words to clear (the next call's control block: it then starts without a memset)
  uint32_t n_zero;
};

__global__ __launch_bounds__(256) void k_mail(MailParams P)
{
  // (the mailbox is host memory: a word written is a word across PCIe -- 16384 words of a list that holds 969 took 35 us)
  uint32_t n_eff[8];
#pragma unroll
  for (uint32_t s = 0; s < 8; ++s)
    n_eff[s] = s < P.count ? (P.limit[s] ? (uint32_t)min((unsigned long long)P.n[s], *P.limit[s]) : P.n[s]) : 0u;
  __syncthreads(); // (a length may sit among the words cleared below: every lane has read it before any lane clears)
#pragma unroll
  for (uint32_t s = 0; s < 8; ++s)
    for (uint32_t i = threadIdx.x; i < n_eff[s]; i += 256) P.mail[P.off[s] + i] = P.src[s][i];
  for (uint32_t i = threadIdx.x; i < P.n_zero; i += 256) P.zero[i] = 0; // (not among the sources of this mail)
  // arrival flag for the polling host: after every lane's values are visible system-wide
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&P.mail[MAIL_WORDS - 1], P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct Mail
{
  MailParams P;
  uint32_t used = 0;
  Mail(nts_ctx* ctx)
  {
    P.count = 0;
    P.mail = ctx->d_mail;
    P.zero = nullptr;
    P.n_zero = 0;
  }
  void clear_after(uint64_t* dev, uint32_t n_words)
  {
    P.zero = dev;
    P.n_zero = n_words;
  }
  // returns the word offset of the range in the mailbox
  uint32_t add(const void* dev, uint32_t n_words, const unsigned long long* dev_limit = nullptr)
  {
    P.src[P.count] = (const uint64_t*)dev;
    P.limit[P.count] = dev_limit;
    P.n[P.count] = n_words;
    P.off[P.count] = used;
    ++P.count;
    used += n_words;
    return used - n_words;
  }
  // launch + wait: afterwards ctx->mail[...] holds the values.  The host polls the arrival flag in the pinned page
  // (a few microseconds after the kernel's last store) instead of sleeping in hipStreamSynchronize (~25 us to wake
  // up); the stream is in order, so the flag also means that everything queued before has finished.
  int post(nts_ctx* ctx)
  {
    if (int rc = launch(ctx)) return rc;
    return wait(ctx);
  }
  // the two halves of post(): work queued between them runs while the values travel and the host gets going again
  int launch(nts_ctx* ctx)
  {
    P.seq = ++ctx->mail_seq;
    NTS_LAUNCH(k_mail, dim3(1), dim3(256), 0, ctx->stream, P);
    HIP_TRY(ctx, hipGetLastError());
    return NTS_OK;
  }
  int wait(nts_ctx* ctx)
  {
    volatile uint64_t* flag = ctx->mail + (MAIL_WORDS - 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0; *flag != P.seq; ++spin) {
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // long-running work in front of us: sleep instead
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag != P.seq) return fail(ctx, NTS_EHIP, "result mailbox: flag did not arrive");
    return NTS_OK;
  }
};

struct SortedOut
{
  uint64_t* d_j = nullptr;   // compact indices of the minimizers, ascending
  uint64_t* d_key = nullptr; // their keys (h0)
  uint64_t count = 0;        // their number -- or an upper bound when d_ctl is set:
  // few uncovered ranges are merged in on the device; d_ctl[1] = number of minimizers, d_ctl[2] = 1 if that path
  // gave up (the call is repeated with ctx->small_gap_path = false)
  uint64_t* d_ctl = nullptr;
  // ... in which case d_j/d_key hold the first `na` of them (the sparse winners) and b_j/b_key the others (the winners of
  // the uncovered ranges, d_ctl[0] of them): k_finalize merges the two lists
  uint64_t* b_j = nullptr;
  uint64_t* b_key = nullptr;
  uint64_t na = 0;
};

// dense kernels over (pseudo-)records into segmented buffers, then a sort: `res` gets the ordered list.
// tiles: optional list of key tiles to hash (uncovered ranges only); nullptr = all tiles.
// sparse: when given (the ordered winners of the pruned pass) and the uncovered ranges are few, their winners are sorted
// and handed on as a second list next to it (k_gap_collect; k_finalize merges): `res` then has res.d_ctl and res.b_j set
// and res.count is an upper bound, and no synchronisation happens here.
int run_dense_sorted(nts_ctx* ctx, const nts_genome* g, const GenomeTables& T, uint32_t k, uint32_t w, const nts_bf* filter,
                     const std::vector<uint64_t>* pseudo_vstart, const std::vector<uint64_t>* pseudo_nv, const std::vector<uint32_t>* tiles,
                     uint64_t est_kmers, const char* slot_prefix, SortedOut& res, const SortedOut* sparse = nullptr,
                     const std::vector<uint32_t>* tile_spans = nullptr)
{
  // tile_spans (with tiles): 2 words per listed tile, the first and last in-tile index inside an uncovered range
  const RunTable& rt = T.rt;
#define DN_WS(ptr, type, name, bytes)                                                               \
  type ptr = (type)ws_get(ctx, name, bytes);                                                        \
  if (!ptr) return NTS_ENOMEM
  const std::string pre(slot_prefix);
  DN_WS(d_seg, unsigned long long*, "seg_count", N_SEG * sizeof(unsigned long long));
  int rc;
  uint32_t* d_tiles = nullptr;
  const uint2* d_spans = nullptr;
  uint64_t n_tile_ids = 0;
  uint64_t n_tiles;
  const uint64_t *d_vs, *d_nv, *d_ts;
  uint32_t n_rec;
  std::vector<uint64_t> tile_start;
  if (pseudo_vstart) {
    // uncovered ranges as pseudo-records: their tables (and the list of key tiles to hash) go up in one copy
    n_tiles = tiles_of(*pseudo_nv, w, tile_start);
    if (pseudo_nv->size() > 0xFFFFFFF0ULL) return fail(ctx, NTS_ERANGE, "too many uncovered ranges");
    const bool use_tiles = tiles && tiles->size() * 2 <= (rt.n_valid + KEY_TILE - 1) / KEY_TILE;
    const bool use_spans = use_tiles && tile_spans && tile_spans->size() == 2 * tiles->size();
    void* dev[5];
    if ((rc = upload_packed(ctx, "gap_tables",
                            { { pseudo_vstart->data(), pseudo_vstart->size() * 8 },
                              { pseudo_nv->data(), pseudo_nv->size() * 8 },
                              { tile_start.data(), tile_start.size() * 8 },
                              { use_tiles ? (const void*)tiles->data() : nullptr, use_tiles ? tiles->size() * 4 : 0 },
                              { use_spans ? (const void*)tile_spans->data() : nullptr, use_spans ? tile_spans->size() * 4 : 0 } },
                            dev)))
      return rc;
    if (use_spans) d_spans = (const uint2*)dev[4];
    d_vs = (const uint64_t*)dev[0];
    d_nv = (const uint64_t*)dev[1];
    d_ts = (const uint64_t*)dev[2];
    if (use_tiles) {
      d_tiles = (uint32_t*)dev[3];
      n_tile_ids = tiles->size();
    }
    n_rec = (uint32_t)pseudo_nv->size();
  } else {
    n_tiles = T.n_win_tiles(w);
    d_vs = T.d_rec_vstart;
    d_nv = T.d_rec_nv;
    if ((rc = T.win_tiles_device(ctx, w, &d_ts))) return rc;
    n_rec = g->n_rec;
  }
  // short windows over the whole genome: the window tiles hash and probe their own k-mers, no key array (k_window_min<true>)
  WinFuse fuse_args;
  const WinFuse* fuse = nullptr;
  if (!pseudo_vstart && w < WIN_FUSE_W && k <= FAST_K_MAX && !ctx->cur_rep && !(ctx->cur_summary && ctx->cur_tile_any) &&
      !(NTS_KNOB("NTS_WIN_FUSE") && atoi(NTS_KNOB("NTS_WIN_FUSE")) == 0)) {
    fuse_args.code = g->d_code + PAD;
    fuse_args.run_pos = T.d_run_pos;
    fuse_args.run_vstart = T.d_run_vstart;
    fuse_args.n_runs = T.n_runs;
    if (int rc_hp = hash_params_for(ctx, k, &fuse_args.hp)) return rc_hp;
    fuse_args.bf = filter ? filter->d_words : nullptr;
    fuse_args.fm = make_fastmod((filter ? filter->bytes : 8) * 8);
    fuse = &fuse_args;
  }
  // keys: one slot per key tile of the genome, or only the listed tiles (uncovered ranges) in a compact buffer
  uint64_t* d_keys = nullptr;
  if (!fuse) {
    d_keys = (uint64_t*)ws_get(ctx, d_tiles ? "gap_keys" : "keys", (d_tiles ? n_tile_ids * KEY_TILE : key_buffer_elems(rt.n_valid)) * 8);
    if (!d_keys) return NTS_ENOMEM;
    if ((rc = launch_hash<MODE_KEYS>(ctx, filter ? "hash_probe" : "hash_only", g, T, k, filter, nullptr, d_keys, d_tiles, n_tile_ids, d_spans))) return rc;
  }
  const char* win_tag = fuse ? (filter ? "hash_probe" : "hash_only") : "window_min"; // (the fused pass is timed as the hashing pass it replaces)
  OutSegs segs;
  segs.d_count = d_seg;
  segs.seg_cap = std::max<uint64_t>(256, ((uint64_t)(std::min(ctx->dense_seg_per_window, (double)w) * (double)est_kmers / (double)w) + 2 * n_tiles) / N_SEG + 64);
  // ---- few uncovered ranges: no host round trip, no sort -----------------------------------------------------
  // the window kernel writes every tile's winners in order to a slot of its own; one workgroup (k_gap_collect) strings
  // the tiles together, k_finalize merges them into the sparse winners; the count is read at the call's end
  const uint64_t expect_winners = 2 * est_kmers / std::max<uint32_t>(w, 1) + n_rec;
  if (sparse && ctx->small_gap_path && n_tiles <= GAP_TILES_MAX && expect_winners <= GAP_LIST_CAP * 3 / 4) {
    OutSegs tl;
    tl.tile_cap = GAP_TILE_CAP;
    tl.d_j = (uint64_t*)ws_get(ctx, "gap_tile_j", n_tiles * GAP_TILE_CAP * 8);
    tl.d_key = (uint64_t*)ws_get(ctx, "gap_tile_key", n_tiles * GAP_TILE_CAP * 8);
    tl.d_tile_cnt = (uint32_t*)ws_get(ctx, "gap_tile_cnt", (n_tiles + 4) * 4);
    DN_WS(d_gj, uint64_t*, "gap_sorted_j", GAP_LIST_CAP * 8);
    DN_WS(d_gk, uint64_t*, "gap_sorted_key", GAP_LIST_CAP * 8);
    DN_WS(d_gctl, uint64_t*, "gap_ctl", 4 * 8);
    if (!tl.d_j || !tl.d_key || !tl.d_tile_cnt) return NTS_ENOMEM;
    if ((rc = launch_window_dense(ctx, d_keys, d_vs, d_nv, d_ts, n_rec, n_tiles, w, tl, win_tag, d_tiles, n_tile_ids, fuse))) return rc;
    {
      ScopedTimer t(ctx, "merge_lists");
      NTS_LAUNCH(k_gap_collect, dim3(1), dim3(GAP_COLLECT_THREADS), 0, ctx->stream, tl.d_tile_cnt, (uint32_t)n_tiles, tl.d_j, tl.d_key,
                         sparse->count, d_gj, d_gk, d_gctl);
    }
    HIP_TRY(ctx, hipGetLastError());
    res.d_j = sparse->d_j;
    res.d_key = sparse->d_key;
    res.na = sparse->count;
    res.b_j = d_gj;
    res.b_key = d_gk;
    res.count = sparse->count + GAP_LIST_CAP;
    res.d_ctl = d_gctl;
    return NTS_OK;
  }
  if (fuse && n_tiles == 0) {
    res.count = 0;
    return NTS_OK;
  }
  if (fuse) {
    // short windows over the whole genome: every tile writes its winners in index order into a segment and leaves (offset, count) in
    // a directory; scan + gather give the ordered list -- no 0xFF fill of the segments, no radix sort of up to half a billion pairs
    // (43 ms at w = 10)
    uint64_t cap = ((uint64_t)(std::min(ctx->fused_seg_per_window, (double)(w + 1)) * (double)est_kmers / (double)(w + 1)) + 2ull * n_rec) / N_SEG + 65536;
    for (int attempt = 0; attempt < 2; ++attempt) {
      OutSegs od;
      od.seg_cap = cap;
      od.d_count = d_seg;
      const uint64_t slots = cap * N_SEG;
      od.d_j = (uint64_t*)ws_get(ctx, (pre + "out_j").c_str(), slots * 8);
      od.d_key = (uint64_t*)ws_get(ctx, (pre + "out_key").c_str(), slots * 8);
      uint64_t* d_doff = (uint64_t*)ws_get(ctx, (pre + "win_dir_off").c_str(), n_tiles * 8);
      uint32_t* d_dcnt = (uint32_t*)ws_get(ctx, (pre + "win_dir_cnt").c_str(), n_tiles * 4 + 8);
      uint64_t* d_dscan = (uint64_t*)ws_get(ctx, (pre + "win_dir_scan").c_str(), n_tiles * 8);
      uint64_t* d_oj2 = (uint64_t*)ws_get(ctx, (pre + "out_j2").c_str(), slots * 8);
      uint64_t* d_ok2 = (uint64_t*)ws_get(ctx, (pre + "out_key2").c_str(), slots * 8);
      unsigned long long* d_ovf = (unsigned long long*)ws_get(ctx, (pre + "win_dir_ovf").c_str(), 8);
      if (!od.d_j || !od.d_key || !d_doff || !d_dcnt || !d_dscan || !d_oj2 || !d_ok2 || !d_ovf) return NTS_ENOMEM;
      HIP_TRY(ctx, hipMemsetAsync(d_seg, 0, N_SEG * sizeof(unsigned long long), ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(d_ovf, 0, 8, ctx->stream));
      fuse_args.dir_off = d_doff;
      fuse_args.dir_cnt = d_dcnt;
      if ((rc = launch_window_dense(ctx, nullptr, d_vs, d_nv, d_ts, n_rec, n_tiles, w, od, win_tag, nullptr, 0, fuse))) return rc;
      {
        ScopedTimer t(ctx, "merge_lists");
        if (int rc_s = scan_counts<uint32_t>(ctx, d_dcnt, n_tiles, d_dscan)) return rc_s;
        NTS_LAUNCH(k_cand_compact_slots, dim3((uint32_t)((n_tiles + CCS_TILES - 1) / CCS_TILES)), dim3(256), 0, ctx->stream, od.d_j, od.d_key, d_doff, d_dcnt,
                           d_dscan, n_tiles, d_oj2, d_ok2, slots, d_ovf);
      }
      HIP_TRY(ctx, hipGetLastError());
      unsigned long long segc[N_SEG];
      uint64_t total = 0;
      {
        Mail m(ctx);
        const uint32_t a_seg = m.add(d_seg, N_SEG);
        const uint32_t a_scan = m.add(d_dscan + (n_tiles - 1), 1);
        const uint32_t a_cnt = m.add((const uint64_t*)(d_dcnt + ((n_tiles - 1) & ~1ull)), 1); // (32-bit counts: the pair holding the last one)
        if ((rc = m.post(ctx))) return rc;
        for (uint32_t q = 0; q < N_SEG; ++q) segc[q] = ctx->mail[a_seg + q];
        total = ctx->mail[a_scan] + (uint32_t)(ctx->mail[a_cnt] >> (32 * ((n_tiles - 1) & 1ull)));
      }
      uint64_t worst = 0;
      for (uint32_t q = 0; q < N_SEG; ++q) worst = std::max<uint64_t>(worst, segc[q]);
      res.d_j = d_oj2;
      res.d_key = d_ok2;
      res.count = total;
      if (worst <= cap) return NTS_OK;
      if (attempt == 1) return fail(ctx, NTS_EHIP, "minimizer segments overflowed twice");
      if (est_kmers >= (1ull << 20)) // (accepted k-mers in clusters -- k = 64 at w = 63: 2.9 per window; the next call starts from what this one needed)
        ctx->fused_seg_per_window = std::max(ctx->fused_seg_per_window, 1.15 * (double)worst * N_SEG * (double)(w + 1) / (double)est_kmers);
      cap = worst + 1024;
    }
  }
  unsigned long long seg_counts[N_SEG];
  uint64_t count = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const uint64_t slots = segs.seg_cap * N_SEG;
    segs.d_j = (uint64_t*)ws_get(ctx, (pre + "out_j").c_str(), slots * 8);
    segs.d_key = (uint64_t*)ws_get(ctx, (pre + "out_key").c_str(), slots * 8);
    if (!segs.d_j || !segs.d_key) return NTS_ENOMEM;
    HIP_TRY(ctx, hipMemsetAsync(segs.d_j, 0xFF, slots * 8, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_seg, 0, N_SEG * sizeof(unsigned long long), ctx->stream));
    if ((rc = launch_window_dense(ctx, d_keys, d_vs, d_nv, d_ts, n_rec, n_tiles, w, segs, win_tag, d_tiles, n_tile_ids, fuse))) return rc;
    {
      Mail m(ctx);
      const uint32_t at = m.add(d_seg, N_SEG);
      if ((rc = m.post(ctx))) return rc;
      for (uint32_t s = 0; s < N_SEG; ++s) seg_counts[s] = ctx->mail[at + s];
    }
    uint64_t worst = 0;
    count = 0;
    for (uint32_t s = 0; s < N_SEG; ++s) {
      worst = std::max<uint64_t>(worst, seg_counts[s]);
      count += seg_counts[s];
    }
    if (worst <= segs.seg_cap) break;
    if (attempt == 1) return fail(ctx, NTS_EHIP, "minimizer segments overflowed twice");
    if (est_kmers >= (1ull << 20)) // (a genome-sized call: the next one starts from what this one needed)
      ctx->dense_seg_per_window = std::max(ctx->dense_seg_per_window, 1.15 * (double)worst * N_SEG * (double)w / (double)est_kmers);
    segs.seg_cap = worst;
  }
  res.count = count;
  if (count == 0) return NTS_OK;
  const uint64_t slots = segs.seg_cap * N_SEG;
  DN_WS(d_oj2, uint64_t*, (pre + "out_j2").c_str(), slots * 8);
  DN_WS(d_ok2, uint64_t*, (pre + "out_key2").c_str(), slots * 8);
  size_t tmp_bytes = 0;
  // compact indices are < n_valid: sort only the bits that can differ (sentinel slots are all ones)
  uint32_t bits = 1;
  while (bits < 64 && (rt.n_valid >> bits) != 0) ++bits;
  const uint32_t end_bit = std::min<uint32_t>(64, bits + 1);
  HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, segs.d_j, d_oj2, segs.d_key, d_ok2, slots, 0, end_bit, ctx->stream));
  DN_WS(d_tmp, void*, "sort_tmp", std::max<size_t>(tmp_bytes, 16));
  {
    ScopedTimer t(ctx, "sort_minimizers");
    HIP_TRY(ctx, rocprim::radix_sort_pairs(d_tmp, tmp_bytes, segs.d_j, d_oj2, segs.d_key, d_ok2, slots, 0, end_bit, ctx->stream));
  }
  res.d_j = d_oj2;
  res.d_key = d_ok2;
  return NTS_OK;
#undef DN_WS
}

// 2-bit image of the genome for k_hash_select (built once per genome)
int ensure_pack(nts_ctx* ctx, const nts_genome* g)
{
  if (g->d_pack) return NTS_OK;
  const uint64_t n_words = (g->n + PAD) / 16; // the trailing pad is readable: look-ahead past the last base stays in bounds
  uint32_t* p = nullptr;
  // (320 more words that nothing looks at: k_hash_select_hi stages 272 words from the word of a tile's first base on)
  HIP_TRY(ctx, dev_malloc((void**)&p, (n_words + 320) * 4));
  {
    ScopedTimer t(ctx, "pack_image");
    if (n_words) NTS_LAUNCH(k_pack2, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, ctx->stream, g->d_code + PAD, n_words, p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dev_free(p);
    HIP_TRY(ctx, e);
  }
  g->d_pack = p;
  return NTS_OK;
}

constexpr uint32_t SUMMARY_LOG2_BITS = 25; // a sparse filter's summary: at most 2^25 bits = 4 MiB (config 4 on one GPU, 2^23 .. 2^26: 324 / 338 / 349 / 331 Gbases/s; NTS_SUMMARY_LOG2_BITS overrides)


// Summary of a sparse filter (one bit per 2^shift filter bits) and its two folded tables, kept with the filter until its contents
// change (k_bf_summary: one streaming pass).  The caller holds filter->mu.
int bf_make_summary(nts_ctx* ctx, const nts_bf* filter, uint32_t shift)
{
  if (filter->summary_version == filter->version && filter->summary_shift == shift) return NTS_OK;
  const uint64_t n_gran = ((uint64_t)filter->bytes * 8 + (1ull << shift) - 1) >> shift;
  const uint64_t words = (n_gran + 31) / 32 + 4;
  if (filter->summary_words < words) {
    if (filter->d_summary) dev_free(filter->d_summary);
    filter->d_summary = nullptr;
    filter->summary_words = 0;
    HIP_TRY(ctx, dev_malloc((void**)&filter->d_summary, words * 4));
    filter->summary_words = words;
  }
  HIP_TRY(ctx, hipMemsetAsync(filter->d_summary, 0, filter->summary_words * 4, ctx->stream));
  if (!filter->d_fold) HIP_TRY(ctx, dev_malloc((void**)&filter->d_fold, 2 * FOLD_WORDS * 4)); // (two tables: k_bf_summary)
  HIP_TRY(ctx, hipMemsetAsync(filter->d_fold, 0, 2 * FOLD_WORDS * 4, ctx->stream));
  const uint64_t n16 = (filter->bytes + 15) / 16;
  {
    ScopedTimer t(ctx, "bf_summary");
    NTS_LAUNCH(k_bf_summary, dim3((uint32_t)std::min<uint64_t>((n16 + 255) / 256, 256 * 16)), dim3(256), 0, ctx->stream,
                       (const uint4*)filter->d_words, n16, shift, filter->d_summary, filter->d_fold, FOLD_WORDS);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (another context may read the summary from its own stream as soon as the lock is free)
  filter->summary_shift = shift;
  filter->summary_version = filter->version;
  return NTS_OK;
}

// The kernel that looks every k-mer of g up in a sparse filter through its summary (and, where they pay, the two folded tables in
// LDS) and lists the accepted ones per 8192-k-mer tile: tile t's list sits in segment t % N_SEG at tile_off[t], tile_cnt[t] entries
// (j, h0) in index order.  Used by the sketch (run_pruned, accept_all) and by the cascade level over a sparse running filter
// (bf_level_sparse).
int launch_accept(nts_ctx* ctx, const nts_genome* g, uint32_t k, const AcceptParams& A, const uint32_t* fold, uint64_t n_kt)
{
  if (fold && k <= 32 && !(NTS_KNOB("NTS_ACCEPT_REG") && atoi(NTS_KNOB("NTS_ACCEPT_REG")) == 0)) {
    // bases from the 2-bit image in registers (k_hash_accept4r); NTS_ACCEPT_REG=0: the LDS-staged kernel (tests)
    if (int rc_pk = ensure_pack(ctx, g)) return rc_pk;
    if (!ctx->acc4r_lds_set) {
#define ACC4R_ATTR(B, F) HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_hash_accept4r<B, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Accept4rLds)))
      ACC4R_ATTR(8, 0);
      ACC4R_ATTR(8, 1);
      ACC4R_ATTR(8, 2);
      ACC4R_ATTR(4, -1);
#undef ACC4R_ATTR
      ctx->acc4r_lds_set = true;
    }
    // (persistent workgroups, one per CU -- 128 KiB of LDS each --, each loops over groups of four tiles; NTS_ACC4R_WGS overrides)
    const uint64_t groups4 = (n_kt + 3) / 4;
    const uint64_t wgs = NTS_KNOB("NTS_ACC4R_WGS") ? (uint64_t)std::max(1, atoi(NTS_KNOB("NTS_ACC4R_WGS"))) : 256ull;
    const dim3 grid4((uint32_t)std::min<uint64_t>(groups4, wgs));
#define ACC4R_RUN(B, F) NTS_LAUNCH((k_hash_accept4r<B, F>), grid4, dim3(ACC4_THREADS), sizeof(Accept4rLds), ctx->stream, A, g->d_pack, fold, n_kt)
    if (NTS_KNOB("NTS_ACC4R_BLOCK") && atoi(NTS_KNOB("NTS_ACC4R_BLOCK")) == 4)
      ACC4R_RUN(4, -1); // (blocks of four, the modulus form read at run time: the kernel as it was, for comparisons)
    else if (A.fm.form == 2)
      ACC4R_RUN(8, 2);
    else if (A.fm.form == 1)
      ACC4R_RUN(8, 1);
    else
      ACC4R_RUN(8, 0);
#undef ACC4R_RUN
  } else if (fold) {
    if (!ctx->acc4_lds_set) {
      HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_hash_accept4), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(Accept4Lds)));
      ctx->acc4_lds_set = true;
    }
    NTS_LAUNCH(k_hash_accept4, dim3((uint32_t)((n_kt + 3) / 4)), dim3(ACC4_THREADS), sizeof(Accept4Lds), ctx->stream, A, fold,
                       n_kt);
  } else {
    NTS_LAUNCH(k_hash_accept, dim3((uint32_t)n_kt), dim3(HASH_THREADS), 0, ctx->stream, A);
  }
  return NTS_OK;
}

// One cascade level over a sparse running filter (nts_bf_sparse.inc): every k-mer of g is looked up in acc (the sketch's accept
// kernels), acc's set bits are cleared granule by granule, the accepted k-mers' bits are set again -- together with the summary and
// the folded tables of the new contents.  acc is touched only once the lists are known to have fitted.
int bf_level_sparse(nts_ctx* ctx, nts_bf* acc, const nts_genome* g, const GenomeTables& T, uint32_t k, int64_t pop_before)
{
  ctx->last_bf_sparse_level = 0;
  ctx->last_bf_sparse_accepted = 0;
  // (a build mode forced by the caller -- tests -- means the build: only the automatic choice comes here)
  if (pop_before < 0 || !acc->owned || ctx->summary_mode != 0 || ctx->bf_build_mode != 0) return 1;
  if (NTS_KNOB("NTS_BF_SPARSE_LEVEL") && atoi(NTS_KNOB("NTS_BF_SPARSE_LEVEL")) == 0) return 1;
  const uint64_t V = T.rt.n_valid;
  if (V == 0 || pop_before == 0) return 1; // (nothing to look up / nothing to keep: the build's own finish handles both)
  const double bits = (double)acc->bytes * 8.0;
  uint32_t shift = 7;
  const uint32_t sum_log2 = NTS_KNOB("NTS_SUMMARY_LOG2_BITS") ? (uint32_t)std::max(16, std::min(28, atoi(NTS_KNOB("NTS_SUMMARY_LOG2_BITS")))) : SUMMARY_LOG2_BITS;
  while ((bits / (double)(1ull << shift)) > (double)(1ull << sum_log2) && shift < 30) ++shift; // (the sketch's choice: nts_sketch_ex)
  // an occupancy of the summary below 0.3 -- stricter than the sketch's own criterion for taking the summary path (nts_sketch_ex: 0.7
  // next to the tiered selection, 3.0 where only the every-k-mer pass is the alternative) -- and the level must beat the build, which
  // with a sparse running filter (k_bin3 skips the residues of empty slices) takes 20-21 ms per 3 Gbp.  Measured level by level on BASELINE configs[3] (scripts/c4_levels.py, profiles/r04_c4_levels.json): through the summary
  // alone the literal level takes 60 / 31 / 22 / 20.3 ms at 27.8 M / 9.0 M / 3.0 M / 1.05 M set bits (17.3 with the tables already in
  // place) -- never ahead; only with the two folded tables in LDS in front of the summary (k_hash_accept4*: 8 ms per genome, up to
  // ~6 * 10^5 set bits) does it win, so that is the automatic choice.  NTS_BF_SPARSE_MAX_OCC replaces both limits by an occupancy
  // (tests and measurements: every accept kernel at any occupancy).
  const double occ = (double)pop_before / bits;
  const bool fold_fits = bits >= (double)(1u << FOLD_BITS_LOG2) && (double)pop_before < 1.2 * (double)(1u << FOLD_BITS_LOG2);
  if (NTS_KNOB("NTS_BF_SPARSE_MAX_OCC")) {
    if (occ >= atof(NTS_KNOB("NTS_BF_SPARSE_MAX_OCC"))) return 1;
  } else if (occ >= 0.3 / (double)(1ull << shift) || !fold_fits || ctx->fold_mode != 0) {
    return 1;
  }
  const uint64_t n_kt = (V + KEY_TILE - 1) / KEY_TILE;
  if (n_kt > 0x7FFFFFFFULL) return 1;
  std::lock_guard<std::mutex> summary_lock(acc->mu);
  if (int rc = bf_make_summary(ctx, acc, shift)) return rc;
  const uint32_t* fold = (ctx->fold_mode == 0 && fold_fits) ? acc->d_fold : nullptr;
  // room for the accepted k-mers: a related genome hits a good share of the set bits, once per copy of the k-mer
  const double own = bits * (1.0 - std::exp(-(double)V / bits));
  const double p = own > 0 ? std::min(1.0, (double)pop_before / own) : 1.0;
  const uint64_t seg_cap = (uint64_t)((double)V * std::min(1.0, 1.5 * p + 1e-4) * 1.25 / N_SEG) + 8192;
#define SL_WS(ptr, type, name, bytes)                                                               \
  type ptr = (type)ws_get(ctx, name, bytes);                                                        \
  if (!ptr) return NTS_ENOMEM
  SL_WS(d_toff, uint64_t*, "spl_tile_off", n_kt * 8);
  SL_WS(d_tcnt, uint32_t*, "spl_tile_cnt", n_kt * 4 + 8);
  SL_WS(d_tord, uint8_t*, "spl_tile_ord", n_kt);
  SL_WS(d_ctl, unsigned long long*, "spl_ctl", (N_SEG + 2) * 8); // [0..63] segment counters, [64] bits turned on
  SL_WS(d_sj, uint64_t*, "spl_seg_j", seg_cap * N_SEG * 8);
  SL_WS(d_sk, uint64_t*, "spl_seg_key", seg_cap * N_SEG * 8);
#undef SL_WS
  HIP_TRY(ctx, hipMemsetAsync(d_ctl, 0, (N_SEG + 2) * 8, ctx->stream));
  AcceptParams A;
  A.code = g->d_code + PAD;
  A.run_pos = T.d_run_pos;
  A.run_vstart = T.d_run_vstart;
  A.n_runs = T.n_runs;
  A.n_valid = V;
  if (int rc_hp = hash_params_for(ctx, k, &A.hp)) return rc_hp;
  A.bf = acc->d_words;
  A.fm = make_fastmod(acc->bytes * 8);
  A.summary = acc->d_summary;
  A.shift = shift;
  A.probe_mask = ~0u;
  A.seg_j = d_sj;
  A.seg_key = d_sk;
  A.seg_cap = seg_cap;
  A.seg_count = d_ctl;
  A.tile_off = d_toff;
  A.tile_cnt = d_tcnt;
  A.tile_ordered = d_tord;
  ScopedTimer t(ctx, "bf_sparse_level", true);
  if (int rc_a = launch_accept(ctx, g, k, A, fold, n_kt)) return rc_a;
  HIP_TRY(ctx, hipGetLastError());
  unsigned long long counts[N_SEG];
  HIP_TRY(ctx, hipMemcpyAsync(counts, d_ctl, N_SEG * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t accepted = 0;
  for (uint32_t sgm = 0; sgm < N_SEG; ++sgm) {
    if (counts[sgm] > seg_cap) return 1; // a list did not fit: acc is as it was, the level goes the build's way
    accepted += counts[sgm];
  }
  const uint64_t n16 = (acc->bytes + 15) / 16;
  const uint64_t n_gran = ((uint64_t)acc->bytes * 8 + (1ull << shift) - 1) >> shift;
  const uint64_t n_sw = (n_gran + 31) / 32;
  acc->popcnt = -1; // from here on the filter changes
  ++acc->version;
  NTS_LAUNCH(k_bf_sparse_clear, dim3((uint32_t)std::min<uint64_t>((n_sw + 3) / 4, 256 * 16)), dim3(256), 0, ctx->stream, (uint4*)acc->d_words, n16,
                     acc->d_summary, n_sw, shift);
  HIP_TRY(ctx, hipMemsetAsync(acc->d_summary, 0, acc->summary_words * 4, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(acc->d_fold, 0, 2 * FOLD_WORDS * 4, ctx->stream));
  NTS_LAUNCH(k_bf_sparse_set, dim3((uint32_t)((n_kt * 32 + 255) / 256)), dim3(256), 0, ctx->stream, acc->d_words, A.fm, d_sk, seg_cap, d_toff, d_tcnt,
                     n_kt, acc->d_summary, shift, acc->d_fold, d_ctl + N_SEG);
  HIP_TRY(ctx, hipGetLastError());
  unsigned long long n_set = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n_set, d_ctl + N_SEG, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  acc->popcnt = (int64_t)n_set;
  acc->summary_shift = shift; // summary and folded tables describe the new contents
  acc->summary_version = acc->version;
  ctx->last_bf_sparse_level = 1;
  ctx->last_bf_sparse_accepted = accepted;
  return 0;
}

// Pruned path; see nts_pruned.inc.  `res` gets every minimizer, ordered (sparse winners come out ordered by
// construction; winners of uncovered ranges, if any, are sorted and merged in).
// accept_all: the filter is sparse and its summary (ctx->cur_summary) is in place -- every k-mer is looked up, the accepted
// ones are the candidates (k_hash_accept), windows without one have no minimizer: no uncovered ranges to evaluate
// (tp: the tiered selection of nts_tiers.inc instead of one threshold -- its list holds every window's minimizer, like the
//  accepted-list path's, so no range is left to the dense kernels)
struct TierPlan
{
  float scale = 0;
  uint32_t exp_shift = 23;
  int32_t exp_bias = 126;
  uint32_t n_tiers = 2, halo = 0, core = 0;
  double c0 = 0;
};

void launch_tiers(nts_ctx* ctx, const TierParams& Q, uint64_t n_tiles)
{
  const dim3 grid((uint32_t)n_tiles), block(TR_THREADS);
  if (Q.fm.form == 2)
    NTS_LAUNCH(k_hash_tiers<2>, grid, block, 0, ctx->stream, Q);
  else if (Q.fm.form == 1)
    NTS_LAUNCH(k_hash_tiers<1>, grid, block, 0, ctx->stream, Q);
  else
    NTS_LAUNCH(k_hash_tiers<0>, grid, block, 0, ctx->stream, Q);
}

__global__ void k_gap_tiers_ctl(const uint64_t* __restrict__ blk_scan_last, const uint64_t* __restrict__ blk_cnt_last, uint64_t n_sparse,
                                const unsigned long long* __restrict__ overflow, uint64_t* __restrict__ ctl)
{
  const uint64_t n = *blk_scan_last + *blk_cnt_last;
  const bool bad = *overflow != 0ULL;
  ctl[0] = bad ? 0 : n;
  ctl[1] = n_sparse + (bad ? 0 : n);
  ctl[2] = bad ? 1 : 0;
}

// The uncovered ranges of the one-threshold selection through the tiered selection (nts_tiers.inc, gap mode) instead of a probe of
// every k-mer in them: the ranges as tiles (position 0 = the range's first index, both ends closed; ranges longer than a tile cut into
// tiles with halos), the thresholds going on from tau; the accepted k-mers found, in index order, are put to the window decision with
// the ranges as records (k_sparse_win: a window lies inside its range), and the winners come back as the second list k_finalize
// merges into the first -- what run_dense_sorted's few-ranges path hands back, in its format (ctl[0] winners, ctl[1] total, ctl[2] gave up).
// Returns NTS_OK with res.d_ctl == nullptr when the path does not apply (the caller goes the dense way).
int run_gap_tiers(nts_ctx* ctx, const nts_genome* g, const GenomeTables& T, uint32_t k, uint32_t w, const nts_bf* filter, uint64_t tau, uint32_t prune_c,
                  const std::vector<uint64_t>& pv, const std::vector<uint64_t>& pn, uint64_t covered, const SortedOut& sparse, SortedOut& res)
{
  res = SortedOut();
  if (!filter || k > FAST_K_MAX || w < 64 || w > 4097 || pv.empty() || ctx->gap_tiers_off) return NTS_OK;
  const uint32_t halo = ((w - 1 + 63) / 64) * 64, core = TR_EXT - 2 * halo;
  std::vector<TierTile> tiles;
  for (size_t i = 0; i < pv.size(); ++i) {
    const uint64_t a = pv[i], n = pn[i];
    if (n <= TR_EXT) {
      TierTile t = {};
      t.e0 = (int64_t)a;
      t.n = (uint32_t)n;
      t.c0 = 0;
      t.c1 = (uint32_t)n;
      t.flags = 3;
      tiles.push_back(t);
      continue;
    }
    for (uint64_t c_lo = 0; c_lo < n; c_lo += core) {
      const uint64_t e_lo = c_lo >= halo ? c_lo - halo : 0, e_hi = std::min<uint64_t>(n, c_lo + core + halo); // [e_lo, e_hi) inside the range
      TierTile t = {};
      t.e0 = (int64_t)(a + e_lo);
      t.n = (uint32_t)(e_hi - e_lo);
      t.c0 = (uint32_t)(c_lo - e_lo);
      t.c1 = (uint32_t)(std::min<uint64_t>(n, c_lo + core) - e_lo);
      t.flags = (e_lo == 0 ? 1u : 0u) | (e_hi == n ? 2u : 0u);
      tiles.push_back(t);
    }
  }
  const uint64_t n_gt = tiles.size();
  if (n_gt > (1u << 20)) return NTS_OK;
#define GT_WS(ptr, type, name, bytes)                                                               \
  type ptr = (type)ws_get(ctx, name, bytes);                                                        \
  if (!ptr) return NTS_ENOMEM
  void* dev[3];
  if (int rc = upload_packed(ctx, "gt_tables", { { tiles.data(), tiles.size() * sizeof(TierTile) }, { pv.data(), pv.size() * 8 }, { pn.data(), pn.size() * 8 } }, dev))
    return rc;
  const TierTile* d_tiles = (const TierTile*)dev[0];
  const uint64_t *d_vs = (const uint64_t*)dev[1], *d_nv = (const uint64_t*)dev[2];
  GT_WS(d_dir, uint32_t*, "gt_dir", n_gt * 12);
  GT_WS(d_toff, uint64_t*, "gt_tile_off", n_gt * 8);
  GT_WS(d_tcnt, uint32_t*, "gt_tile_cnt", n_gt * 4 + 8);
  GT_WS(d_tord, uint8_t*, "gt_tile_ord", n_gt);
  GT_WS(d_tscan, uint64_t*, "gt_tile_scan", n_gt * 8);
  GT_WS(d_ctl, unsigned long long*, "gt_ctl", (N_SEG + 4) * 8); // [0..63] segment counters, [64] ranges (unused), [65] overflow, [66..67] tier statistics
  // the accepted k-mers of the ranges: a few per window at most where the filter accepts anything at all (the ranges are what the
  // other genomes do not share); a list that does not fit raises the flag and the call is repeated the dense way
  // Room per output segment: 5 % of the ranges' k-mers dealt out over the segments -- and never less than what the tiles of one
  // segment can list when every one of them is as rich as a conserved stretch (tiles go to segments round robin; a tile of TR_CORE
  // k-mers over sequence the other genomes DO share lists up to an eighth of its k-mers): a few long ranges over conserved sequence
  // used to overflow a segment sized from the average, and the whole sketch then ran again the dense way.
  const uint64_t tiles_per_seg = (n_gt + N_SEG - 1) / N_SEG;
  const uint64_t rich = std::min<uint64_t>(tiles_per_seg, 8) * (uint64_t)(core / 8 + 64);
  const uint64_t seg_cap = std::max<uint64_t>((uint64_t)((double)covered * 0.05 / N_SEG) + 2048, rich);
  const uint64_t m_max = seg_cap * N_SEG, n_blk = (m_max + SPARSE_BLOCK - 1) / SPARSE_BLOCK;
  GT_WS(d_sj, uint64_t*, "gt_seg_j", m_max * 8);
  GT_WS(d_sk, uint64_t*, "gt_seg_key", m_max * 8);
  GT_WS(d_pj, uint64_t*, "gt_cand_j", m_max * 8);
  GT_WS(d_pk, uint64_t*, "gt_cand_key", m_max * 8);
  GT_WS(d_stj, uint64_t*, "gt_stage_j", n_blk * SPARSE_BLOCK * 8);
  GT_WS(d_stk, uint64_t*, "gt_stage_k", n_blk * SPARSE_BLOCK * 8);
  GT_WS(d_bcnt, uint64_t*, "gt_blk_cnt", n_blk * 8);
  GT_WS(d_bscan, uint64_t*, "gt_blk_scan", n_blk * 8);
  GT_WS(d_gj, uint64_t*, "gt_win_j", n_blk * SPARSE_BLOCK * 8);
  GT_WS(d_gk, uint64_t*, "gt_win_key", n_blk * SPARSE_BLOCK * 8);
  GT_WS(d_gctl, uint64_t*, "gap_ctl", 4 * 8);
  HIP_TRY(ctx, hipMemsetAsync(d_ctl, 0, (N_SEG + 4) * 8, ctx->stream));
  TierParams Q;
  Q.code = g->d_code + PAD;
  Q.pack = g->d_pack;
  Q.run_pos = T.d_run_pos;
  Q.run_vstart = T.d_run_vstart;
  Q.n_runs = T.n_runs;
  Q.n_valid = T.rt.n_valid;
  Q.rec_vstart = T.d_rec_vstart;
  Q.n_rec = g->n_rec;
  Q.tiles = d_tiles;
  Q.excl_on = 1;
  Q.excl_hi = (uint32_t)(tau >> 32);
  Q.dir = d_dir;
  if (int rc_hp = hash_params_for(ctx, k, &Q.hp)) return rc_hp;
  Q.bf = filter->d_words;
  Q.fm = make_fastmod(filter->bytes * 8);
  Q.w = w;
  Q.halo = halo;
  Q.core = core;
  // tiers going on from tau: (tau, 2 tau], (2 tau, 4 tau], ... while the threshold stays below ~3/4 of all hashes; the last takes the rest
  Q.scale = (float)(1.0 / ((double)(tau >> 32) + 1.0));
  Q.exp_shift = 23;
  Q.exp_bias = 127;
  uint32_t n_exp = 1;
  while (n_exp < TR_TIERS_MAX - 1 && (double)prune_c * (double)(2u << n_exp) <= 0.75 * (double)w) ++n_exp;
  Q.n_tiers = n_exp + 1;
  Q.seg_j = d_sj;
  Q.seg_key = d_sk;
  Q.seg_cap = seg_cap;
  Q.seg_count = d_ctl;
  Q.tile_off = d_toff;
  Q.tile_cnt = d_tcnt;
  Q.tile_ordered = d_tord;
  Q.stats = d_ctl + N_SEG + 2;
  Q.exp_stop = 0;
  {
    ScopedTimer t(ctx, "hash_probe"); // (the group the dense pass over the ranges is timed under)
    NTS_LAUNCH(k_tier_dir, dim3((uint32_t)((n_gt + 255) / 256)), dim3(256), 0, ctx->stream, T.d_run_vstart, T.n_runs, T.d_rec_vstart, g->n_rec,
                       T.rt.n_valid, halo, core, n_gt, d_tiles, d_dir);
    launch_tiers(ctx, Q, n_gt);
  }
  {
    ScopedTimer t(ctx, "window_min");
    if (int rc_s = scan_counts<uint32_t>(ctx, d_tcnt, n_gt, d_tscan)) return rc_s;
    NTS_LAUNCH(k_cand_compact, dim3((uint32_t)((n_gt + CC_TILES - 1) / CC_TILES)), dim3(256), 0, ctx->stream, d_sj, d_sk, seg_cap, d_toff, d_tcnt, d_tord,
                       d_tscan, n_gt, d_pj, d_pk, m_max, d_ctl + N_SEG + 1, false);
    SparseParams S;
    S.pj = d_pj;
    S.pk = d_pk;
    S.m_scan_last = d_tscan + (n_gt - 1);
    S.m_cnt_last = d_tcnt + (n_gt - 1);
    S.m_max = m_max;
    S.rec_vstart = d_vs;
    S.rec_nv = d_nv;
    S.n_valid = T.rt.n_valid;
    S.n_rec = (uint32_t)pv.size();
    S.w = w;
    S.stage_j = d_stj;
    S.stage_k = d_stk;
    S.blk_cnt = d_bcnt;
    S.gap_lo = S.gap_hi = nullptr;
    S.gap_count = d_ctl + N_SEG;
    S.gap_cap = 0;
    S.overflow = d_ctl + N_SEG + 1;
    S.rec_holes = 1;
    NTS_LAUNCH(k_sparse_win, dim3((uint32_t)n_blk), dim3(SPARSE_THREADS), 0, ctx->stream, S);
    if (int rc_s = scan_counts<uint64_t>(ctx, d_bcnt, n_blk, d_bscan)) return rc_s;
    NTS_LAUNCH(k_gather_winners, dim3((uint32_t)n_blk), dim3(256), 0, ctx->stream, d_stj, d_stk, d_bcnt, d_bscan, d_gj, d_gk);
    NTS_LAUNCH(k_gap_tiers_ctl, dim3(1), dim3(1), 0, ctx->stream, d_bscan + (n_blk - 1), d_bcnt + (n_blk - 1), sparse.count, d_ctl + N_SEG + 1, d_gctl);
  }
  HIP_TRY(ctx, hipGetLastError());
  res.d_j = sparse.d_j;
  res.d_key = sparse.d_key;
  res.na = sparse.count;
  res.b_j = d_gj;
  res.b_key = d_gk;
  res.count = sparse.count + std::min<uint64_t>(m_max, covered); // (an upper bound: the real number comes with the call's last mail)
  res.d_ctl = d_gctl;
  return NTS_OK;
#undef GT_WS
}

int run_pruned(nts_ctx* ctx, const nts_genome* g, const GenomeTables& T, uint32_t k, uint32_t w, const nts_bf* filter, uint32_t prune_c,
               double p_accept, SortedOut& res, bool accept_all = false, const TierPlan* tp = nullptr)
{
  const RunTable& rt = T.rt;
  const uint64_t V = rt.n_valid;
  // the upper-halves select kernel (k <= 32, threshold below half the hash range) works on wave tiles
  // (and a listing that fits one round of 4 per lane with room to spare: ~4096 c/w k-mers per tile; beyond that k_hash_select)
  // (and an assembly that is not in pieces: a tile that spans runs lists 64 k-mers per boundary and looks positions up
  // per k-mer.  3 Gbp in 24 / 5,000 / 100,000 / 1,000,000 contigs: 1.43 / 1.56 / 2.12 / 7.29 ms against k_hash_select's
  // 2.19 / 2.26 / 2.82 / 4.39 -- it keeps the assemblies with more than one run per two tiles)
  // (listed k-mers per tile: mean 4096 c/w, spread ~ its square root; a tile with more than a round holds takes the slow path.
  // Margins measured at 3 Gbp, pairs at 2-8 % divergence: two per lane up to a mean of 107 (c = 22: select 1.79 ms against 2.00
  // with four per lane), four per lane up to 223 (c = 53: 4.32 ms against 4.79 for k_hash_select); NTS_HI_M2 / NTS_HI_M4 override)
  const double hi_m4 = NTS_KNOB("NTS_HI_M4") ? atof(NTS_KNOB("NTS_HI_M4")) : 1.15, hi_m2 = NTS_KNOB("NTS_HI_M2") ? atof(NTS_KNOB("NTS_HI_M2")) : 1.2;
  const bool sel_hi = !accept_all && !tp && ctx->select_impl != 1 && k <= HI_K_MAX && 4096.0 * prune_c / w * hi_m4 <= 256.0 &&
                      (ctx->select_impl == 2 || 2ull * T.n_runs <= (V + HIW_TILE - 1) / HIW_TILE + 64);
  const uint32_t hi_per = 4096.0 * prune_c / w * hi_m2 <= 128.0 ? 2u : 4u; // listed k-mers per lane and round
  const uint64_t sel_tile = tp ? (uint64_t)tp->core : accept_all ? (uint64_t)KEY_TILE : sel_hi ? (uint64_t)HIW_TILE : (uint64_t)SEL_TILE;
  const uint64_t n_kt = (V + sel_tile - 1) / sel_tile; // tiles of the select kernel (16384 indices each; 8192 for k_hash_accept)
  if (n_kt > 0x7FFFFFFFULL) return fail(ctx, NTS_ERANGE, "genome too large for one launch");
  // threshold: a fraction c/w of all hashes
  const unsigned __int128 full = ((unsigned __int128)1) << 64;
  unsigned __int128 t128 = full / w * prune_c;
  // low 32 bits set: "h0 <= tau" is then a test of the high word alone (k_hash_select's rolling loop relies on it)
  uint64_t tau = t128 >= full - 1 ? KEY_MAX - 1 : ((uint64_t)t128 | 0xFFFFFFFFULL);
  if (tau == KEY_MAX) tau = KEY_MAX - 1;
  const double frac = accept_all ? 1.0 : std::min(1.0, (double)prune_c / (double)w);
  if (tp) accept_all = true; // (everything behind the selection is the accepted-list path's)
#define PR_WS(ptr, type, name, bytes)                                                               \
  type ptr = (type)ws_get(ctx, name, bytes);                                                        \
  if (!ptr) return NTS_ENOMEM
  PR_WS(d_toff, uint64_t*, "sel_tile_off", n_kt * 8);
  PR_WS(d_tcnt, uint32_t*, "sel_tile_cnt", n_kt * 4 + 8);
  PR_WS(d_tord, uint8_t*, "sel_tile_ord", n_kt);
  PR_WS(d_tscan, uint64_t*, "sel_tile_scan", n_kt * 8);
  // control block: [0..63] candidate segment counters, [64] uncovered-range counter, [65] "a tile list did not fit"
  PR_WS(d_ctl, unsigned long long*, "sel_ctl", (N_SEG + 2) * 8);
  const uint64_t gap_cap = V / w + g->n_rec + 16;
  PR_WS(d_glo, uint64_t*, "gap_lo", gap_cap * 8);
  PR_WS(d_ghi, uint64_t*, "gap_hi", gap_cap * 8);
  // room for the ACCEPTED candidates (share p_accept of the candidates, 1 if unknown); too little is seen and retried
  uint64_t cseg_cap = (uint64_t)((double)V * frac * std::min(1.0, 1.5 * p_accept + (accept_all ? 1e-4 : 0.02)) * 1.25 / N_SEG) + 8192;
  // tiers: the accepted k-mers found are ~p x (3.4/p probes per window) plus what conserved stretches add; a retry follows if it was too small
  if (tp) cseg_cap = (uint64_t)((double)V * (std::max(6.0, 2.5 * ctx->tier_x0) / (double)w + 0.002) * 1.25 / N_SEG) + 8192;
  // the upper-halves kernel drops about 70 % of the accepted k-mers again (those that cannot win a window) -- where its tiles lie inside
  // one run; the window and gather kernels are launched over the capacity, so half of it is what they get until a call has
  // needed more (an assembly in pieces: the retry below, once per context)
  const bool elim_on = sel_hi && !(NTS_KNOB("NTS_SELECT_ELIM") && atoi(NTS_KNOB("NTS_SELECT_ELIM")) == 0);
  if (elim_on && !ctx->elim_needs_full_cap) cseg_cap = cseg_cap / 2 + 8192;
  unsigned long long ctl[N_SEG + 1];
  std::vector<uint64_t> glo(GAP_PEEK), ghi(GAP_PEEK);
  uint64_t m = 0, n_gap = 0, n_sparse = 0;
  uint64_t *d_sj = nullptr, *d_sk = nullptr;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const uint64_t m_max = cseg_cap * N_SEG;
    const uint64_t n_blk = (m_max + SPARSE_BLOCK - 1) / SPARSE_BLOCK;
    // (upper-halves select kernel: one slot of 64 * hi_per entries per tile in front of the segments)
    const uint64_t slot_words = sel_hi ? n_kt * 64ull * hi_per : 0ull;
    d_sj = (uint64_t*)ws_get(ctx, "sel_seg_j", (slot_words + m_max) * 8);
    d_sk = (uint64_t*)ws_get(ctx, "sel_seg_key", (slot_words + m_max) * 8);
    PR_WS(d_pj, uint64_t*, "cand_j", m_max * 8);
    PR_WS(d_pk, uint64_t*, "cand_key", m_max * 8);
    PR_WS(d_stj, uint64_t*, "stage_j", n_blk * SPARSE_BLOCK * 8);
    PR_WS(d_stk, uint64_t*, "stage_k", n_blk * SPARSE_BLOCK * 8);
    PR_WS(d_bcnt, uint64_t*, "blk_cnt", n_blk * 8);
    PR_WS(d_bscan, uint64_t*, "blk_scan", n_blk * 8);
    if (!d_sj || !d_sk) return NTS_ENOMEM;
    if (!ctx->sel_ctl_clean) HIP_TRY(ctx, hipMemsetAsync(d_ctl, 0, (N_SEG + 2) * 8, ctx->stream)); // (else: cleared by the previous call's last kernel)
    ctx->sel_ctl_clean = false;
    SelParams S;
    S.code = g->d_code + PAD;
    if (!accept_all || tp) {
      if (int rc_pk = ensure_pack(ctx, g)) return rc_pk;
    }
    S.pack = g->d_pack;
    S.run_pos = T.d_run_pos;
    S.run_vstart = T.d_run_vstart;
    S.n_runs = T.n_runs;
    S.n_valid = V;
    if (int rc_hp = hash_params_for(ctx, k, &S.hp)) return rc_hp;
    S.bf = filter ? filter->d_words : nullptr;
    S.fm = make_fastmod((filter ? filter->bytes : 8) * 8);
    S.tau = tau;
    S.seg_j = d_sj;
    S.seg_key = d_sk;
    S.seg_cap = cseg_cap;
    S.seg_count = d_ctl;
    S.tile_off = d_toff;
    S.tile_cnt = d_tcnt;
    S.tile_ordered = d_tord;
    // k_hash_select_hi drops accepted k-mers that cannot be a window's minimum (NTS_SELECT_ELIM=0: keeps them all; same result)
    S.w_elim = (NTS_KNOB("NTS_SELECT_ELIM") && atoi(NTS_KNOB("NTS_SELECT_ELIM")) == 0) ? 0u : w;
    if (tp) {
      PR_WS(d_dir, uint32_t*, "tier_dir", n_kt * 12);
      PR_WS(d_tstats, unsigned long long*, "tier_stats", 16);
      HIP_TRY(ctx, hipMemsetAsync(d_tstats, 0, 16, ctx->stream));
      TierParams Q;
      Q.code = S.code;
      Q.pack = S.pack;
      Q.run_pos = S.run_pos;
      Q.run_vstart = S.run_vstart;
      Q.n_runs = S.n_runs;
      Q.n_valid = V;
      Q.rec_vstart = T.d_rec_vstart;
      Q.n_rec = g->n_rec;
      Q.dir = d_dir;
      Q.hp = S.hp;
      Q.bf = S.bf;
      Q.fm = S.fm;
      Q.w = w;
      Q.halo = tp->halo;
      Q.core = tp->core;
      Q.scale = tp->scale;
      Q.exp_shift = tp->exp_shift;
      Q.exp_bias = tp->exp_bias;
      Q.n_tiers = tp->n_tiers;
      Q.seg_j = d_sj;
      Q.seg_key = d_sk;
      Q.seg_cap = cseg_cap;
      Q.seg_count = d_ctl;
      Q.tile_off = d_toff;
      Q.tile_cnt = d_tcnt;
      Q.tile_ordered = d_tord;
      Q.stats = d_tstats;
      Q.exp_stop = NTS_KNOB("NTS_TR_STOP") ? (uint32_t)atoi(NTS_KNOB("NTS_TR_STOP")) : 0u;
      Q.tiles = nullptr;
      Q.excl_on = Q.excl_hi = 0;
      ScopedTimer t(ctx, "hash_tiers", true);
      NTS_LAUNCH(k_tier_dir, dim3((uint32_t)((n_kt + 255) / 256)), dim3(256), 0, ctx->stream, T.d_run_vstart, T.n_runs, T.d_rec_vstart, g->n_rec, V,
                         tp->halo, tp->core, n_kt, (const TierTile*)nullptr, d_dir);
      launch_tiers(ctx, Q, n_kt);
    } else if (accept_all) {
      AcceptParams A;
      A.code = S.code;
      A.run_pos = S.run_pos;
      A.run_vstart = S.run_vstart;
      A.n_runs = S.n_runs;
      A.n_valid = V;
      A.hp = S.hp;
      A.bf = S.bf;
      A.fm = S.fm;
      A.summary = ctx->cur_summary;
      A.shift = ctx->cur_summary_shift;
      A.probe_mask = ~0u;
#ifdef NTS_EXPERIMENTS
      if (NTS_KNOB("NTS_ACC_NO_LOOKUP") && atoi(NTS_KNOB("NTS_ACC_NO_LOOKUP"))) { // a measurement switch that changes the RESULT: never silently, never in the product build
        A.probe_mask = 0u;
        static std::once_flag warned;
        std::call_once(warned, [] { fprintf(stderr, "[ntsynt_hip] NTS_ACC_NO_LOOKUP is set: sparse-filter sketches return WRONG results (timing experiment only)\n"); });
      }
#endif
      A.seg_j = d_sj;
      A.seg_key = d_sk;
      A.seg_cap = cseg_cap;
      A.seg_count = d_ctl;
      A.tile_off = d_toff;
      A.tile_cnt = d_tcnt;
      A.tile_ordered = d_tord;
      ScopedTimer t(ctx, "hash_accept", true);
      if (int rc_a = launch_accept(ctx, g, k, A, ctx->cur_fold, n_kt)) return rc_a;
    } else {
      const bool chained = g_live_contexts.load() > 1 && ctx->device < 32 && !(NTS_KNOB("NTS_SELECT_CHAIN") && atoi(NTS_KNOB("NTS_SELECT_CHAIN")) == 0);
      std::unique_lock<std::mutex> chain_lock(nts_chain::mu, std::defer_lock);
      if (chained) {
        chain_lock.lock();
        if (nts_chain::live[ctx->device]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, nts_chain::ev[ctx->device], 0));
      }
      struct ChainRecord // (behind the timer's closing event, whichever way the block is left)
      {
        nts_ctx* c;
        bool on;
        ~ChainRecord()
        {
          if (!on) return;
          hipEvent_t& e = nts_chain::ev[c->device];
          if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return;
          if (hipEventRecord(e, c->stream) == hipSuccess) nts_chain::live[c->device] = true;
        }
      } chain_record{ ctx, chained };
      ScopedTimer t(ctx, filter ? "hash_select" : "hash_select_nofilter", true);
      // a lane rolls 64 k-mers and keeps 64*c/w candidates on average: 8 private slots while that is small, 16 beyond
      // (a lane that runs out sends its tile through the general path, i.e. hashes it twice)
      {
        const uint32_t slots = 64.0 * frac > 2.5 ? 16u : 8u;
        // a block of 16 steps adds 16*c/w candidates on average; leave room for three times that plus four
        const uint32_t need = (uint32_t)std::min(16.0, std::ceil(48.0 * frac + 4.0));
        S.flush_at = slots == 8 ? 8u : slots - need;
      }
      const uint32_t nch = (k + 3) / 4;
      if (sel_hi) {
        // tiles per wave: enough to amortise the table load and to overlap probes with rolling, not so many that a small genome leaves CUs idle
        const uint64_t waves_wanted = 256ull * 32ull * 2ull;
        uint32_t tpw = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, n_kt / waves_wanted)); // (3 Gbp: 4 / 8 / 16 / 32 -> 1430 / 1460 / 1487 / 1404 Gbases/s)
        if (const char* e = NTS_KNOB("NTS_HI_TPW")) tpw = (uint32_t)std::max(1, std::min(64, atoi(e))); // (tests: small inputs through the multi-tile loop)
        const uint64_t per_wg = (uint64_t)HIW_WAVES * tpw;
        const dim3 grid((uint32_t)((n_kt + per_wg - 1) / per_wg));
        if (filter && hi_per == 2)
          NTS_LAUNCH((k_hash_select_hi<true, 2>), grid, dim3(HIW_THREADS), nch * 4096u, ctx->stream, S, nch, tpw, n_kt);
        else if (filter)
          NTS_LAUNCH((k_hash_select_hi<true, 4>), grid, dim3(HIW_THREADS), nch * 4096u, ctx->stream, S, nch, tpw, n_kt);
        else if (hi_per == 2)
          NTS_LAUNCH((k_hash_select_hi<false, 2>), grid, dim3(HIW_THREADS), nch * 4096u, ctx->stream, S, nch, tpw, n_kt);
        else
          NTS_LAUNCH((k_hash_select_hi<false, 4>), grid, dim3(HIW_THREADS), nch * 4096u, ctx->stream, S, nch, tpw, n_kt);
      }
      else if (64.0 * frac > 2.5)
        NTS_LAUNCH(k_hash_select<16>, dim3((uint32_t)n_kt), dim3(HASH_THREADS), 0, ctx->stream, S);
      else
        NTS_LAUNCH(k_hash_select<8>, dim3((uint32_t)n_kt), dim3(HASH_THREADS), 0, ctx->stream, S);
    }
    {
      ScopedTimer t(ctx, "cand_compact");
      if (int rc_s = scan_counts<uint32_t>(ctx, d_tcnt, n_kt, d_tscan)) return rc_s;
      if (sel_hi)
        NTS_LAUNCH(k_cand_compact_slots, dim3((uint32_t)((n_kt + CCS_TILES - 1) / CCS_TILES)), dim3(256), 0, ctx->stream, d_sj, d_sk, d_toff, d_tcnt, d_tscan, n_kt,
                           d_pj, d_pk, m_max, d_ctl + N_SEG + 1);
      else
        NTS_LAUNCH(k_cand_compact, dim3((uint32_t)((n_kt + CC_TILES - 1) / CC_TILES)), dim3(256), 0, ctx->stream, d_sj, d_sk, cseg_cap, d_toff, d_tcnt, d_tord, d_tscan, n_kt,
                           d_pj, d_pk, m_max, d_ctl + N_SEG + 1, false);
    }
    SparseParams Q;
    Q.pj = d_pj;
    Q.pk = d_pk;
    Q.m_scan_last = d_tscan + (n_kt - 1);
    Q.m_cnt_last = d_tcnt + (n_kt - 1);
    Q.m_max = m_max;
    Q.rec_vstart = T.d_rec_vstart;
    Q.rec_nv = T.d_rec_nv;
    Q.n_valid = V;
    Q.n_rec = g->n_rec;
    Q.w = w;
    Q.stage_j = d_stj;
    Q.stage_k = d_stk;
    Q.blk_cnt = d_bcnt;
    Q.gap_lo = d_glo;
    Q.gap_hi = d_ghi;
    Q.gap_count = d_ctl + N_SEG;
    Q.gap_cap = gap_cap;
    Q.overflow = d_ctl + N_SEG + 1;
    Q.no_gaps = accept_all ? 1u : 0u; // (the list holds every accepted k-mer that can win: a window without one has no minimizer)
    {
      ScopedTimer t(ctx, "sparse_win");
      NTS_LAUNCH(k_sparse_win, dim3((uint32_t)n_blk), dim3(SPARSE_THREADS), 0, ctx->stream, Q);
      // ordered output without a sort: scan the per-workgroup counts, gather (below; the candidate segments are free again)
      if (int rc_s = scan_counts<uint64_t>(ctx, d_bcnt, n_blk, d_bscan)) return rc_s;
    }
    HIP_TRY(ctx, hipGetLastError());
    // one synchronisation: candidate counters, uncovered-range count + a first batch of ranges, winner count
    uint64_t last_scan = 0, last_cnt = 0, listed = 0;
    {
      static_assert(N_SEG + 7 + 2 * GAP_PEEK < MAIL_WORDS - 8, "mailbox too small (the last word is the arrival flag)");
      const uint32_t peek = (uint32_t)std::min<uint64_t>(GAP_PEEK, gap_cap);
      Mail mb(ctx);
      const uint32_t a_ctl = mb.add(d_ctl, N_SEG + 1);
      const uint32_t a_scan = mb.add(d_bscan + (n_blk - 1), 1);
      const uint32_t a_cnt = mb.add(d_bcnt + (n_blk - 1), 1);
      const uint32_t a_tscan = mb.add(d_tscan + (n_kt - 1), 1);
      const uint32_t a_tcnt = mb.add((const uint64_t*)(d_tcnt + ((n_kt - 1) & ~1ull)), 1); // (32-bit counts: the pair holding the last one)
      const uint32_t a_lo = mb.add(d_glo, peek, d_ctl + N_SEG); // (the first n_gap of them: the count travels in the same mail)
      const uint32_t a_hi = mb.add(d_ghi, peek, d_ctl + N_SEG);
      const uint32_t a_tier = tp ? mb.add((const uint64_t*)ws_get(ctx, "tier_stats", 16), 2) : 0u;
      // the gather runs behind the mail kernel: the counters are on their way to the host while it works
      int rc_m = mb.launch(ctx);
      if (rc_m) return rc_m;
      {
        ScopedTimer t(ctx, "gather_winners");
        NTS_LAUNCH(k_gather_winners, dim3((uint32_t)n_blk), dim3(256), 0, ctx->stream, d_stj, d_stk, d_bcnt, d_bscan, d_sj, d_sk);
      }
      HIP_TRY(ctx, hipGetLastError());
      if ((rc_m = mb.wait(ctx))) return rc_m;
      for (uint32_t i = 0; i <= N_SEG; ++i) ctl[i] = ctx->mail[a_ctl + i];
      last_scan = ctx->mail[a_scan];
      last_cnt = ctx->mail[a_cnt];
      listed = ctx->mail[a_tscan] + (uint32_t)(ctx->mail[a_tcnt] >> (32 * ((n_kt - 1) & 1ull)));
      memcpy(glo.data(), ctx->mail + a_lo, (size_t)peek * 8);
      memcpy(ghi.data(), ctx->mail + a_hi, (size_t)peek * 8);
      if (tp) {
        ctx->last_tier_probes = ctx->mail[a_tier];
        ctx->last_tier_rounds = ctx->mail[a_tier + 1];
      }
    }
    uint64_t worst = 0;
    m = 0;
    for (uint32_t s = 0; s < N_SEG; ++s) {
      worst = std::max<uint64_t>(worst, ctl[s]);
      m += ctl[s];
    }
    n_gap = ctl[N_SEG];
    n_sparse = last_scan + last_cnt;
    ctx->last_many_listed = sel_hi ? m : 0;
    if (sel_hi) { // the tiles' own slots are not counted by the segment counters: the directory's total is
      m = listed;
      if (listed > m_max) worst = std::max<uint64_t>(worst, listed / N_SEG + 1); // the compacted array was cut short
    }
    if (NTS_KNOB("NTS_DEBUG_RETRY")) fprintf(stderr, "[select] attempt %d: worst segment %llu of %llu, listed %llu, tiers %d\n", attempt, (unsigned long long)worst, (unsigned long long)cseg_cap, (unsigned long long)m, tp ? 1 : 0);
    if (worst <= cseg_cap) break;
    if (elim_on) ctx->elim_needs_full_cap = true;
    if (attempt == 1) return fail(ctx, NTS_EHIP, "candidate segments overflowed twice");
    cseg_cap = worst + 1024; // candidate lists were truncated: everything downstream of them is void; run again
  }
  ctx->last_candidates = m;
  if (accept_all) n_gap = 0; // every accepted k-mer is a candidate: a window without candidates has no minimizer
  ctx->last_gaps = n_gap;
  res.d_j = d_sj;
  res.d_key = d_sk;
  res.count = n_sparse;
  if (n_gap == 0) return NTS_OK;
  if (n_gap > gap_cap) return fail(ctx, NTS_EHIP, "uncovered-range list overflowed");
  // ---- dense evaluation of the uncovered ranges ------------------------------------------------------------
  if (n_gap > GAP_PEEK) {
    glo.resize(n_gap);
    ghi.resize(n_gap);
    HIP_TRY(ctx, hipMemcpyAsync(glo.data(), d_glo, n_gap * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ghi.data(), d_ghi, n_gap * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<size_t> ord(n_gap);
  for (size_t i = 0; i < n_gap; ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return glo[a] < glo[b]; });
  std::vector<uint64_t> pv(n_gap), pn(n_gap);
  std::vector<uint32_t> tiles, spans; // spans: per listed tile, first and last in-tile index inside a range
  uint64_t covered = 0;
  for (size_t i = 0; i < n_gap; ++i) {
    const uint64_t a = glo[ord[i]], b = ghi[ord[i]];
    pv[i] = a;
    pn[i] = b - a + 1;
    covered += pn[i];
    for (uint64_t t = a / KEY_TILE; t <= b / KEY_TILE; ++t) {
      const uint32_t lo_in = (uint32_t)(std::max(a, t * KEY_TILE) - t * KEY_TILE);
      const uint32_t hi_in = (uint32_t)(std::min(b, t * KEY_TILE + KEY_TILE - 1) - t * KEY_TILE);
      if (tiles.empty() || tiles.back() != (uint32_t)t) {
        tiles.push_back((uint32_t)t);
        spans.push_back(lo_in);
        spans.push_back(hi_in);
      } else { // (ranges ascend: the tile's first index stays, its last grows)
        spans[spans.size() - 1] = std::max(spans.back(), hi_in);
      }
    }
  }
  ctx->last_gap_kmers = covered;
  SortedOut dense;
  SortedOut sparse_list;
  sparse_list.d_j = d_sj;
  sparse_list.d_key = d_sk;
  sparse_list.count = n_sparse;
  if (ctx->small_gap_path) { // the ranges through the tiered selection; not applicable, or a retry after it gave up: the dense kernels
    int rc_g = run_gap_tiers(ctx, g, T, k, w, filter, tau, prune_c, pv, pn, covered, sparse_list, dense);
    if (rc_g) return rc_g;
    if (dense.d_ctl) {
      res = dense;
      return NTS_OK;
    }
  }
  int rc = run_dense_sorted(ctx, g, T, k, w, filter, &pv, &pn, &tiles, covered, "gap_", dense, &sparse_list, &spans);
  if (rc) return rc;
  if (dense.d_ctl) { // merged on the device: the caller reads the count after its own synchronisation
    res = dense;
    return NTS_OK;
  }
  if (dense.count == 0) return NTS_OK;
  const uint64_t total = n_sparse + dense.count;
  PR_WS(d_mj, uint64_t*, "merged_j", total * 8);
  PR_WS(d_mk, uint64_t*, "merged_key", total * 8);
  {
    ScopedTimer t(ctx, "merge_lists");
    NTS_LAUNCH(k_merge_lists, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_sj, d_sk, n_sparse, dense.d_j, dense.d_key,
                       dense.count, d_mj, d_mk);
  }
  HIP_TRY(ctx, hipGetLastError());
  res.d_j = d_mj;
  res.d_key = d_mk;
  res.count = total;
  return NTS_OK;
#undef PR_WS
}


} // namespace

extern "C" int nts_sketch_mode(nts_ctx* ctx, int mode, uint32_t prune_c)
{
  if (!ctx || mode < 0 || mode > 2) return fail(ctx, NTS_EINVAL, "nts_sketch_mode: mode must be 0 (auto), 1 (dense) or 2 (pruned)");
  ctx->sketch_mode = mode;
  ctx->prune_c = prune_c;
  return NTS_OK;
}

extern "C" int nts_sketch_summary(nts_ctx* ctx, int mode, uint32_t* last_shift)
{
  if (!ctx || mode < -1 || mode > 2) return fail(ctx, NTS_EINVAL, "nts_sketch_summary: mode is -1 (query), 0 (auto), 1 (never) or 2 (no LDS copy)");
  if (mode >= 0) {
    ctx->summary_mode = mode == 1 ? 1 : 0;
    ctx->fold_mode = mode == 2 ? 1 : 0;
  }
  if (last_shift) *last_shift = ctx->last_summary;
  return NTS_OK;
}

extern "C" int nts_sketch_select(nts_ctx* ctx, int impl)
{
  if (!ctx || impl < 0 || impl > 2)
    return fail(ctx, NTS_EINVAL, "nts_sketch_select: impl is 0 (auto), 1 (full-width rolling) or 2 (upper halves wherever the kernel applies)");
  ctx->select_impl = impl;
  return NTS_OK;
}

extern "C" int nts_sketch_tiers(nts_ctx* ctx, int mode, double x0, int half_steps, uint64_t* last_probes, uint64_t* last_rounds, uint32_t* last_tiers)
{
  if (!ctx || mode < -1 || mode > 2 || x0 < 0 || x0 > 64)
    return fail(ctx, NTS_EINVAL, "nts_sketch_tiers: mode is -1 (query), 0 (auto), 1 (never) or 2 (wherever the kernel applies); 0 <= x0 <= 64");
  if (mode >= 0) {
    ctx->tier_mode = mode;
    ctx->gap_tiers_off = mode == 1 ? 1 : 0;
    ctx->tier_x0 = x0;
    ctx->tier_half = half_steps ? 1u : 0u;
  }
  if (last_probes) *last_probes = ctx->last_tier_probes;
  if (last_rounds) *last_rounds = ctx->last_tier_rounds;
  if (last_tiers) *last_tiers = (uint32_t)ctx->last_tiers;
  return NTS_OK;
}

extern "C" int nts_path_stats(nts_ctx* ctx, uint64_t* sketch_many_listed, uint64_t* bf_direct_indices, uint32_t* bf_list_fallback)
{
  if (!ctx) return NTS_EINVAL;
  if (sketch_many_listed) *sketch_many_listed = ctx->last_many_listed;
  if (bf_direct_indices) *bf_direct_indices = ctx->last_bf_direct;
  if (bf_list_fallback) *bf_list_fallback = ctx->last_bf_fallback;
  return NTS_OK;
}

extern "C" int nts_bf_level_stats(nts_ctx* ctx, uint32_t* sparse_level, uint64_t* accepted_kmers)
{
  if (!ctx) return NTS_EINVAL;
  if (sparse_level) *sparse_level = ctx->last_bf_sparse_level;
  if (accepted_kmers) *accepted_kmers = ctx->last_bf_sparse_accepted;
  return NTS_OK;
}

extern "C" int nts_sketch_stats(nts_ctx* ctx, uint64_t* candidates, uint64_t* uncovered_ranges, uint64_t* uncovered_kmers, uint32_t* prune_c_used)
{
  if (!ctx) return NTS_EINVAL;
  if (prune_c_used) *prune_c_used = ctx->last_c;
  if (candidates) *candidates = ctx->last_candidates;
  if (uncovered_ranges) *uncovered_ranges = ctx->last_gaps;
  if (uncovered_kmers) *uncovered_kmers = ctx->last_gap_kmers;
  return NTS_OK;
}

extern "C" int nts_sketch(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint32_t w, const nts_bf* filter, const nts_interval* mask,
                          uint64_t n_mask, nts_mx** out)
{
  return nts_sketch_ex(ctx, g, k, w, filter, nullptr, mask, n_mask, out);
}

extern "C" int nts_sketch_ex(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint32_t w, const nts_bf* filter, const nts_bf* filter_out,
                             const nts_interval* mask, uint64_t n_mask, nts_mx** out)
{
  if (!ctx || !g || !out || k == 0 || w == 0 || (n_mask && !mask)) return fail(ctx, NTS_EINVAL, "nts_sketch: bad arguments");
  ctx->cur_rep = filter_out; // (read by the key kernel's launch; cleared on every way out)
  struct RepGuard
  {
    nts_ctx* c;
    ~RepGuard() { c->cur_rep = nullptr; }
  } rep_guard{ ctx };
  if (w > WIN_MAX_W) return fail(ctx, NTS_ERANGE, "nts_sketch: w exceeds the LDS-resident window limit (12000)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  GenomeTables scratch;
  const GenomeTables* T = nullptr;
  int rc = get_tables(ctx, g, k, mask, n_mask, scratch, &T);
  if (rc) return rc;
  const RunTable& rt = T->rt;
  nts_mx* mx = new nts_mx();
  ctx->last_candidates = ctx->last_gaps = ctx->last_gap_kmers = ctx->last_many_listed = 0;
  ctx->small_gap_path = true;
  if (rt.n_valid == 0 || T->n_win_tiles(w) == 0) {
    *out = mx;
    return NTS_OK;
  }
  auto bail = [&](int code) {
    hipStreamSynchronize(ctx->stream);
    nts_mx_free(ctx, mx);
    ctx->cur_summary = nullptr;
    ctx->cur_tile_any = nullptr;
    return code;
  };
#define SK_TRY(expr)                                                                                \
  do {                                                                                              \
    int rc_ = (expr);                                                                               \
    if (rc_ != NTS_OK) return bail(rc_);                                                            \
  } while (0)
#define SK_HIP(expr)                                                                                \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                 \
      return bail(e_ == hipErrorOutOfMemory ? NTS_ENOMEM : NTS_EHIP);                               \
    }                                                                                               \
  } while (0)
#define SK_WS(ptr, type, name, bytes)                                                               \
  type ptr = (type)ws_get(ctx, name, bytes);                                                        \
  if (!ptr) return bail(NTS_ENOMEM)

  // Pruning policy.  p = share of this genome's k-mers the filter accepts, estimated from occupancies: the
  // genome alone would set about bits*(1-exp(-V/bits)) bits, the common filter kept popcount of them.  A window
  // of w k-mers holds ~c*p accepted candidates when hashes <= (c/w)*2^64 are kept; c*p = 12 leaves ~6e-6 of the
  // windows uncovered (they are re-evaluated densely).  The select kernel probes its candidates in batches and keeps
  // only the accepted ones, so c may grow until a quarter of the k-mers are candidates (p down to 48/w); below that
  // the dense kernels take over.
  // (one threshold from w = 200; below it the tiers, with or without a filter -- without one, ntSynt --no-common, every listed k-mer is accepted and
  //  the rounds only decide which k-mers are hashed in full: 3 Gbp at w = 10 / 33 / 63 / 90 / 150 / 200: 35 / 14 / 10 / 8 / 6.5 / 6.1 ms, against 64 / 57 /
  //  49 ms through the window tiles and 23 / 8.5 / 7.0 with one threshold; scripts/mode_sweep.py)
  const uint32_t prune_min_w = 200u;
  bool pruned = ctx->sketch_mode == 2 || (ctx->sketch_mode == 0 && w >= prune_min_w);
  // a filter-out filter (indexlr -r: experimental in the reference) is served by the every-k-mer-probed kernels only
  if (filter_out) pruned = false;
  uint32_t prune_c = ctx->prune_c;
  double p = 1.0; // accepted share of the candidates (1 when unknown: sizes the candidate arrays)
  bool tiered = false;
  TierPlan plan;
  // (tiers forced: wherever the kernel applies.  Windows of 64 .. 199 k-mers, where one threshold never paid and every k-mer was probed:
  //  the tiered selection is looked at there too -- a whole 3 Gbp genome at w = 100 against its family's filter: 90 ms the dense way)
  const bool tiers_forced = ctx->tier_mode == 2 && ctx->sketch_mode == 0 && !filter_out && prune_c == 0;
  // (windows below WIN_FUSE_W = 64, round 6: the same selection down to w = 8 where its estimated cost stays below tier_small_c of the every-k-mer pass)
  const uint32_t tier_min_w = NTS_KNOB("NTS_TIER_MIN_W") ? (uint32_t)atoi(NTS_KNOB("NTS_TIER_MIN_W")) : 8u;
  const double tier_small_c = NTS_KNOB("NTS_TIER_SMALL_C") ? atof(NTS_KNOB("NTS_TIER_SMALL_C")) : 0.85;
  // (without a filter -- ntSynt --no-common -- below the window where one threshold takes over: every listed k-mer is accepted, the tiers only
  //  decide which k-mers are hashed in full at all)
  const bool tiers_small_w = ctx->tier_mode == 0 && ctx->sketch_mode == 0 && !filter_out && prune_c == 0 && w >= tier_min_w && w < 200;
  if ((pruned || tiers_forced || tiers_small_w) && prune_c == 0) {
    if (filter) {
      uint64_t pc = 0;
      SK_TRY(nts_bf_popcount(ctx, filter, &pc));
      const double bits = (double)filter->bytes * 8.0;
      auto share = [&](double kmers) { // of a genome with that many (distinct) k-mers
        const double own = bits * (1.0 - std::exp(-kmers / bits));
        return own > 0 ? std::min(1.0, (double)pc / own) : 1.0;
      };
      if (g->part_bases.size() > 1 && g->total_bases) {
        // a batch: every part is a genome of its own as far as the filter is concerned (the parts of a batch are
        // assemblies of one family: their k-mers are largely the same ones, not three times as many)
        double acc = 0;
        for (uint64_t b : g->part_bases) {
          const double f = (double)b / (double)g->total_bases;
          acc += f * share((double)rt.n_valid * f);
        }
        p = acc;
      } else {
        p = share((double)rt.n_valid);
      }
    }
    // c*p = 11 accepted candidates per window on average (`cp` below; 10.5 until round 3).  (More would not empty the list of uncovered ranges:
    // beyond the ~V*(cp/w)*exp(-cp) chance ones there are the stretches the other genomes do not share at all.
    // Measured at 3 x 3 Gbp, w = 1000, p = 0.70, with k_hash_select_hi and the uncovered ranges probed only where a window
    // reads them: c = 12 / 13 / 14 / 15 / 16 -> 1021 / 1085 / 1124 / 1134 / 1124 Gbases/s; with k_hash_select, whose rolling
    // cost twice as much per k-mer, and whole key tiles probed around every range, the optimum was c = 18, cp = 12.)
    // (round 3, with the select kernel dropping hopeless candidates and a family with insertions: c = 13 .. 17 ->
    //  1280 / 1397 / 1420 / 1476 / 1413 Gbases/s; c = 16 at p = 0.70.  Round 5, same family: c = 12 .. 17 -> 1118 / 1250 / 1432 / 1468 /
    //  1522 / 1516.)  An assembly in thousands of pieces (more than one run of valid bases per 2^20 k-mers) has its uncovered
    //  ranges whatever c is -- scaffold ends, gaps, repeats -- and a listing cost that rises faster with c (tiles that list more than
    //  their slots hold): the assembly-like family at 3 x 3 Gbp, p = 0.55: c = 14 / 16 / 18 / 21 -> 986 / 1134 / 1121 / 1070 Gbases/s; with the uncovered
    //  ranges through the tiered selection (run_gap_tiers): c = 12 / 14 / 16 / 18 -> 933 / 1193 / 1291 / 1250.
    const bool in_pieces = (uint64_t)T->n_runs > (rt.n_valid >> 20) + 64;
    const double cp = in_pieces ? 8.7 : 11.0;
    const double want = std::max(8.0, std::ceil(cp / std::max(p, 1e-4)));
    // (measured: at a quarter of the k-mers as candidates the pruned pass is still twice as fast as the dense one;
    // at 40 % single lanes run out of slots in most tiles and it is half as fast)
    // (with short windows the accepted candidates per lane of k_hash_select get dense sooner: measured at w = 250,
    // 3 x 100 Mbp: c = 35 -> 56 Gbases/s against 34 dense, c = 50 -> 17; at w = 1000, c = 203 still gives 94)
    const double cap = (w >= 512 ? 0.25 : 0.15) * (double)w;
    prune_c = (uint32_t)std::min(want, cap);
    if ((ctx->sketch_mode == 0 && want > cap) || w < prune_min_w) pruned = ctx->sketch_mode == 2;
    // Tiered selection (nts_tiers.inc): thresholds tau_0 2^t instead of one threshold, each probed only where a window is still
    // without an accepted k-mer -- ~3.4/p probes per window instead of 11/p.  It takes over where one threshold lists so many
    // k-mers that the upper-halves kernel no longer applies, down to accepted shares where even the first tier is half of all k-mers.
    if (!filter_out && ctx->sketch_mode == 0 && ctx->tier_mode != 1 && k <= FAST_K_MAX && w >= tier_min_w && w <= 4097) {
      // (first tier: 2.4 accepted k-mers per window on average; 1.2 below w = 64, where every tier is a large share of the k-mers and a
      //  thinner first one saves 8-12 % of the probes: scripts/tiers_x0_sweep.py)
      const double x0 = ctx->tier_x0 > 0 ? ctx->tier_x0 : (w < WIN_FUSE_W ? 1.2 : 2.4);
      const double c0 = x0 / std::max(p, 1e-6);
      const double switch_c = 54.0 * (double)w / 1000.0; // (beyond it k_hash_select takes over from k_hash_select_hi)
      // Below w = 200 the other way probes every k-mer (k_window_min<true> under 64: ~85 ms per 3 Gbp at k = 24, 105 at k = 100; the key array
      // above: 110-120).  The tiers cost ~91 ms per (probe per k-mer) at k = 24 and more with k -- a listed k-mer is hashed from scratch,
      // ceil(k / 4) table reads: 290 ms at k = 100 -- and where w <= k they probe every k-mer of the stretch a substitution empties.  Probes
      // per k-mer, fitted to scripts/tiers_small_w.py, tiers_x0_sweep.py and the k = 16 / 100 runs: 0.9 (1 - p) min(1, (k/w)^2) + 2.5 / (p w).
      bool pays = true;
      if (w < 200 && ctx->tier_mode != 2) {
        const double kw = std::min(1.0, (double)k / (double)w);
        const double est = std::min(1.0, 0.9 * (1.0 - p) * kw * kw + 2.5 / std::max(p * (double)w, 1e-9));
        const double per_probe = 0.3 + 0.7 * (double)k / 24.0;
        pays = per_probe * est <= tier_small_c * (1.0 + 0.002 * ((double)k - 24.0));
      }
      const double c0_max = 0.5 * (double)w;
      const bool fits = w >= WIN_FUSE_W || 48.0 * 7.5 * (double)rt.n_valid / (double)w <= 120e9;
      // (an assembly in pieces lists fewer candidates per window -- cp = 8.7 -- but that says nothing about where the tiers overtake the one
      //  threshold: the comparison is made with the whole-genome figure; 200,000 contigs at w = 250, p = 0.7: tiers 7.4 ms, one threshold 10.4)
      const double want_whole = std::max(8.0, std::ceil(11.0 / std::max(p, 1e-4)));
      if (c0 <= c0_max && fits && pays && (ctx->tier_mode == 2 || want_whole > switch_c)) {
        tiered = true;
        pruned = false;
        plan.c0 = c0;
        const double t0 = std::max(1.0, std::floor(c0 / (double)w * 4294967296.0));
        plan.scale = (float)(1.0 / (t0 + 1.0));
        plan.exp_shift = ctx->tier_half ? 22u : 23u;
        plan.exp_bias = ctx->tier_half ? 253 : 126;
        // explicit tiers while their threshold stays below ~3/4 of all hashes; the last tier takes the rest
        uint32_t n_exp = 1;
        auto c_of = [&](uint32_t t) { return ctx->tier_half ? c0 * ((t & 1u) ? 1.5 : 1.0) * (double)(1u << (t >> 1)) : c0 * (double)(1u << t); };
        while (n_exp < TR_TIERS_MAX - 1 && c_of(n_exp) <= 0.75 * (double)w) ++n_exp;
        plan.n_tiers = n_exp + 1;
        plan.halo = ((w - 1 + 63) / 64) * 64;
        plan.core = TR_EXT - 2 * plan.halo;
        prune_c = (uint32_t)std::max(1.0, std::ceil(c0));
      }
    }
  }
  ctx->last_c = (pruned || tiered) ? prune_c : 0;
  ctx->last_tiers = tiered ? plan.n_tiers : 0;
  ctx->last_tier_probes = ctx->last_tier_rounds = 0;
  // Dense pass over a sparse filter: with occupancy o, a summary bit covering 2^shift filter bits is set with probability
  // ~ o * 2^shift; when that is small the summary answers nearly every probe from the L2 (k_hash_keys_sparse).
  ctx->cur_summary = nullptr;
  ctx->cur_fold = nullptr;
  ctx->cur_tile_any = nullptr;
  ctx->last_summary = 0;
  bool accept_all = false;
  if (!pruned && !(tiered && ctx->tier_mode == 2) && filter && filter->owned && ctx->summary_mode == 0 && !filter_out) {
    std::lock_guard<std::mutex> summary_lock(filter->mu);
    uint64_t pc = 0;
    SK_TRY(nts_bf_popcount(ctx, filter, &pc));
    const double bits = (double)filter->bytes * 8.0;
    uint32_t shift = 7;
    const uint32_t sum_log2 = NTS_KNOB("NTS_SUMMARY_LOG2_BITS") ? (uint32_t)std::max(16, std::min(28, atoi(NTS_KNOB("NTS_SUMMARY_LOG2_BITS")))) : SUMMARY_LOG2_BITS;
    while ((bits / (double)(1ull << shift)) > (double)(1ull << sum_log2) && shift < 30) ++shift; // summary <= 2^sum_log2 bits
    // How full may the summary be?  A set summary bit sends the k-mer to HBM, so the path costs ~8 ms of hashing and look-ups per 3 Gbp plus
    // P(bit set) = 1 - exp(-occupancy 2^shift) of the every-k-mer pass (75 ms).  Against the tiered selection (families of 4 .. 7 genomes at
    // 6-10 %, scripts/summary_switch.py): 32 against 92 ms at 0.31 (five genomes at 10 %: neither path applied there until round 5 --
    // every k-mer was probed), 52 against 50 at 0.82, 56 against 50 at 0.96: it wins below ~0.7; against every k-mer probed, where the
    // tiers do not apply, as long as a summary bit says anything at all.
    const double sum_max = NTS_KNOB("NTS_SUMMARY_MAX") ? atof(NTS_KNOB("NTS_SUMMARY_MAX")) : (tiered ? 0.7 : 3.0);
    if ((double)pc / bits * (double)(1ull << shift) < sum_max) {
      SK_TRY(bf_make_summary(ctx, filter, shift));
      const uint64_t key_tiles = (rt.n_valid + KEY_TILE - 1) / KEY_TILE;
      SK_WS(d_any, uint32_t*, "tile_any", (key_tiles + 4) * 4);
      ctx->cur_summary = filter->d_summary;
      ctx->cur_summary_shift = shift;
      ctx->cur_tile_any = d_any;
      ctx->last_summary = shift;
      // accepted share of the k-mers, for sizing the candidate segments (a retry follows if it was too small)
      const double own = bits * (1.0 - std::exp(-(double)rt.n_valid / bits));
      p = own > 0 ? std::min(1.0, (double)pc / own) : 1.0;
      accept_all = ctx->sketch_mode != 1; // (mode "dense" keeps the key / window kernels, with the summary in front of the probes)
      if (accept_all) {
        tiered = false;
        ctx->last_tiers = 0;
      }
      // the folded copy pays while it rejects a good share of the k-mers: set bits / 2^19 below ~1.2 (about 70 % of its bits set)
      ctx->cur_fold = (ctx->fold_mode == 0 && bits >= (double)(1u << FOLD_BITS_LOG2) && (double)pc < 1.2 * (double)(1u << FOLD_BITS_LOG2))
                        ? filter->d_fold : nullptr;
    }
  }

  for (int attempt = 0;; ++attempt) {
    SortedOut res;
    if (pruned || accept_all || tiered) {
      const int rc_p = run_pruned(ctx, g, *T, k, w, filter, prune_c, p, res, accept_all, tiered ? &plan : nullptr);
      if (rc_p == NTS_ENOMEM && tiered && !pruned && !accept_all && w < WIN_FUSE_W) {
        // short windows have a second way that keeps nothing but the minimizers (k_window_min<true>): the lists of accepted k-mers
        // the tiers fill did not fit the device next to what the caller holds
        tiered = false;
        ctx->last_c = ctx->last_tiers = 0;
        SK_TRY(run_dense_sorted(ctx, g, *T, k, w, filter, nullptr, nullptr, nullptr, rt.n_valid, "", res));
      } else {
        SK_TRY(rc_p);
      }
    } else
      SK_TRY(run_dense_sorted(ctx, g, *T, k, w, filter, nullptr, nullptr, nullptr, rt.n_valid, "", res));
    const uint64_t count = res.count; // exact, or an upper bound when the count still lives on the device
    mx->n = count;
    if (count) {
      SK_TRY(alloc_result(ctx, mx, count));
      {
        // one search per FIN_CHUNK elements of the first list (k_fin_chunks), the elements walk on from there
        const uint64_t n_first = res.b_j ? res.na : count;
        const uint64_t n_chunks = n_first / FIN_CHUNK + 1;
        SK_WS(d_cb, uint64_t*, "fin_chunks", n_chunks * 16);
        uint32_t* const d_crun = (uint32_t*)(d_cb + n_chunks);
        uint32_t* const d_crec = d_crun + n_chunks;
        ScopedTimer t(ctx, "finalize");
        NTS_LAUNCH(k_fin_chunks, dim3((uint32_t)((n_chunks + 255) / 256)), dim3(256), 0, ctx->stream, res.d_j, n_first,
                           res.d_ctl ? res.d_ctl + 1 : nullptr, res.b_j, res.b_j ? res.d_ctl : nullptr, T->d_run_pos, T->d_run_vstart, T->n_runs,
                           g->d_rec_off, g->n_rec, d_cb, d_crun, d_crec);
        NTS_LAUNCH(k_finalize, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, res.d_j, res.d_key, (uint64_t)count,
                           res.d_ctl ? res.d_ctl + 1 : nullptr, res.b_j, res.b_key, res.b_j ? res.d_ctl : nullptr, res.na, T->d_run_pos, T->d_run_vstart, T->n_runs, g->d_rec_off, g->n_rec, k,
                           d_cb, d_crun, d_crec, mx->d_h1, mx->d_rec, mx->d_pos);
      }
      SK_HIP(hipGetLastError());
    }
    // the last kernel of the call also clears the pruned pass's control block for the next call (one fill launch less
    // at the head of every sketch)
    uint64_t* const sel_ctl = (pruned || accept_all || tiered) ? (uint64_t*)ws_get(ctx, "sel_ctl", (N_SEG + 2) * 8) : nullptr;
    if (!res.d_ctl) {
      Mail done(ctx); // nothing to fetch: only the arrival flag
      if (sel_ctl) done.clear_after(sel_ctl, N_SEG + 2);
      SK_TRY(done.post(ctx));
      ctx->sel_ctl_clean = sel_ctl != nullptr;
      break;
    }
    Mail mb(ctx);
    const uint32_t at = mb.add(res.d_ctl, 3);
    if (sel_ctl) mb.clear_after(sel_ctl, N_SEG + 2);
    SK_TRY(mb.post(ctx));
    ctx->sel_ctl_clean = sel_ctl != nullptr;
    if (ctx->mail[at + 2] == 0) {
      mx->n = ctx->mail[at + 1];
      break;
    }
    // the device-side merge of the uncovered ranges gave up (too many winners): once more on the general path
    if (attempt == 1) return bail(fail(ctx, NTS_EHIP, "nts_sketch: uncovered ranges could not be merged"));
    ctx->mx_pool.push_back({ mx->d_h1, mx->cap_bytes });
    mx->d_h1 = nullptr;
    mx->d_pos = nullptr;
    mx->d_rec = nullptr;
    ctx->small_gap_path = false;
  }
  ctx->small_gap_path = true;
  ctx->cur_summary = nullptr;
  ctx->cur_tile_any = nullptr;
  *out = mx;
  return NTS_OK;
#undef SK_TRY
#undef SK_HIP
#undef SK_WS
}

extern "C" {

uint64_t nts_mx_count(const nts_mx* mx)
{
  return mx ? mx->n : 0;
}

void nts_mx_free(nts_ctx* ctx, nts_mx* mx)
{
  if (!mx) return;
  if (ctx) hipSetDevice(ctx->device);
  if (mx->d_h1) { // h1 | pos | rec share one allocation
    if (ctx && ctx->mx_pool.size() < 32 && mx->cap_bytes)
      ctx->mx_pool.push_back({ mx->d_h1, mx->cap_bytes });
    else
      dev_free(mx->d_h1);
  }
  delete mx;
}

int nts_mx_download(nts_ctx* ctx, const nts_mx* mx, uint64_t* h1, uint32_t* rec, uint64_t* pos)
{
  if (!ctx || !mx) return fail(ctx, NTS_EINVAL, "nts_mx_download: bad arguments");
  if (mx->n == 0) return NTS_OK;
  if (!h1 || !rec || !pos) return fail(ctx, NTS_EINVAL, "nts_mx_download: NULL output");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(h1, mx->d_h1, mx->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(rec, mx->d_rec, mx->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(pos, mx->d_pos, mx->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

int nts_mx_export_async(nts_ctx* ctx, const nts_mx* mx, void* h1_dev, void* rec_dev, void* pos_dev)
{
  if (!ctx || !mx) return fail(ctx, NTS_EINVAL, "nts_mx_export: bad arguments");
  if (mx->n == 0) return NTS_OK;
  if (!h1_dev || !rec_dev || !pos_dev) return fail(ctx, NTS_EINVAL, "nts_mx_export: NULL destination");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(h1_dev, mx->d_h1, mx->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(rec_dev, mx->d_rec, mx->n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(pos_dev, mx->d_pos, mx->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  return NTS_OK;
}

int nts_mx_export(nts_ctx* ctx, const nts_mx* mx, void* h1_dev, void* rec_dev, void* pos_dev)
{
  const int rc = nts_mx_export_async(ctx, mx, h1_dev, rec_dev, pos_dev);
  if (rc != NTS_OK || mx->n == 0) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return NTS_OK;
}

int nts_mx_device_ptrs(const nts_mx* mx, void** h1, void** rec, void** pos)
{
  if (!mx) return NTS_EINVAL;
  if (h1) *h1 = mx->d_h1;
  if (rec) *rec = mx->d_rec;
  if (pos) *pos = mx->d_pos;
  return NTS_OK;
}

int nts_mx_upload(nts_ctx* ctx, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos, uint64_t n, nts_mx** out)
{
  if (!ctx || !out || (n && (!h1 || !rec || !pos))) return fail(ctx, NTS_EINVAL, "nts_mx_upload: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  nts_mx* mx = new nts_mx();
  mx->n = n;
  if (n) {
    if (dev_malloc((void**)&mx->d_h1, n * 20) != hipSuccess) {
      nts_mx_free(ctx, mx);
      return fail(ctx, NTS_ENOMEM, "nts_mx_upload: hipMalloc");
    }
    mx->cap_bytes = n * 20;
    mx->d_pos = mx->d_h1 + n;
    mx->d_rec = (uint32_t*)(mx->d_pos + n);
    hipMemcpyAsync(mx->d_h1, h1, n * 8, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(mx->d_rec, rec, n * 4, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(mx->d_pos, pos, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      nts_mx_free(ctx, mx);
      return fail(ctx, NTS_EHIP, "nts_mx_upload: copy");
    }
  }
  *out = mx;
  return NTS_OK;
}

void nts_free(void* p)
{
  free(p);
}

} // extern "C"
