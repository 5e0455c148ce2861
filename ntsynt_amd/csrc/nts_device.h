// Device-side arithmetic shared by the kernels: canonical ntHash2,
// Bloom bit addressing (A2), exact 64-bit modulo by a runtime divisor.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nts {

constexpr uint64_t SEED_A = 0x3c8bfbb395c60474ULL;
constexpr uint64_t SEED_C = 0x3193c18562a02b4cULL;
constexpr uint64_t SEED_G = 0x20323ed082572324ULL;
constexpr uint64_t SEED_T = 0x295549f54be24456ULL;
constexpr uint64_t MULTISEED = 0x90b45d39fb6da1faULL;
constexpr uint64_t KEY_MAX = 0xFFFFFFFFFFFFFFFFULL;
constexpr uint8_t CODE_INVALID = 4;

// split rotate left by one: bits 0..32 form a 33-bit ring, bits 33..63 a 31-bit ring.
// On the device the rotation is written on the two 32-bit halves (alignbit / and-or / shift-or, six full-rate
// instructions); 64-bit shifts issue at half rate on CDNA and the rolling loops are VALU-bound.
__host__ __device__ __forceinline__ uint64_t srol1(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  const uint32_t lo2 = (lo << 1) | (hi & 1u);                        // bit 32 wraps to bit 0
  const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 31);          // (hi << 1) | (lo >> 31)
  const uint32_t hi2 = (t & ~2u) | ((hi >> 30) & 2u);                // bit 63 wraps to bit 33
  return ((uint64_t)hi2 << 32) | lo2;
#else
  const uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
  return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
#endif
}

__host__ __device__ __forceinline__ uint64_t sror1(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  const uint32_t lo2 = __builtin_amdgcn_alignbit(hi, lo, 1);         // (lo >> 1) | (hi << 31): bit 32 moves to bit 31
  // bit 0 -> 32, bits 34..63 move down, bit 33 -> 63: four operations (written with the bit-field insert spelled out;
  // left to itself the optimiser rewrites it as three masks and a three-way or)
  uint32_t low;
  asm("v_bfi_b32 %0, 1, %1, %2" : "=v"(low) : "v"(lo), "v"(hi >> 1)); // (lo & 1) | ((hi >> 1) & ~1)
  const uint32_t hi2 = ((hi & 2u) << 30) | low;
  return ((uint64_t)hi2 << 32) | lo2;
#else
  const uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
  return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
#endif
}

__host__ __device__ __forceinline__ uint64_t extend_h1(uint64_t h0, uint32_t k)
{
  uint64_t t = h0 * (1ULL ^ ((uint64_t)k * MULTISEED));
  t ^= t >> 27;
  return t;
}

// Kernel-argument block for the hashing kernels.
//   roll_f[cin*4+cout] = seed[cin] ^ srol^k(seed[cout])             (forward strand update)
//   roll_r[cin*4+cout] = srol^k(seed[3-cin]) ^ seed[3-cout]         (reverse strand update)
//   init[(i*4 + b)*2 + {0,1}] = { srol^(k-1-i)(seed[b]), srol^i(seed[3-b]) }   (device memory; the hash of a
//                                lane's first k-mer is the XOR of k such pairs: no rotations, independent loads)
//   init4[(g*256 + v)*2 + {0,1}] = XOR over j < 4, 4g+j < k of init[4g+j][(v >> 2j) & 3]: the same sum taken four
//                                bases (one byte of the 2-bit genome image) at a time, ceil(k/4) table reads per k-mer
struct HashParams
{
  uint64_t seed[4];
  uint64_t roll_f[16];
  uint64_t roll_r[16];
  const uint64_t* init;  // [k][4][2]
  const uint64_t* init4; // [ceil(k/4)][256][2]
  uint32_t k;
};

// forward / reverse strand hashes of the k-mer whose bases are fetched by base(i), i = 0..k-1
template <typename BaseAt>
__device__ __forceinline__ void hash_init(const HashParams& hp, BaseAt&& base, uint64_t& f, uint64_t& r)
{
  f = 0;
  r = 0;
  const ulonglong2* tab = reinterpret_cast<const ulonglong2*>(hp.init);
  for (uint32_t i0 = 0; i0 < hp.k; i0 += 8) { // eight table reads in flight at a time (L1-resident: k x 64 bytes)
    ulonglong2 e[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
      e[q] = make_ulonglong2(0, 0);
      if (i0 + q < hp.k) e[q] = tab[(i0 + q) * 4u + base(i0 + q)];
    }
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
      f ^= e[q].x;
      r ^= e[q].y;
    }
  }
}

// The same for a k-mer of at most 64 bases held 2 bits per base in four words (base i = bits 2i.. of the 128-bit
// string): one table read per four bases.
__device__ __forceinline__ void hash_init_packed(const HashParams& hp, const uint32_t (&bases)[4], uint64_t& f, uint64_t& r)
{
  f = 0;
  r = 0;
  const ulonglong2* tab = reinterpret_cast<const ulonglong2*>(hp.init4);
  const uint32_t ng = (hp.k + 3u) >> 2;
#pragma unroll
  for (uint32_t g0 = 0; g0 < 16; g0 += 8) {
    if (g0 >= ng) break;
    ulonglong2 e[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
      const uint32_t g = g0 + q;
      e[q] = make_ulonglong2(0, 0);
      if (g < ng) e[q] = tab[g * 256u + ((bases[g >> 2] >> (8u * (g & 3u))) & 255u)];
    }
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
      f ^= e[q].x;
      r ^= e[q].y;
    }
  }
}

// Exact h % m for a runtime m: q = mulhi(h, floor(2^64/m)) is floor(h/m) or one less.
struct FastMod
{
  uint64_t m;
  uint64_t inv; // floor(2^64 / m)   (m >= 2)
  // The words the short forms below work on, as fields of their own: derived from m and inv inside the kernel, the optimiser
  // proves `(uint32_t)inv == inv` on the branch where inv fits 32 bits and goes back to the generic 64 x 64 product (six of the
  // quarter-rate multiplies instead of three).
  uint32_t inv32, m_lo, m_hi;
  uint32_t form; // 0 generic; 1: m > 2^32 (inv and the quotient fit 32 bits); 2: 2^32 < m < 2^38
  // The same with the form fixed at compile time (kernels instantiated per form: k_bin1).  Left as a run-time field, `form` stays a
  // branch per k-mer inside unrolled loops -- the compiler keeps all three forms there, materialises the uniform conditions through
  // VCC for every k-mer and runs out of scalar registers (a dozen v_readlane per k-mer in k_bin1's hot loop).
  template <int FORM> // (FORM < 0: the run-time field decides)
  __device__ __forceinline__ uint64_t mod(uint64_t h) const
  {
    if (FORM < 0) return (*this)(h);
    if (FORM != 0) {
      const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
      const uint64_t u = (uint64_t)hi * inv32 + __umulhi(lo, inv32);
      const uint32_t q = (uint32_t)(u >> 32);
      if (FORM == 2) {
        const uint64_t p = (uint64_t)q * m_lo;
        const uint32_t p_lo = (uint32_t)p;
        const uint32_t r_lo = lo - p_lo;
        uint32_t qm_hi;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(qm_hi) : "s"(m_hi & 0x3Fu), "v"(q & 0x7Fu));
        const uint32_t r_hi = (hi - (uint32_t)(p >> 32) - qm_hi - (lo < p_lo ? 1u : 0u)) & 0x7Fu;
        uint64_t r = ((uint64_t)r_hi << 32) | r_lo;
        if (r >= m) r -= m;
        return r;
      }
      const uint64_t qm = (uint64_t)q * m_lo + ((uint64_t)(q * m_hi) << 32);
      uint64_t r = h - qm;
      if (r >= m) r -= m;
      return r;
    }
    const uint64_t q = __umul64hi(h, inv);
    uint64_t r = h - q * m;
    if (r >= m) r -= m;
    return r;
  }
  __device__ __forceinline__ uint64_t operator()(uint64_t h) const
  {
    // Filters above 512 MiB (m > 2^32 bits, every genome beyond ~100 Mbp): inv and the quotient fit 32 bits, and the
    // 64 x 64 -> 128-bit product and the 64 x 64 multiply-back collapse to 32-bit multiplies, in kernels that are bound by
    // VALU issue (k_bin1, the sparse-filter select kernels).
    if (form != 0) { // (uniform: the same for every lane of a launch)
      const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
      const uint64_t u = (uint64_t)hi * inv32 + __umulhi(lo, inv32); // (h * inv) >> 32; its high word is floor(h * inv / 2^64)
      const uint32_t q = (uint32_t)(u >> 32);
      if (form == 2) {
        // filters below 32 GiB: the remainder r = h - q * m is below 2 m < 2^39, so its upper word is wanted modulo 128 only
        // and q * (m >> 32) shrinks to a 24-bit multiply of two small numbers -- three quarter-rate multiplies in all, each of
        // them the issue time of four plain instructions
        const uint64_t p = (uint64_t)q * m_lo;
        const uint32_t p_lo = (uint32_t)p;
        const uint32_t r_lo = lo - p_lo;
        uint32_t qm_hi; // (spelled out: left to itself the optimiser folds this product into a third 64-bit multiply-add)
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(qm_hi) : "s"(m_hi & 0x3Fu), "v"(q & 0x7Fu));
        const uint32_t r_hi = (hi - (uint32_t)(p >> 32) - qm_hi - (lo < p_lo ? 1u : 0u)) & 0x7Fu;
        uint64_t r = ((uint64_t)r_hi << 32) | r_lo;
        if (r >= m) r -= m;
        return r;
      }
      const uint64_t qm = (uint64_t)q * m_lo + ((uint64_t)(q * m_hi) << 32); // q * m mod 2^64 (q * m <= h)
      uint64_t r = h - qm;
      if (r >= m) r -= m;
      return r;
    }
    const uint64_t q = __umul64hi(h, inv);
    uint64_t r = h - q * m;
    if (r >= m) r -= m;
    return r;
  }
};

// slot of filter bit `idx` in the second folded table of a sparse filter (2^19 bits): a multiplicative hash of the index
// (filters hold fewer than 2^38 bits: idx >> 6 fits 32 bits)
__host__ __device__ __forceinline__ uint32_t fold2_slot(uint64_t idx)
{
  return ((uint32_t)(idx >> 6) * 0x9E3779B1u + (uint32_t)(idx & 63u) * 0x85EBCA6Bu) >> 13;
}

// Bloom bit `idx` lives in byte idx/8 at bit idx%8 (LSB first) == bit idx%32 of little-endian
// 32-bit word idx/32.
__device__ __forceinline__ bool bf_test(const uint32_t* __restrict__ words, uint64_t idx)
{
  return (words[idx >> 5] >> (idx & 31)) & 1u;
}

__device__ __forceinline__ void bf_set(uint32_t* words, uint64_t idx)
{
  atomicOr(&words[idx >> 5], 1u << (idx & 31));
}

// The same, looking first (a load past the L1, which the atomics of other workgroups do not update): the overflow of the
// partitioned build is made of the copies of a few k-mers -- satellite arrays -- and after the first copy of each the
// bit is there; millions of atomics on a few hundred words are what this avoids.
__device__ __forceinline__ void bf_set_unless_set(uint32_t* words, uint64_t idx)
{
  uint32_t* w = &words[idx >> 5];
  const uint32_t bit = 1u << (idx & 31);
  if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(w, bit);
}

} // namespace nts
