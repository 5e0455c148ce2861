/*
 * oracle/nts_oracle.c -- CPU restatement of ntSynt's sketch + common-Bloom-filter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ntsynt_amd/ (the product) may import, link or
 * call this file.  It is used by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg as the checker / CPU baseline, never as the thing shipped.
 *
 * What it restates (citations are paths under /root/reference, or SURVEY.md rows when the
 * arithmetic lives in a dependency whose source is absent from the reference tree):
 *
 *   - canonical ntHash2 of every k-mer (btllib NtHash; used
 *     at src/ntsynt_make_common_bf.cpp:147-151 and inside `indexlr`, smk:85)
 *   - Bloom filter sizing            src/ntsynt_make_common_bf.cpp:28-40   (row A1)
 *   - level-1 Bloom insert           src/ntsynt_make_common_bf.cpp:121-132 (row A2)
 *   - cascade contains->insert       src/ntsynt_make_common_bf.cpp:134-160 (row A3)
 *   - occupancy FPR                  src/ntsynt_make_common_bf.cpp:132,154,162 (row A4)
 *   - indexlr minimizer selection    SURVEY.md section 3.3 / rows B2, B3 (btllib Indexlr:
 *     ring buffer of w+1 hashed k-mers, `<=` rescan => rightmost minimum, UINT64_MAX
 *     sentinel for k-mers rejected by the `-s` filter-in Bloom filter)
 *
 * Third-party algorithm source: bcgsc/btllib "v1.6.2+" (README.md:111), not vendored in
 * the reference tree.  Parity pins: the 295,028 hash:pos:kmer known answers in
 * tests/expected_result/<genome>.k{20,24}.w1000.tsv pin h0/h1 (tests/test_oracle_golden.py).
 * Bloom bit addressing and the window/tie rule are restated from the published btllib
 * algorithm: "parity unpinned" for those.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- ntHash2 constants ------------------------------------------ */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
#define MULTISEED 0x90b45d39fb6da1faULL
#define MULTISHIFT 27

/* base code: 0..3 = A,C,G,T(U); 4 = anything else (k-mer skipped, as NtHash::roll does) */
static inline int
base_code(unsigned char c)
{
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
  }
}

static const uint64_t SEEDS[5] = { SEED_A, SEED_C, SEED_G, SEED_T, 0 };

/* split rotate left by one: bits 0..32 rotate as a 33-bit word, bits 33..63 as a 31-bit word */
static inline uint64_t
srol1(uint64_t x)
{
  uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
  return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}

static inline uint64_t
sror1(uint64_t x)
{
  uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
  return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}

static inline uint64_t
sroln(uint64_t x, unsigned d)
{
  for (unsigned i = 0; i < d; ++i) x = srol1(x);
  return x;
}

/* forward / reverse-complement hash of the k-mer starting at s (all k bases valid) */
static void
base_hashes(const unsigned char* s, unsigned k, uint64_t* fwd, uint64_t* rev)
{
  uint64_t f = 0, r = 0;
  for (unsigned i = 0; i < k; ++i) {
    f = srol1(f) ^ SEEDS[base_code(s[i])];
    r = srol1(r) ^ SEEDS[3 - base_code(s[k - 1 - i])];
  }
  *fwd = f;
  *rev = r;
}

static inline uint64_t
extend_h1(uint64_t h0, unsigned k)
{
  uint64_t t = h0 * (1ULL ^ ((uint64_t)k * MULTISEED));
  t ^= t >> MULTISHIFT;
  return t;
}

/*
 * Rolling iterator with NtHash::roll() semantics: yields, in order of position, every k-mer
 * that contains only A/C/G/T(U) (case-insensitive); k-mers with any other byte are skipped.
 */
typedef struct
{
  const unsigned char* seq;
  uint64_t len;
  unsigned k;
  uint64_t pos; /* position of the current k-mer */
  uint64_t fwd, rev;
  uint64_t out_rot[4]; /* srol^k(seed[c]) */
  int started;
} roller;

static void
roller_init(roller* R, const unsigned char* seq, uint64_t len, unsigned k)
{
  R->seq = seq;
  R->len = len;
  R->k = k;
  R->pos = 0;
  R->started = 0;
  for (int c = 0; c < 4; ++c) R->out_rot[c] = sroln(SEEDS[c], k);
}

/* find the first all-valid k-mer at or after `from`; returns 0 if none */
static int
roller_seek(roller* R, uint64_t from)
{
  const uint64_t len = R->len;
  const unsigned k = R->k;
  if (len < k) return 0;
  uint64_t p = from;
  while (p + k <= len) {
    /* scan the window right-to-left for the last invalid base */
    int64_t bad = -1;
    for (int64_t j = (int64_t)k - 1; j >= 0; --j) {
      if (base_code(R->seq[p + (uint64_t)j]) == 4) {
        bad = j;
        break;
      }
    }
    if (bad < 0) {
      base_hashes(R->seq + p, k, &R->fwd, &R->rev);
      R->pos = p;
      return 1;
    }
    p += (uint64_t)bad + 1;
  }
  return 0;
}

static int
roller_next(roller* R)
{
  if (!R->started) {
    R->started = 1;
    return roller_seek(R, 0);
  }
  const unsigned k = R->k;
  if (R->pos + k >= R->len) return 0;
  const int cin = base_code(R->seq[R->pos + k]);
  if (cin == 4) return roller_seek(R, R->pos + k + 1);
  const int cout = base_code(R->seq[R->pos]);
  R->fwd = srol1(R->fwd) ^ SEEDS[cin] ^ R->out_rot[cout];
  R->rev = sror1(R->rev ^ R->out_rot[3 - cin] ^ SEEDS[3 - cout]);
  R->pos += 1;
  return 1;
}

/* ---- exported: hashing ------------------------------------------------------------------ */

/* h0/h1 of one k-mer given as text; returns 0 if the k-mer has a non-ACGT byte */
int
nts_o_hash_kmer(const char* kmer, unsigned k, uint64_t* h0, uint64_t* h1)
{
  for (unsigned i = 0; i < k; ++i)
    if (base_code((unsigned char)kmer[i]) == 4) return 0;
  uint64_t f, r;
  base_hashes((const unsigned char*)kmer, k, &f, &r);
  *h0 = f + r;
  *h1 = extend_h1(*h0, k);
  return 1;
}

/* every valid k-mer of seq, in order: pos[], h0[] ; returns count (arrays sized >= len) */
uint64_t
nts_o_hash_all(const char* seq, uint64_t len, unsigned k, uint64_t* pos, uint64_t* h0)
{
  roller R;
  roller_init(&R, (const unsigned char*)seq, len, k);
  uint64_t n = 0;
  while (roller_next(&R)) {
    pos[n] = R.pos;
    h0[n] = R.fwd + R.rev;
    ++n;
  }
  return n;
}

uint64_t
nts_o_h1_from_h0(uint64_t h0, unsigned k)
{
  return extend_h1(h0, k);
}

/* ---- exported: Bloom filter --------------------------------------------------------------- */

/* src/ntsynt_make_common_bf.cpp:28-40 : size_bits = ceil(-genome/ln(1-fpr)); return bits/8 */
long long
nts_o_bf_approx_bytes(long long genome_size, double fpr)
{
  long long size_bits = (long long)ceil(((double)(-1 * genome_size)) / log(1 - fpr));
  return size_bits / 8;
}

/* btllib BloomFilter ctor: byte count rounded up to a multiple of sizeof(uint64_t) (u1) */
uint64_t
nts_o_bf_ctor_bytes(uint64_t bytes)
{
  return (uint64_t)(ceil((double)bytes / 8.0) * 8.0);
}

/* The same with the rounding as a switch (u1 is recalled, not verifiable in the reference tree):
 * 0 = up (above), 1 = down ((bytes / 8) * 8, an integer division inside the ceil), 2 = none. */
uint64_t
nts_o_bf_ctor_bytes_mode(uint64_t bytes, int rounding)
{
  if (rounding == 1) return bytes / 8 * 8;
  if (rounding == 2) return bytes;
  return nts_o_bf_ctor_bytes(bytes);
}

static inline void
bf_set(uint8_t* bf, uint64_t bits, uint64_t h)
{
  const uint64_t idx = h % bits;
  const uint8_t mask = (uint8_t)(1u << (idx & 7));
#ifdef _OPENMP
  __atomic_fetch_or(&bf[idx >> 3], mask, __ATOMIC_RELAXED);
#else
  bf[idx >> 3] |= mask;
#endif
}

static inline int
bf_get(const uint8_t* bf, uint64_t bits, uint64_t h)
{
  const uint64_t idx = h % bits;
  return (bf[idx >> 3] >> (idx & 7)) & 1;
}

/* src/ntsynt_make_common_bf.cpp:128-131 : bf->insert(record.seq) for one record */
void
nts_o_bf_insert_seq(uint8_t* bf, uint64_t bf_bytes, const char* seq, uint64_t len, unsigned k)
{
  roller R;
  roller_init(&R, (const unsigned char*)seq, len, k);
  const uint64_t bits = bf_bytes * 8;
  while (roller_next(&R)) bf_set(bf, bits, R.fwd + R.rev);
}

/* src/ntsynt_make_common_bf.cpp:146-152 : if prev.contains(h) next.insert(h), one record */
void
nts_o_bf_cascade_seq(const uint8_t* prev,
                     uint8_t* next,
                     uint64_t bf_bytes,
                     const char* seq,
                     uint64_t len,
                     unsigned k)
{
  roller R;
  roller_init(&R, (const unsigned char*)seq, len, k);
  const uint64_t bits = bf_bytes * 8;
  while (roller_next(&R)) {
    const uint64_t h = R.fwd + R.rev;
    if (bf_get(prev, bits, h)) bf_set(next, bits, h);
  }
}

/* bin/ntsynt_make_repeat_bfs.py:56-67 for one record: if genome_bf.contains(h): rep_bf.insert(h) else genome_bf.insert(h) */
void
nts_o_bf_repeats_seq(uint8_t* genome_bf, uint8_t* rep_bf, uint64_t bf_bytes, const char* seq, uint64_t len, unsigned k)
{
  roller R;
  roller_init(&R, (const unsigned char*)seq, len, k);
  const uint64_t bits = bf_bytes * 8;
  while (roller_next(&R)) {
    const uint64_t h = R.fwd + R.rev;
    if (bf_get(genome_bf, bits, h))
      bf_set(rep_bf, bits, h);
    else
      bf_set(genome_bf, bits, h);
  }
}

/*
 * Many records at once, parallel over records like the reference's `#pragma omp parallel`
 * over SeqReader records (src/ntsynt_make_common_bf.cpp:128,145).  prev == NULL => plain insert.
 */
void
nts_o_bf_records(const uint8_t* prev,
                 uint8_t* next,
                 uint64_t bf_bytes,
                 const char* seq,
                 const uint64_t* rec_off,
                 const uint64_t* rec_len,
                 uint32_t n_rec,
                 unsigned k,
                 int threads)
{
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (uint32_t r = 0; r < n_rec; ++r) {
    if (prev)
      nts_o_bf_cascade_seq(prev, next, bf_bytes, seq + rec_off[r], rec_len[r], k);
    else
      nts_o_bf_insert_seq(next, bf_bytes, seq + rec_off[r], rec_len[r], k);
  }
}

uint64_t
nts_o_bf_popcount(const uint8_t* bf, uint64_t bf_bytes)
{
  uint64_t n = 0;
  for (uint64_t i = 0; i < bf_bytes; ++i) n += (uint64_t)__builtin_popcount(bf[i]);
  return n;
}

int
nts_o_bf_contains(const uint8_t* bf, uint64_t bf_bytes, uint64_t h0)
{
  return bf_get(bf, bf_bytes * 8, h0);
}

/* ---- exported: indexlr minimizer selection -------------------------------------------- */

typedef struct
{
  uint64_t min_hash, out_hash, pos;
} hashed_kmer;

/*
 * One record.  Restates btllib Indexlr::minimize + calc_minimizer:
 *   - returns nothing if k > len or w > len-k+1
 *   - idx counts VALID k-mers only; ring buffer of w+1 slots
 *   - with a filter-in Bloom filter, k-mers whose h0 is absent get min_hash = UINT64_MAX
 *   - when the current minimum slid out of the window: rescan left-to-right with `<=`
 *     (rightmost minimum); else the newest k-mer replaces it if `<=`
 *   - emit when pos > last emitted pos and min_hash != UINT64_MAX
 * Output: out_h1[] (the printed hash, hashes()[1]), out_pos[]; returns the count.
 */
/*   - with a filter-out Bloom filter as well (indexlr -r, the experimental repeat filter: ntsynt_run_pipeline.smk:83),
 *     a k-mer present in it is rejected too; with only a filter-out filter, presence alone rejects
 *     (btllib Indexlr::filter_hashed_kmer, recalled: unpinned) */
/* The window decision of Indexlr::minimize for the k-mer just written to buf[idx % ring] (idx-th VALID k-mer of the
 * record): returns the k-mer to emit, or NULL.  `strict` is 0 in the restatement (`<=`: the rightmost of equal hashes
 * wins); 1 (`<`) exists only so that a test can show the reference's own output rules it out. */
static const hashed_kmer*
window_decide(const hashed_kmer* buf, uint64_t ring, uint64_t idx, unsigned w, const hashed_kmer** cur_io, int64_t* min_pos_prev, int strict)
{
  if (idx + 1 < w) return NULL;
  const hashed_kmer* cur = *cur_io;
  const uint64_t left = idx + 1 - w, right = idx + 1;
  const hashed_kmer* min_left = &buf[left % ring];
  const hashed_kmer* min_right = &buf[(right - 1) % ring];
  if (cur == NULL || cur->pos < min_left->pos) {
    cur = min_left;
    for (uint64_t i = left; i < right; ++i) {
      const hashed_kmer* mi = &buf[i % ring];
      if (strict ? mi->min_hash < cur->min_hash : mi->min_hash <= cur->min_hash) cur = mi;
    }
  } else if (strict ? min_right->min_hash < cur->min_hash : min_right->min_hash <= cur->min_hash) {
    cur = min_right;
  }
  *cur_io = cur;
  if ((int64_t)cur->pos > *min_pos_prev && cur->min_hash != UINT64_MAX) {
    *min_pos_prev = (int64_t)cur->pos;
    return cur;
  }
  return NULL;
}

uint64_t
nts_o_minimize2(const char* seq,
                uint64_t len,
                unsigned k,
                unsigned w,
                const uint8_t* bf,
                uint64_t bf_bytes,
                const uint8_t* bf_out,
                uint64_t bf_out_bytes,
                uint64_t* out_h1,
                uint64_t* out_pos,
                uint64_t cap)
{
  if ((uint64_t)k > len || (uint64_t)w > len - k + 1) return 0;
  const uint64_t ring = (uint64_t)w + 1;
  hashed_kmer* buf = (hashed_kmer*)malloc(sizeof(hashed_kmer) * ring);
  const hashed_kmer* cur = NULL;
  int64_t min_pos_prev = -1;
  uint64_t n_out = 0;
  const uint64_t bits = bf_bytes * 8;
  roller R;
  roller_init(&R, (const unsigned char*)seq, len, k);
  uint64_t idx = 0;
  for (; roller_next(&R); ++idx) {
    hashed_kmer* hk = &buf[idx % ring];
    const uint64_t h0 = R.fwd + R.rev;
    hk->min_hash = h0;
    hk->out_hash = extend_h1(h0, k);
    hk->pos = R.pos;
    if (bf && !bf_get(bf, bits, h0)) hk->min_hash = UINT64_MAX;
    if (bf_out && bf_get(bf_out, bf_out_bytes * 8, h0)) hk->min_hash = UINT64_MAX;
    const hashed_kmer* emit = window_decide(buf, ring, idx, w, &cur, &min_pos_prev, 0);
    if (emit) {
      if (n_out < cap) {
        out_h1[n_out] = emit->out_hash;
        out_pos[n_out] = emit->pos;
      }
      ++n_out;
    }
  }
  free(buf);
  return n_out;
}

/* The same decision logic over a stream of comparison keys instead of a sequence: keys[i] is the key of the k-mer at
 * position i (UINT64_MAX = no accepted k-mer there), every position counts as a valid k-mer.  Used to pin the window rule
 * to the reference's own output (tests/test_oracle_golden.py): fed only the minimizers a reference TSV lists -- h0 recovered
 * from the printed hash -- it must emit exactly those again.  Returns the count, positions into out_pos. */
uint64_t
nts_o_minimize_keys(const uint64_t* keys, uint64_t n, unsigned w, int strict, uint64_t* out_pos, uint64_t cap)
{
  if ((uint64_t)w > n || w == 0) return 0;
  const uint64_t ring = (uint64_t)w + 1;
  hashed_kmer* buf = (hashed_kmer*)malloc(sizeof(hashed_kmer) * ring);
  const hashed_kmer* cur = NULL;
  int64_t min_pos_prev = -1;
  uint64_t n_out = 0;
  for (uint64_t idx = 0; idx < n; ++idx) {
    hashed_kmer* hk = &buf[idx % ring];
    hk->min_hash = keys[idx];
    hk->out_hash = keys[idx];
    hk->pos = idx;
    const hashed_kmer* emit = window_decide(buf, ring, idx, w, &cur, &min_pos_prev, strict);
    if (emit) {
      if (n_out < cap) out_pos[n_out] = emit->pos;
      ++n_out;
    }
  }
  free(buf);
  return n_out;
}

uint64_t
nts_o_minimize(const char* seq,
               uint64_t len,
               unsigned k,
               unsigned w,
               const uint8_t* bf,
               uint64_t bf_bytes,
               uint64_t* out_h1,
               uint64_t* out_pos,
               uint64_t cap)
{
  return nts_o_minimize2(seq, len, k, w, bf, bf_bytes, NULL, 0, out_h1, out_pos, cap);
}

/*
 * All records of a genome, `threads` records in flight (indexlr -t N).  Outputs are written
 * per record into [rec_cap_off[r], rec_cap_off[r+1]) of out_h1/out_pos; counts into rec_cnt[r].
 */
void
nts_o_minimize_records2(const char* seq,
                        const uint64_t* rec_off,
                        const uint64_t* rec_len,
                        uint32_t n_rec,
                        unsigned k,
                        unsigned w,
                        const uint8_t* bf,
                        uint64_t bf_bytes,
                        const uint8_t* bf_out,
                        uint64_t bf_out_bytes,
                        uint64_t* out_h1,
                       uint64_t* out_pos,
                       const uint64_t* rec_cap_off,
                       uint64_t* rec_cnt,
                       int threads)
{
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (uint32_t r = 0; r < n_rec; ++r) {
    const uint64_t cap = rec_cap_off[r + 1] - rec_cap_off[r];
    rec_cnt[r] = nts_o_minimize2(seq + rec_off[r],
                                 rec_len[r],
                                 k,
                                 w,
                                 bf,
                                 bf_bytes,
                                 bf_out,
                                 bf_out_bytes,
                                 out_h1 + rec_cap_off[r],
                                 out_pos + rec_cap_off[r],
                                 cap);
  }
}

void
nts_o_minimize_records(const char* seq,
                       const uint64_t* rec_off,
                       const uint64_t* rec_len,
                       uint32_t n_rec,
                       unsigned k,
                       unsigned w,
                       const uint8_t* bf,
                       uint64_t bf_bytes,
                       uint64_t* out_h1,
                       uint64_t* out_pos,
                       const uint64_t* rec_cap_off,
                       uint64_t* rec_cnt,
                       int threads)
{
  nts_o_minimize_records2(seq, rec_off, rec_len, n_rec, k, w, bf, bf_bytes, NULL, 0, out_h1, out_pos, rec_cap_off, rec_cnt, threads);
}
