/*
 * ntsynt_hip.h -- C ABI of libntsynt_hip.so: the MI355X (gfx950) implementation of ntSynt's
 * hot path (canonical-ntHash minimizer sketch + common-k-mer Bloom filter + minimizer-graph
 * chaining).  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * The reference has no in-process FFI for this path: its boundary is CLI + files
 * (SURVEY.md 8(b)).  Each entry point below names the reference interface it replaces
 * (paths relative to the reference tree); INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions: every call returns 0 on success or a negative code (NTS_E*); the message is
 * retrievable with nts_last_error(ctx) (ctx == NULL: the message of a failed nts_init).
 * Handles are opaque.  One nts_ctx per GPU (owns one HIP stream); a ctx is not thread-safe.
 * Buffers returned through `T**` out-parameters are allocated by the library on the host and
 * released with nts_free().
 */
#ifndef NTSYNT_HIP_H
#define NTSYNT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTS_OK 0
#define NTS_EINVAL (-22)
#define NTS_ENOMEM (-12)
#define NTS_EHIP (-5)
#define NTS_ERANGE (-34)
#define NTS_EFORMAT (-74) /* input file is not FASTA (e.g. FASTQ) */
#define NTS_ECOMM (-70)   /* RCCL unavailable or a collective failed */

typedef struct nts_ctx nts_ctx;
typedef struct nts_genome nts_genome; /* one FASTA resident in HBM */
typedef struct nts_bf nts_bf;         /* Bloom bit array resident in HBM */
typedef struct nts_mx nts_mx;         /* minimizer list resident in HBM */
typedef struct nts_comm nts_comm;     /* RCCL communicator of the multi-GPU path (one rank per GPU) */

/* hard-mask interval [start, end) in record coordinates (bedtools maskfasta semantics) */
typedef struct
{
  uint32_t rec;
  uint64_t start;
  uint64_t end;
} nts_interval;

/* ---- context ------------------------------------------------------------------------------- */
int nts_init(int device, nts_ctx** out);
void nts_destroy(nts_ctx* ctx);
const char* nts_last_error(nts_ctx* ctx);
int nts_sync(nts_ctx* ctx);
/* HIP stream of the context (hipStream_t as void*), for callers that enqueue their own work */
void* nts_stream(nts_ctx* ctx);

/* Per-kernel timing with HIP events on the context's stream (bench.py's roofline leg).
 * nts_profile(ctx,1) resets and enables; nts_timing() reports total ms and launch count of the
 * kernel called `name` since then ("hash_probe", "window_min", "bf_insert", ...).  nts_profile(ctx,2) times only
 * the dominant kernels (hash_select, hash_probe, bf_insert): every event pair costs ~10 us of bubble in the stream,
 * which a short call feels.  0 disables. */
int nts_profile(nts_ctx* ctx, int enable);
int nts_timing(nts_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);

/* Device memory of this process's library allocations (all contexts): bytes live now and their high-water mark since the
 * last reset, plus what the device reports as in use / in total (hipMemGetInfo on the context's device: includes other
 * users of the GPU, e.g. a caller's torch tensors).  Replaces the peak-memory column of the reference's `--benchmark`
 * wrappers (`/usr/bin/time -v` / memusg per rule, bin/ntsynt_run_pipeline.smk:26-35; README.md:156-158 quotes wall clock and
 * peak memory per run).  Any out-pointer may be NULL; ctx may be NULL (device figures are then 0).
 * nts_mem_reset_peak: the mark restarts from the bytes live now. */
int nts_mem_stats(nts_ctx* ctx, uint64_t* live_bytes, uint64_t* peak_bytes, uint64_t* device_used_bytes, uint64_t* device_total_bytes);
void nts_mem_reset_peak(void);
/* hipMalloc / hipFree calls the library has made in this process and the host time they took, in ms: what a call that meets a
 * context without workspaces (the first sketch of a run: rule indexlr runs once per genome, bin/ntsynt_run_pipeline.smk:74-85) spends
 * allocating -- bench.py's `cold` leg takes the difference around a call. */
int nts_alloc_stats(uint64_t* calls, double* ms);
/* Device blocks the library frees are kept (up to 96 GB in all) and handed out again, whole or in pieces: on some boxes a hipMalloc of
 * a few hundred MB takes 20-100 ms, memory a process has just given back costs the next allocation a wait, and every genome, filter
 * and context of a run allocates and frees.  Requests below 64 KB live in 8 MB slabs of their own (a long-lived small workspace never
 * holds a large kept allocation).  live / peak of nts_mem_stats count blocks in use, not cached ones (device_used_bytes sees both).
 * nts_mem_trim gives every cached block back to the driver (bytes released); an allocation that fails does so by itself and tries
 * again; so does nts_destroy of the process's last context.  nts_mem_cache_stats: bytes cached now, allocations served from the cache.
 * nts_mem_reserve: ONE driver allocation of `bytes` on `device` (less when the device has less to give; *reserved_bytes says), kept
 * for the allocations to come -- a run that knows its plan (file sizes -> filter, build workspaces, resident genomes) asks the driver
 * once, before its first file is read, and never again; may be called from a thread of its own.  Replaces nothing in the reference
 * (it allocates two filters once, src/ntsynt_make_common_bf.cpp:121-137): it is what makes "allocated once" true of the GPU run.
 * nts_mem_events: out[0..7] = hipMalloc + hipFree calls, ns spent in them, allocations served from the cache, allocations tried again
 * after the cache was emptied, bytes taken from the driver, bytes given back, ns spent waiting for the device before a freed block was
 * kept, nts_mem_reserve calls that got memory (process-wide, monotonic: take differences around a stage). */
uint64_t nts_mem_trim(void);
int nts_mem_cache_stats(uint64_t* cached_bytes, uint64_t* hits);
int nts_mem_reserve(int device, uint64_t bytes, uint64_t* reserved_bytes);
int nts_mem_events(uint64_t out[8]);

/* ---- A1: Bloom filter sizing ----------------------------------------------------------------
 * replaces approximate_bf_size(), src/ntsynt_make_common_bf.cpp:28-40, and the byte rounding of
 * the btllib::KmerBloomFilter constructor used at :122-123.  approx_bytes = ceil(-n/ln(1-fpr))/8
 * (truncating), ctor_bytes = approx_bytes rounded up to a multiple of 8. */
int nts_bf_size_bytes(uint64_t genome_bp, double fpr, uint64_t* approx_bytes, uint64_t* ctor_bytes);
/* The constructor's rounding is recalled from btllib's source, which is not in the reference tree,
 * and it decides the modulus of every bit index: a maintainer holding btllib can settle it by flipping this switch.
 *   NTS_BF_ROUND_UP   ceil(double(bytes) / 8) * 8   -- the default, what nts_bf_size_bytes() returns
 *   NTS_BF_ROUND_DOWN (bytes / 8) * 8               -- an integer division inside the ceil
 *   NTS_BF_ROUND_NONE bytes as approximate_bf_size() returned them
 * Every filter entry point accepts any positive byte count. */
#define NTS_BF_ROUND_UP 0
#define NTS_BF_ROUND_DOWN 1
#define NTS_BF_ROUND_NONE 2
int nts_bf_size_bytes_ex(uint64_t genome_bp, double fpr, int rounding, uint64_t* approx_bytes, uint64_t* ctor_bytes);

/* ---- genome ---------------------------------------------------------------------------------
 * replaces btllib::SeqReader(path, LONG_MODE) record streaming (src/...cpp:32-36,125-131).
 * `seq` = the records' bases concatenated (ASCII, any case, no separators); record r occupies
 * [rec_off[r], rec_off[r]+rec_len[r]).  Copied to HBM once. */
int nts_genome_upload(nts_ctx* ctx,
                      const uint8_t* seq,
                      uint64_t n,
                      const uint64_t* rec_off,
                      const uint64_t* rec_len,
                      uint32_t n_rec,
                      nts_genome** out);
/* A batch of uploaded genomes as one device genome (device-to-device copy): the records of part 0, then those of
 * part 1, ...; record ids of part p start at the number of records of the parts before it.  Sketching the batch is
 * the reference's "indexlr per assembly" (bin/ntsynt_run_pipeline.smk:74-85) for all assemblies with one sequence of launches --
 * records never share k-mers, so the minimizers of a record are those of the same record sketched alone. */
int nts_genome_concat(nts_ctx* ctx, uint32_t n_parts, const nts_genome* const* parts, nts_genome** out);
/* Records [rec0, rec1) of a resident genome as a resident genome of their own (device-to-device copy): the shard one rank of a
 * genome's group works on when genomes < GPUs.  Windows never cross records (Indexlr minimizes per record), so the shards'
 * minimizer lists concatenate to the genome's (nts_mx_concat) and their filters OR to the genome's (nts_bf_allreduce_groups). */
int nts_genome_slice(nts_ctx* ctx, const nts_genome* g, uint32_t rec0, uint32_t rec1, nts_genome** out);
void nts_genome_free(nts_ctx* ctx, nts_genome* g);
/* Bench / scale-test utilities (no counterpart in the reference): a synthetic genome generated directly in
 * HBM -- `n_contigs` equal records of i.i.d. bases drawn from `seed_ancestor`, with independent substitutions
 * at `substitution_rate` keyed by `seed_genome` (genomes sharing seed_ancestor are relatives) -- and the
 * read-back of a slice of any resident genome as upper-case ASCII (concatenated-sequence coordinates). */
int nts_genome_synth(nts_ctx* ctx, uint64_t total_bp, uint32_t n_contigs, uint64_t seed_ancestor, uint64_t seed_genome,
                     double substitution_rate, nts_genome** out);
/* The same family with structural events (inversions, translocations, indels, N runs, so that the block rules
 * of bin/ntsynt_synteny.py:391-409 (indel split), :312-340 (erosion) and :434-472 (collinear merge) have work at full scale):
 * the genome is given as a tiling of pieces in ascending `dst` order -- each a stretch [src, src + len) of the ancestor
 * (coordinates of nts_genome_synth's concatenated ancestor), forwards or reverse-complemented, `len` bases of sequence of the
 * genome's own (an insertion; `src` = offset in that stream), or a run of N -- cut into records of rec_len[r] bases.
 * The plan is the caller's (ntsynt_amd/synth.py structural_plan); substitutions as in nts_genome_synth. */
#define NTS_SYNTH_REVCOMP 1u
#define NTS_SYNTH_NOVEL 2u
#define NTS_SYNTH_NRUN 4u
typedef struct
{
  uint64_t dst, len, src;
  uint32_t flags, reserved;
} nts_synth_piece;
int nts_genome_synth_plan(nts_ctx* ctx, uint32_t n_rec, const uint64_t* rec_len, uint32_t n_pieces, const nts_synth_piece* pieces,
                          uint64_t seed_ancestor, uint64_t seed_genome, double substitution_rate, nts_genome** out);
/* The assembly-like family (BASELINE config 5 stands on real mammalian assemblies, /root/reference README.md:157; they are not in
 * the container): the same tiling, over an ancestor that is not i.i.d.  `rep` describes interspersed repeat families as a function
 * of the ancestor coordinate s -- every cell of 2^cell_log2 bases holds, with probability prob_256/256, one copy of one of
 * `families` consensus elements (SINE-like: sine_len bases; LINE-like: the last line_min_len..line_len bases of a line_len
 * consensus, i.e. 5'-truncated), forwards or reverse-complemented, each copy diverged from its consensus by one of sixteen levels
 * between div_min_1024/1024 and div_max_1024/1024 (young copies share k-mers: within-assembly duplicate minimizers, Bloom buckets
 * that overflow); SINE copies overwrite LINE copies.  Pieces flagged NTS_SYNTH_TANDEM are satellite arrays: `reserved` = the array
 * family, base = that family's sat_unit-base unit at (source coordinate mod sat_unit), diverged at sat_div_1024/1024 per base
 * (keyed by the source coordinate, so that the copies of an array in the genomes of a family agree).  rep == NULL: an i.i.d.
 * ancestor, tandem pieces rejected.  ntsynt_amd/synth.py (realistic_plan, plan_bases) holds the plan and the numpy statement of
 * the same generator that the tests compare with. */
#define NTS_SYNTH_TANDEM 8u
typedef struct
{
  uint32_t sine_cell_log2, sine_len, sine_prob_256, sine_families;
  uint32_t line_cell_log2, line_len, line_min_len, line_prob_256, line_families;
  uint32_t div_min_1024, div_max_1024;
  uint32_t sat_unit, sat_div_1024;
} nts_synth_repeats;
int nts_genome_synth_plan_ex(nts_ctx* ctx, uint32_t n_rec, const uint64_t* rec_len, uint32_t n_pieces, const nts_synth_piece* pieces,
                             uint64_t seed_ancestor, uint64_t seed_genome, double substitution_rate, const nts_synth_repeats* rep,
                             nts_genome** out);
int nts_genome_download(nts_ctx* ctx, const nts_genome* g, uint64_t offset, uint64_t len, uint8_t* ascii);
/* total bases (sum of record lengths, what approximate_bf_size() counts) */
uint64_t nts_genome_bases(const nts_genome* g);
/* number of k-mers made only of A/C/G/T(U), i.e. the k-mers NtHash::roll() visits */
int nts_genome_valid_kmers(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint64_t* n_valid);

/* ---- A2-A4: Bloom filter ----------------------------------------------------------------------
 * nts_bf_create     : btllib::KmerBloomFilter(bytes, 1, k), cpp:122-123,136-137 (bytes = ctor_bytes)
 * nts_bf_insert     : bf->insert(record.seq) over all records, cpp:128-131  (level-1 filter)
 * nts_bf_cascade    : `if (bf->contains(h)) new_bf->insert(h)`, cpp:145-153 (literal cascade level)
 * nts_bf_and        : acc &= other -- with one hash function the cascade equals the AND of the
 *                     per-genome filters; this is the form the multi-GPU path reduces
 * nts_bf_insert_and : one whole cascade level, cpp:134-160, as acc &= (filter of genome g) inside the partitioned build's
 *                     last pass -- no second filter, no clearing, no AND pass; bit for bit what nts_bf_insert into a
 *                     cleared filter followed by nts_bf_and gives (and what nts_bf_cascade gives)
 * nts_bf_popcount   : numerator of bf->get_fpr(), cpp:132,154,162
 * nts_bf_download / nts_bf_upload : raw bit array for bf->save()/load (cpp:164; smk:76).  The download sees
 *                     everything queued before the call and runs on the context's copy stream: it is the one
 *                     call that may be issued from a second host thread while the first keeps sketching. */
int nts_bf_create(nts_ctx* ctx, uint64_t bytes, nts_bf** out);
void nts_bf_free(nts_ctx* ctx, nts_bf* bf);
uint64_t nts_bf_bytes(const nts_bf* bf);
void* nts_bf_device_ptr(nts_bf* bf);
int nts_bf_clear(nts_ctx* ctx, nts_bf* bf);
int nts_bf_insert_and(nts_ctx* ctx, nts_bf* acc, const nts_genome* g, uint32_t k);
int nts_bf_insert(nts_ctx* ctx, nts_bf* bf, const nts_genome* g, uint32_t k);
/* How nts_bf_insert sets the bits (same filter either way): 0 = auto (large genomes: hash, partition the bit
 * indices by filter segment in two streaming passes, set them in LDS bitmaps and OR whole segments into the
 * filter; small ones: one atomic OR per k-mer), 1 = always one atomic OR per k-mer, 2 = partitioned build whenever
 * the filter layout allows it (tests). */
int nts_bf_build_mode(nts_ctx* ctx, int mode);
int nts_bf_cascade(nts_ctx* ctx, const nts_bf* prev, nts_bf* next, const nts_genome* g, uint32_t k);
/* the per-genome loop of the experimental repeat filter (bin/ntsynt_make_repeat_bfs.py:56-67; rule make_repeat_bf,
 * smk:65-72): `if genome_bf.contains(h): repeat_bf.insert(h) else: genome_bf.insert(h)` over every k-mer of g */
int nts_bf_insert_repeats(nts_ctx* ctx, nts_bf* genome_bf, nts_bf* repeat_bf, const nts_genome* g, uint32_t k);
int nts_bf_and(nts_ctx* ctx, nts_bf* acc, const nts_bf* other);
int nts_bf_popcount(nts_ctx* ctx, const nts_bf* bf, uint64_t* bits_set);
int nts_bf_download(nts_ctx* ctx, const nts_bf* bf, uint8_t* host, uint64_t bytes);
int nts_bf_upload(nts_ctx* ctx, nts_bf* bf, const uint8_t* host, uint64_t bytes);
/* bf->save(path) (cpp:164) straight out of HBM: `header` (the caller's btllib-style text header), then the bit array, written
 * by n_threads host threads (0 = default) through pinned staging and positional writes -- no host copy of the filter.
 * Like nts_bf_download it sees everything queued before the call and may run on a second host thread. */
int nts_bf_save(nts_ctx* ctx, const nts_bf* bf, const char* path, const void* header, uint64_t header_bytes, uint32_t n_threads);
/* Microbenchmark: n_probes pseudo-random single-bit reads of the filter with the access shape of the sketch's
 * probe batches and no hashing -- the empirical ceiling for sector-granular random reads on this GPU. */
int nts_bench_random_probe(nts_ctx* ctx, const nts_bf* bf, uint64_t n_probes, uint32_t repeats, double* avg_ms, uint64_t* hits);
/* Microbenchmark: issue rate of the integer VALU instructions ntHash is made of (the roof of the VALU-bound kernels
 * k_hash_select and k_bin1).  kind: 0 v_xor_b32, 1 v_alignbit_b32, 2 v_lshl_add_u64, 3 v_lshlrev_b64, 4 v_mul_lo_u32,
 * 5 v_mad_u64_u32, 6 v_add_co_u32 + v_addc_co_u32 (a 64-bit add as two instructions), 7 a dependent chain of v_xor_b32
 * (latency).  Every CU runs `waves_per_simd` (1..8) waves per SIMD, each wave `instr_per_wave` instructions in eight
 * independent chains; cycles_per_instr = shader cycles (s_memtime) a SIMD needs per wave-instruction; wall_ms = the
 * launch, from which wave-instructions per second per CU = 4 * waves_per_simd * instr_per_wave / wall follow. */
int nts_bench_valu(nts_ctx* ctx, int kind, uint32_t waves_per_simd, uint32_t iters, double* wall_ms, double* cycles_per_instr,
                   double* instr_per_wave);
/* Diagnostic: out[i] = h[i] mod bits computed by the device code every probe and insert goes through (nts::FastMod: a
 * generic form and two short ones for bits > 2^32 and 2^32 < bits < 2^38; `form` = -1 takes the one the library would, 0..2
 * caps it) -- btllib's `hashes[0] % array_bits`.  h and out are host arrays of n words. */
int nts_mod_indices(nts_ctx* ctx, uint64_t bits, int form, const uint64_t* h, uint64_t n, uint64_t* out);
/* Non-owning filter over caller-provided HBM (e.g. the buffer a collective runs on): `device_ptr` must be
 * 16-byte aligned and hold `bytes` rounded up to a multiple of 16, the tail zeroed.  nts_bf_free() on a
 * wrapped filter releases only the handle. */
int nts_bf_wrap(nts_ctx* ctx, void* device_ptr, uint64_t bytes, nts_bf** out);
/* acc &= other on raw device buffers (bytes a multiple of 16): the local reduction step of the
 * bitwise-AND all-reduce of per-genome filters across GPUs (SURVEY.md 8(e) exchange 1; RCCL has no
 * bitwise reduction). */
int nts_and_raw(nts_ctx* ctx, void* acc_dev, const void* other_dev, uint64_t bytes);

/* ---- multi-GPU exchange steps: one process per GPU, genomes partitioned over the ranks -------------
 * The reference has no counterpart (it is a single-node Snakemake workflow); these calls are what a sharded
 * make_common_bf (src/ntsynt_make_common_bf.cpp:134-160) and the hand-over of the per-genome indexlr outputs to
 * ntsynt_run.py (bin/ntsynt_run_pipeline.smk:87-103) become across GPUs.
 *
 * nts_comm_unique_id : 128 opaque bytes (ncclGetUniqueId) made on one rank and handed to all by the launcher
 *                      (torch.distributed store, MPI, a file)
 * nts_comm_init      : ncclCommInitRank on the context's GPU
 * nts_comm_wrap      : adopt an existing ncclComm_t (not destroyed by nts_comm_destroy)
 * nts_comm_library   : which library serves the collectives: "librccl" (the process's copy or the system's), the path
 *                      given in NTS_RCCL_LIB (a site's own build; the test suite's stand-in for ranks that share one
 *                      GPU), or "" when none could be loaded.  (The environment variables the product build reads are four:
 *                      NTS_RCCL_LIB, NTS_COMM_PIECE, NTS_IO_THREADS, NTS_HOST_THREADS -- csrc/nts_knobs.h; NTS_COMM_PIECE and
 *                      NTS_IO_THREADS are read ONCE per context, by nts_init: set them before the context is created.  The
 *                      experiment switches exist only in libntsynt_hip_exp.so.)
 * nts_bf_create_sharded : a filter whose allocation is `world` chunks of a multiple of 16 bytes -- the layout the
 *                      all-reduce exchanges; otherwise identical to nts_bf_create
 * nts_bf_fill_ones   : identity of AND, for a rank that owns no genome
 * nts_bf_allreduce_and : exchange 1 -- in place, every rank ends with the bitwise AND of all ranks' filters (= the
 *                      cascade's result).  RCCL has no bitwise reduction: direct reduce-scatter with
 *                      ncclSend/ncclRecv over all xGMI links at once through a (world-1) x 256 MiB scratch, a local AND
 *                      kernel, ncclAllGather in place.  comm == NULL or world 1: no-op.
 * nts_mx_allgather   : exchange 2 -- the ranks' minimizer lists (n_local handles, genome numbers in local_ids; a rank
 *                      holds at most ceil(n_total / world)) gathered on every rank: out[g] = list of genome g,
 *                      resident in HBM, g in [0, n_total); released with nts_mx_free.  Two collectives per call. */
int nts_comm_unique_id(uint8_t* id128);
int nts_comm_init(nts_ctx* ctx, const uint8_t* id128, int world, int rank, nts_comm** out);
int nts_comm_wrap(nts_ctx* ctx, void* rccl_comm, nts_comm** out);
void nts_comm_destroy(nts_comm* comm);
int nts_comm_world(const nts_comm* comm);
int nts_comm_rank(const nts_comm* comm);
void* nts_comm_handle(const nts_comm* comm);
const char* nts_comm_library(void);
int nts_bf_create_sharded(nts_ctx* ctx, uint64_t bytes, int world, nts_bf** out);
int nts_bf_fill_ones(nts_ctx* ctx, nts_bf* bf);
int nts_bf_allreduce_and(nts_ctx* ctx, nts_bf* bf, nts_comm* comm);
/* Exchange 1 when there are fewer genomes than GPUs (last paragraph; what the reference parallelises over is
 * records, src/ntsynt_make_common_bf.cpp:128-131,145-153): the ranks of group g hold the filters of the shards (record ranges,
 * nts_genome_slice) of genome g.  In place, every rank ends with AND over groups of (OR over the group's ranks); group_of[r] in
 * [0, n_groups) for each of the communicator's ranks, every group with at least one rank.  group_of == NULL: nts_bf_allreduce_and.
 * nts_comm_last_sparse: 1 if the context's last all-reduce gathered the set bits' indices instead of the reduced chunks (a
 * reduced filter that is all but empty -- BASELINE config 4 -- moves megabytes instead of the filter's size). */
int nts_bf_allreduce_groups(nts_ctx* ctx, nts_bf* bf, nts_comm* comm, const int32_t* group_of, uint32_t n_groups);
int nts_comm_last_sparse(const nts_ctx* ctx);
/* The context's last exchange 2 (nts_mx_allgather[_ex]): the lists travel packed -- h1 as it is, the position in 32 bits when the
 * list's largest fits, the record as one start index per RECORD (a list is in record order): 12 bytes per minimizer and a few KB where
 * the lists themselves hold 20.  packed_bytes / unpacked_bytes: all ranks' lists in the two forms; sent_bytes: what this rank sent. */
int nts_comm_last_exchange2(const nts_ctx* ctx, uint64_t* packed_bytes, uint64_t* unpacked_bytes, uint64_t* sent_bytes);
/* Exchange 1 when a family's records are shared out over the ranks by bases, across genome boundaries (ntsynt_amd/pipeline.py
 * partition_plan: three genomes on eight GPUs are eight ranges of 1.125 Gbp, not 3/3/2 ranks per genome): a rank holds one filter per
 * genome its range of records touches.  bfs[0 .. n_local): this rank's filters, all of one size (nts_bf_create_sharded); the first
 * n_of[rank] of them are contributions, and bfs[0] receives the result on every rank (a rank without records still passes one
 * filter, the destination).  slot_group[r * n_slots + s] = group (genome) of rank r's s-th filter, -1 = none (a rank's contributions
 * are its first slots); every group in [0, n_groups) needs a filter somewhere; at most 128 filters in all.  Result as
 * nts_bf_allreduce_groups: AND over groups of (OR over the group's filters) -- the reference's cascade with the OR of
 * src/ntsynt_make_common_bf.cpp:128-131 (the records of a file inserted into one filter) spread over ranks.  comm == NULL or world 1:
 * the reduction over this rank's own filters. */
int nts_bf_allreduce_parts(nts_ctx* ctx, nts_bf* const* bfs, uint32_t n_local, nts_comm* comm, uint32_t n_slots, const int32_t* slot_group,
                           uint32_t n_groups);
int nts_mx_allgather(nts_ctx* ctx, nts_comm* comm, uint32_t n_local, const nts_mx* const* local, const uint32_t* local_ids,
                     uint32_t n_total, nts_mx** out);
/* nts_mx_allgather with the list slots per rank said by the caller (0: ceil(n_total / world)): with records shared out by bases a
 * rank holds one list per genome its range touches, and the plan's maximum is the slot count on every rank. */
int nts_mx_allgather_ex(nts_ctx* ctx, nts_comm* comm, uint32_t n_local, const nts_mx* const* local, const uint32_t* local_ids,
                        uint32_t n_total, uint32_t slots_per_rank, nts_mx** out);

/* ---- B1-B3, B5: minimizer sketch -----------------------------------------------------------------
 * replaces `indexlr -k K -w W --long --pos -s common.bf genome.fa` (smk:81-85) and the re-sketch of
 * hard-masked assemblies in the refinement rounds (bin/ntsynt_synteny.py:134-157,167-192): the mask
 * intervals are applied to the resident sequence instead of writing masked FASTA files.
 * Result: minimizers in (record, position) order; h1 = the hash indexlr prints. */
int nts_sketch(nts_ctx* ctx,
               const nts_genome* g,
               uint32_t k,
               uint32_t w,
               const nts_bf* filter_or_null,
               const nts_interval* mask,
               uint64_t n_mask,
               nts_mx** out);

/* The same with a filter-out Bloom filter as well (`indexlr -r <bf>`, the reference's experimental repeat filter: rule indexlr,
   bin/ntsynt_run_pipeline.smk:74-85, and ntsynt_synteny.py:172-180): a k-mer present in `filter_out` is rejected like one
   absent from `filter`; either may be NULL.  With a filter-out filter the call takes the every-k-mer-probed kernels. */
int nts_sketch_ex(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint32_t w, const nts_bf* filter, const nts_bf* filter_out,
                  const nts_interval* mask, uint64_t n_mask, nts_mx** out);
/* Sketch policy.  mode 0 = auto (pruned when w >= 200 and c = 11 / accepted share stays below w/4, below 0.15 w for w < 512; for
 * 8 <= w < 200 the tiered selection, nts_sketch_tiers, where its estimated probes pay -- always so without a filter),
 * 1 = dense (probe the filter for every k-mer), 2 = pruned: only k-mers whose hash is <= (c / w) * 2^64 are probed;
 * windows holding no accepted candidate are re-evaluated densely, so the result is identical
 * (ntsynt_amd/csrc/nts_pruned.inc).  prune_c = 0: c is chosen per call from the filter's occupancy
 * (c = 11 / accepted share, at least 8: DESIGN.md 4.1); otherwise c = prune_c. */
int nts_sketch_mode(nts_ctx* ctx, int mode, uint32_t prune_c);
/* Dense sketch over a sparse filter (many divergent genomes: nearly no k-mer is common to all): when occupancy x 2^shift is
 * small, a summary of the filter with one bit per 2^shift filter bits (<= 4 MiB: mostly L2-resident; built once per filter state) is
 * consulted first and only a set summary bit leads to a read of the filter; key tiles without an accepted k-mer are skipped
 * by the window kernel.  In auto mode the accepted k-mers themselves become the candidate list (no keys, no window kernel), and
 * when the filter holds few enough set bits a copy of it folded onto 2^19 bits is looked at first, from LDS.  Identical output.
 * mode 0 = auto (default), 1 = never, 2 = auto without the LDS copy, -1 = leave as is; last_shift = the shift the last
 * nts_sketch call used (0: it did not use a summary). */
int nts_sketch_summary(nts_ctx* ctx, int mode, uint32_t* last_shift);
/* Candidate selection of the pruned sketch (k_hash_select_hi / k_hash_select): 0 = automatic (the kernel that rolls only the
   upper halves of the strand hashes where it applies -- k <= 32, at most ~180 candidates per 4096 k-mers -- and pays -- fewer
   than one run of valid bases per two tiles), 1 = always the full-width kernel, 2 = the upper-halves kernel wherever it
   applies, fragmented assemblies included.  Results are identical; tests and measurements switch it. */
int nts_sketch_select(nts_ctx* ctx, int impl);
/* Tiered selection (k_hash_tiers, ntsynt_amd/csrc/nts_tiers.inc): for filters that accept a few per cent of a genome's k-mers
 * (three genomes at 10 % divergence, eight at 4 %, the reference's eleven-genome benchmark row, README.md:158) one threshold would list a
 * quarter of the k-mers or more.  Thresholds tau_0 2^t, t = 0, 1, ..., are then probed one after the other, each only where
 * some window still holds no accepted k-mer: ~3.4 / p probes per window instead of 11 / p (p = accepted share), identical
 * output (the filter-in semantics of indexlr -s, bin/ntsynt_run_pipeline.smk:81-85).  mode 0 = automatic (default), 1 = never,
 * 2 = wherever the kernel applies, -1 = leave as is; x0 = accepted k-mers per window the first tier aims at (0: default 2.4, 1.2 for
 * w < 64).  Automatic also means: short windows (8 <= w < 64, the last refinement round's w = 10, bin/ntSynt:89-91) go this way while the
 * filter accepts about five k-mers per window -- a fifth to a half of the probes of the every-k-mer pass;
 * half_steps: thresholds grow by 1.5 / 1.33 instead of 2.  Of the last nts_sketch call that went this way: k-mers probed,
 * rounds run summed over the tiles, tiers planned (0: it did not go this way). */
int nts_sketch_tiers(nts_ctx* ctx, int mode, double x0, int half_steps, uint64_t* last_probes, uint64_t* last_rounds, uint32_t* last_tiers);
/* of the last nts_sketch call: accepted candidates, uncovered ranges handed to the dense kernels, the number of
 * k-mers in them, and the c that was used (all 0 for a dense-mode call) */
int nts_sketch_stats(nts_ctx* ctx, uint64_t* candidates, uint64_t* uncovered_ranges, uint64_t* uncovered_kmers, uint32_t* prune_c_used);
/* Which of the paths for repeat-rich or fragmented input the last calls took (tests assert that an assembly-like family reaches
 * them): of the last nts_sketch call, the candidates of k_hash_select_hi tiles that listed more k-mers than their slots hold (the
 * two-sweep path: copies of a repeat, tiles in pieces); of the last partitioned Bloom build (nts_bf_insert / nts_bf_insert_and), the
 * indices that bypassed the buckets (a full bucket: the copies of a repeat family; lanes whose k-mers span a run boundary), and
 * whether its list of such indices ran full (store-only build: fell back to read-and-OR on the device; fused AND build: redone
 * with a filter of the genome's own).  The Bloom figures are read at the next synchronising call on the filter
 * (nts_bf_popcount, nts_bf_download, nts_sync). */
int nts_path_stats(nts_ctx* ctx, uint64_t* sketch_many_listed, uint64_t* bf_direct_indices, uint32_t* bf_list_fallback);
/* Of the last nts_bf_insert_and: whether the level went the literal way of src/ntsynt_make_common_bf.cpp:134-160 -- every k-mer of
 * the genome looked up in the running filter, the bits that were hit kept -- which the library chooses by itself once the running
 * filter is all but empty (its popcount known and below ~6 * 10^5 bits, where two folded tables in LDS answer most look-ups: what is
 * left of BASELINE configs[3]'s filter after its eighth genome), and how many k-mers were accepted.  Otherwise the level is the partitioned build with the AND in its last pass.
 * Same bits either way (tests/test_gpu_parity.py).  NTS_BF_SPARSE_LEVEL=0 in the environment keeps every level on the build. */
int nts_bf_level_stats(nts_ctx* ctx, uint32_t* sparse_level, uint64_t* accepted_kmers);
uint64_t nts_mx_count(const nts_mx* mx);
void nts_mx_free(nts_ctx* ctx, nts_mx* mx);
/* copy a minimizer list to caller-provided host arrays of nts_mx_count() elements */
int nts_mx_download(nts_ctx* ctx, const nts_mx* mx, uint64_t* h1, uint32_t* rec, uint64_t* pos);
/* device pointers (for an all-gather over RCCL): arrays of nts_mx_count() elements */
int nts_mx_device_ptrs(const nts_mx* mx, void** h1, void** rec, void** pos);
/* device-to-device copy into caller buffers of nts_mx_count() elements (send side of the all-gather) */
int nts_mx_export(nts_ctx* ctx, const nts_mx* mx, void* h1_dev, void* rec_dev, void* pos_dev);
/* the same without waiting: the copies are queued on the context's stream; nts_sync() before the buffers are read
 * by another stream and before the list is freed (several lists, one wait) */
int nts_mx_export_async(nts_ctx* ctx, const nts_mx* mx, void* h1_dev, void* rec_dev, void* pos_dev);
/* k-mer text (upper case, k bytes per minimizer, no separators) of a list sketched from `g`, to a host buffer of
 * nts_mx_count() * k bytes: what `indexlr --seq` prints (smk:81) when the host never held the bases (nts_genome_from_fasta) */
int nts_mx_kmers(nts_ctx* ctx, const nts_genome* g, const nts_mx* mx, uint32_t k, uint8_t* host);
/* the list of a batch genome (nts_genome_concat) taken apart on the device: out[p] = the minimizers of records
 * [rec_base[p], rec_base[p+1]) with record ids rebased to the part (rec_base has n_parts + 1 entries) */
int nts_mx_split(nts_ctx* ctx, const nts_mx* mx, uint32_t n_parts, const uint32_t* rec_base, nts_mx** out);
/* the inverse: the lists of a genome's shards (nts_genome_slice), in record order, as the genome's list -- part p's record
 * numbers raised by rec_offset[p] */
int nts_mx_concat(nts_ctx* ctx, uint32_t n_parts, const nts_mx* const* parts, const uint32_t* rec_offset, nts_mx** out);
/* ntJoin's read_minimizers(file, repeat_bf) -- stage 3's `--filter Filter --repeat <bf>` (bin/ntsynt_synteny.py:183-184,601-604; the
 * function itself is ntJoin's and not in the reference tree: [RECALLED]): *out = the minimizers of `mx` whose k-mer
 * (k bases of `g` at the minimizer's record and position) the filter does NOT hold, in order; released with nts_mx_free. */
int nts_mx_screen(nts_ctx* ctx, const nts_genome* g, const nts_mx* mx, uint32_t k, const nts_bf* filter_out, nts_mx** out);
/* build a device list from host arrays (receiving side of the all-gather, tests) */
int nts_mx_upload(nts_ctx* ctx,
                  const uint64_t* h1,
                  const uint32_t* rec,
                  const uint64_t* pos,
                  uint64_t n,
                  nts_mx** out);

/* test/bench hook: canonical h0 of every valid k-mer in (record, position) order (row B1) */
int nts_hash_all(nts_ctx* ctx, const nts_genome* g, uint32_t k, uint64_t** h0, uint64_t* n_out);

/* ---- C1-C5: minimizer graph -> collinear chains -----------------------------------------------------
 * replaces ntjoin_utils.read_minimizers' duplicate removal, filter_minimizers and build_graph
 * (call sites bin/ntsynt_synteny.py:607-612, 483, 539) and the path walk of Ntjoin.find_paths
 * (bin/ntsynt_synteny.py:492, 620).
 *
 * nts_graph_build -- on the GPU (radix sorts + scans over all minimizers of all assemblies):
 *   C1  per assembly, keep hashes seen exactly once (over ALL elements given);
 *       then AND with the caller's `keep` mask (refinement rounds: rows C11's position filter);
 *   C2a keep hashes that survive in every assembly; number them 0..nv-1 by ascending hash;
 *   C2b undirected edges between list-adjacent survivors; a "list" is a maximal run of equal
 *       `list_id` inside one assembly (NULL: the record index, i.e. one list per FASTA record);
 *       weight = number of assemblies in which the pair is adjacent.
 * Assemblies are given in the reference's order (descending file name).  Output (host arrays owned by
 * the library; release with nts_graph_free):
 *   v_hash[nv]            surviving hashes, ascending
 *   occ_rec/occ_pos       record and position of vertex v in assembly a at [a*nv + v]
 *   e_u/e_v               endpoints of each distinct edge, in the orientation of its first sighting
 *   e_w                   weight
 *   e_first               sequence number of the first sighting in the reference's traversal order
 *                         (assembly, list, index)
 * Edges are returned in the order ntJoin's dict of dicts iterates them, `[(s, t) for s in edges for t in
 * edges[s]]`: sources by the time they first became a source, then by creation time (bin/ntsynt_synteny.py:573
 * walks the edges in that order, and the order decides which bubbles are removed).
 */
typedef struct
{
  const uint64_t* h1;
  const uint32_t* rec;
  const uint64_t* pos;
  const uint8_t* keep;      /* NULL = keep all */
  const uint32_t* list_id;  /* NULL = rec */
  uint64_t n;
} nts_mxlist;

typedef struct
{
  uint64_t nv;
  uint64_t* v_hash;
  uint32_t* occ_rec;
  uint64_t* occ_pos;
  uint64_t ne;
  uint32_t* e_u;
  uint32_t* e_v;
  uint32_t* e_w;
  uint64_t* e_first;
} nts_graph;
int nts_graph_build(nts_ctx* ctx, uint32_t n_asm, const nts_mxlist* lists, nts_graph* out);
void nts_graph_free(nts_graph* g);

/* ---- C1-C9, C11, C12 (erosion): the minimizer graph of a run, resident in HBM across rounds ---------------------------
 * replaces the graph object the reference's stage 3 keeps in igraph / Python dicts (bin/ntsynt_synteny.py:31-32, ntJoin's
 * Ntjoin.graph) and every walk over it; only block tables (a few numbers per synteny block) and the rare-event lists of
 * the bubble rule return to the host.  Kernels and rule-by-rule citations: ntsynt_amd/csrc/nts_dgraph.inc; host driver:
 * ntsynt_amd/synteny_device.py.
 *
 * nts_engine_create(n_asm, ref_asm): assemblies in the reference's order (descending file name, S:34); ref_asm = the one
 *     whose positions orient the paths (ntJoin: the last = lexicographically smallest file).
 * nts_engine_add: build_graph(..., graph=self.graph, black_list=terminals) (S:483, S:612) from minimizer lists resident
 *     in HBM (nts_sketch / nts_mx_allgather results go in without touching the host).  spans == NULL: initial round.
 *     Refinement round (S:476-491, S:256-290): spans[a] = the block interiors of assembly a as composite keys
 *     record * 2^40 + position sorted by `start`, `end_max` = running maximum of the composite ends.
 * nts_engine_bubbles / nts_engine_apply: input and outcome of run_graph_simplification (S:566-590); the rule is order
 *     dependent and touches a handful of edges, it runs on the host over this table.
 * nts_engine_filter: filter_graph_global(_flag_overlaps) (S:292-303, S:491, S:617).
 * nts_engine_erode: refine_graph / erode_edges (S:305-362) over the pairs flagged by the last filter.
 * nts_engine_blocks: find_paths (ntJoin, called at S:492, S:620) by pointer jumping, then find_synteny_blocks,
 *     determine_orientations, check_for_indels, filter_synteny_blocks (S:66-106, synteny_block.py:48-70, S:364-426);
 *     per kept block: first / last vertex, number of minimizers, and per assembly [a * n_blocks + b] the contig, first and
 *     last position and orientation code (0 '+', 1 '-').
 * nts_engine_read: field read-back for tests ("v_hash", "v_alive", "v_rec", "v_pos", "internal", "terminal", "e_u",
 *     "e_v", "e_w", "e_alive", "path_verts", "path_off"). */
typedef struct nts_engine nts_engine;
typedef struct
{
  const uint64_t* start;
  const uint64_t* end_max;
  uint64_t n;
} nts_spans;
typedef struct
{
  uint64_t n_cand;
  uint32_t* cand_edge; /* candidate edges, ascending = the reference's edge order */
  uint64_t n_inc;      /* live edges incident to a candidate's end point */
  uint32_t* inc_edge;
  uint32_t* inc_u;
  uint32_t* inc_v;
  uint32_t* inc_w;
} nts_bubbles;
typedef struct
{
  uint64_t n_blocks;
  uint32_t* first_vid;
  uint32_t* last_vid;
  uint32_t* n_mx;
  uint32_t* rec;       /* [n_asm * n_blocks] */
  uint64_t* first_pos;
  uint64_t* last_pos;
  uint8_t* ori;
  uint64_t stats_paths, stats_unoriented, stats_indel_cuts, stats_small;
} nts_blocks;
int nts_engine_create(nts_ctx* ctx, uint32_t n_asm, uint32_t ref_asm, nts_engine** out);
void nts_engine_free(nts_ctx* ctx, nts_engine* eng);
int nts_engine_size(const nts_engine* eng, uint64_t* nv, uint64_t* ne);
int nts_engine_add(nts_ctx* ctx, nts_engine* eng, const nts_mx* const* lists, const nts_spans* spans, uint64_t* nv, uint64_t* ne);
int nts_engine_bubbles(nts_ctx* ctx, nts_engine* eng, nts_bubbles* out);
void nts_bubbles_free(nts_bubbles* b);
int nts_engine_apply(nts_ctx* ctx, nts_engine* eng, const uint32_t* dead_vertices, uint64_t n_dead, const uint32_t* promote_edges,
                     uint64_t n_promote, uint32_t weight);
int nts_engine_filter(nts_ctx* ctx, nts_engine* eng, uint32_t min_weight, int flag, uint64_t* n_light);
int nts_engine_erode(nts_ctx* ctx, nts_engine* eng, uint32_t k, uint64_t* n_dead_edges);
int nts_engine_blocks(nts_ctx* ctx, nts_engine* eng, int64_t bp, double m_percent, uint32_t min_mx, nts_blocks* out);
void nts_blocks_free(nts_blocks* b);
int nts_engine_paths(const nts_engine* eng, uint64_t* n_paths, uint64_t* n_verts);
int nts_engine_read(nts_ctx* ctx, const nts_engine* eng, const char* field, void* dst, uint64_t bytes);

/* ---- block-level rules over flat tables (host side, no GPU work): what is left of the reference's per-object Python once the
 * graph lives in HBM -- sequential where the reference's rule is order dependent.
 * nts_bubble_rule : run_graph_simplification (bin/ntsynt_synteny.py:566-590) over nts_engine_bubbles' table; doomed[i] /
 *     promoted[i] (caller-allocated, n_cand entries) = middle vertex and candidate edge of the i-th bubble, *n_out of them.
 * nts_blocks_merge: merge_collinear_blocks (S:428-472) in place over blocks sorted like SyntenyBlock.__lt__; tables are
 *     [a * n + b]; ori 0 '+' 1 '-'; reason 0 None, 1 id_change, 2 ori_change, 3 inconsistent_order, 4 indel, 5 merge.
 * nts_blocks_text : get_block_string (bin/synteny_block.py:72-85) for a table: blocks shorter than z in any assembly skipped,
 *     the others numbered from 0, rows in out_order; `reason` != NULL adds the verbose column.  names = NUL-separated
 *     strings: the G assembly names, then every assembly's contig names; contig_base[a] = index of assembly a's first contig
 *     among the contig names.  *text is released with nts_free. */
int nts_bubble_rule(uint64_t n_cand, const uint32_t* cand_edge, uint64_t n_inc, const uint32_t* inc_edge, const uint32_t* inc_u,
                    const uint32_t* inc_v, const uint32_t* inc_w, uint32_t wmax, uint32_t* doomed, uint32_t* promoted, uint64_t* n_out);
int nts_blocks_merge(uint32_t n_asm, uint64_t n, int64_t k, int64_t bp, int64_t collinear_merge, uint32_t* rec, int64_t* first,
                     int64_t* last, uint8_t* ori, int64_t* n_mx, uint8_t* reason, uint64_t* n_out, uint64_t* n_merged);
int nts_blocks_text(uint32_t n_asm, uint64_t n, int64_t k, int64_t z, const uint32_t* out_order, const char* names, uint64_t names_bytes,
                    const uint64_t* contig_base, const uint32_t* rec, const int64_t* first, const int64_t* last, const uint8_t* ori,
                    const int64_t* n_mx, const uint8_t* reason, char** text, uint64_t* text_bytes);

/* Host-side helper (no GPU work): connected components of an undirected graph given as edge arrays
 * that are simple paths (two degree-1 ends, everything else degree 2, at least 2 vertices) -- what
 * Ntjoin.find_paths keeps.  Path i is verts[off[i] .. off[i+1]); each path starts at its end with the
 * smaller vertex id.  Paths are listed by ascending first vertex id.  Release with nts_free(). */
int nts_walk_chains(uint64_t nv, uint64_t ne, const uint32_t* e_u, const uint32_t* e_v,
                    uint64_t** off, uint32_t** verts, uint64_t* n_paths);

/* The same walk on the engine's own arrays: 64-bit edge ends, an optional liveness byte per edge (NULL = all
 * live) and an optional per-vertex key: with a key, each path is written starting at the end with the smaller
 * key (row C5: ntJoin's find_paths, called at bin/ntsynt_synteny.py:620, starts a path at the end with the
 * smaller position in the reference assembly) while the paths keep the order above (ascending smaller-id end). */
int nts_walk_paths(uint64_t nv, uint64_t ne, const int64_t* e_u, const int64_t* e_v, const uint8_t* e_alive,
                   const int64_t* key, uint64_t** off, int64_t** verts, uint64_t* n_paths);

/* Host-side helper (host threads): dst[i] = src[i] as a 64-bit integer, src holding 32-bit (src_bytes 4) or 64-bit (8)
 * unsigned values -- how a caller that keeps everything in int64 (the Python engine) takes over nts_graph's arrays. */
int nts_to_i64(const void* src, uint32_t src_bytes, uint64_t n, int64_t* dst);

/* Host-side helper: deg[v] = number of live edge ends at v, saturating at 255 (bubble detection asks
 * "degree 3", bin/ntsynt_synteny.py:548-590; path ends ask "degree 1"). */
int nts_edge_degrees(uint64_t nv, uint64_t ne, const int64_t* e_u, const int64_t* e_v, const uint8_t* e_alive,
                     uint8_t* deg);

/* Host-side helper (no GPU work, host threads): one pass over the paths of a round -- path i is
 * verts[off[i] .. off[i+1]) -- against the per-assembly vertex tables v_rec / v_pos ([a*nv + v]):
 *   start[i]          index into verts of the first vertex kept: a path whose contig changes in any assembly
 *                     keeps only its last run (bin/ntsynt_synteny.py:71-77);
 *   n_up[a*n_paths+i] steps of the kept run along which the position in assembly a rises -- the orientation
 *                     rule's input (bin/synteny_block.py:48-65);
 *   over[j]           1 where the distance from verts[j] to verts[j+1] differs between two assemblies by more
 *                     than bp (indel split, bin/ntsynt_synteny.py:364-409); 0 outside kept runs.
 * All outputs are caller-allocated (n_paths, n_asm*n_paths, off[n_paths] elements). */
int nts_path_scan(uint32_t n_asm, uint64_t nv, const int64_t* v_rec, const int64_t* v_pos, uint64_t n_paths,
                  const uint64_t* off, const int64_t* verts, int64_t bp, uint64_t* start, uint64_t* n_up,
                  uint8_t* over);

/* ---- host-side I/O (no GPU work) ---------------------------------------------------------------------
 * nts_fasta_read: plain or gzip FASTA, single- or multi-line, LF or CRLF -> concatenated bases + record table +
 *   record ids (header up to the first whitespace, NUL-separated) + the `samtools faidx` columns; NTS_EFORMAT for a
 *   non-empty file that does not start with a '>' header (FASTQ is not accepted).  Replaces
 *   btllib::SeqReader(LONG_MODE) record streaming (src/ntsynt_make_common_bf.cpp:32-36) and rule faidx (smk:48-53).
 * nts_write_indexlr_tsv: `indexlr --long --pos [--seq]` text (smk:81-85): "id\thash:pos[:KMER] ...\n" per record;
 *   minimizers given in (record, position) order as nts_mx_download returns them. */
typedef struct
{
  uint8_t* seq;
  uint64_t n;
  uint32_t n_rec;
  uint64_t* rec_off;
  uint64_t* rec_len;
  char* names;
  uint64_t names_bytes;
  uint64_t* fai_offset;
  uint32_t* fai_linebases;
  uint32_t* fai_linewidth;
} nts_fasta;
int nts_fasta_read(const char* path, nts_fasta* out);
/* The same ingest with the parse on the GPU: the file's bytes (plain: mapped; .gz: inflated on the
 * host) go to HBM through pinned staging, kernels find the header lines, drop line ends, compact and encode the bases into
 * a resident genome and produce the record table and the faidx columns; the host reads only the header lines.  `meta` as
 * nts_fasta_read fills it, except seq == NULL (the bases never exist on the host; nts_mx_kmers serves `--seq` output).
 * Same rules as nts_fasta_read (white space at either end of a sequence line is dropped, inside a line it is an invalid base).
 * nts_ingest_trim: gives back what the ingest keeps between files -- the image of the largest file read so far (as large as
 * that file, in HBM) and the pinned staging lanes -- once the last genome of a run is resident. */
int nts_genome_from_fasta(nts_ctx* ctx, const char* path, nts_genome** out, nts_fasta* meta);
int nts_ingest_trim(nts_ctx* ctx);
/* the Bloom build's bucket arrays (24 GB at 3 Gbp) back to the library's allocation cache once the last filter of a run is made (rules
 * indexlr and ntsynt_synteny build none: bin/ntsynt_run_pipeline.smk:74-103): later allocations are cut from them; the next build, if
 * there is one, takes them up again */
int nts_bf_build_trim(nts_ctx* ctx);
/* nts_write_indexlr_tsv for records whose bases are not on the host (fa->seq == NULL): `kmers` = nts_mx_kmers' output, or
 * NULL without --seq */
int nts_write_indexlr_tsv_kmers(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                                uint64_t n, uint32_t k, const uint8_t* kmers);
void nts_fasta_free(nts_fasta* f);
/* nts_read_indexlr_tsv: ntJoin's read_minimizers on an `indexlr --long --pos [--seq]` file -- the input of the reference's stage 3
 * (bin/ntsynt_run.py:12 FILES, bin/ntsynt_synteny.py:607-609; file format: rule indexlr, smk:74-85): per line the record id and its
 * tokens "hash:pos[:KMER]".  Out: the ids of all lines (NUL-separated, in file order), and per token the printed hash, the position
 * and the number of its line, in file order.  NTS_EFORMAT for a token without a position.  Released with nts_mx_tsv_free. */
typedef struct
{
  uint64_t n_lines;
  char* names;
  uint64_t names_bytes;
  uint64_t n;
  uint64_t* h1;
  uint64_t* pos;
  uint32_t* line;
} nts_mx_tsv;
int nts_read_indexlr_tsv(const char* path, nts_mx_tsv* out);
void nts_mx_tsv_free(nts_mx_tsv* t);
int nts_write_indexlr_tsv(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                          uint64_t n, uint32_t k, int with_seq);

void nts_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
