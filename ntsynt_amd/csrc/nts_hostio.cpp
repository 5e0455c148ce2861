// Host side of libntsynt_hip.so (no GPU work): FASTA ingest, `.fai` columns, indexlr-format
// minimizer TSV writer, and the chain walk over the minimizer graph (Ntjoin.find_paths).  Stands in for btllib::SeqReader (src/ntsynt_make_common_bf.cpp:32-36,125-131),
// `samtools faidx` (bin/ntsynt_run_pipeline.smk:48-53) and indexlr's output stage (smk:81-85).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ntsynt_hip.h"

namespace {

bool ends_with(const std::string& s, const char* suf)
{
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// whole file into memory (gzip-transparent)
bool slurp(const char* path, std::vector<uint8_t>& data)
{
  const std::string p(path);
  if (ends_with(p, ".gz")) {
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    size_t used = 0;
    data.resize(1 << 24);
    for (;;) {
      if (used == data.size()) data.resize(data.size() * 2);
      const int got = gzread(f, data.data() + used, (unsigned)std::min<size_t>(data.size() - used, 1u << 30));
      if (got < 0) {
        gzclose(f);
        return false;
      }
      if (got == 0) break;
      used += (size_t)got;
    }
    gzclose(f);
    data.resize(used);
    return true;
  }
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long long sz = ftello(f);
  fseek(f, 0, SEEK_SET);
  data.resize((size_t)sz);
  size_t used = 0;
  while (used < (size_t)sz) {
    const size_t got = fread(data.data() + used, 1, (size_t)sz - used, f);
    if (got == 0) break;
    used += got;
  }
  fclose(f);
  data.resize(used);
  return true;
}

// Bytes of an input file: plain files are mapped (no read() copy, no second buffer to fault in), gzip files are
// inflated into memory.
struct FileBytes
{
  const uint8_t* p = nullptr;
  size_t n = 0;
  void* map = nullptr;
  size_t map_len = 0;
  std::vector<uint8_t> owned;
  bool open(const char* path)
  {
    if (!ends_with(path, ".gz")) {
      const int fd = ::open(path, O_RDONLY);
      if (fd < 0) return false;
      struct stat st;
      if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return slurp(path, owned) && ((p = owned.data()), (n = owned.size()), true);
      }
      if (st.st_size == 0) {
        ::close(fd);
        return true;
      }
      void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      ::close(fd);
      if (m == MAP_FAILED) return slurp(path, owned) && ((p = owned.data()), (n = owned.size()), true);
      madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
      madvise(m, (size_t)st.st_size, MADV_WILLNEED);
      map = m;
      map_len = (size_t)st.st_size;
      p = (const uint8_t*)m;
      n = map_len;
      return true;
    }
    if (!slurp(path, owned)) return false;
    p = owned.data();
    n = owned.size();
    return true;
  }
  ~FileBytes()
  {
    if (map) munmap(map, map_len);
  }
};

} // namespace

// internal (not part of the C ABI in include/ntsynt_hip.h): the bytes of an input file for the device-side FASTA parse in
// ntsynt_hip.hip -- mapped for plain files, inflated for .gz; released with nts_internal_file_close
extern "C" int nts_internal_file_open(const char* path, const uint8_t** p, uint64_t* n, void** handle)
{
  FileBytes* f = new FileBytes();
  if (!f->open(path)) {
    delete f;
    return NTS_EINVAL;
  }
  *p = f->p;
  *n = f->n;
  *handle = f;
  return NTS_OK;
}

extern "C" void nts_internal_file_close(void* handle)
{
  delete (FileBytes*)handle;
}

namespace {

// Large host buffers (a genome's bases): 2 MiB-aligned and advised to use huge pages, so that first touch costs
// thousands of page faults rather than millions.  Released with free().
void* big_alloc(size_t bytes)
{
  constexpr size_t HUGE = (size_t)1 << 21;
  if (bytes < 8 * HUGE) return malloc(std::max<size_t>(bytes, 1));
  void* p = nullptr;
  if (posix_memalign(&p, HUGE, (bytes + HUGE - 1) / HUGE * HUGE) != 0) return nullptr;
  madvise(p, (bytes + HUGE - 1) / HUGE * HUGE, MADV_HUGEPAGE);
  return p;
}

template <typename T>
T* dup_vec(const std::vector<T>& v)
{
  T* p = (T*)malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

inline char* put_u64(char* p, uint64_t v)
{
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}

} // namespace

extern "C" int nts_fasta_read(const char* path, nts_fasta* out)
{
  if (!path || !out) return NTS_EINVAL;
  memset(out, 0, sizeof(*out));
  FileBytes file;
  if (!file.open(path)) return NTS_EINVAL;
  const uint8_t* const data = file.p;
  const size_t n = file.n;
  bool blank = true;
  {
    // FASTA only: FASTQ (first non-blank byte '@') would parse as zero records and the run would go on with an empty
    // assembly; so would any other non-blank file without a single '>' header (checked after the parse)
    size_t q = 0;
    while (q < n && (data[q] == '\n' || data[q] == '\r' || data[q] == ' ' || data[q] == '\t')) ++q;
    blank = q == n;
    if (!blank && data[q] == '@') return NTS_EFORMAT;
  }
  uint8_t* seq = (uint8_t*)big_alloc(n);
  if (!seq) return NTS_ENOMEM;
  std::vector<uint64_t> rec_off, rec_len, fai_off;
  std::vector<uint32_t> fai_bases, fai_width;
  std::string names;
  size_t w = 0; // write cursor in seq
  size_t i = 0;
  bool in_record = false;
  bool first_line = false;
  while (i < n) {
    const uint8_t* nl = (const uint8_t*)memchr(data + i, '\n', n - i);
    const size_t line_end = nl ? (size_t)(nl - data) : n; // exclusive, without the newline
    if (data[i] == '>') {
      if (in_record) rec_len.back() = w - rec_off.back();
      // record id = header up to the first whitespace
      size_t s = i + 1, e = s;
      while (e < line_end && data[e] != ' ' && data[e] != '\t' && data[e] != '\r' && data[e] != '\v' && data[e] != '\f') ++e;
      names.append((const char*)data + s, e - s);
      names.push_back('\0');
      rec_off.push_back(w);
      rec_len.push_back(0);
      fai_off.push_back(nl ? line_end + 1 : n);
      fai_bases.push_back(0);
      fai_width.push_back(0);
      in_record = true;
      first_line = true;
    } else if (in_record) {
      size_t len = line_end - i;
      size_t bases = len;
      if (bases && data[i + bases - 1] == '\r') --bases;
      if (first_line) {
        if (rec_len.size() && line_end > i) {
          fai_bases.back() = (uint32_t)bases;
          fai_width.back() = (uint32_t)(len + (nl ? 1 : 0));
        }
        first_line = false;
      }
      // sequence bytes: everything except CR (LF is already excluded)
      if (memchr(data + i, '\r', len) == nullptr) {
        memcpy(seq + w, data + i, len);
        w += len;
      } else {
        for (size_t q = i; q < line_end; ++q)
          if (data[q] != '\r') seq[w++] = data[q];
      }
    }
    i = nl ? line_end + 1 : n;
  }
  if (in_record) rec_len.back() = w - rec_off.back();
  if (rec_off.empty() && !blank) {
    free(seq);
    return NTS_EFORMAT;
  }
  for (size_t r = 0; r < rec_len.size(); ++r)
    if (rec_len[r] == 0) fai_bases[r] = fai_width[r] = 0;
  out->seq = seq;
  out->n = w;
  out->n_rec = (uint32_t)rec_off.size();
  out->rec_off = dup_vec(rec_off);
  out->rec_len = dup_vec(rec_len);
  out->names_bytes = names.size();
  out->names = (char*)malloc(std::max<size_t>(names.size(), 1));
  if (out->names && !names.empty()) memcpy(out->names, names.data(), names.size());
  out->fai_offset = dup_vec(fai_off);
  out->fai_linebases = dup_vec(fai_bases);
  out->fai_linewidth = dup_vec(fai_width);
  if (!out->rec_off || !out->rec_len || !out->names || !out->fai_offset || !out->fai_linebases || !out->fai_linewidth) {
    nts_fasta_free(out);
    return NTS_ENOMEM;
  }
  return NTS_OK;
}

extern "C" void nts_fasta_free(nts_fasta* f)
{
  if (!f) return;
  free(f->seq);
  free(f->rec_off);
  free(f->rec_len);
  free(f->names);
  free(f->fai_offset);
  free(f->fai_linebases);
  free(f->fai_linewidth);
  memset(f, 0, sizeof(*f));
}

static int write_indexlr_tsv_impl(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                                  uint64_t n, uint32_t k, int with_seq, const uint8_t* kmers);

// `indexlr --long --pos [--seq]`: one line per record, "id \t hash:pos[:KMER] hash:pos[:KMER] ...\n"
extern "C" int nts_write_indexlr_tsv(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                                     uint64_t n, uint32_t k, int with_seq)
{
  if (fa && with_seq && !fa->seq) return NTS_EINVAL; // bases not on the host: nts_write_indexlr_tsv_kmers
  return write_indexlr_tsv_impl(path, fa, h1, rec, pos, n, k, with_seq, nullptr);
}

extern "C" int nts_write_indexlr_tsv_kmers(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                                           uint64_t n, uint32_t k, const uint8_t* kmers)
{
  return write_indexlr_tsv_impl(path, fa, h1, rec, pos, n, k, kmers != nullptr, kmers);
}

static int write_indexlr_tsv_impl(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                                  uint64_t n, uint32_t k, int with_seq, const uint8_t* kmers)
{
  if (!path || !fa || (n && (!h1 || !rec || !pos))) return NTS_EINVAL;
  FILE* f = fopen(path, "wb");
  if (!f) return NTS_EINVAL;
  std::vector<char> buf(1 << 22);
  size_t used = 0;
  auto flush = [&]() {
    if (used) fwrite(buf.data(), 1, used, f);
    used = 0;
  };
  const char* name = fa->names;
  uint64_t i = 0;
  for (uint32_t r = 0; r < fa->n_rec; ++r) {
    const size_t name_len = strlen(name);
    if (used + name_len + 2 > buf.size()) flush();
    memcpy(buf.data() + used, name, name_len);
    used += name_len;
    buf[used++] = '\t';
    bool first = true;
    while (i < n && rec[i] == r) {
      if (used + 64 + k > buf.size()) flush();
      char* p = buf.data() + used;
      if (!first) *p++ = ' ';
      first = false;
      p = put_u64(p, h1[i]);
      *p++ = ':';
      p = put_u64(p, pos[i]);
      if (with_seq) {
        *p++ = ':';
        const uint8_t* s = kmers ? kmers + i * (uint64_t)k : fa->seq + fa->rec_off[r] + pos[i];
        for (uint32_t q = 0; q < k; ++q) {
          const uint8_t c = s[q];
          *p++ = (char)((c >= 'a' && c <= 'z') ? c - 32 : c);
        }
      }
      used = (size_t)(p - buf.data());
      ++i;
    }
    buf[used++] = '\n';
    name += name_len + 1;
  }
  flush();
  const bool ok = fclose(f) == 0 && i == n;
  return ok ? NTS_OK : NTS_EINVAL;
}

// ---- chain walk (row C5) ------------------------------------------------------------------------------------
namespace {

// host threads worth starting: the cgroup CPU quota when there is one (a container may see far more logical
// CPUs than it is allowed to use), capped
unsigned host_threads(unsigned cap)
{
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[64];
    long long period = 0;
    if (fscanf(f, "%63s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
      const long long q = atoll(quota);
      if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / period));
    }
    fclose(f);
  }
  if (const char* e = getenv("NTS_HOST_THREADS")) { // (an upper limit set by the user, e.g. to leave cores to other work)
    const long v = atol(e);
    if (v > 0) n = std::min<unsigned>(n, (unsigned)v);
  }
  return std::max(1u, std::min(n, cap));
}

template <typename F>
void parallel_ranges(uint64_t n, unsigned n_threads, F&& body)
{
  if (n_threads <= 1 || n < 4096) {
    body(0u, (uint64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  const uint64_t per = (n + n_threads - 1) / n_threads;
  for (unsigned t = 0; t < n_threads; ++t) {
    const uint64_t lo = std::min<uint64_t>(n, t * per), hi = std::min<uint64_t>(n, lo + per);
    pool.emplace_back([&body, t, lo, hi]() { body(t, lo, hi); });
  }
  for (auto& th : pool) th.join();
}

} // namespace

// Components that are simple paths (Ntjoin.find_paths keeps exactly those).  Every step of a walk is a dependent
// cache miss on a multi-million-vertex graph, so the walks are spread over host threads: each degree-1 vertex walks
// to the other end of its chain and the walk that started at the smaller vertex id is the one kept, which is the
// path (and the order, by ascending first vertex) a sequential sweep over the vertex ids yields.
//
// EdgeT/OutT let the engine hand over its int64 edge arrays and liveness mask as they are and get int64 vertex ids
// back (no conversion passes); `key`, when given, orients each path to start at the end with the smaller key
// (ntJoin's determine_source_vertex on reference positions) without changing the order of the paths.
template <typename EdgeT, typename OutT>
static int walk_impl(uint64_t nv, uint64_t ne, const EdgeT* e_u, const EdgeT* e_v, const uint8_t* e_alive, const int64_t* key, uint64_t** off,
                     OutT** verts, uint64_t* n_paths)
{
  if (!off || !verts || !n_paths || (ne && (!e_u || !e_v)) || nv > 0xFFFFFFFEULL) return NTS_EINVAL;
  const unsigned T = host_threads(16);
  const uint32_t NONE = 0xFFFFFFFFu;
  auto t_last = std::chrono::steady_clock::now();
  const bool debug = getenv("NTS_HOST_DEBUG") != nullptr;
  auto lap = [&](const char* what) {
    if (!debug) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "nts_walk_chains %s %.3f s (%u threads)\n", what, std::chrono::duration<double>(now - t_last).count(), T);
    t_last = now;
  };
  // degree (saturating at 3) and the first two neighbours of every vertex.  Each thread owns a range of vertex ids
  // and streams over the whole edge list: the random accesses stay inside a cache-sized slice and need no atomics.
  std::vector<uint8_t> deg(nv, 0);
  std::vector<uint32_t> nb(2 * nv, NONE);
  std::atomic<bool> bad(false);
  parallel_ranges(nv, T, [&](unsigned, uint64_t lo, uint64_t hi) {
    auto touch = [&](uint32_t x, uint32_t other) {
      if (x < lo || x >= hi) return;
      const uint8_t c = deg[x];
      if (c < 3) deg[x] = c + 1;
      if (c < 2) nb[2 * (uint64_t)x + c] = other;
    };
    for (uint64_t e = 0; e < ne; ++e) {
      if (e_alive && !e_alive[e]) continue;
      const uint64_t u = (uint64_t)e_u[e], v = (uint64_t)e_v[e];
      if (u >= nv || v >= nv) {
        bad.store(true);
        return;
      }
      touch((uint32_t)u, (uint32_t)v);
      touch((uint32_t)v, (uint32_t)u);
    }
  });
  if (bad.load()) return NTS_EINVAL;
  lap("neighbour table");
  auto degree = [&](uint32_t v) { return deg[v]; };
  // walks, handed out dynamically: in refinement rounds almost every chain end is a freshly numbered vertex, so
  // splitting the vertex-id range evenly would leave one thread with all the work
  std::vector<uint32_t> ends;
  for (uint64_t v = 0; v < nv; ++v)
    if (deg[v] == 1) ends.push_back((uint32_t)v);
  struct Kept
  {
    uint32_t thread;
    uint64_t begin, len;
  };
  std::vector<Kept> kept(ends.size(), Kept{ 0, 0, 0 });
  std::vector<std::vector<uint32_t>> t_out(T);
  std::vector<uint64_t> t_steps(T, 0);
  std::atomic<uint64_t> next_end(0);
  auto walker = [&](unsigned t) {
    std::vector<uint32_t>& out = t_out[t];
    constexpr uint64_t GRAIN = 4;
    for (;;) {
      const uint64_t i0 = next_end.fetch_add(GRAIN);
      if (i0 >= ends.size()) return;
      for (uint64_t i = i0; i < std::min<uint64_t>(ends.size(), i0 + GRAIN); ++i) {
        const uint32_t s = ends[i];
        const size_t mark = out.size();
        uint32_t prev = NONE, cur = s;
        bool ok = true;
        for (uint64_t steps = 0;; ++steps) {
          ++t_steps[t];
          out.push_back(cur);
          if (degree(cur) > 2 || steps > nv) { // a branching vertex poisons the component
            ok = false;
            break;
          }
          const uint32_t a = nb[2 * (uint64_t)cur], b = nb[2 * (uint64_t)cur + 1];
          uint32_t nxt = NONE;
          if (a != NONE && a != prev)
            nxt = a;
          else if (b != NONE && b != prev)
            nxt = b;
          if (nxt == NONE) break;
          prev = cur;
          cur = nxt;
        }
        // keep the walk that started at the smaller end
        if (ok && out.size() - mark >= 2 && degree(out.back()) == 1 && out.back() > s)
          kept[i] = Kept{ t, mark, out.size() - mark };
        else
          out.resize(mark);
      }
    }
  };
  if (T <= 1 || ends.size() < 64) {
    walker(0);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t) pool.emplace_back(walker, t);
    for (auto& th : pool) th.join();
  }
  lap("walks");
  uint64_t total_paths = 0, total_verts = 0;
  for (const Kept& k : kept) {
    if (!k.len) continue;
    ++total_paths;
    total_verts += k.len;
  }
  *n_paths = total_paths;
  *off = (uint64_t*)malloc((total_paths + 1) * sizeof(uint64_t));
  *verts = (OutT*)malloc(std::max<uint64_t>(total_verts, 1) * sizeof(OutT));
  if (!*off || !*verts) return NTS_ENOMEM;
  std::vector<uint64_t> kept_at; // output offset of every kept walk, ascending first vertex
  kept_at.reserve(total_paths);
  uint64_t po = 0, vo = 0;
  (*off)[0] = 0;
  for (uint64_t i = 0; i < kept.size(); ++i) {
    if (!kept[i].len) continue;
    kept_at.push_back(i);
    vo += kept[i].len;
    (*off)[++po] = vo;
  }
  parallel_ranges(total_paths, T, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t p = lo; p < hi; ++p) {
      const Kept& k = kept[kept_at[p]];
      const uint32_t* src = t_out[k.thread].data() + k.begin;
      OutT* dst = *verts + (*off)[p];
      if (key && key[src[k.len - 1]] < key[src[0]])
        for (uint64_t j = 0; j < k.len; ++j) dst[j] = (OutT)src[k.len - 1 - j];
      else
        for (uint64_t j = 0; j < k.len; ++j) dst[j] = (OutT)src[j];
    }
  });
  lap("concatenate");
  if (debug) {
    uint64_t longest = 0, steps = 0;
    for (uint64_t i = 0; i < total_paths; ++i) longest = std::max(longest, (*off)[i + 1] - (*off)[i]);
    for (unsigned t = 0; t < T; ++t) steps += t_steps[t];
    for (unsigned t = 0; t < T; ++t) fprintf(stderr, "  thread %u: %llu steps\n", t, (unsigned long long)t_steps[t]);
    fprintf(stderr, "nts_walk_chains nv=%llu ne=%llu paths=%llu vertices on paths=%llu longest=%llu steps walked=%llu\n", (unsigned long long)nv,
            (unsigned long long)ne, (unsigned long long)total_paths, (unsigned long long)total_verts, (unsigned long long)longest,
            (unsigned long long)steps);
  }
  return NTS_OK;
}

extern "C" int nts_walk_chains(uint64_t nv, uint64_t ne, const uint32_t* e_u, const uint32_t* e_v, uint64_t** off, uint32_t** verts,
                               uint64_t* n_paths)
{
  return walk_impl<uint32_t, uint32_t>(nv, ne, e_u, e_v, nullptr, nullptr, off, verts, n_paths);
}

extern "C" int nts_walk_paths(uint64_t nv, uint64_t ne, const int64_t* e_u, const int64_t* e_v, const uint8_t* e_alive, const int64_t* key,
                              uint64_t** off, int64_t** verts, uint64_t* n_paths)
{
  return walk_impl<int64_t, int64_t>(nv, ne, e_u, e_v, e_alive, key, off, verts, n_paths);
}

// dst[i] = src[i] widened to 64 bits (src_bytes = 4) or copied (8), over host threads: the graph build hands its arrays to
// the engine this way (numpy's astype does the same on one thread: 0.15 s for the 65 M elements of a 3 x 3 Gbp run).
extern "C" int nts_to_i64(const void* src, uint32_t src_bytes, uint64_t n, int64_t* dst)
{
  if (n && (!src || !dst)) return NTS_EINVAL;
  if (src_bytes != 4 && src_bytes != 8) return NTS_EINVAL;
  parallel_ranges(n, host_threads(16), [&](unsigned, uint64_t lo, uint64_t hi) {
    if (src_bytes == 4) {
      const uint32_t* s4 = (const uint32_t*)src;
      for (uint64_t i = lo; i < hi; ++i) dst[i] = (int64_t)s4[i];
    } else {
      memcpy(dst + lo, (const uint64_t*)src + lo, (hi - lo) * 8);
    }
  });
  return NTS_OK;
}

// Degree of every vertex over the live edges, saturating at 255 (the engine asks "== 1" and "== 3").
extern "C" int nts_edge_degrees(uint64_t nv, uint64_t ne, const int64_t* e_u, const int64_t* e_v, const uint8_t* e_alive, uint8_t* deg)
{
  if ((nv && !deg) || (ne && (!e_u || !e_v))) return NTS_EINVAL;
  std::atomic<bool> bad(false);
  parallel_ranges(nv, host_threads(16), [&](unsigned, uint64_t lo, uint64_t hi) {
    memset(deg + lo, 0, hi - lo);
    auto touch = [&](uint64_t x) {
      if (x >= lo && x < hi && deg[x] != 255) ++deg[x];
    };
    for (uint64_t e = 0; e < ne; ++e) {
      if (e_alive && !e_alive[e]) continue;
      const uint64_t u = (uint64_t)e_u[e], v = (uint64_t)e_v[e];
      if (u >= nv || v >= nv) {
        bad.store(true);
        return;
      }
      touch(u);
      touch(v);
    }
  });
  return bad.load() ? NTS_EINVAL : NTS_OK;
}

// ---- per-path scan (rows C6-C8) -------------------------------------------------------------------------------
// One pass over every path of the round, spread over host threads (each vertex costs a cache miss per table):
//   * a path whose contig changes in any assembly keeps only its last run (bin/ntsynt_synteny.py:71-77):
//     start[i] = index into verts of the first vertex kept;
//   * rising steps per assembly over the kept run, for the orientation rule (bin/synteny_block.py:48-65);
//   * over[j] = 1 where the gap to the next vertex differs between assemblies by more than `bp`
//     (bin/ntsynt_synteny.py:364-409), inside kept runs only.
extern "C" int nts_path_scan(uint32_t n_asm, uint64_t nv, const int64_t* v_rec, const int64_t* v_pos, uint64_t n_paths, const uint64_t* off,
                             const int64_t* verts, int64_t bp, uint64_t* start, uint64_t* n_up, uint8_t* over)
{
  if (n_asm == 0 || (n_paths && (!off || !verts || !v_rec || !v_pos || !start || !n_up || !over))) return NTS_EINVAL;
  std::atomic<uint64_t> next(0);
  std::atomic<bool> bad(false);
  const unsigned T = n_paths ? host_threads(16) : 1;
  auto work = [&]() {
    constexpr uint64_t GRAIN = 8;
    for (;;) {
      const uint64_t p0 = next.fetch_add(GRAIN);
      if (p0 >= n_paths) return;
      for (uint64_t p = p0; p < std::min(n_paths, p0 + GRAIN); ++p) {
        const uint64_t lo = off[p], hi = off[p + 1];
        for (uint64_t j = lo; j < hi; ++j) {
          over[j] = 0;
          if (verts[j] < 0 || (uint64_t)verts[j] >= nv) {
            bad.store(true);
            return;
          }
        }
        uint64_t st = lo;
        for (uint64_t j = lo; j + 1 < hi; ++j) {
          const uint64_t x = (uint64_t)verts[j], y = (uint64_t)verts[j + 1];
          for (uint32_t a = 0; a < n_asm; ++a)
            if (v_rec[(uint64_t)a * nv + x] != v_rec[(uint64_t)a * nv + y]) st = j + 1;
        }
        start[p] = st;
        for (uint32_t a = 0; a < n_asm; ++a) n_up[(uint64_t)a * n_paths + p] = 0;
        for (uint64_t j = st; j + 1 < hi; ++j) {
          const uint64_t x = (uint64_t)verts[j], y = (uint64_t)verts[j + 1];
          int64_t g_min = INT64_MAX, g_max = INT64_MIN;
          for (uint32_t a = 0; a < n_asm; ++a) {
            const int64_t px = v_pos[(uint64_t)a * nv + x], py = v_pos[(uint64_t)a * nv + y];
            if (py > px) ++n_up[(uint64_t)a * n_paths + p];
            const int64_t g = px > py ? px - py : py - px;
            g_min = std::min(g_min, g);
            g_max = std::max(g_max, g);
          }
          over[j] = (g_max - g_min > bp) ? 1 : 0;
        }
      }
    }
  };
  if (T <= 1 || n_paths < 64) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  return bad.load() ? NTS_EINVAL : NTS_OK;
}
