// Host side of libntsynt_hip.so (no GPU work): FASTA ingest, `.fai` columns, indexlr-format
// minimizer TSV writer, and the chain walk over the minimizer graph (Ntjoin.find_paths).  Stands in for btllib::SeqReader (src/ntsynt_make_common_bf.cpp:32-36,125-131),
// `samtools faidx` (bin/ntsynt_run_pipeline.smk:48-53) and indexlr's output stage (smk:81-85).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "nts_knobs.h"

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
      const size_t len = line_end - i;
      // white space at either end of the line goes (same rule as csrc/nts_fasta_dev.inc); carriage returns go wherever they are
      auto blank = [](uint8_t c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r'; };
      size_t a = i, b = line_end;
      while (a < b && blank(data[a])) ++a;
      while (b > a && blank(data[b - 1])) --b;
      const size_t w0 = w;
      if (memchr(data + a, '\r', b - a) == nullptr) {
        memcpy(seq + w, data + a, b - a);
        w += b - a;
      } else {
        for (size_t q = a; q < b; ++q)
          if (data[q] != '\r') seq[w++] = data[q];
      }
      if (first_line) {
        if (rec_len.size() && line_end > i) {
          fai_bases.back() = (uint32_t)(w - w0);
          fai_width.back() = (uint32_t)(len + (nl ? 1 : 0));
        }
        first_line = false;
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
// ntJoin's read_minimizers on an `indexlr --long --pos [--seq]` file (the stage-3 input of the reference: bin/ntsynt_run.py FILES,
// bin/ntsynt_synteny.py:607-609): one line per record, "id\thash:pos[:KMER] hash:pos[:KMER] ...".  Parsed on host threads
// (a 3 Gbp genome's file: ~6 M tokens, 300 MB with --seq).
extern "C" int nts_read_indexlr_tsv(const char* path, nts_mx_tsv* out)
{
  if (!path || !out) return NTS_EINVAL;
  memset(out, 0, sizeof(*out));
  FileBytes f;
  if (!f.open(path)) return NTS_EINVAL;
  const uint8_t* p = f.p;
  const uint64_t n = f.n;
  // lines
  std::vector<uint64_t> ls;
  for (uint64_t at = 0; at < n;) {
    ls.push_back(at);
    const void* nl = memchr(p + at, '\n', n - at);
    at = nl ? (uint64_t)((const uint8_t*)nl - p) + 1 : n;
  }
  ls.push_back(n);
  const uint64_t n_lines = ls.size() - 1;
  const unsigned n_thr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, n / (8u << 20) + 1));
  std::vector<std::vector<uint64_t>> th1(n_thr), tpos(n_thr);
  std::vector<std::vector<uint32_t>> tline(n_thr);
  std::vector<int> bad(n_thr, 0);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < n_thr; ++t)
    pool.emplace_back([&, t] {
      const uint64_t l0 = n_lines * t / n_thr, l1 = n_lines * (t + 1) / n_thr;
      for (uint64_t l = l0; l < l1; ++l) {
        const uint8_t* q = p + ls[l];
        const uint8_t* e = p + ls[l + 1];
        while (e > q && (e[-1] == '\n' || e[-1] == '\r')) --e;
        const uint8_t* tab = (const uint8_t*)memchr(q, '\t', (size_t)(e - q));
        if (!tab) continue; // a record id alone: no minimizers
        q = tab + 1;
        while (q < e) {
          while (q < e && *q == ' ') ++q;
          if (q >= e) break;
          uint64_t h = 0, ps = 0;
          const uint8_t* d0 = q;
          while (q < e && *q >= '0' && *q <= '9') h = h * 10 + (uint64_t)(*q++ - '0');
          if (q == d0 || q >= e || *q != ':') { // ntJoin needs hash:pos
            bad[t] = 1;
            return;
          }
          ++q;
          d0 = q;
          while (q < e && *q >= '0' && *q <= '9') ps = ps * 10 + (uint64_t)(*q++ - '0');
          if (q == d0) {
            bad[t] = 1;
            return;
          }
          while (q < e && *q != ' ') ++q; // (:KMER)
          th1[t].push_back(h);
          tpos[t].push_back(ps);
          tline[t].push_back((uint32_t)l);
        }
      }
    });
  for (auto& th : pool) th.join();
  for (int b : bad)
    if (b) return NTS_EFORMAT;
  uint64_t tot = 0, name_bytes = 0;
  for (unsigned t = 0; t < n_thr; ++t) tot += th1[t].size();
  for (uint64_t l = 0; l < n_lines; ++l) {
    const uint8_t* q = p + ls[l];
    const uint8_t* e = p + ls[l + 1];
    const uint8_t* tab = (const uint8_t*)memchr(q, '\t', (size_t)(e - q));
    const uint8_t* ne = tab ? tab : e;
    while (ne > q && (ne[-1] == '\n' || ne[-1] == '\r')) --ne;
    name_bytes += (uint64_t)(ne - q) + 1;
  }
  out->n_lines = n_lines;
  out->n = tot;
  out->names = (char*)malloc(std::max<uint64_t>(name_bytes, 1));
  out->h1 = (uint64_t*)malloc(std::max<uint64_t>(tot, 1) * 8);
  out->pos = (uint64_t*)malloc(std::max<uint64_t>(tot, 1) * 8);
  out->line = (uint32_t*)malloc(std::max<uint64_t>(tot, 1) * 4);
  if (!out->names || !out->h1 || !out->pos || !out->line) {
    nts_mx_tsv_free(out);
    return NTS_ENOMEM;
  }
  out->names_bytes = name_bytes;
  char* w = out->names;
  for (uint64_t l = 0; l < n_lines; ++l) {
    const uint8_t* q = p + ls[l];
    const uint8_t* e = p + ls[l + 1];
    const uint8_t* tab = (const uint8_t*)memchr(q, '\t', (size_t)(e - q));
    const uint8_t* ne = tab ? tab : e;
    while (ne > q && (ne[-1] == '\n' || ne[-1] == '\r')) --ne;
    memcpy(w, q, (size_t)(ne - q));
    w += ne - q;
    *w++ = 0;
  }
  uint64_t at = 0;
  for (unsigned t = 0; t < n_thr; ++t) {
    const size_t m = th1[t].size();
    if (m) {
      memcpy(out->h1 + at, th1[t].data(), m * 8);
      memcpy(out->pos + at, tpos[t].data(), m * 8);
      memcpy(out->line + at, tline[t].data(), m * 4);
    }
    at += m;
  }
  return NTS_OK;
}

extern "C" void nts_mx_tsv_free(nts_mx_tsv* t)
{
  if (!t) return;
  free(t->names);
  free(t->h1);
  free(t->pos);
  free(t->line);
  memset(t, 0, sizeof(*t));
}

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
  static const char* const e = getenv("NTS_HOST_THREADS"); // (an upper limit set by the user, e.g. to leave cores to other work; read once)
  if (e) {
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
  const bool debug = NTS_KNOB("NTS_HOST_DEBUG") != nullptr;
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

// ---- block-level rules of the graph stage as array passes (rows C3, C10, C12-merge) ---------------------------------------------
// The HBM-resident engine (nts_engine_*) returns tables with a few numbers per block / candidate edge; at 10^5 blocks the
// reference's per-object Python over them (bin/ntsynt_synteny.py:428-472, 566-590; bin/synteny_block.py:72-109) is what is left
// of the stage's wall clock.  These helpers are the same rules over flat arrays, sequential where the reference's rule is
// order dependent.

// run_graph_simplification (bin/ntsynt_synteny.py:566-590) over the table nts_engine_bubbles returns: candidate edges in
// ascending order (= the reference's edge order), and every live edge incident to one of their end points (ascending too).
// For a candidate (s, t): if s and t each have exactly one incident edge of weight wmax and exactly one common neighbour m,
// then m is doomed and the candidate's weight becomes wmax -- which later candidates see (S:586).
extern "C" int nts_bubble_rule(uint64_t n_cand, const uint32_t* cand_edge, uint64_t n_inc, const uint32_t* inc_edge, const uint32_t* inc_u,
                               const uint32_t* inc_v, const uint32_t* inc_w, uint32_t wmax, uint32_t* doomed, uint32_t* promoted,
                               uint64_t* n_out)
{
  if (!n_out || (n_cand && (!cand_edge || !doomed || !promoted)) || (n_inc && (!inc_edge || !inc_u || !inc_v || !inc_w))) return NTS_EINVAL;
  *n_out = 0;
  if (n_cand == 0 || n_inc == 0) return NTS_OK;
  // vertices of the table, numbered densely in ascending order.  Vertex ids are indices into the engine's vertex table: a direct table
  // id -> dense number (presence marks, then a running count) unless the ids are far sparser than the table is long -- sorting 2 n_inc ids
  // and a binary search per end point were 20 of this function's 25 seconds on a table of 12.8 M edges (short windows on Gbp genomes:
  // 3 M candidate edges per round).
  std::vector<uint32_t> verts, dense;
  uint32_t vmax = 0;
  for (uint64_t i = 0; i < n_inc; ++i) vmax = std::max(vmax, std::max(inc_u[i], inc_v[i]));
  const bool direct = (uint64_t)vmax < 64ull * n_inc + (1ull << 20);
  if (direct) {
    dense.assign((size_t)vmax + 1, 0u);
    for (uint64_t i = 0; i < n_inc; ++i) dense[inc_u[i]] = dense[inc_v[i]] = 1u;
    uint32_t count = 0;
    for (size_t v = 0; v <= vmax; ++v)
      if (dense[v]) {
        dense[v] = count++;
        verts.push_back((uint32_t)v);
      }
  } else {
    verts.reserve(2 * n_inc);
    for (uint64_t i = 0; i < n_inc; ++i) {
      verts.push_back(inc_u[i]);
      verts.push_back(inc_v[i]);
    }
    std::sort(verts.begin(), verts.end());
    verts.erase(std::unique(verts.begin(), verts.end()), verts.end());
  }
  auto vid = [&](uint32_t v) { return direct ? dense[v] : (uint32_t)(std::lower_bound(verts.begin(), verts.end(), v) - verts.begin()); };
  const size_t nv = verts.size();
  // adjacency in CSR form; a later edge between the same pair replaces the earlier one (dict assignment), which cannot happen
  // among live edges but is kept
  std::vector<uint32_t> deg(nv + 1, 0), lu(n_inc), lv(n_inc);
  for (uint64_t i = 0; i < n_inc; ++i) {
    lu[i] = vid(inc_u[i]);
    lv[i] = vid(inc_v[i]);
    ++deg[lu[i] + 1];
    ++deg[lv[i] + 1];
  }
  for (size_t v = 0; v < nv; ++v) deg[v + 1] += deg[v];
  std::vector<uint32_t> fill(deg.begin(), deg.end() - 1), nb(2 * n_inc), ne(2 * n_inc);
  for (uint64_t i = 0; i < n_inc; ++i) {
    nb[fill[lu[i]]] = lv[i];
    ne[fill[lu[i]]++] = (uint32_t)i;
    nb[fill[lv[i]]] = lu[i];
    ne[fill[lv[i]]++] = (uint32_t)i;
  }
  std::vector<uint32_t> w(inc_w, inc_w + n_inc);
  auto full_edges = [&](uint32_t v) {
    uint32_t c = 0;
    for (uint32_t q = deg[v]; q < deg[v + 1]; ++q) c += w[ne[q]] == wmax;
    return c;
  };
  uint64_t out = 0;
  const uint32_t* at = inc_edge; // (both lists ascend: the look-up walks on from the candidate before)
  for (uint64_t c = 0; c < n_cand; ++c) {
    if (c && cand_edge[c] < cand_edge[c - 1]) at = inc_edge; // (not ascending after all: search from the start)
    at = std::lower_bound(at, inc_edge + n_inc, cand_edge[c]);
    if (at == inc_edge + n_inc || *at != cand_edge[c]) return NTS_EINVAL; // a candidate is incident to its own end points
    const uint64_t i = (uint64_t)(at - inc_edge);
    const uint32_t s = lu[i], t = lv[i];
    if (full_edges(s) != 1 || full_edges(t) != 1) continue;
    uint32_t n_common = 0, common = 0;
    for (uint32_t q = deg[s]; q < deg[s + 1]; ++q) {
      const uint32_t u = nb[q];
      if (u == t) continue;
      bool also = false;
      for (uint32_t r = deg[t]; r < deg[t + 1]; ++r) also |= nb[r] == u;
      if (also) {
        ++n_common;
        common = u;
      }
    }
    if (n_common != 1) continue;
    doomed[out] = verts[common];
    promoted[out] = cand_edge[c];
    ++out;
    w[i] = wmax;
  }
  *n_out = out;
  return NTS_OK;
}

namespace {
inline int64_t blk_start(const int64_t* f, const int64_t* l, uint64_t i) { return std::min(f[i], l[i]); }
inline int64_t blk_end(const int64_t* f, const int64_t* l, uint64_t i, int64_t k) { return std::max(f[i], l[i]) + k; }
} // namespace

// merge_collinear_blocks (bin/ntsynt_synteny.py:428-472) over blocks sorted like SyntenyBlock.__lt__: tables are [a * n + b]
// (assembly-major); ori codes 0 '+', 1 '-'; reason codes 0 None, 1 id_change, 2 ori_change, 3 inconsistent_order, 4 indel,
// 5 merge.  Works in place: the first *n_out blocks of every table are the result (a merged block keeps its first block's
// contigs, orientations, first positions and reason, takes the last block's last positions and the sum of the counts).
extern "C" int nts_blocks_merge(uint32_t G, uint64_t n, int64_t k, int64_t bp, int64_t collinear_merge, uint32_t* rec, int64_t* first,
                                int64_t* last, uint8_t* ori, int64_t* n_mx, uint8_t* reason, uint64_t* n_out, uint64_t* n_merged)
{
  if (!n_out || G == 0 || (n && (!rec || !first || !last || !ori || !n_mx || !reason))) return NTS_EINVAL;
  uint64_t merged = 0;
  if (n == 0) {
    *n_out = 0;
    if (n_merged) *n_merged = 0;
    return NTS_OK;
  }
  // cur = slot `o` (being built in place: o <= b always, so nothing unread is overwritten)
  uint64_t o = 0;
  for (uint64_t b = 1; b < n; ++b) {
    bool same_ori = true, same_ctg = true, negative = false;
    int64_t d_min = INT64_MAX, d_max = INT64_MIN;
    for (uint32_t a = 0; a < G; ++a) {
      const uint64_t ic = (uint64_t)a * n + o, ib = (uint64_t)a * n + b;
      same_ori &= ori[ic] == ori[ib];
      same_ctg &= rec[ic] == rec[ib];
      const int64_t d = (ori[ic] == 1 && ori[ib] == 1) ? blk_start(first, last, ic) - blk_end(first, last, ib, k)
                                                       : blk_start(first, last, ib) - blk_end(first, last, ic, k);
      d_min = std::min(d_min, d);
      d_max = std::max(d_max, d);
      negative |= d < 0;
    }
    const int64_t spread = d_max - d_min;
    if (!same_ori || !same_ctg || spread > bp - k || d_max >= collinear_merge) {
      uint8_t why = reason[b];
      if (!same_ctg)
        why = 1;
      else if (!same_ori)
        why = 2;
      else if (negative)
        why = 3;
      else if (spread > bp - k)
        why = 4;
      else if (d_max >= collinear_merge)
        why = 5;
      ++o;
      for (uint32_t a = 0; a < G; ++a) {
        const uint64_t io = (uint64_t)a * n + o, ib = (uint64_t)a * n + b;
        rec[io] = rec[ib];
        first[io] = first[ib];
        last[io] = last[ib];
        ori[io] = ori[ib];
      }
      n_mx[o] = n_mx[b];
      reason[o] = why;
    } else {
      for (uint32_t a = 0; a < G; ++a) last[(uint64_t)a * n + o] = last[(uint64_t)a * n + b]; // minimizers.extend(): only ends and count matter
      n_mx[o] += n_mx[b];
      ++merged;
    }
  }
  *n_out = o + 1;
  if (n_merged) *n_merged = merged;
  return NTS_OK;
}

// get_block_string (bin/synteny_block.py:72-85) for a whole table: blocks shorter than z in any assembly are skipped, the others
// numbered from 0; rows of a block in `out_order` (assemblies by name); with `reason` the verbose column.  names: NUL-separated
// strings -- assembly names [G], then the contig names of assembly 0, 1, ...; contig_base[a] = index of assembly a's first contig
// name among them.  Returns a malloc'ed buffer (nts_free).
extern "C" int nts_blocks_text(uint32_t G, uint64_t n, int64_t k, int64_t z, const uint32_t* out_order, const char* names, uint64_t names_bytes,
                               const uint64_t* contig_base, const uint32_t* rec, const int64_t* first, const int64_t* last, const uint8_t* ori,
                               const int64_t* n_mx, const uint8_t* reason, char** text, uint64_t* text_bytes)
{
  if (!text || !text_bytes || G == 0 || !out_order || !names || !contig_base || (n && (!rec || !first || !last || !ori || !n_mx))) return NTS_EINVAL;
  std::vector<const char*> str;
  for (uint64_t i = 0; i < names_bytes;) {
    str.push_back(names + i);
    i += strlen(names + i) + 1;
  }
  static const char* const REASON[6] = { "None", "id_change", "ori_change", "inconsistent_order", "indel", "merge" };
  std::string out;
  out.reserve(n * G * 48);
  char num[32];
  uint64_t id = 0;
  for (uint64_t b = 0; b < n; ++b) {
    bool keep = true;
    for (uint32_t a = 0; a < G; ++a) {
      const int64_t d = first[(uint64_t)a * n + b] - last[(uint64_t)a * n + b];
      keep &= (d < 0 ? -d : d) >= z - k; // end - start = |first - last| + k
    }
    if (!keep) continue;
    for (uint32_t q = 0; q < G; ++q) {
      const uint32_t a = out_order[q];
      const uint64_t i = (uint64_t)a * n + b;
      const uint64_t ci = G + contig_base[a] + rec[i];
      if (a >= str.size() || ci >= str.size()) return NTS_EINVAL;
      char* e = put_u64(num, id);
      out.append(num, e - num);
      out.push_back('\t');
      out.append(str[a]);
      out.push_back('\t');
      out.append(str[ci]);
      out.push_back('\t');
      e = put_u64(num, (uint64_t)blk_start(first, last, i));
      out.append(num, e - num);
      out.push_back('\t');
      e = put_u64(num, (uint64_t)blk_end(first, last, i, k));
      out.append(num, e - num);
      out.push_back('\t');
      out.push_back(ori[i] == 0 ? '+' : ori[i] == 1 ? '-' : '?');
      out.push_back('\t');
      e = put_u64(num, (uint64_t)n_mx[b]);
      out.append(num, e - num);
      if (reason) {
        out.push_back('\t');
        out.append(REASON[reason[b] < 6 ? reason[b] : 0]);
      }
      out.push_back('\n');
    }
    ++id;
  }
  char* buf = (char*)malloc(std::max<size_t>(out.size(), 1));
  if (!buf) return NTS_ENOMEM;
  memcpy(buf, out.data(), out.size());
  *text = buf;
  *text_bytes = out.size();
  return NTS_OK;
}
