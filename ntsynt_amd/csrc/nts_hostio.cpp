// Host-side I/O of libntsynt_hip.so (no GPU work): FASTA ingest, `.fai` columns, indexlr-format
// minimizer TSV writer.  Stands in for btllib::SeqReader (src/ntsynt_make_common_bf.cpp:32-36,125-131),
// `samtools faidx` (bin/ntsynt_run_pipeline.smk:48-53) and indexlr's output stage (smk:81-85).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
  std::vector<uint8_t> data;
  if (!slurp(path, data)) return NTS_EINVAL;
  const size_t n = data.size();
  uint8_t* seq = (uint8_t*)malloc(std::max<size_t>(n, 1));
  if (!seq) return NTS_ENOMEM;
  std::vector<uint64_t> rec_off, rec_len, fai_off;
  std::vector<uint32_t> fai_bases, fai_width;
  std::string names;
  size_t w = 0; // write cursor in seq
  size_t i = 0;
  bool in_record = false;
  bool first_line = false;
  while (i < n) {
    const uint8_t* nl = (const uint8_t*)memchr(data.data() + i, '\n', n - i);
    const size_t line_end = nl ? (size_t)(nl - data.data()) : n; // exclusive, without the newline
    if (data[i] == '>') {
      if (in_record) rec_len.back() = w - rec_off.back();
      // record id = header up to the first whitespace
      size_t s = i + 1, e = s;
      while (e < line_end && data[e] != ' ' && data[e] != '\t' && data[e] != '\r' && data[e] != '\v' && data[e] != '\f') ++e;
      names.append((const char*)data.data() + s, e - s);
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
      if (memchr(data.data() + i, '\r', len) == nullptr) {
        memcpy(seq + w, data.data() + i, len);
        w += len;
      } else {
        for (size_t q = i; q < line_end; ++q)
          if (data[q] != '\r') seq[w++] = data[q];
      }
    }
    i = nl ? line_end + 1 : n;
  }
  if (in_record) rec_len.back() = w - rec_off.back();
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

// `indexlr --long --pos [--seq]`: one line per record, "id \t hash:pos[:KMER] hash:pos[:KMER] ...\n"
extern "C" int nts_write_indexlr_tsv(const char* path, const nts_fasta* fa, const uint64_t* h1, const uint32_t* rec, const uint64_t* pos,
                                     uint64_t n, uint32_t k, int with_seq)
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
        const uint8_t* s = fa->seq + fa->rec_off[r] + pos[i];
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
