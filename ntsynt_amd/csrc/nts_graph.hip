// libntsynt_hip.so, third translation unit: the minimizer graph (rows C1, C2: nts_graph_build) and the graph stage resident in HBM
// (nts_dgraph.inc: nts_engine_*).  Shared state and helpers: nts_internal.h.
#include "nts_internal.h"

// ---- minimizer graph build (rows C1, C2a, C2b) --------------------------------------------------------
namespace {

// after a stable sort by hash, duplicates of one assembly are adjacent (global element index is
// assembly-major): mark elements whose hash is unique within their assembly and kept by the caller
__global__ __launch_bounds__(256) void k_g_valid(const uint64_t* __restrict__ h_sorted, const uint64_t* __restrict__ idx_sorted,
                                                 const uint32_t* __restrict__ asm_of, const uint8_t* __restrict__ keep, uint64_t n,
                                                 uint8_t* __restrict__ valid)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = h_sorted[i];
  const uint64_t e = idx_sorted[i];
  const uint32_t a = asm_of[e];
  bool dup = false;
  if (i > 0 && h_sorted[i - 1] == h && asm_of[idx_sorted[i - 1]] == a) dup = true;
  if (i + 1 < n && h_sorted[i + 1] == h && asm_of[idx_sorted[i + 1]] == a) dup = true;
  valid[i] = (!dup && keep[e]) ? 1 : 0;
}

// group heads (first element of each run of equal hashes): the hash is common iff exactly n_asm
// valid elements carry it (each assembly contributes at most one)
__global__ __launch_bounds__(256) void k_g_common(const uint64_t* __restrict__ h_sorted, const uint8_t* __restrict__ valid, uint64_t n,
                                                  uint32_t n_asm, uint64_t* __restrict__ head_common)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t flag = 0;
  const uint64_t h = h_sorted[i];
  if (i == 0 || h_sorted[i - 1] != h) {
    uint32_t cnt = 0;
    for (uint64_t j = i; j < n && h_sorted[j] == h; ++j) cnt += valid[j];
    flag = (cnt == n_asm) ? 1 : 0;
  }
  head_common[i] = flag;
}

// vid_scan = exclusive scan of head_common: every valid element of a common group gets the group's id
__global__ __launch_bounds__(256) void k_g_assign(const uint64_t* __restrict__ h_sorted, const uint64_t* __restrict__ idx_sorted,
                                                  const uint8_t* __restrict__ valid, const uint64_t* __restrict__ head_common,
                                                  const uint64_t* __restrict__ vid_scan, uint64_t n, const uint32_t* __restrict__ asm_of,
                                                  const uint32_t* __restrict__ rec, const uint64_t* __restrict__ pos, uint64_t nv,
                                                  uint32_t* __restrict__ elem_vid, uint64_t* __restrict__ v_hash,
                                                  uint32_t* __restrict__ occ_rec, uint64_t* __restrict__ occ_pos)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = h_sorted[i];
  if (!(i == 0 || h_sorted[i - 1] != h) || !head_common[i]) return;
  const uint64_t vid = vid_scan[i];
  v_hash[vid] = h;
  for (uint64_t j = i; j < n && h_sorted[j] == h; ++j) {
    if (!valid[j]) continue;
    const uint64_t e = idx_sorted[j];
    elem_vid[e] = (uint32_t)vid;
    const uint32_t a = asm_of[e];
    occ_rec[(uint64_t)a * nv + vid] = rec[e];
    occ_pos[(uint64_t)a * nv + vid] = pos[e];
  }
}

__global__ __launch_bounds__(256) void k_g_flag_kept(const uint32_t* __restrict__ elem_vid, uint64_t n, uint64_t* __restrict__ flag)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = elem_vid[i] != 0xFFFFFFFFu ? 1 : 0;
}

// compact the survivors in traversal order: c_vid / c_asm / c_list
__global__ __launch_bounds__(256) void k_g_compact(const uint32_t* __restrict__ elem_vid, const uint64_t* __restrict__ where, uint64_t n,
                                                   const uint32_t* __restrict__ asm_of, const uint32_t* __restrict__ list_id,
                                                   uint32_t* __restrict__ c_vid, uint32_t* __restrict__ c_asm, uint32_t* __restrict__ c_list)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || elem_vid[i] == 0xFFFFFFFFu) return;
  const uint64_t c = where[i];
  c_vid[c] = elem_vid[i];
  c_asm[c] = asm_of[i];
  c_list[c] = list_id[i];
}

// adjacent survivors of one list -> edge occurrence (canonical key, sequence number); others get key ~0
__global__ __launch_bounds__(256) void k_g_pairs(const uint32_t* __restrict__ c_vid, const uint32_t* __restrict__ c_asm,
                                                 const uint32_t* __restrict__ c_list, uint64_t m, uint64_t* __restrict__ key,
                                                 uint64_t* __restrict__ seq)
{
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  uint64_t kk = ~0ULL;
  if (c + 1 < m && c_asm[c] == c_asm[c + 1] && c_list[c] == c_list[c + 1]) {
    const uint64_t u = c_vid[c], v = c_vid[c + 1];
    kk = u < v ? ((u << 32) | v) : ((v << 32) | u);
  }
  key[c] = kk;
  seq[c] = c;
}

__global__ __launch_bounds__(256) void k_g_edge_heads(const uint64_t* __restrict__ key_sorted, uint64_t m, uint64_t* __restrict__ head)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint64_t kk = key_sorted[i];
  head[i] = (kk != ~0ULL && (i == 0 || key_sorted[i - 1] != kk)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_g_edges(const uint64_t* __restrict__ key_sorted, const uint64_t* __restrict__ seq_sorted,
                                                 const uint64_t* __restrict__ head, const uint64_t* __restrict__ head_scan, uint64_t m,
                                                 const uint32_t* __restrict__ c_vid, uint32_t* __restrict__ e_u, uint32_t* __restrict__ e_v,
                                                 uint32_t* __restrict__ e_w, uint64_t* __restrict__ e_first)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || !head[i]) return;
  const uint64_t kk = key_sorted[i];
  uint32_t cnt = 0;
  for (uint64_t j = i; j < m && key_sorted[j] == kk; ++j) ++cnt;
  const uint64_t e = head_scan[i];
  const uint64_t s = seq_sorted[i]; // stable sort: the first sighting leads its group
  e_u[e] = c_vid[s];
  e_v[e] = c_vid[s + 1];
  e_w[e] = cnt;
  e_first[e] = s;
}

// ---- edge order of the reference: `[(s, t) for s in edges for t in edges[s]]` over ntJoin's dict of dicts ----
// sources by the time they first became a source, then by creation time; both are sequence numbers < 2^32
__global__ __launch_bounds__(256) void k_g_src_rank(const uint32_t* __restrict__ e_u, const uint64_t* __restrict__ e_first, uint64_t ne,
                                                    unsigned long long* __restrict__ src_rank)
{
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) atomicMin(&src_rank[e_u[e]], (unsigned long long)e_first[e]);
}

__global__ __launch_bounds__(256) void k_g_order_keys(const uint32_t* __restrict__ e_u, const uint64_t* __restrict__ e_first, uint64_t ne,
                                                      const unsigned long long* __restrict__ src_rank, uint64_t* __restrict__ key,
                                                      uint64_t* __restrict__ idx)
{
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  key[e] = ((uint64_t)src_rank[e_u[e]] << 32) | e_first[e];
  idx[e] = e;
}

// element numbers base .. base+m-1 and the assembly id of one list of the concatenation
__global__ __launch_bounds__(256) void k_g_number(uint64_t* __restrict__ idx, uint32_t* __restrict__ asm_id, uint64_t m, uint64_t base, uint32_t a)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  idx[i] = base + i;
  asm_id[i] = a;
}

__global__ __launch_bounds__(256) void k_g_permute_edges(const uint64_t* __restrict__ idx_sorted, uint64_t ne, const uint32_t* __restrict__ e_u,
                                                         const uint32_t* __restrict__ e_v, const uint32_t* __restrict__ e_w,
                                                         const uint64_t* __restrict__ e_first, uint32_t* __restrict__ o_u,
                                                         uint32_t* __restrict__ o_v, uint32_t* __restrict__ o_w, uint64_t* __restrict__ o_first)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ne) return;
  const uint64_t e = idx_sorted[i];
  o_u[i] = e_u[e];
  o_v[i] = e_v[e];
  o_w[i] = e_w[e];
  o_first[i] = e_first[e];
}

template <typename T>
T* host_copy(nts_ctx* ctx, const T* d, uint64_t n)
{
  T* h = (T*)malloc(std::max<uint64_t>(n, 1) * sizeof(T));
  if (h && n) hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream);
  return h;
}

} // namespace

namespace {

// Device-side result of one build, in the context's scratch (valid until the next build on this context)
struct GraphDev
{
  uint64_t n = 0;   // elements given
  uint64_t nv = 0, ne = 0;
  uint64_t* v_hash = nullptr; // [nv] ascending
  uint32_t* occ_rec = nullptr; // [n_asm * nv]
  uint64_t* occ_pos = nullptr;
  uint32_t *e_u = nullptr, *e_v = nullptr, *e_w = nullptr; // [ne] dict order
  uint64_t* e_first = nullptr;
};

// Hook between duplicate removal and the cross-assembly intersection: given valid[e] per element (in element order),
// a caller may rewrite the list ids (refinement rounds cut lists between consecutive *kept* minimizers, row C11).
struct ListHook
{
  virtual int operator()(nts_ctx* ctx, uint64_t n, const uint8_t* d_valid_elem, const uint32_t* d_asm, const uint32_t* d_rec, const uint64_t* d_pos,
                         uint32_t* d_list) = 0;
  virtual ~ListHook() {}
};

__global__ __launch_bounds__(256) void k_g_valid_scatter(const uint64_t* __restrict__ idx_sorted, const uint8_t* __restrict__ valid, uint64_t n,
                                                         uint8_t* __restrict__ valid_elem)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) valid_elem[idx_sorted[i]] = valid[i];
}

// The build proper.  Expects the concatenated elements (assembly-major) already in the scratch buffers g_h / g_rec / g_pos /
// g_keep / g_list / g_asm / g_idx (filled by the callers below); leaves the graph in scratch and describes it in `G`.
int graph_build_core(nts_ctx* ctx, uint32_t n_asm, uint64_t n, GraphDev* G, ListHook* hook)
{
  *G = GraphDev();
  G->n = n;
  if (n == 0) return NTS_OK;
#define G_WS(ptr, type, name, bytes)                                                                \
  type ptr = (type)ws_get(ctx, name, bytes);                                                        \
  if (!ptr) return NTS_ENOMEM
  G_WS(d_h, uint64_t*, "g_h", n * 8);
  G_WS(d_idx, uint64_t*, "g_idx", n * 8);
  G_WS(d_h2, uint64_t*, "g_h2", n * 8);
  G_WS(d_idx2, uint64_t*, "g_idx2", n * 8);
  G_WS(d_asm, uint32_t*, "g_asm", n * 4);
  G_WS(d_rec, uint32_t*, "g_rec", n * 4);
  G_WS(d_pos, uint64_t*, "g_pos", n * 8);
  G_WS(d_keep, uint8_t*, "g_keep", n);
  G_WS(d_list, uint32_t*, "g_list", n * 4);
  G_WS(d_valid, uint8_t*, "g_valid", n);
  G_WS(d_flag, uint64_t*, "g_flag", n * 8);
  G_WS(d_scan, uint64_t*, "g_scan", (n + 1) * 8);
  G_WS(d_evid, uint32_t*, "g_evid", n * 4);
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  size_t tmp_sort = 0, tmp_scan = 0;
  HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_sort, d_h, d_h2, d_idx, d_idx2, n, 0, 64, ctx->stream));
  HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, tmp_scan, d_flag, d_scan, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
  G_WS(d_tmp, void*, "g_tmp", std::max<size_t>(std::max(tmp_sort, tmp_scan), 16));
  {
    ScopedTimer t(ctx, "graph_build");
    // C1 + keep mask
    HIP_TRY(ctx, rocprim::radix_sort_pairs(d_tmp, tmp_sort, d_h, d_h2, d_idx, d_idx2, n, 0, 64, ctx->stream));
    NTS_LAUNCH(k_g_valid, dim3(nb), dim3(256), 0, ctx->stream, d_h2, d_idx2, d_asm, d_keep, n, d_valid);
  }
  if (hook) {
    G_WS(d_valid_elem, uint8_t*, "g_valid_elem", n);
    NTS_LAUNCH(k_g_valid_scatter, dim3(nb), dim3(256), 0, ctx->stream, d_idx2, d_valid, n, d_valid_elem);
    if (int rc = (*hook)(ctx, n, d_valid_elem, d_asm, d_rec, d_pos, d_list)) return rc;
  }
  {
    ScopedTimer t(ctx, "graph_build");
    // C2a
    NTS_LAUNCH(k_g_common, dim3(nb), dim3(256), 0, ctx->stream, d_h2, d_valid, n, n_asm, d_flag);
    HIP_TRY(ctx, rocprim::exclusive_scan(d_tmp, tmp_scan, d_flag, d_scan, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
  }
  uint64_t last_flag = 0, last_scan = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&last_flag, d_flag + (n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&last_scan, d_scan + (n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const uint64_t nv = last_scan + last_flag;
  G->nv = nv;
  if (nv == 0) return NTS_OK;
  G_WS(d_vhash, uint64_t*, "g_vhash", nv * 8);
  G_WS(d_orec, uint32_t*, "g_orec", (uint64_t)n_asm * nv * 4);
  G_WS(d_opos, uint64_t*, "g_opos", (uint64_t)n_asm * nv * 8);
  HIP_TRY(ctx, hipMemsetAsync(d_evid, 0xFF, n * 4, ctx->stream));
  {
    ScopedTimer t(ctx, "graph_build");
    NTS_LAUNCH(k_g_assign, dim3(nb), dim3(256), 0, ctx->stream, d_h2, d_idx2, d_valid, d_flag, d_scan, n, d_asm, d_rec, d_pos, nv,
                       d_evid, d_vhash, d_orec, d_opos);
    // survivors in traversal order
    NTS_LAUNCH(k_g_flag_kept, dim3(nb), dim3(256), 0, ctx->stream, d_evid, n, d_flag);
    HIP_TRY(ctx, rocprim::exclusive_scan(d_tmp, tmp_scan, d_flag, d_scan, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
  }
  const uint64_t m = (uint64_t)n_asm * nv; // every common hash occurs once per assembly
  G_WS(d_cvid, uint32_t*, "g_cvid", (m + 1) * 4);
  G_WS(d_casm, uint32_t*, "g_casm", m * 4);
  G_WS(d_clist, uint32_t*, "g_clist", m * 4);
  G_WS(d_key, uint64_t*, "g_key", m * 8);
  G_WS(d_seq, uint64_t*, "g_seq", m * 8);
  G_WS(d_key2, uint64_t*, "g_key2", m * 8);
  G_WS(d_seq2, uint64_t*, "g_seq2", m * 8);
  G_WS(d_eh, uint64_t*, "g_eh", m * 8);
  G_WS(d_es, uint64_t*, "g_es", (m + 1) * 8);
  size_t tmp_sort2 = 0, tmp_scan2 = 0;
  HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_sort2, d_key, d_key2, d_seq, d_seq2, m, 0, 64, ctx->stream));
  HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, tmp_scan2, d_eh, d_es, (uint64_t)0, m, rocprim::plus<uint64_t>(), ctx->stream));
  G_WS(d_tmp2, void*, "g_tmp2", std::max<size_t>(std::max(tmp_sort2, tmp_scan2), 16));
  const uint32_t mb = (uint32_t)((m + 255) / 256);
  {
    ScopedTimer t(ctx, "graph_build");
    NTS_LAUNCH(k_g_compact, dim3(nb), dim3(256), 0, ctx->stream, d_evid, d_scan, n, d_asm, d_list, d_cvid, d_casm, d_clist);
    NTS_LAUNCH(k_g_pairs, dim3(mb), dim3(256), 0, ctx->stream, d_cvid, d_casm, d_clist, m, d_key, d_seq);
    HIP_TRY(ctx, rocprim::radix_sort_pairs(d_tmp2, tmp_sort2, d_key, d_key2, d_seq, d_seq2, m, 0, 64, ctx->stream));
    NTS_LAUNCH(k_g_edge_heads, dim3(mb), dim3(256), 0, ctx->stream, d_key2, m, d_eh);
    HIP_TRY(ctx, rocprim::exclusive_scan(d_tmp2, tmp_scan2, d_eh, d_es, (uint64_t)0, m, rocprim::plus<uint64_t>(), ctx->stream));
  }
  HIP_TRY(ctx, hipMemcpyAsync(&last_flag, d_eh + (m - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&last_scan, d_es + (m - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const uint64_t ne = last_scan + last_flag;
  G->ne = ne;
  G_WS(d_eu, uint32_t*, "g_eu", std::max<uint64_t>(ne, 1) * 4);
  G_WS(d_ev, uint32_t*, "g_ev", std::max<uint64_t>(ne, 1) * 4);
  G_WS(d_ew, uint32_t*, "g_ew", std::max<uint64_t>(ne, 1) * 4);
  G_WS(d_ef, uint64_t*, "g_ef", std::max<uint64_t>(ne, 1) * 8);
  if (ne) {
    // the unordered edge arrays reuse buffers the pair stage is done with; d_key/d_seq become sort keys again
    G_WS(d_eu0, uint32_t*, "g_eu0", ne * 4);
    G_WS(d_ev0, uint32_t*, "g_ev0", ne * 4);
    G_WS(d_ew0, uint32_t*, "g_ew0", ne * 4);
    G_WS(d_ef0, uint64_t*, "g_ef0", ne * 8);
    G_WS(d_srank, unsigned long long*, "g_srank", nv * 8);
    size_t tmp_sort3 = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_sort3, d_key, d_key2, d_seq, d_seq2, ne, 0, 64, ctx->stream));
    G_WS(d_tmp3, void*, "g_tmp3", std::max<size_t>(tmp_sort3, 16));
    const uint32_t eb = (uint32_t)((ne + 255) / 256);
    ScopedTimer t(ctx, "graph_build");
    NTS_LAUNCH(k_g_edges, dim3(mb), dim3(256), 0, ctx->stream, d_key2, d_seq2, d_eh, d_es, m, d_cvid, d_eu0, d_ev0, d_ew0, d_ef0);
    HIP_TRY(ctx, hipMemsetAsync(d_srank, 0xFF, nv * 8, ctx->stream));
    NTS_LAUNCH(k_g_src_rank, dim3(eb), dim3(256), 0, ctx->stream, d_eu0, d_ef0, ne, d_srank);
    NTS_LAUNCH(k_g_order_keys, dim3(eb), dim3(256), 0, ctx->stream, d_eu0, d_ef0, ne, d_srank, d_key, d_seq);
    HIP_TRY(ctx, rocprim::radix_sort_pairs(d_tmp3, tmp_sort3, d_key, d_key2, d_seq, d_seq2, ne, 0, 64, ctx->stream));
    NTS_LAUNCH(k_g_permute_edges, dim3(eb), dim3(256), 0, ctx->stream, d_seq2, ne, d_eu0, d_ev0, d_ew0, d_ef0, d_eu, d_ev, d_ew, d_ef);
  }
  HIP_TRY(ctx, hipGetLastError());
  G->v_hash = d_vhash;
  G->occ_rec = d_orec;
  G->occ_pos = d_opos;
  G->e_u = d_eu;
  G->e_v = d_ev;
  G->e_w = d_ew;
  G->e_first = d_ef;
  return NTS_OK;
#undef G_WS
}

// scratch buffers of the concatenation, sized for n elements
struct GraphIn
{
  uint64_t* h = nullptr;
  uint64_t* idx = nullptr;
  uint32_t* asm_id = nullptr;
  uint32_t* rec = nullptr;
  uint64_t* pos = nullptr;
  uint8_t* keep = nullptr;
  uint32_t* list = nullptr;
};

int graph_inputs(nts_ctx* ctx, uint64_t n, GraphIn* in)
{
  const uint64_t c = std::max<uint64_t>(n, 1);
  in->h = (uint64_t*)ws_get(ctx, "g_h", c * 8);
  in->idx = (uint64_t*)ws_get(ctx, "g_idx", c * 8);
  in->asm_id = (uint32_t*)ws_get(ctx, "g_asm", c * 4);
  in->rec = (uint32_t*)ws_get(ctx, "g_rec", c * 4);
  in->pos = (uint64_t*)ws_get(ctx, "g_pos", c * 8);
  in->keep = (uint8_t*)ws_get(ctx, "g_keep", c);
  in->list = (uint32_t*)ws_get(ctx, "g_list", c * 4);
  return (in->h && in->idx && in->asm_id && in->rec && in->pos && in->keep && in->list) ? NTS_OK : NTS_ENOMEM;
}

} // namespace

extern "C" int nts_graph_build(nts_ctx* ctx, uint32_t n_asm, const nts_mxlist* lists, nts_graph* out)
{
  if (!ctx || !out || n_asm == 0 || !lists) return fail(ctx, NTS_EINVAL, "nts_graph_build: bad arguments");
  memset(out, 0, sizeof(*out));
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t n = 0;
  for (uint32_t a = 0; a < n_asm; ++a) {
    if (lists[a].n && (!lists[a].h1 || !lists[a].rec || !lists[a].pos)) return fail(ctx, NTS_EINVAL, "nts_graph_build: NULL list arrays");
    n += lists[a].n;
  }
  if (n >= 0xFFFFFFFFULL) return fail(ctx, NTS_ERANGE, "nts_graph_build: more than 2^32 minimizers");
  auto finish_empty = [&]() {
    out->v_hash = (uint64_t*)malloc(8);
    out->occ_rec = (uint32_t*)malloc(8);
    out->occ_pos = (uint64_t*)malloc(8);
    out->e_u = (uint32_t*)malloc(8);
    out->e_v = (uint32_t*)malloc(8);
    out->e_w = (uint32_t*)malloc(8);
    out->e_first = (uint64_t*)malloc(8);
    return NTS_OK;
  };
  if (n == 0) return finish_empty();
  GraphIn in;
  if (graph_inputs(ctx, n, &in) != NTS_OK) return NTS_ENOMEM;
  // assembly-major concatenation; element numbers and assembly ids are generated on the device (each assembly's range is
  // one launch), the keep mask is uploaded only where a list brings one
  uint64_t o = 0;
  for (uint32_t a = 0; a < n_asm; ++a) {
    const uint64_t m = lists[a].n;
    if (m) {
      HIP_TRY(ctx, hipMemcpyAsync(in.h + o, lists[a].h1, m * 8, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(in.rec + o, lists[a].rec, m * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(in.pos + o, lists[a].pos, m * 8, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(in.list + o, lists[a].list_id ? lists[a].list_id : lists[a].rec, m * 4, hipMemcpyHostToDevice, ctx->stream));
      if (lists[a].keep)
        HIP_TRY(ctx, hipMemcpyAsync(in.keep + o, lists[a].keep, m, hipMemcpyHostToDevice, ctx->stream));
      else
        HIP_TRY(ctx, hipMemsetAsync(in.keep + o, 1, m, ctx->stream));
      NTS_LAUNCH(k_g_number, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, ctx->stream, in.idx + o, in.asm_id + o, m, o, a);
    }
    o += m;
  }
  GraphDev G;
  if (int rc = graph_build_core(ctx, n_asm, n, &G, nullptr)) return rc;
  const uint64_t nv = G.nv, ne = G.ne;
  out->nv = nv;
  out->ne = ne;
  if (nv == 0) return finish_empty();
  out->v_hash = host_copy(ctx, G.v_hash, nv);
  out->occ_rec = host_copy(ctx, G.occ_rec, (uint64_t)n_asm * nv);
  out->occ_pos = host_copy(ctx, G.occ_pos, (uint64_t)n_asm * nv);
  out->e_u = host_copy(ctx, G.e_u, ne);
  out->e_v = host_copy(ctx, G.e_v, ne);
  out->e_w = host_copy(ctx, G.e_w, ne);
  out->e_first = host_copy(ctx, G.e_first, ne);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (!out->v_hash || !out->occ_rec || !out->occ_pos || !out->e_u || !out->e_v || !out->e_w || !out->e_first) {
    nts_graph_free(out);
    return fail(ctx, NTS_ENOMEM, "nts_graph_build: host allocation failed");
  }
  return NTS_OK;
}

#include "nts_dgraph.inc"

extern "C" void nts_graph_free(nts_graph* g)
{
  if (!g) return;
  free(g->v_hash);
  free(g->occ_rec);
  free(g->occ_pos);
  free(g->e_u);
  free(g->e_v);
  free(g->e_w);
  free(g->e_first);
  memset(g, 0, sizeof(*g));
}
