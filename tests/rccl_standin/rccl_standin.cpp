// rccl_standin.cpp -- TEST-ONLY stand-in for the eleven librccl entry points libntsynt_hip.so resolves (csrc/nts_comm.inc
// rccl_api()), so that the product's exchange code (nts_bf_allreduce_and, nts_mx_allgather) can execute with more than
// one rank on a box with ONE GPU: real RCCL refuses two ranks on the same device.  Ranks are processes that share the
// GPU; data travels through files in a directory both see (default /dev/shm): device -> mapped file -> device.
// Correctness only, no performance.  Selected with NTS_RCCL_LIB=<path to this .so>; never loaded otherwise.
//
// Semantics kept from NCCL: point-to-point operations between a pair of ranks match in issue order; operations between
// ncclGroupStart/End complete together (any order of sends and receives inside a group must not deadlock); all
// operations are ordered after the work already queued on the stream they are given (here: the stream is drained, then
// the copies are synchronous).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

constexpr int MAXW = 16;
constexpr uint32_t MAGIC = 0x4e545346; // "NTSF"
constexpr uint64_t PIECE = 64ull << 20; // a message is cut into pieces of this size (bounds the mailbox files)

struct Box
{
  std::atomic<uint64_t> posted, consumed; // messages written by the source / taken by the destination
  std::atomic<uint64_t> bytes;            // size of the message in the box
};

struct Ctl
{
  std::atomic<uint32_t> arrived, departed;
  Box box[MAXW * MAXW]; // [src * MAXW + dst]
};

struct Map
{
  int fd = -1;
  uint8_t* p = nullptr;
  uint64_t len = 0;
};

struct Comm
{
  uint32_t magic = MAGIC;
  int world = 0, rank = 0;
  std::string base; // path prefix of this communicator's files
  Ctl* ctl = nullptr;
  Map out[MAXW], in[MAXW];
};

struct Op
{
  bool send;
  uint8_t* dev;
  uint64_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
  uint64_t done = 0; // bytes moved so far
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local std::string g_err;

size_t type_size(ncclDataType_t t)
{
  switch (t) {
  case ncclInt8: case ncclUint8: return 1;
  case ncclFloat16: case ncclBfloat16: return 2;
  case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
  case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
  default: return 0;
  }
}

bool map_at_least(Map& m, const std::string& path, uint64_t need, bool grow)
{
  if (m.p && m.len >= need) return true;
  if (m.p) munmap(m.p, m.len), m.p = nullptr;
  if (m.fd < 0) m.fd = open(path.c_str(), O_RDWR | O_CREAT, 0600);
  if (m.fd < 0) return false;
  struct stat st;
  if (fstat(m.fd, &st) != 0) return false;
  uint64_t len = (uint64_t)st.st_size;
  if (len < need) {
    if (!grow) return false;
    len = (need + 4095) & ~4095ull;
    if (ftruncate(m.fd, (off_t)len) != 0) return false;
  }
  void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, m.fd, 0);
  if (p == MAP_FAILED) return false;
  m.p = (uint8_t*)p;
  m.len = len;
  return true;
}

std::string box_path(const Comm* c, int src, int dst) { return c->base + "." + std::to_string(src) + "." + std::to_string(dst); }

// one attempt at moving the next piece of op; returns 1 if it moved something, 0 if it has to wait, -1 on error
int try_piece(Op& op)
{
  Comm* c = op.comm;
  const uint64_t len = std::min(PIECE, op.bytes - op.done);
  if (op.send) {
    Box& b = c->ctl->box[c->rank * MAXW + op.peer];
    if (b.posted.load(std::memory_order_acquire) != b.consumed.load(std::memory_order_acquire)) return 0; // box still full
    if (!map_at_least(c->out[op.peer], box_path(c, c->rank, op.peer), std::max<uint64_t>(len, 1), true)) {
      g_err = "stand-in: cannot map the outgoing box";
      return -1;
    }
    if (len && hipMemcpy(c->out[op.peer].p, op.dev + op.done, len, hipMemcpyDeviceToHost) != hipSuccess) {
      g_err = "stand-in: device -> box copy failed";
      return -1;
    }
    b.bytes.store(len, std::memory_order_relaxed);
    b.posted.fetch_add(1, std::memory_order_release);
  } else {
    Box& b = c->ctl->box[op.peer * MAXW + c->rank];
    if (b.posted.load(std::memory_order_acquire) == b.consumed.load(std::memory_order_acquire)) return 0; // nothing there yet
    const uint64_t have = b.bytes.load(std::memory_order_relaxed);
    if (have != len) { // NCCL: matching send and receive carry the same count
      g_err = "stand-in: receive of " + std::to_string(len) + " bytes met a send of " + std::to_string(have);
      return -1;
    }
    if (!map_at_least(c->in[op.peer], box_path(c, op.peer, c->rank), std::max<uint64_t>(len, 1), false)) {
      g_err = "stand-in: cannot map the incoming box";
      return -1;
    }
    if (len && hipMemcpy(op.dev + op.done, c->in[op.peer].p, len, hipMemcpyHostToDevice) != hipSuccess) {
      g_err = "stand-in: box -> device copy failed";
      return -1;
    }
    b.consumed.fetch_add(1, std::memory_order_release);
  }
  op.done += len;
  if (op.bytes == 0) op.done = ~0ull; // a zero-byte message is one empty piece
  return 1;
}

inline bool finished(const Op& op) { return op.bytes == 0 ? op.done == ~0ull : op.done == op.bytes; }

ncclResult_t run_ops(std::vector<Op>& ops)
{
  std::vector<hipStream_t> seen;
  for (const Op& op : ops) {
    bool dup = false;
    for (hipStream_t s : seen) dup = dup || s == op.stream;
    if (!dup) {
      seen.push_back(op.stream);
      if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    }
  }
  const double limit = getenv("NTS_STANDIN_TIMEOUT") ? atof(getenv("NTS_STANDIN_TIMEOUT")) : 120.0;
  auto last = std::chrono::steady_clock::now();
  size_t left = 0;
  for (const Op& op : ops) left += !finished(op);
  while (left) {
    bool moved = false;
    // per (peer, direction) only the first unfinished operation may move: messages of a pair match in issue order
    bool busy_send[MAXW] = {}, busy_recv[MAXW] = {};
    for (Op& op : ops) {
      if (finished(op)) continue;
      bool& busy = op.send ? busy_send[op.peer] : busy_recv[op.peer];
      if (busy) continue;
      busy = true;
      const int r = try_piece(op);
      if (r < 0) return ncclInternalError;
      if (r > 0) {
        moved = true;
        if (finished(op)) --left;
      }
    }
    if (moved)
      last = std::chrono::steady_clock::now();
    else {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - last).count() > limit) {
        g_err = "stand-in: no progress for " + std::to_string(limit) + " s (a peer is missing or the calls do not match)";
        return ncclInternalError;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  return ncclSuccess;
}

Comm* as_comm(ncclComm_t h)
{
  Comm* c = (Comm*)h;
  return c && c->magic == MAGIC ? c : nullptr;
}

ncclResult_t enqueue(bool send, const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t h, hipStream_t stream)
{
  Comm* c = as_comm(h);
  const size_t sz = type_size(t);
  if (!c || !sz || peer < 0 || peer >= c->world || peer == c->rank || (count && !buf)) return ncclInvalidArgument;
  Op op{ send, (uint8_t*)buf, (uint64_t)count * sz, peer, c, stream };
  if (g_depth > 0) {
    g_ops.push_back(op);
    return ncclSuccess;
  }
  std::vector<Op> one{ op };
  return run_ops(one);
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  const char* dir = getenv("NTS_STANDIN_DIR");
  static std::atomic<uint32_t> serial{ 0 };
  const uint64_t now = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof(id->internal), "%s/ntsfake-%d-%u-%llx", dir ? dir : "/dev/shm", (int)getpid(), serial.fetch_add(1),
           (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
  if (!out || world < 1 || world > MAXW || rank < 0 || rank >= world) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  Comm* c = new Comm();
  c->world = world;
  c->rank = rank;
  c->base = id.internal;
  const std::string path = c->base + ".ctl";
  int fd = open(path.c_str(), O_RDWR | O_CREAT, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Ctl)) != 0) { // a new file reads as zeros: every counter starts at 0
    g_err = "stand-in: cannot create " + path;
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->ctl = (Ctl*)p;
  c->ctl->arrived.fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->ctl->arrived.load() < (uint32_t)world) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
      g_err = "stand-in: the other ranks did not arrive";
      return ncclInternalError;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t h)
{
  Comm* c = as_comm(h);
  if (!c) return ncclInvalidArgument;
  for (int r = 0; r < MAXW; ++r)
    for (Map* m : { &c->out[r], &c->in[r] }) {
      if (m->p) munmap(m->p, m->len);
      if (m->fd >= 0) close(m->fd);
    }
  const bool last = c->ctl->departed.fetch_add(1) + 1 == (uint32_t)c->world;
  munmap(c->ctl, sizeof(Ctl));
  if (last) { // the last rank to leave removes the files
    for (int s = 0; s < c->world; ++s)
      for (int d = 0; d < c->world; ++d) unlink(box_path(c, s, d).c_str());
    unlink((c->base + ".ctl").c_str());
  }
  c->magic = 0;
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t h, int* n)
{
  Comm* c = as_comm(h);
  if (!c || !n) return ncclInvalidArgument;
  *n = c->world;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t h, int* r)
{
  Comm* c = as_comm(h);
  if (!c || !r) return ncclInvalidArgument;
  *r = c->rank;
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t h, hipStream_t s) { return enqueue(true, buf, count, t, peer, h, s); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t h, hipStream_t s) { return enqueue(false, buf, count, t, peer, h, s); }

ncclResult_t ncclGroupStart()
{
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t h, hipStream_t stream)
{
  Comm* c = as_comm(h);
  const size_t sz = type_size(t);
  if (!c || !sz || (count && (!send || !recv))) return ncclInvalidArgument;
  const uint64_t bytes = (uint64_t)count * sz;
  uint8_t* mine = (uint8_t*)recv + (uint64_t)c->rank * bytes;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if ((const void*)mine != send && bytes && hipMemcpy(mine, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<Op> ops;
  for (int step = 1; step < c->world; ++step) {
    const int to = (c->rank + step) % c->world, from = (c->rank - step + c->world) % c->world;
    ops.push_back(Op{ true, mine, bytes, to, c, stream });
    ops.push_back(Op{ false, (uint8_t*)recv + (uint64_t)from * bytes, bytes, from, c, stream });
  }
  if (g_depth > 0) {
    g_ops.insert(g_ops.end(), ops.begin(), ops.end());
    return ncclSuccess;
  }
  return run_ops(ops);
}

const char* ncclGetErrorString(ncclResult_t r)
{
  static thread_local std::string text;
  text = "rccl stand-in (tests/rccl_standin): result " + std::to_string((int)r) + (g_err.empty() ? "" : " -- " + g_err);
  return text.c_str();
}

} // extern "C"
