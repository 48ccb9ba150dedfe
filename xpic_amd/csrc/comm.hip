// comm.hip -- the exchange layer of the z-slab decomposition (replaces the reference's MPI / PetscSF traffic,
// SURVEY.md section 2.2 C1-C11): ring send/receive with the two z-neighbours and a sum all-reduce.
//   kind 1: RCCL over xGMI, straight on device buffers, on the context's stream (production);
//   kind 2: host callbacks (buffers staged through pinned host memory) -- lets tests drive two ranks that
//           share one GPU through torch.distributed/gloo;
//   kind 0: single rank, every call is a no-op.
#include <rccl/rccl.h>

#include <cstring>

#include <cstdlib>

#include <unistd.h>

#include "common.h"

namespace xpic {

#define XPIC_NCCL(call)                                                                              \
  do {                                                                                               \
    ncclResult_t r_ = (call);                                                                        \
    if (r_ != ncclSuccess) {                                                                         \
      set_error(std::string(#call) + ": " + ncclGetErrorString(r_) + " at " __FILE__ ":" + std::to_string(__LINE__)); \
      return 7;                                                                                      \
    }                                                                                                \
  } while (0)

static int ensure_host(xpic_ctx* c, size_t bytes)
{
  if (c->comm.host_bytes >= bytes) return 0;
  for (int i = 0; i < 4; ++i) {
    if (c->comm.host[i]) XPIC_HIP(hipHostFree(c->comm.host[i]));
    XPIC_HIP(hipHostMalloc(&c->comm.host[i], bytes));
  }
  c->comm.host_bytes = bytes;
  return 0;
}

// send `down` to rank-1 and `up` to rank+1; receive the up-neighbour's `down` message into from_up and the
// down-neighbour's `up` message into from_down.  All pointers are device pointers; sizes in bytes.
int comm_ring(xpic_ctx* c, const void* down, size_t ndown, const void* up, size_t nup, void* from_up, size_t nfrom_up,
  void* from_down, size_t nfrom_down)
{
  Comm& m = c->comm;
  XPIC_CHECK(m.kind != 0, "comm_ring called on a single-rank context");
  XPIC_CALL(peer_order(c));
  c->peer_exchanges += 1;
  m.sent_msgs += (ndown ? 1 : 0) + (nup ? 1 : 0);
  m.sent_bytes += (int64_t)(ndown + nup);
  const int lo = (m.rank - 1 + m.nranks) % m.nranks, hi = (m.rank + 1) % m.nranks;
  if (m.kind == 1) {
    ncclComm_t nc = (ncclComm_t)m.nccl;
    XPIC_NCCL(ncclGroupStart());
    if (ndown) XPIC_NCCL(ncclSend(down, ndown, ncclChar, lo, nc, c->stream));
    if (nup) XPIC_NCCL(ncclSend(up, nup, ncclChar, hi, nc, c->stream));
    if (nfrom_up) XPIC_NCCL(ncclRecv(from_up, nfrom_up, ncclChar, hi, nc, c->stream));
    if (nfrom_down) XPIC_NCCL(ncclRecv(from_down, nfrom_down, ncclChar, lo, nc, c->stream));
    XPIC_NCCL(ncclGroupEnd());
    return 0;
  }
  size_t mx = ndown;
  if (nup > mx) mx = nup;
  if (nfrom_up > mx) mx = nfrom_up;
  if (nfrom_down > mx) mx = nfrom_down;
  XPIC_CALL(ensure_host(c, mx));
  if (ndown) XPIC_HIP(hipMemcpyAsync(m.host[0], down, ndown, hipMemcpyDeviceToHost, c->stream));
  if (nup) XPIC_HIP(hipMemcpyAsync(m.host[1], up, nup, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  int rc = m.cb.sendrecv(m.cb.user, m.host[0], ndown, m.host[1], nup, m.host[2], nfrom_up, m.host[3], nfrom_down);
  XPIC_CHECK(rc == 0, "comm callback sendrecv failed");
  if (nfrom_up) XPIC_HIP(hipMemcpyAsync(from_up, m.host[2], nfrom_up, hipMemcpyHostToDevice, c->stream));
  if (nfrom_down) XPIC_HIP(hipMemcpyAsync(from_down, m.host[3], nfrom_down, hipMemcpyHostToDevice, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int comm_allreduce_sum(xpic_ctx* c, double* dbuf, int n)
{
  Comm& m = c->comm;
  // every reduction that is an MPI_Allreduce on slabs is counted (on a single slab too): xpic_profile_get("allreduce")
  if (c->profiling) c->prof["allreduce"].launches += 1;
  if (m.kind == 0 || n == 0) return 0;
  XPIC_CALL(peer_order(c));
  m.allreduces += 1;
  m.allreduce_bytes += (int64_t)sizeof(double) * n;
  if (m.kind == 1) {
    XPIC_NCCL(ncclAllReduce(dbuf, dbuf, n, ncclDouble, ncclSum, (ncclComm_t)m.nccl, c->stream));
    return 0;
  }
  XPIC_CALL(ensure_host(c, sizeof(double) * n));
  XPIC_HIP(hipMemcpyAsync(m.host[0], dbuf, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  int rc = m.cb.allreduce_sum(m.cb.user, (double*)m.host[0], n);
  XPIC_CHECK(rc == 0, "comm callback allreduce failed");
  XPIC_HIP(hipMemcpyAsync(dbuf, m.host[0], sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  return 0;
}

int comm_allreduce_sum_host(xpic_ctx* c, double* hbuf, int n)
{
  if (c->comm.kind == 0 || n == 0) return 0;
  XPIC_CHECK(n <= 32, "host all-reduce limited to 32 values");
  double* d = c->red_out + 32;
  XPIC_HIP(hipMemcpyAsync(d, hbuf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  XPIC_CALL(comm_allreduce_sum(c, d, n));
  XPIC_HIP(hipMemcpyAsync(hbuf, d, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// max of one non-negative value over the slabs, through the sum transport: every rank contributes its own slot
int comm_allreduce_max_host(xpic_ctx* c, double* v)
{
  if (c->comm.kind == 0) return 0;
  // (in chunks of 32 ranks -- the host all-reduce's width: any number of slabs, one more small all-reduce per 32 of them)
  double m = 0.0;
  for (int base = 0; base < c->comm.nranks; base += 32) {
    const int n = c->comm.nranks - base < 32 ? c->comm.nranks - base : 32;
    double buf[32] = {};
    if (c->comm.rank >= base && c->comm.rank < base + n) buf[c->comm.rank - base] = *v;
    XPIC_CALL(comm_allreduce_sum_host(c, buf, n));
    for (int i = 0; i < n; ++i) m = buf[i] > m ? buf[i] : m;
  }
  *v = m;
  return 0;
}

// Peer copies run on copy_stream; whatever this rank sends next -- the message that tells a neighbour its ghost rows have
// landed -- is issued on the compute (or communication) stream: make that stream wait for the copies first.
int peer_order(xpic_ctx* c)
{
  if (!c->peer_pending) return 0;
  c->peer_pending = false;
  XPIC_HIP(hipStreamWaitEvent(c->stream, c->copy_ev[1], 0));
  return 0;
}

void comm_free(xpic_ctx* c)
{
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(c->copy_ev[i]);
    (void)hipStreamDestroy(c->copy_stream);
    c->copy_stream = nullptr;
  }
  for (int i = 0; i < 2; ++i) {
    if (c->peer_mapped[i]) (void)hipIpcCloseMemHandle(c->peer_mapped[i]);
    c->peer_mapped[i] = nullptr; c->peer_lrow[i] = nullptr;
  }
  if (c->comm_stream) {
    (void)hipStreamSynchronize(c->comm_stream);
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(c->comm_ev[i]);
    (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
  }
  if (c->comm.kind == 1 && c->comm.nccl) (void)ncclCommDestroy((ncclComm_t)c->comm.nccl);
  for (int i = 0; i < 4; ++i)
    if (c->comm.host[i]) (void)hipHostFree(c->comm.host[i]);
  c->comm = Comm{};
}

}  // namespace xpic

using namespace xpic;

extern "C" {

int xpic_comm_rccl_unique_id(void* id128)
{
  XPIC_CHECK(id128, "null argument");
  ncclUniqueId id;
  XPIC_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int xpic_comm_init_rccl(xpic_ctx* ctx, const void* id128)
{
  XPIC_CHECK(ctx && id128, "null argument");
  XPIC_CHECK(ctx->g.G > 0, "context was created without ghost planes (nranks == 1)");
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t nc;
  XPIC_HIP(hipSetDevice(ctx->geom.device));
  XPIC_NCCL(ncclCommInitRank(&nc, ctx->geom.nranks, id, ctx->geom.rank));
  if (!ctx->comm_stream) {
    // highest priority: when workgroup slots free up, the dispatcher serves the exchange's few workgroups before the
    // next colour launch of the assembly queued behind 512 of its own (at equal priority the RCCL kernels of a posted
    // exchange did not start before the compute stream ran dry: profiles/r04_trace_overlap_selfring.txt)
    int least = 0, greatest = 0;
    XPIC_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    XPIC_HIP(hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, greatest));
    for (int i = 0; i < 2; ++i) XPIC_HIP(hipEventCreateWithFlags(&ctx->comm_ev[i], hipEventDisableTiming));
  }
  ctx->comm.kind = 1;
  ctx->comm.nccl = nc;
  ctx->comm.rank = ctx->geom.rank;
  ctx->comm.nranks = ctx->geom.nranks;
  // The overlapped exchanges (fields.hip: op_apply_overlapped, the matL ghost rows beside the interior colours) put
  // send/recv on comm_stream while all-reduces go on the compute stream, on ONE communicator.  Every rank issues them
  // in the same program order, which is what RCCL asks for, but the only hardware that has run this branch is a
  // one-GPU self-ring: until a run on two distinct GPUs exists, N > 1 over RCCL exchanges first (no second stream)
  // unless the caller asked for the overlap (xpic_set_overlap(ctx, 1) or XPIC_RCCL_OVERLAP=1).
  if (ctx->geom.nranks > 1 && !ctx->overlap_explicit) {
    const char* e = getenv("XPIC_RCCL_OVERLAP");
    ctx->overlap = e && e[0] == '1';
  }
  return 0;
}

// ---- copy-engine path: the neighbours' receive buffers, mapped ------------------------------------------------------
namespace {
struct PeerBlob { // what a rank publishes (fits XPIC_PEER_BLOB_BYTES)
  uint64_t magic;
  int64_t pid;
  int32_t device, valid;
  uint64_t raw[2];   // the buffers' addresses in the exporting process (peers inside that process use them as they are)
  uint64_t bytes;
  hipIpcMemHandle_t h[2];
};
static_assert(sizeof(PeerBlob) <= XPIC_PEER_BLOB_BYTES, "peer blob");
constexpr uint64_t kPeerMagic = 0x7870696370656572ull; // "xpicpeer"
}  // namespace

int xpic_comm_peer_export(xpic_ctx* ctx, void* blob)
{
  XPIC_CHECK(ctx && blob, "null argument");
  XPIC_CHECK(ctx->g.G > 0 && ctx->lrow_buf[0] && ctx->lrow_buf[1], "no ghost-row buffers: a single slab, or a scheme without matL");
  PeerBlob b{};
  b.magic = kPeerMagic;
  b.pid = (int64_t)getpid();
  b.device = ctx->geom.device;
  b.bytes = sizeof(double) * 3 * (uint64_t)ctx->g.lplane();
  b.valid = 1;
  for (int i = 0; i < 2; ++i) {
    b.raw[i] = (uint64_t)(uintptr_t)ctx->lrow_buf[i];
    if (hipIpcGetMemHandle(&b.h[i], ctx->lrow_buf[i]) != hipSuccess) { (void)hipGetLastError(); b.valid = 2; } // same-process peers only
  }
  memcpy(blob, &b, sizeof(b));
  return 0;
}

int xpic_comm_peer_import(xpic_ctx* ctx, const void* lower_blob, const void* upper_blob)
{
  XPIC_CHECK(ctx && lower_blob && upper_blob, "null argument");
  XPIC_CHECK(ctx->g.G > 0 && ctx->lrow_buf[0], "no ghost-row buffers: a single slab, or a scheme without matL");
  XPIC_CHECK(!ctx->peer_lrow[0] && !ctx->peer_lrow[1], "peer buffers imported twice");
  XPIC_HIP(hipSetDevice(ctx->geom.device));
  const void* blobs[2] = {lower_blob, upper_blob};
  for (int side = 0; side < 2; ++side) {
    PeerBlob b;
    memcpy(&b, blobs[side], sizeof(b));
    XPIC_CHECK(b.magic == kPeerMagic && b.valid != 0, "not a peer blob of xpic_comm_peer_export");
    XPIC_CHECK(b.bytes == sizeof(double) * 3 * (uint64_t)ctx->g.lplane(), "the neighbour's ghost-row buffer has another size (another grid?)");
    // my "down" message lands in the lower neighbour's from-above buffer (its lrow_buf[0]), my "up" message in the upper
    // neighbour's from-below buffer (its lrow_buf[1]): two different allocations even when both neighbours are one process
    const int which = side;
    if (b.pid == (int64_t)getpid()) ctx->peer_lrow[side] = (double*)(uintptr_t)b.raw[which]; // threads of one process, a self-ring
    else {
      XPIC_CHECK(b.valid == 1, "the neighbour could not export an IPC handle of its ghost-row buffer");
      void* p = nullptr;
      XPIC_HIP(hipIpcOpenMemHandle(&p, b.h[which], hipIpcMemLazyEnablePeerAccess));
      ctx->peer_mapped[side] = p;
      ctx->peer_lrow[side] = (double*)p;
    }
  }
  if (!ctx->copy_stream) {
    XPIC_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) XPIC_HIP(hipEventCreateWithFlags(&ctx->copy_ev[i], hipEventDisableTiming));
  }
  return 0;
}

int xpic_comm_stats(xpic_ctx* ctx, int64_t* out4, int reset)
{
  XPIC_CHECK(ctx && out4, "null argument");
  Comm& m = ctx->comm;
  out4[0] = m.sent_msgs; out4[1] = m.sent_bytes; out4[2] = m.allreduces; out4[3] = m.allreduce_bytes;
  if (reset) m.sent_msgs = m.sent_bytes = m.allreduces = m.allreduce_bytes = 0;
  return 0;
}

int xpic_comm_size(xpic_ctx* ctx, int* nranks)
{
  XPIC_CHECK(ctx && nranks, "null argument");
  *nranks = 1;
  if (ctx->comm.kind == 1) {
    int n = 0;
    XPIC_NCCL(ncclCommCount((ncclComm_t)ctx->comm.nccl, &n)); // what RCCL itself holds, not what the caller asked for
    *nranks = n;
  }
  else if (ctx->comm.kind == 2) *nranks = ctx->comm.nranks;
  return 0;
}

int xpic_comm_init_callbacks(xpic_ctx* ctx, const xpic_comm_callbacks* cb)
{
  XPIC_CHECK(ctx && cb && cb->sendrecv && cb->allreduce_sum, "null argument");
  XPIC_CHECK(ctx->g.G > 0, "context was created without ghost planes (nranks == 1)");
  ctx->comm.kind = 2;
  ctx->comm.cb = *cb;
  ctx->comm.rank = ctx->geom.rank;
  ctx->comm.nranks = ctx->geom.nranks;
  return 0;
}

}  // extern "C"
