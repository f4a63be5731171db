// Native ghost-feature halo exchange over RCCL (xGMI point-to-point), usable as the engine's
// snet_halo_fn pair without any host staging and without Python.
//
// Replaces, for hosts that sit on libsnet_hip.so directly, the reference's ghost exchange
//   PairE3GNNParallel::{pack,unpack}_{forward,reverse}_comm_gnn   sevenn/pair_e3gnn/pair_e3gnn_parallel.cpp:747-911
//   CommBrick::forward_comm / reverse_comm (float overloads)       sevenn/pair_e3gnn/comm_brick.cpp:1057-1123
// (six sequential blocking MPI swaps with ghost-of-ghost forwarding, optional host staging, :806-809) by ONE
// grouped exchange per call: every peer's ncclSend / ncclRecv pair sits in one ncclGroup, so all peers move
// concurrently, each over its own xGMI link, on the caller's HIP stream (ordered with the kernels around it).
//
//   forward:  pack rows send_idx -> send buffer (snet_gather_rows); peer p receives its share straight into
//             the ghost rows x[n_local + recv_off[p] ..] (ghost rows are laid out contiguously per peer)
//   reverse:  ghost rows gx[n_local ..] go back to their owners; what arrives is summed per target row in
//             fixed peer order (snet_segment_sum_rows) and added with one snet_scatter_add_rows: deterministic
//
// RCCL is bound at run time (dlopen): the library itself has no link-time dependency on it, a host that never
// creates a communicator never loads it, and under PyTorch-ROCm the copy PyTorch already loaded is reused.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <numeric>
#include <vector>

#include "snet_common.h"

namespace {

// the few RCCL entry points used, with the types of <rccl/rccl.h> (opaque handle, 128-byte id, enum ints)
struct UniqueId {
  char internal[128];
};
using Comm = void *;
constexpr int kFloat = 7;  // ncclFloat32
constexpr int kSum = 0;    // ncclSum
constexpr int kFloat64 = 8;

struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*CommCount)(Comm, int *) = nullptr;      // optional: diagnostics only
  int (*CommUserRank)(Comm, int *) = nullptr;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;   // several host threads may create communicators / halos at the same time
  std::call_once(once, [] {
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)  // a copy some other component (PyTorch-ROCm) already loaded wins
      if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    for (const char *n : names) {
      if (r.h) break;
      r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (r.h) {
      auto sym = [&](const char *s) { return dlsym(r.h, s); };
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
      r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
      r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
      r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
      r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
      r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
      r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
      r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
      if (!(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv &&
            r.AllReduce))
        r.h = nullptr;
    }
  });
  return r.h ? &r : nullptr;
}

int fail(const char *what, int rc) {
  Rccl *r = rccl();
  snet::set_error(std::string(what) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error") +
                  " (" + std::to_string(rc) + ")");
  return 1;
}

// ---- transport: RCCL, or an in-process stand-in with the same group semantics ---------------------------------
// The in-process ("loopback") transport exists for tests on ONE GPU: W host threads play W ranks, a send posts
// {device pointer, ready event} into a mailbox, the matching receive copies device-to-device on the receiver's stream
// and acknowledges with an event the sender's stream then waits on -- i.e. ncclSend / ncclRecv inside one ncclGroup
// (FIFO per ordered pair, group end blocks until every peer arrived).  Everything above the transport -- the pack /
// land / permute / reverse-accumulate plan, its offsets for several peers -- is the code the RCCL path runs.
struct LoopMsg {
  const void *ptr;
  size_t bytes;
  hipEvent_t ready;
};
struct LoopHub {
  int world = 1;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::deque<LoopMsg>> box;      // [src * world + dst]
  std::vector<std::deque<hipEvent_t>> ack;   // [src * world + dst]: "dst has copied src's message"
  bool aborted = false;
};
struct CommBox {  // what the C-ABI's opaque communicator pointer points to
  int kind = 0;   // 0: RCCL, 1: loopback
  Comm nccl = nullptr;
  LoopHub *hub = nullptr;
  int world = 1, rank = 0;
  struct Pending {
    void *dst;
    size_t bytes;
    int peer;
    hipStream_t st;
    bool is_send;
  };
  std::vector<Pending> pending;  // loopback: the operations of the open group
};

int x_group_start(CommBox *c) {
  if (c->kind == 0) {
    const int rc = rccl()->GroupStart();
    return rc ? fail("ncclGroupStart", rc) : 0;
  }
  c->pending.clear();
  return 0;
}
int x_send(CommBox *c, const float *buf, size_t count, int peer, hipStream_t st) {
  if (c->kind == 0) {
    const int rc = rccl()->Send(buf, count, kFloat, peer, c->nccl, st);
    return rc ? fail("ncclSend", rc) : 0;
  }
  LoopMsg m{buf, count * sizeof(float), nullptr};
  if (hipEventCreateWithFlags(&m.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(m.ready, st) != hipSuccess) {
    snet::set_error("loopback send: event creation failed");
    return 1;
  }
  {
    std::lock_guard<std::mutex> lk(c->hub->mu);
    c->hub->box[(size_t)c->rank * c->world + peer].push_back(m);
  }
  c->hub->cv.notify_all();
  c->pending.push_back({nullptr, 0, peer, st, true});
  return 0;
}
int x_recv(CommBox *c, float *buf, size_t count, int peer, hipStream_t st) {
  if (c->kind == 0) {
    const int rc = rccl()->Recv(buf, count, kFloat, peer, c->nccl, st);
    return rc ? fail("ncclRecv", rc) : 0;
  }
  c->pending.push_back({buf, count * sizeof(float), peer, st, false});
  return 0;
}
int x_group_end(CommBox *c) {
  if (c->kind == 0) {
    const int rc = rccl()->GroupEnd();
    return rc ? fail("ncclGroupEnd", rc) : 0;
  }
  LoopHub *h = c->hub;
  for (const auto &p : c->pending) {  // receives first: nobody waits for an acknowledgement before having copied
    if (p.is_send) continue;
    LoopMsg m;
    {
      std::unique_lock<std::mutex> lk(h->mu);
      auto &q = h->box[(size_t)p.peer * c->world + c->rank];
      h->cv.wait(lk, [&] { return !q.empty() || h->aborted; });
      if (h->aborted) {
        snet::set_error("loopback transport aborted");
        return 1;
      }
      m = q.front();
      q.pop_front();
    }
    hipEvent_t done = nullptr;
    bool ok = m.bytes == p.bytes;
    ok = ok && hipStreamWaitEvent(p.st, m.ready, 0) == hipSuccess;
    ok = ok && (p.bytes == 0 || hipMemcpyAsync(p.dst, m.ptr, p.bytes, hipMemcpyDeviceToDevice, p.st) == hipSuccess);
    ok = ok && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess && hipEventRecord(done, p.st) == hipSuccess;
    (void)hipEventDestroy(m.ready);
    {
      std::lock_guard<std::mutex> lk(h->mu);
      if (!ok) h->aborted = true;
      else h->ack[(size_t)p.peer * c->world + c->rank].push_back(done);
    }
    h->cv.notify_all();
    if (!ok) {
      snet::set_error("loopback receive: size mismatch between the two ends of an exchange, or a HIP call failed");
      return 1;
    }
  }
  for (const auto &p : c->pending) {  // a send is complete (its buffer reusable) once the peer's copy is ordered before us
    if (!p.is_send) continue;
    hipEvent_t done;
    {
      std::unique_lock<std::mutex> lk(h->mu);
      auto &q = h->ack[(size_t)c->rank * c->world + p.peer];
      h->cv.wait(lk, [&] { return !q.empty() || h->aborted; });
      if (h->aborted) {
        snet::set_error("loopback transport aborted");
        return 1;
      }
      done = q.front();
      q.pop_front();
    }
    const bool ok = hipStreamWaitEvent(p.st, done, 0) == hipSuccess;
    (void)hipEventDestroy(done);
    if (!ok) {
      snet::set_error("loopback send: hipStreamWaitEvent failed");
      return 1;
    }
  }
  c->pending.clear();
  return 0;
}

template <class T>
struct Dev {
  T *p = nullptr;
  size_t n = 0;
  bool ensure(size_t want) {
    if (want <= n) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) return false;
    n = want;
    return true;
  }
  bool upload(const std::vector<T> &h) {
    if (h.empty()) return true;
    return ensure(h.size()) && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
  }
  ~Dev() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

struct snet_halo {
  CommBox *comm = nullptr;
  int world = 1, rank = 0;
  std::vector<int64_t> send_cnt, recv_cnt, send_off, recv_off;  // rows per peer and their prefix sums
  int64_t n_send = 0, n_ghost = 0, n_seg = 0;
  int32_t seg_dim = 0;  // row width of the reverse exchange staged in seg_buf (0: none)
  Dev<int32_t> send_idx, red_rows, red_perm, red_ptr;
  int64_t max_send_row = -1;  // send_idx entries are local rows: max_send_row < n_local must hold on every call
  Dev<int32_t> ghost_perm, ghost_inv;      // optional: k-th received row (peer order) <-> ghost row (host's node order)
  bool permuted = false;
  Dev<float> send_buf, recv_buf, seg_buf, ghost_buf;  // grown to the widest row seen
};

extern "C" {

int snet_rccl_unique_id(void *id128) {
  SNET_REQUIRE(id128 != nullptr, "snet_rccl_unique_id: null argument");
  Rccl *r = rccl();
  SNET_REQUIRE(r != nullptr, "snet_rccl_unique_id: librccl.so could not be loaded");
  UniqueId id;
  const int rc = r->GetUniqueId(&id);
  if (rc) return fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, 128);
  return 0;
}

int snet_rccl_comm_create(const void *id128, int32_t world, int32_t rank, void **comm) {
  SNET_REQUIRE(id128 != nullptr && comm != nullptr && world >= 1 && rank >= 0 && rank < world,
               "snet_rccl_comm_create: bad argument");
  Rccl *r = rccl();
  SNET_REQUIRE(r != nullptr, "snet_rccl_comm_create: librccl.so could not be loaded");
  UniqueId id;
  memcpy(id.internal, id128, 128);
  Comm c = nullptr;
  const int rc = r->CommInitRank(&c, world, id, rank);
  if (rc) return fail("ncclCommInitRank", rc);
  auto *box = new CommBox;
  box->kind = 0;
  box->nccl = c;
  box->world = world;
  box->rank = rank;
  *comm = box;
  return 0;
}

// 1 when librccl.so can be bound in this process (dlopen + every symbol the exchange needs), 0 otherwise: a LOCAL, non-collective
// probe -- hosts agree on the transport with it BEFORE any rank enters the collective communicator construction
int snet_rccl_available(void) { return rccl() != nullptr ? 1 : 0; }

// what RCCL itself reports for a communicator (ncclCommCount / ncclCommUserRank); the in-process test transport reports its own
int snet_rccl_comm_info(void *comm, int32_t *world_out, int32_t *rank_out) {
  auto *box = static_cast<CommBox *>(comm);
  SNET_REQUIRE(box != nullptr && world_out != nullptr && rank_out != nullptr, "snet_rccl_comm_info: bad argument");
  *world_out = box->world;
  *rank_out = box->rank;
  if (box->kind == 0) {
    Rccl *r = rccl();
    SNET_REQUIRE(r != nullptr && r->CommCount && r->CommUserRank, "snet_rccl_comm_info: ncclCommCount not available");
    int n = 0, me = 0;
    int rc = r->CommCount(box->nccl, &n);
    if (rc) return fail("ncclCommCount", rc);
    rc = r->CommUserRank(box->nccl, &me);
    if (rc) return fail("ncclCommUserRank", rc);
    *world_out = n;
    *rank_out = me;
  }
  return 0;
}

void snet_rccl_comm_destroy(void *comm) {
  auto *box = static_cast<CommBox *>(comm);
  if (!box) return;
  Rccl *r = rccl();
  if (box->kind == 0 && r && box->nccl) (void)r->CommDestroy(box->nccl);
  delete box;
}

// sum of n doubles over all ranks, in place (total energy, virial): one tiny all-reduce per step
int snet_rccl_allreduce_sum_f64(void *comm, double *dev_values, int64_t n, void *stream) {
  auto *box = static_cast<CommBox *>(comm);
  SNET_REQUIRE(box != nullptr && dev_values != nullptr, "snet_rccl_allreduce_sum_f64: bad argument");
  SNET_REQUIRE(box->kind == 0, "snet_rccl_allreduce_sum_f64: not available on the in-process test transport");
  Rccl *r = rccl();
  SNET_REQUIRE(r != nullptr, "snet_rccl_allreduce_sum_f64: RCCL not loaded");
  const int rc = r->AllReduce(dev_values, dev_values, (size_t)n, kFloat64, kSum, box->nccl, static_cast<hipStream_t>(stream));
  return rc ? fail("ncclAllReduce", rc) : 0;
}

// ---- in-process transport for tests (one GPU, one host thread per rank) ----
int snet_loopback_hub_create(int32_t world, void **hub) {
  SNET_REQUIRE(hub != nullptr && world >= 1 && world <= 64, "snet_loopback_hub_create: bad argument");
  auto *h = new LoopHub;
  h->world = world;
  h->box.resize((size_t)world * world);
  h->ack.resize((size_t)world * world);
  *hub = h;
  return 0;
}
void snet_loopback_hub_abort(void *hub) {  // wake every rank blocked in an exchange (a test thread died)
  auto *h = static_cast<LoopHub *>(hub);
  if (!h) return;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->aborted = true;
  }
  h->cv.notify_all();
}
void snet_loopback_hub_destroy(void *hub) { delete static_cast<LoopHub *>(hub); }
int snet_loopback_comm_create(void *hub, int32_t rank, void **comm) {
  auto *h = static_cast<LoopHub *>(hub);
  SNET_REQUIRE(h != nullptr && comm != nullptr && rank >= 0 && rank < h->world, "snet_loopback_comm_create: bad argument");
  auto *box = new CommBox;
  box->kind = 1;
  box->hub = h;
  box->world = h->world;
  box->rank = rank;
  *comm = box;
  return 0;
}

int snet_halo_create(void *comm, int32_t world, int32_t rank, const int32_t *send_counts, const int32_t *send_idx_host,
                     const int32_t *recv_counts, const int32_t *recv_perm_host, snet_halo **out) {
  SNET_REQUIRE(comm != nullptr && out != nullptr && send_counts && recv_counts && world >= 1 && rank >= 0 && rank < world,
               "snet_halo_create: bad argument");
  auto *box = static_cast<CommBox *>(comm);
  SNET_REQUIRE(box->world == world && box->rank == rank, "snet_halo_create: world / rank differ from the communicator's");
  auto *h = new snet_halo;
  h->comm = box;
  h->world = world;
  h->rank = rank;
  h->send_cnt.assign(send_counts, send_counts + world);
  h->recv_cnt.assign(recv_counts, recv_counts + world);
  h->send_off.assign(world + 1, 0);
  h->recv_off.assign(world + 1, 0);
  for (int p = 0; p < world; ++p) {
    h->send_off[p + 1] = h->send_off[p] + h->send_cnt[p];
    h->recv_off[p + 1] = h->recv_off[p] + h->recv_cnt[p];
  }
  h->n_send = h->send_off[world];
  h->n_ghost = h->recv_off[world];
  bool ok = true;
  if (h->n_send > 0) {
    SNET_REQUIRE(send_idx_host != nullptr, "snet_halo_create: send_idx missing");
    std::vector<int32_t> idx(send_idx_host, send_idx_host + h->n_send);
    for (int32_t v : idx) {  // the largest row a peer asks for: checked against n_local on every exchange
      if (v < 0) {
        delete h;
        snet::set_error("snet_halo_create: negative send index");
        return 1;
      }
      h->max_send_row = std::max(h->max_send_row, (int64_t)v);
    }
    ok = h->send_idx.upload(idx);
    // reverse unpack plan: received rows grouped by the local row they add into; a stable sort keeps the
    // contributions of one row in peer order, so the floating-point sum order is fixed
    std::vector<int32_t> perm(h->n_send);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return idx[a] < idx[b]; });
    std::vector<int32_t> rows, ptr{0};
    for (int64_t k = 0; k < h->n_send; ++k) {
      if (rows.empty() || rows.back() != idx[perm[k]]) {
        if (!rows.empty()) ptr.push_back((int32_t)k);
        rows.push_back(idx[perm[k]]);
      }
    }
    ptr.push_back((int32_t)h->n_send);
    h->n_seg = (int64_t)rows.size();
    ok = ok && h->red_rows.upload(rows) && h->red_perm.upload(perm) && h->red_ptr.upload(ptr);
  }
  if (ok && recv_perm_host != nullptr && h->n_ghost > 0) {
    // the host numbers its ghost rows in its own order: the k-th row of the peer-ordered stream is ghost row perm[k]
    std::vector<int32_t> perm(recv_perm_host, recv_perm_host + h->n_ghost), inv(h->n_ghost, -1);
    for (int64_t k = 0; k < h->n_ghost && ok; ++k) {
      ok = perm[k] >= 0 && perm[k] < h->n_ghost && inv[perm[k]] < 0;
      if (ok) inv[perm[k]] = (int32_t)k;
    }
    if (!ok) {
      delete h;
      snet::set_error("snet_halo_create: recv_perm is not a permutation of the ghost rows");
      return 2;
    }
    ok = h->ghost_perm.upload(perm) && h->ghost_inv.upload(inv);
    h->permuted = true;
  }
  if (!ok) {
    delete h;
    snet::set_error("snet_halo_create: device allocation / upload failed");
    return 1;
  }
  *out = h;
  return 0;
}

void snet_halo_destroy(snet_halo *h) { delete h; }

int64_t snet_halo_ghost_rows(const snet_halo *h) { return h ? h->n_ghost : 0; }
int64_t snet_halo_send_rows(const snet_halo *h) { return h ? h->n_send : 0; }

// snet_halo_fn: fill ghost rows x[n_local ..] with their owners' rows
int snet_halo_forward(void *user, float *x, int64_t n_total, int64_t n_local, int32_t dim, void *stream) {
  auto *h = static_cast<snet_halo *>(user);
  SNET_REQUIRE(h != nullptr && x != nullptr && dim > 0, "snet_halo_forward: bad argument");
  SNET_REQUIRE(n_total - n_local == h->n_ghost, "snet_halo_forward: ghost row count does not match the exchange plan");
  SNET_REQUIRE(h->max_send_row < n_local, "snet_halo_forward: the exchange plan sends a row beyond n_local");
  SNET_REQUIRE(h->comm->kind != 0 || rccl() != nullptr, "snet_halo_forward: RCCL not loaded");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->n_send > 0) {
    SNET_REQUIRE(h->send_buf.ensure((size_t)h->n_send * dim), "snet_halo_forward: allocation failed");
    if (int rc = snet_gather_rows(x, h->send_idx.p, h->send_buf.p, h->n_send, dim, stream)) return rc;
  }
  float *land = x + n_local * dim;  // where the peer-ordered stream lands
  if (h->permuted) {
    SNET_REQUIRE(h->ghost_buf.ensure((size_t)h->n_ghost * dim), "snet_halo_forward: allocation failed");
    land = h->ghost_buf.p;
  }
  if (int rc = x_group_start(h->comm)) return rc;
  int rc = 0;
  for (int p = 0; p < h->world && !rc; ++p) {
    if (h->send_cnt[p]) rc = x_send(h->comm, h->send_buf.p + h->send_off[p] * dim, (size_t)h->send_cnt[p] * dim, p, st);
    if (!rc && h->recv_cnt[p]) rc = x_recv(h->comm, land + h->recv_off[p] * dim, (size_t)h->recv_cnt[p] * dim, p, st);
  }
  const int rc2 = x_group_end(h->comm);
  if (rc) return rc;
  if (rc2) return rc2;
  if (h->permuted)  // ghost row g <- stream row inv[g]
    return snet_gather_rows(h->ghost_buf.p, h->ghost_inv.p, x + n_local * dim, h->n_ghost, dim, stream);
  return 0;
}

// Reverse exchange in two halves so that a host can overlap it with work that WRITES the local rows (round 4: interior /
// boundary split of the convolution): _exchange reads the ghost rows gx[n_local ..] only, sends them home and reduces what
// the peers return into the halo's own staging rows; _accumulate adds those into the owners' rows gx[.. n_local].
int snet_halo_reverse_exchange(void *user, const float *gx, int64_t n_total, int64_t n_local, int32_t dim, void *stream) {
  auto *h = static_cast<snet_halo *>(user);
  SNET_REQUIRE(h != nullptr && gx != nullptr && dim > 0, "snet_halo_reverse: bad argument");
  SNET_REQUIRE(n_total - n_local == h->n_ghost, "snet_halo_reverse: ghost row count does not match the exchange plan");
  SNET_REQUIRE(h->max_send_row < n_local, "snet_halo_reverse: the exchange plan sends a row beyond n_local");
  SNET_REQUIRE(h->comm->kind != 0 || rccl() != nullptr, "snet_halo_reverse: RCCL not loaded");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->n_send > 0)
    SNET_REQUIRE(h->recv_buf.ensure((size_t)h->n_send * dim) && h->seg_buf.ensure((size_t)h->n_seg * dim),
                 "snet_halo_reverse: allocation failed");
  const float *home = gx + n_local * dim;  // ghost rows in peer order
  if (h->permuted) {
    SNET_REQUIRE(h->ghost_buf.ensure((size_t)h->n_ghost * dim), "snet_halo_reverse: allocation failed");
    if (int e = snet_gather_rows(gx + n_local * dim, h->ghost_perm.p, h->ghost_buf.p, h->n_ghost, dim, stream)) return e;
    home = h->ghost_buf.p;
  }
  if (int rc = x_group_start(h->comm)) return rc;
  int rc = 0;
  for (int p = 0; p < h->world && !rc; ++p) {
    if (h->recv_cnt[p])  // my ghost rows owned by p go home
      rc = x_send(h->comm, home + h->recv_off[p] * dim, (size_t)h->recv_cnt[p] * dim, p, st);
    if (!rc && h->send_cnt[p])  // p returns the gradients of the rows I sent it
      rc = x_recv(h->comm, h->recv_buf.p + h->send_off[p] * dim, (size_t)h->send_cnt[p] * dim, p, st);
  }
  const int rc2 = x_group_end(h->comm);
  if (rc) return rc;
  if (rc2) return rc2;
  if (h->n_seg > 0)
    if (int e = snet_segment_sum_rows(h->recv_buf.p, h->red_ptr.p, h->red_perm.p, h->n_seg, dim, h->seg_buf.p, stream)) return e;
  h->seg_dim = dim;
  return 0;
}

int snet_halo_reverse_accumulate(void *user, float *gx, int32_t dim, void *stream) {
  auto *h = static_cast<snet_halo *>(user);
  SNET_REQUIRE(h != nullptr && gx != nullptr && dim > 0, "snet_halo_reverse_accumulate: bad argument");
  SNET_REQUIRE(h->seg_dim == dim, "snet_halo_reverse_accumulate: no exchange of this row width is staged");
  h->seg_dim = 0;
  if (h->n_seg > 0) return snet_scatter_add_rows(h->seg_buf.p, h->red_rows.p, gx, h->n_seg, dim, stream);
  return 0;
}

// snet_halo_fn: add ghost rows gx[n_local ..] into their owners' rows
int snet_halo_reverse(void *user, float *gx, int64_t n_total, int64_t n_local, int32_t dim, void *stream) {
  if (int rc = snet_halo_reverse_exchange(user, gx, n_total, n_local, dim, stream)) return rc;
  return snet_halo_reverse_accumulate(user, gx, dim, stream);
}

// convenience: install the exchange on a model (ghost forces / atomic virials folded by the same hooks)
int snet_model_set_rccl_halo(snet_model *model, snet_halo *halo, int32_t fold_forces) {
  SNET_REQUIRE(model != nullptr, "snet_model_set_rccl_halo: null model");
  if (halo == nullptr) return snet_model_set_halo(model, nullptr, nullptr, nullptr, 1);
  return snet_model_set_halo(model, &snet_halo_forward, &snet_halo_reverse, halo, fold_forces);
}

}  // extern "C"
