// fotg_node.hip -- one process, several GPUs ("one process per node with one host thread + stream set per GPU").
// Built on the public C-ABI only (fotg_pipe_* of include/fotg.h): a node owns one pipe and one host thread per device slot.
// Frame pairs are independent, so a batch of n pairs is cut into contiguous shards (fotg_node_shard: the first n % ndev slots get one pair more)
// and every slot runs its shard through its own pipe; there is no data-path exchange between the GPUs.  Two ways in:
//   resident   the caller's frames of shard d already live on device d (per-slot pointers) -- what a server with per-GPU
//              decoders does, and what the scaling bench measures;
//   scatter    the whole batch lives on the first slot's device; the other slots pull their shards over xGMI in chunks
//              (hipMemcpyPeerAsync into depth + 1 staging buffers on a copy stream of their own), compute chunk t while chunk
//              t + 1 travels, and push the flows back into the caller's array -- the C++ twin of
//              flowonthego_amd/shard.py: pipelined_scatter_compute (RCCL send / recv between processes there, peer copies here).
// The host threads only ISSUE work (a step is 23 launches, ~0.07 ms of host time per device: one thread could not feed eight
// GPUs at 0.36 ms per step); they never wait for the GPU.  fotg_node_wait synchronises with the pipes' own completion events of
// the job's pieces (no stream of the node's own on the resident path: a stream that only waits still occupies a hardware queue
// and blocks the slot stream that shares it -- measured: 134 k instead of 187 k pairs/s on one GPU).  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "../../include/fotg.h"

namespace {

constexpr int RING = 16;                       // jobs that may be issued and not yet waited for

struct Job {
  long id = -1;
  int kind = 0;                                // 0 resident, 1 scatter
  int n = 0, u8 = 0, chunk = 0;
  const void *I0[FOTG_NODE_MAX_DEV], *I1[FOTG_NODE_MAX_DEV];
  float *out[FOTG_NODE_MAX_DEV];
};

struct Slot {
  int device = 0, index = 0;
  fotg_pipe *pipe = nullptr;
  std::thread th;
  hipStream_t copy = nullptr;                  // peer copies of the scatter mode (created with the staging buffers)
  hipEvent_t done[RING] = {};                  // scatter jobs: completion of job id on this slot (recorded on `copy`): done[id % RING]
  std::vector<long> piece_tk[RING];            // the pipe tickets of the job's pieces: fotg_node_wait waits for them through the pipe (its own
                                               // completion events -- no extra stream: a stream that only waits would still occupy, and
                                               // block, a hardware queue), which also verifies / recomputes a stalled piece
  bool on_copy[RING] = {};
  int status[RING] = {};                       // issue status of job id on this slot
  int hip_err[RING] = {};                      // fotg_last_hip_error() of the issuing thread when status is FOTG_ERR_HIP
  long issued = 0;                             // jobs this slot's thread has issued (guarded by the node's mutex)
  // scatter mode: depth + 1 staging buffers of 2 x chunk frames + chunk flows each
  std::vector<void *> stage_in;
  std::vector<float *> stage_out;
};

}  // namespace

struct fotg_node {
  int ndev = 0, depth = 0, max_batch = 0, w = 0, h = 0, noc = 1, nch = 2, ow = 0, oh = 0, u8_color = 0;
  size_t frame_elems = 0, flow_elems = 0;
  Slot slot[FOTG_NODE_MAX_DEV];
  std::mutex mu;
  std::condition_variable cv_job, cv_issued;
  Job ring[RING];
  long submitted = 0, waited = 0;              // jobs handed in; jobs [0, waited) have been waited for (their ring entries are free)
  int jstatus[RING] = {};                      // final status of job id (id % RING), valid for ids in [waited - RING, waited): a repeated or
  long jstatus_id[RING];                       // out-of-order wait reports the job's OWN status
  int last_hip = 0;                            // HIP error code behind the last FOTG_ERR_HIP a wait returned (raised on a worker thread)
  std::mutex wait_mu;                          // one waiting thread at a time
  bool stop = false;
};

namespace {

struct OnDevice {
  int prev = -1;
  explicit OnDevice(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~OnDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// completion of (job, slot): the pipe tickets of the job's pieces (resident frames, the source slot of a scatter), or an event
// behind the last copy on the slot's copy stream (the pulling slots of a scatter) and the tickets for the stall check
int finish_job(Slot &s, const Job &j, const std::vector<long> &tickets)
{
  s.piece_tk[j.id % RING] = tickets;
  s.on_copy[j.id % RING] = false;
  return FOTG_OK;
}

int issue_resident(fotg_node *nd, Slot &s, const Job &j)
{
  int b = 0, cnt = 0;
  fotg_node_shard(j.n, nd->ndev, s.index, &b, &cnt);
  std::vector<long> tickets;
  const size_t esz = j.u8 ? (nd->u8_color ? 3 : 1) : 4;       // (8-bit colour frames of a gray context: three bytes per pixel)
  // the shard in pieces of at most max_batch pairs, consecutive pieces on consecutive slots of the pipe
  for (int o = 0; o < cnt; o += nd->max_batch) {
    const int m = cnt - o < nd->max_batch ? cnt - o : nd->max_batch;
    const char *a = (const char *)j.I0[s.index] + (size_t)o * nd->frame_elems * esz, *bb = (const char *)j.I1[s.index] + (size_t)o * nd->frame_elems * esz;
    float *out = j.out[s.index] + (size_t)o * nd->flow_elems;
    long t = -1;
    const int st = j.u8 ? fotg_pipe_submit_u8(s.pipe, m, (const unsigned char *)a, (const unsigned char *)bb, nullptr, out, FOTG_NO_STREAM, &t)
                        : fotg_pipe_submit(s.pipe, m, (const float *)a, (const float *)bb, nullptr, out, FOTG_NO_STREAM, &t);
    if (st != FOTG_OK) return st;
    tickets.push_back(t);
  }
  return finish_job(s, j, tickets);
}

// whole batch on the source slot's device (slot 0): this slot pulls its shard chunk by chunk
int issue_scatter(fotg_node *nd, Slot &s, const Job &j)
{
  int b = 0, cnt = 0;
  fotg_node_shard(j.n, nd->ndev, s.index, &b, &cnt);
  const size_t fbytes = nd->frame_elems * (j.u8 ? (nd->u8_color ? 3 : 1) : 4);      // (8-bit frames travel as bytes: a quarter of the link traffic)
  // pulled pieces live in staging buffers that are recycled behind device-side waits: never recomputed (FOTG_SUBMIT_NO_RECOMPUTE at
  // SUBMIT, not only at the wait: a host wait of another job on this pipe may verify the slot while the piece is still in flight)
  auto submit = [&](int m, const void *a, const void *bb, float *out, void *after, long *t) {
    return fotg_pipe_submit_ex(s.pipe, m, a, bb, j.u8, nullptr, out, after, s.index == 0 ? 0 : FOTG_SUBMIT_NO_RECOMPUTE, t);
  };
  const char *G0 = (const char *)j.I0[0] + (size_t)b * fbytes, *G1 = (const char *)j.I1[0] + (size_t)b * fbytes;
  float *GO = j.out[0] + (size_t)b * nd->flow_elems;
  const int src_dev = nd->slot[0].device, chunk = j.chunk;
  std::vector<long> tickets;
  if (s.index == 0) {
    // the source slot computes on views of the caller's arrays: no copy at all
    for (int o = 0; o < cnt; o += chunk) {
      const int m = cnt - o < chunk ? cnt - o : chunk;
      long t = -1;
      const int st = submit(m, G0 + (size_t)o * fbytes, G1 + (size_t)o * fbytes, GO + (size_t)o * nd->flow_elems, FOTG_NO_STREAM, &t);
      if (st != FOTG_OK) return st;
      tickets.push_back(t);
    }
    return finish_job(s, j, tickets);
  }
  const int nbuf = (int)s.stage_in.size();
  std::vector<int> piece_m;
  auto push_back_flows = [&](int piece) -> int {          // piece's flows -> the caller's array on the source device, behind its compute
    const int bi = piece % nbuf;
    int st = fotg_pipe_wait(s.pipe, tickets[piece], s.copy, 0);
    if (st != FOTG_OK) return st;
    if (hipMemcpyPeerAsync(GO + (size_t)piece * chunk * nd->flow_elems, src_dev, s.stage_out[bi], s.device,
                           (size_t)piece_m[piece] * nd->flow_elems * 4, s.copy) != hipSuccess) return FOTG_ERR_HIP;
    return FOTG_OK;
  };
  int piece = 0;
  for (int o = 0; o < cnt; o += chunk, ++piece) {
    const int m = cnt - o < chunk ? cnt - o : chunk, bi = piece % nbuf;
    // the buffer piece lands in was read (frames) and written (flows) by piece - nbuf: its flows leave first, on the same
    // copy stream and behind that piece's compute, so the incoming frames are ordered behind both
    if (piece >= nbuf) { const int st = push_back_flows(piece - nbuf); if (st != FOTG_OK) return st; }
    char *in0 = (char *)s.stage_in[bi], *in1 = in0 + (size_t)chunk * fbytes;
    if (hipMemcpyPeerAsync(in0, s.device, G0 + (size_t)o * fbytes, src_dev, (size_t)m * fbytes, s.copy) != hipSuccess ||
        hipMemcpyPeerAsync(in1, s.device, G1 + (size_t)o * fbytes, src_dev, (size_t)m * fbytes, s.copy) != hipSuccess) return FOTG_ERR_HIP;
    long t = -1;
    const int st = submit(m, in0, in1, s.stage_out[bi], s.copy, &t);
    if (st != FOTG_OK) return st;
    tickets.push_back(t);
    piece_m.push_back(m);
  }
  for (int q = piece > nbuf ? piece - nbuf : 0; q < piece; ++q) { const int st = push_back_flows(q); if (st != FOTG_OK) return st; }
  // the job's event: behind the last copy on the copy stream (which is behind every compute of the job)
  if (hipEventRecord(s.done[j.id % RING], s.copy) != hipSuccess) return FOTG_ERR_HIP;
  s.on_copy[j.id % RING] = true;
  s.piece_tk[j.id % RING] = tickets;
  return FOTG_OK;
}

void worker(fotg_node *nd, int k)
{
  Slot &s = nd->slot[k];
  (void)hipSetDevice(s.device);
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> lk(nd->mu);
      nd->cv_job.wait(lk, [&] { return nd->stop || nd->submitted > s.issued; });
      if (nd->submitted <= s.issued) return;            // stop, nothing left to issue
      j = nd->ring[s.issued % RING];
    }
    int st = j.kind == 0 ? issue_resident(nd, s, j) : issue_scatter(nd, s, j);
    {
      std::lock_guard<std::mutex> lk(nd->mu);
      s.status[j.id % RING] = st;
      s.hip_err[j.id % RING] = st == FOTG_ERR_HIP ? fotg_last_hip_error() : 0;
      ++s.issued;
    }
    nd->cv_issued.notify_all();
  }
}

int node_submit(fotg_node *nd, int kind, int u8, int n, const void *const *I0, const void *const *I1, float *const *out, int chunk, long *ticket)
{
  if (!nd || !I0 || !I1 || !out || n < 1) return FOTG_ERR_ARG;
  if (kind == 1) {
    if (chunk < 1 || chunk > nd->max_batch) return FOTG_ERR_BATCH;
    if (!I0[0] || !I1[0] || !out[0]) return FOTG_ERR_ARG;
  } else {
    for (int d = 0; d < nd->ndev; ++d) {
      int b, cnt;
      fotg_node_shard(n, nd->ndev, d, &b, &cnt);
      if (cnt > 0 && (!I0[d] || !I1[d] || !out[d])) return FOTG_ERR_ARG;
    }
  }
  std::unique_lock<std::mutex> lk(nd->mu);
  // the ring entry (and its events) of job id - RING must have been waited for
  if (nd->submitted - nd->waited >= RING) return FOTG_ERR_BATCH;
  Job &j = nd->ring[nd->submitted % RING];
  j.id = nd->submitted; j.kind = kind; j.n = n; j.u8 = u8; j.chunk = chunk;
  for (int d = 0; d < nd->ndev; ++d) {
    j.I0[d] = kind == 1 ? I0[0] : I0[d]; j.I1[d] = kind == 1 ? I1[0] : I1[d]; j.out[d] = kind == 1 ? out[0] : out[d];
  }
  if (ticket) *ticket = nd->submitted;
  ++nd->submitted;
  lk.unlock();
  nd->cv_job.notify_all();
  return FOTG_OK;
}

}  // namespace

extern "C" {

int fotg_node_shard(int n, int ndev, int d, int *begin, int *count)
{
  if (n < 0 || ndev < 1 || d < 0 || d >= ndev) return FOTG_ERR_ARG;
  const int base = n / ndev, rem = n % ndev;
  if (begin) *begin = d * base + (d < rem ? d : rem);
  if (count) *count = base + (d < rem ? 1 : 0);
  return FOTG_OK;
}

void fotg_node_destroy(fotg_node *nd)
{
  if (!nd) return;
  {
    std::lock_guard<std::mutex> lk(nd->mu);
    nd->stop = true;
  }
  nd->cv_job.notify_all();
  for (int k = 0; k < nd->ndev; ++k) if (nd->slot[k].th.joinable()) nd->slot[k].th.join();
  for (int k = 0; k < nd->ndev; ++k) {
    Slot &s = nd->slot[k];
    OnDevice od(s.device);
    if (s.copy) (void)hipStreamSynchronize(s.copy);
    if (s.pipe) fotg_pipe_destroy(s.pipe);
    for (void *p : s.stage_in) (void)hipFree(p);
    for (float *p : s.stage_out) (void)hipFree(p);
    for (auto &e : s.done) if (e) (void)hipEventDestroy(e);
    if (s.copy) (void)hipStreamDestroy(s.copy);
  }
  delete nd;
}

int fotg_node_create(const fotg_params *p, int w_org, int h_org, const int *devices, int ndev, int max_batch, int depth, fotg_node **out)
{
  if (!p || !devices || !out || ndev < 1 || ndev > FOTG_NODE_MAX_DEV || max_batch < 1 || depth < 1 || depth > FOTG_PIPE_MAX_DEPTH) return FOTG_ERR_ARG;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess) return FOTG_ERR_HIP;
  for (int k = 0; k < ndev; ++k) if (devices[k] < 0 || devices[k] >= have) return FOTG_ERR_ARG;
  fotg_node *nd = new (std::nothrow) fotg_node();
  if (!nd) return FOTG_ERR_ARG;
  nd->ndev = ndev; nd->depth = depth; nd->max_batch = max_batch; nd->w = w_org; nd->h = h_org; nd->noc = p->noc; nd->nch = p->depth ? 1 : 2; nd->u8_color = p->u8_color;
  nd->frame_elems = (size_t)w_org * h_org * p->noc;
  for (auto &v : nd->jstatus_id) v = -1;
  for (int k = 0; k < ndev; ++k) {
    Slot &s = nd->slot[k];
    s.device = devices[k]; s.index = k;
    OnDevice od(s.device);
    int st = fotg_pipe_create(p, w_org, h_org, s.device, max_batch, depth, &s.pipe);
    if (st != FOTG_OK) { fotg_node_destroy(nd); return st; }
    for (auto &e : s.done) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { fotg_node_destroy(nd); return FOTG_ERR_HIP; }
    // peer access to the source device of the scatter mode (slot 0's): direct xGMI copies instead of staging through the host
    if (k > 0 && s.device != nd->slot[0].device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, s.device, nd->slot[0].device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(nd->slot[0].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { fotg_node_destroy(nd); return FOTG_ERR_HIP; }
        (void)hipGetLastError();
      }
    }
  }
  fotg_ctx *c0 = nullptr;
  if (fotg_pipe_context(nd->slot[0].pipe, 0, &c0) != FOTG_OK || fotg_out_size(c0, &nd->ow, &nd->oh) != FOTG_OK) { fotg_node_destroy(nd); return FOTG_ERR_ARG; }
  nd->flow_elems = (size_t)nd->ow * nd->oh * nd->nch;
  for (int k = 0; k < ndev; ++k) nd->slot[k].th = std::thread(worker, nd, k);
  *out = nd;
  return FOTG_OK;
}

int fotg_node_submit(fotg_node *nd, int n, const float *const *I0, const float *const *I1, float *const *outflow, long *ticket)
{
  return node_submit(nd, 0, 0, n, (const void *const *)I0, (const void *const *)I1, outflow, 0, ticket);
}

int fotg_node_submit_u8(fotg_node *nd, int n, const unsigned char *const *I0, const unsigned char *const *I1, float *const *outflow, long *ticket)
{
  return node_submit(nd, 0, 1, n, (const void *const *)I0, (const void *const *)I1, outflow, 0, ticket);
}

static int node_scatter(fotg_node *nd, int u8, int n, const void *I0, const void *I1, float *outflow, int chunk, long *ticket)
{
  if (!nd) return FOTG_ERR_ARG;
  // the staging buffers of the pulling slots, sized for max_batch pairs per chunk, on first use: all of a slot's buffers or none
  // (a partial set would silently run the slot without the copy / compute overlap), under the node's mutex (two submitting threads)
  {
    std::lock_guard<std::mutex> lk(nd->mu);
    for (int k = 1; k < nd->ndev; ++k) {
      Slot &s = nd->slot[k];
      if (!s.stage_in.empty()) continue;
      OnDevice od(s.device);
      if (!s.copy && hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking) != hipSuccess) return FOTG_ERR_HIP;
      std::vector<void *> in;
      std::vector<float *> outb;
      bool ok = true;
      for (int b = 0; b < nd->depth + 1 && ok; ++b) {
        void *pi = nullptr; float *po = nullptr;
        ok = hipMalloc(&pi, 2 * (size_t)nd->max_batch * nd->frame_elems * 4) == hipSuccess;
        if (ok) { in.push_back(pi); ok = hipMalloc((void **)&po, (size_t)nd->max_batch * nd->flow_elems * 4) == hipSuccess; }
        if (ok) outb.push_back(po);
      }
      if (!ok) {
        nd->last_hip = (int)hipGetLastError();
        for (void *q : in) (void)hipFree(q);
        for (float *q : outb) (void)hipFree(q);
        return FOTG_ERR_HIP;
      }
      s.stage_in = in; s.stage_out = outb;
    }
  }
  const void *a[1] = {I0}, *b[1] = {I1};
  float *o[1] = {outflow};
  return node_submit(nd, 1, u8, n, a, b, o, chunk, ticket);
}
int fotg_node_submit_scatter(fotg_node *nd, int n, const float *I0, const float *I1, float *outflow, int chunk, long *ticket)
{
  return node_scatter(nd, 0, n, I0, I1, outflow, chunk, ticket);
}
int fotg_node_submit_scatter_u8(fotg_node *nd, int n, const unsigned char *I0, const unsigned char *I1, float *outflow, int chunk, long *ticket)
{
  return node_scatter(nd, 1, n, I0, I1, outflow, chunk, ticket);
}

int fotg_node_wait(fotg_node *nd, long ticket)
{
  if (!nd || ticket < 0) return FOTG_ERR_ARG;
  std::lock_guard<std::mutex> one(nd->wait_mu);
  long first;
  {
    std::unique_lock<std::mutex> lk(nd->mu);
    if (ticket >= nd->submitted) return FOTG_ERR_ARG;
    // a job that has been waited for before (tickets are waited for in order): its OWN status, as long as the ring remembers it
    if (ticket < nd->waited) return nd->jstatus_id[ticket % RING] == ticket ? nd->jstatus[ticket % RING] : FOTG_OK;
    nd->cv_issued.wait(lk, [&] { for (int k = 0; k < nd->ndev; ++k) if (nd->slot[k].issued <= ticket) return false; return true; });
    first = nd->waited;
  }
  // every job up to `ticket`, in order (their ring entries become free).  A piece whose tile solver gave up a bounded wait is
  // recomputed by the pipe's host wait where its frames are still in place (resident shards, the source slot of a scatter) and
  // reported (FOTG_ERR_STALL for THIS job, on every wait for it) where they are not (pulled pieces: their staging buffers have been
  // recycled).  The return value is the worst status of the jobs this call covers; each job keeps its own for later waits.
  int worst = FOTG_OK;
  for (long id = first; id <= ticket; ++id) {
    int st = FOTG_OK;
    for (int k = 0; k < nd->ndev; ++k) {
      Slot &s = nd->slot[k];
      if (s.status[id % RING] != FOTG_OK) { st = s.status[id % RING]; if (st == FOTG_ERR_HIP) nd->last_hip = s.hip_err[id % RING]; continue; }
      OnDevice od(s.device);
      if (s.on_copy[id % RING] && hipEventSynchronize(s.done[id % RING]) != hipSuccess) { st = FOTG_ERR_HIP; nd->last_hip = (int)hipGetLastError(); }
      for (long t : s.piece_tk[id % RING]) {
        const int sp = fotg_pipe_wait(s.pipe, t, nullptr, s.on_copy[id % RING] ? 2 : 1);
        if (sp == FOTG_ERR_HIP) nd->last_hip = fotg_last_hip_error();
        if (sp != FOTG_OK && st == FOTG_OK) st = sp;
      }
    }
    {
      std::lock_guard<std::mutex> lk(nd->mu);
      nd->jstatus[id % RING] = st; nd->jstatus_id[id % RING] = id;
      nd->waited = id + 1;
    }
    if (st != FOTG_OK && worst == FOTG_OK) worst = st;
  }
  return worst;
}

int fotg_node_last_hip_error(const fotg_node *nd) { return nd ? nd->last_hip : 0; }

int fotg_node_sync(fotg_node *nd)
{
  if (!nd) return FOTG_ERR_ARG;
  long last;
  {
    std::lock_guard<std::mutex> lk(nd->mu);
    last = nd->submitted - 1;
  }
  return last < 0 ? FOTG_OK : fotg_node_wait(nd, last);
}

int fotg_node_info(const fotg_node *nd, int *ndev, int *out_w, int *out_h, int *flow_channels)
{
  if (!nd) return FOTG_ERR_ARG;
  if (ndev) *ndev = nd->ndev;
  if (out_w) *out_w = nd->ow;
  if (out_h) *out_h = nd->oh;
  if (flow_channels) *flow_channels = nd->nch;
  return FOTG_OK;
}

int fotg_node_pipe(fotg_node *nd, int slot, fotg_pipe **pipe)
{
  if (!nd || !pipe || slot < 0 || slot >= nd->ndev) return FOTG_ERR_ARG;
  *pipe = nd->slot[slot].pipe;
  return FOTG_OK;
}

}  // extern "C"
