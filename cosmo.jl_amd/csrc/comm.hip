// comm.hip -- clique-sharded projections over the GPUs of one node (option 1: replicated affine step,
// sharded projection, ONE exchange step per iteration).
//
// Every rank (one process per GPU) holds the whole problem and runs the identical, bit-reproducible x-/w-steps; the cone
// projections -- which dominate SDP iterations (`proj_time`, src/solver.jl:15,152; docs/src/performance.md:35) -- are
// partitioned: rank r projects only the SOC / PSD cones of a CONTIGUOUS range of cones, whose rows form one contiguous slice
// of `s`.  After the local projections each rank's slice is broadcast in place from its owner (RCCL ncclBroadcast calls
// inside one group on the handle's stream = an all-gather with unequal counts).  On a fully connected xGMI node every link
// carries 1/N of the payload once.  No other collective exists: all scalars (CG, rho, residuals) are recomputed redundantly
// and agree bit for bit because the gathered `s` is bit-identical everywhere.
//
// RCCL is loaded lazily with dlopen so that the library has no RCCL dependency for single-GPU use and shares the RCCL
// instance of the hosting process (torch bundles its own librccl.so; loading a second copy would clash).
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <stdarg.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include "internal.h"

typedef struct { char internal[128]; } nccl_uid_t;
typedef void* nccl_comm_t;
enum { NCCL_REAL = REAL_IS_FLOAT ? 7 : 8 };   // ncclFloat32 = 7 / ncclFloat64 = 8 (rccl.h ncclDataType_t): the element type of `real`

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(nccl_uid_t*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_uid_t, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
static RcclApi g_rccl;

static const char* rccl_load() {
  if (g_rccl.lib) return nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
  if (!lib) return "cannot dlopen librccl.so";
  g_rccl.GetUniqueId = (int (*)(nccl_uid_t*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(nccl_comm_t*, int, nccl_uid_t, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(nccl_comm_t))dlsym(lib, "ncclCommDestroy");
  g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(lib, "ncclBroadcast");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(lib, "ncclAllReduce");
  g_rccl.GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  g_rccl.GetVersion = (int (*)(int*))dlsym(lib, "ncclGetVersion");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.Broadcast || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd)
    return "librccl.so lacks an expected symbol";
  g_rccl.lib = lib;
  return nullptr;
}

#define NCHK(h, call)                                                                                                     \
  do {                                                                                                                    \
    int r__ = (call);                                                                                                     \
    if (r__ != 0)                                                                                                         \
      return cosmo_fail((h), COSMO_HIP_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); \
  } while (0)

// Host-staged transport (cosmo_hip_comm_init_hostshm): the ranks are processes that may share ONE GPU (RCCL refuses two ranks
// on one device), the slices travel through a POSIX shared-memory segment.  It exists so that the sharded loop -- ownership
// ranges, row_lo / row_hi slices, the exchange point of the iteration, the flag reduction of the certificates -- executes with
// nranks > 1 on a single-GPU box; it is synchronous and slow by construction and is not a production path.
#define COSMO_SHM_MAX_RANKS 16
struct ShmSeg {
  std::atomic<int> arrive;     // sense-reversing barrier
  std::atomic<int> gen;
  int flag[COSMO_SHM_MAX_RANKS];
  double hvals[COSMO_SHM_MAX_RANKS][8];   // comm_allreduce_host
  long long capacity;          // reals in data[]
  real data[1];
};

struct CommState {
  nccl_comm_t comm = nullptr;
  int rank = 0, nranks = 1;
  std::vector<long long> first_cone;   // nranks + 1 boundaries (cone indices)
  std::vector<long long> row_lo, row_hi;
  int* d_flag = nullptr;               // 2 ints: send / receive buffer of comm_allreduce_flag
  // host-staged transport
  ShmSeg* shm = nullptr;
  size_t shm_bytes = 0;
  std::string shm_name;
  long long exchanges = 0;             // exchange steps executed with nranks > 1 (cosmo_hip_comm_stats)
  long long bytes = 0;                 // payload bytes this rank contributed to / received from collectives of the loop
  double* d_hv = nullptr;              // 16 doubles: send / receive buffer of comm_allreduce_host
  std::vector<real> hsum;              // host-staged all-reduce: the summed vector
};

int comm_nranks(const cosmo_hip_handle* h) { return h->comm ? ((const CommState*)h->comm)->nranks : 1; }
int comm_rank(const cosmo_hip_handle* h) { return h->comm ? ((const CommState*)h->comm)->rank : 0; }

static int32_t shm_barrier(cosmo_hip_handle* h, CommState* c) {
  ShmSeg* g = c->shm;
  const int gen = g->gen.load(std::memory_order_acquire);
  if (g->arrive.fetch_add(1, std::memory_order_acq_rel) == c->nranks - 1) {
    g->arrive.store(0, std::memory_order_relaxed);
    g->gen.store(gen + 1, std::memory_order_release);
    return COSMO_HIP_OK;
  }
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (long spin = 0; g->gen.load(std::memory_order_acquire) == gen; ++spin) {
    sched_yield();
    if ((spin & 1023) == 1023) {
      timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
      if (t1.tv_sec - t0.tv_sec > 60) return cosmo_fail(h, COSMO_HIP_ERR_COMM, "host-staged barrier timed out (a peer rank died?)");
    }
  }
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_comm_unique_id(uint8_t id[128]) {
  if (!id) return COSMO_HIP_ERR_INVALID;
  if (rccl_load()) return COSMO_HIP_ERR_COMM;
  nccl_uid_t u;
  if (g_rccl.GetUniqueId(&u) != 0) return COSMO_HIP_ERR_COMM;
  memcpy(id, u.internal, 128);
  return COSMO_HIP_OK;
}

extern "C" int32_t cosmo_hip_comm_destroy(cosmo_hip_handle* h) {
  if (!h || !h->comm) return COSMO_HIP_OK;
  CommState* c = (CommState*)h->comm;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->shm) { (void)munmap(c->shm, c->shm_bytes); if (c->rank == 0) (void)shm_unlink(c->shm_name.c_str()); }
  if (c->d_flag) (void)hipFree(c->d_flag);
  if (c->d_hv) (void)hipFree(c->d_hv);
  delete c;
  h->comm = nullptr;
  h->cone_lo = 0; h->cone_hi = -1;
  return COSMO_HIP_OK;
}

// One communicator per handle (one process per GPU).  `id` comes from cosmo_hip_comm_unique_id on rank 0 and is
// distributed by the host layer (torch.distributed / MPI / a Julia Distributed broadcast).
extern "C" int32_t cosmo_hip_comm_init(cosmo_hip_handle* h, int32_t rank, int32_t nranks, const uint8_t id[128]) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_init: bad arguments");
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (const char* e = rccl_load()) return cosmo_fail(h, COSMO_HIP_ERR_COMM, "%s", e);
  (void)cosmo_hip_comm_destroy(h);
  CommState* c = new CommState();
  c->rank = rank; c->nranks = nranks;
  nccl_uid_t u; memcpy(u.internal, id, 128);
  const int rc = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
  if (rc != 0) { delete c; return cosmo_fail(h, COSMO_HIP_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"); }
  h->comm = c;
  return COSMO_HIP_OK;
}

// Host-staged communicator (see ShmSeg): `name` is a POSIX shared-memory name ("/cosmo_...") unique to the job; rank 0 creates
// the segment, the other ranks attach.  Requires set_problem (the segment holds m doubles).
extern "C" int32_t cosmo_hip_comm_init_hostshm(cosmo_hip_handle* h, int32_t rank, int32_t nranks, const char* name) {
  if (!h || !name || name[0] != '/' || nranks < 1 || nranks > COSMO_SHM_MAX_RANKS || rank < 0 || rank >= nranks)
    return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_init_hostshm: bad arguments");
  if (!h->have_problem) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_init_hostshm: set_problem first");
  (void)cosmo_hip_comm_destroy(h);
  CommState* c = new CommState();
  c->rank = rank; c->nranks = nranks; c->shm_name = name;
  // room for a full-length row vector (all-gather of slices) and for nranks partial n-vectors (+ 2 nranks scalars) of the all-reduce
  // ... and for the partial arrays of the accelerator's inner products in row-sharded accelerated runs (anderson.hip: (AA_MAX_MEM + 1) slots)
  const long long cap = std::max<long long>(std::max<long long>(std::max<long long>(h->m, 1), (long long)nranks * (h->n + 2LL * nranks + 8)),
                                             (long long)nranks * 32LL * COSMO_MAX_PARTIALS);
  c->shm_bytes = sizeof(ShmSeg) + sizeof(real) * (size_t)cap;
  int fd = -1;
  if (rank == 0) {
    (void)shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) { if (fd >= 0) close(fd); delete c; return cosmo_fail(h, COSMO_HIP_ERR_COMM, "shm_open/ftruncate(%s) failed", name); }
  } else {
    for (int tries = 0; tries < 6000 && fd < 0; ++tries) {          // wait (<= 60 s) for rank 0 to create and size the segment
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) { struct stat sb; if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < c->shm_bytes) { close(fd); fd = -1; } }
      if (fd < 0) usleep(10000);
    }
    if (fd < 0) { delete c; return cosmo_fail(h, COSMO_HIP_ERR_COMM, "cannot attach to shared segment %s", name); }
  }
  void* p = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return cosmo_fail(h, COSMO_HIP_ERR_COMM, "mmap of %s failed", name); }
  c->shm = (ShmSeg*)p;
  if (rank == 0) c->shm->capacity = cap;       // ftruncate zero-filled the barrier words
  h->comm = c;
  return shm_barrier(h, c);                    // everybody attached
}

// out = {nranks, rank, exchange steps executed with nranks > 1, transport (0 none, 1 RCCL, 2 host-staged)}
extern "C" int32_t cosmo_hip_comm_stats(cosmo_hip_handle* h, int64_t out[4]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  out[0] = 1; out[1] = 0; out[2] = 0; out[3] = 0;
  if (!h->comm) return COSMO_HIP_OK;
  const CommState* c = (const CommState*)h->comm;
  out[0] = c->nranks; out[1] = c->rank; out[2] = c->exchanges; out[3] = c->shm ? 2 : 1;
  return COSMO_HIP_OK;
}

// out = {nranks, rank, collectives of the loop executed with nranks > 1, transport, mode (0 none, 1 cone-sharded projections, 2 row-sharded),
//        payload bytes of those collectives (this rank's view), all-reduces of n-vectors, elements per all-reduce (last)}
extern "C" int32_t cosmo_hip_comm_stats_ex(cosmo_hip_handle* h, int64_t out[8]) {
  if (!h || !out) return COSMO_HIP_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  out[0] = 1;
  if (!h->comm) return COSMO_HIP_OK;
  const CommState* c = (const CommState*)h->comm;
  out[0] = c->nranks; out[1] = c->rank; out[2] = c->exchanges; out[3] = c->shm ? 2 : 1;
  out[4] = h->row_shard ? 2 : (c->first_cone.empty() ? 0 : 1);
  out[5] = c->bytes; out[6] = h->rs_allreduces; out[7] = h->rs_allreduce_elems;
  return COSMO_HIP_OK;
}

// first_cone: nranks+1 non-decreasing cone indices, first_cone[0] = 0, first_cone[nranks] = ncones.  Rank r projects the
// SOC / PSD cones first_cone[r] <= k < first_cone[r+1]; Zero / Nonnegatives / Box rows are projected by everyone (they are
// part of the elementwise copy kernel).  Must be called after cosmo_hip_set_cones (rebuilds the SOC table and PSD plan).
int32_t rebuild_cone_plans(cosmo_hip_handle* h);   // api.hip
// boundaries -> CommState (first_cone, row_lo / row_hi of every rank), validated against the composite set currently installed
int32_t comm_set_partition(cosmo_hip_handle* h, const int64_t* first_cone, const char* who) {
  if (!h->comm) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "%s: comm_init first", who);
  if (!h->have_cones) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "%s: set_cones first", who);
  CommState* c = (CommState*)h->comm;
  const long long nc = (long long)h->cones.type.size();
  if (first_cone[0] != 0 || first_cone[c->nranks] != nc) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "%s: boundaries must span all cones", who);
  c->first_cone.assign(first_cone, first_cone + c->nranks + 1);
  c->row_lo.assign(c->nranks, 0); c->row_hi.assign(c->nranks, 0);
  for (int r = 0; r < c->nranks; ++r) {
    if (first_cone[r + 1] < first_cone[r]) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "%s: boundaries must be non-decreasing", who);
    const long long a = first_cone[r], b = first_cone[r + 1];
    c->row_lo[r] = (a < nc) ? h->cones.off[a] : h->m;
    c->row_hi[r] = (b < nc) ? h->cones.off[b] : h->m;
  }
  return COSMO_HIP_OK;
}
void comm_my_range(const cosmo_hip_handle* h, long long* cone_lo, long long* cone_hi, long long* row_lo, long long* row_hi) {
  const CommState* c = (const CommState*)h->comm;
  *cone_lo = c->first_cone[c->rank]; *cone_hi = c->first_cone[c->rank + 1];
  *row_lo = c->row_lo[c->rank]; *row_hi = c->row_hi[c->rank];
}

extern "C" int32_t cosmo_hip_set_cone_shard(cosmo_hip_handle* h, const int64_t* first_cone) {
  if (!h || !first_cone) return COSMO_HIP_ERR_INVALID;
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_cone_shard: the handle is row-sharded");
  CHK(comm_set_partition(h, first_cone, "set_cone_shard"));
  CommState* c = (CommState*)h->comm;
  h->cone_lo = first_cone[c->rank];
  h->cone_hi = first_cone[c->rank + 1];
  return rebuild_cone_plans(h);
}

// Ownership only (no communicator): this handle projects the SOC / PSD cones cone_lo <= k < cone_hi and leaves the other
// cones' rows untouched.  Building block of cosmo_hip_set_cone_shard; also lets the sharded projection be verified on a
// single GPU (project with two handles owning complementary ranges, merge the slices).
extern "C" int32_t cosmo_hip_set_cone_ownership(cosmo_hip_handle* h, int64_t cone_lo, int64_t cone_hi) {
  if (!h) return COSMO_HIP_ERR_INVALID;
  if (!h->have_cones) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_cone_ownership: set_cones first");
  if (cone_lo < 0 || cone_hi < cone_lo || cone_hi > (int64_t)h->cones.type.size()) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "set_cone_ownership: bad range");
  h->cone_lo = cone_lo; h->cone_hi = cone_hi;
  return rebuild_cone_plans(h);
}

// in-place all-gather of a full-length (m_g) row vector: every owner broadcasts its slice (enqueued on the handle's stream)
int32_t comm_allgather_rows(cosmo_hip_handle* h, real* s) {
  if (!h->comm) return COSMO_HIP_OK;
  CommState* c = (CommState*)h->comm;
  if (c->nranks == 1 || c->first_cone.empty()) return COSMO_HIP_OK;
  long long total = 0;
  for (int r = 0; r < c->nranks; ++r) total += c->row_hi[r] - c->row_lo[r];
  c->bytes += (long long)sizeof(real) * total;
  if (c->shm) {
    // host-staged: owner slice -> segment, barrier, the other ranks' slices <- segment, barrier (the segment is reused next time)
    const long long lo = c->row_lo[c->rank], hi = c->row_hi[c->rank];
    if (hi > c->shm->capacity) return cosmo_fail(h, COSMO_HIP_ERR_COMM, "shared segment too small");
    if (hi > lo) HIPCHK(h, hipMemcpyAsync(c->shm->data + lo, s + lo, sizeof(real) * (size_t)(hi - lo), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(shm_barrier(h, c));
    for (int r = 0; r < c->nranks; ++r) {
      const long long cnt = c->row_hi[r] - c->row_lo[r];
      if (r == c->rank || cnt <= 0) continue;
      HIPCHK(h, hipMemcpyAsync(s + c->row_lo[r], c->shm->data + c->row_lo[r], sizeof(real) * (size_t)cnt, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return shm_barrier(h, c);
  }
  NCHK(h, g_rccl.GroupStart());
  for (int r = 0; r < c->nranks; ++r) {
    const long long cnt = c->row_hi[r] - c->row_lo[r];
    if (cnt <= 0) continue;
    NCHK(h, g_rccl.Broadcast(s + c->row_lo[r], s + c->row_lo[r], (size_t)cnt, NCCL_REAL, r, c->comm, h->stream));
  }
  NCHK(h, g_rccl.GroupEnd());
  return COSMO_HIP_OK;
}

// clique sharding (option 1): the one exchange step of the iteration.  Row-sharded handles (option 2) never exchange s.
int32_t comm_enqueue_exchange(cosmo_hip_handle* h, real* s) {
  if (!h->comm || h->row_shard) return COSMO_HIP_OK;
  CommState* c = (CommState*)h->comm;
  if (c->nranks == 1 || c->first_cone.empty()) return COSMO_HIP_OK;
  c->exchanges += 1;
  return comm_allgather_rows(h, s);
}

// In-place sum over the ranks of `count` reals at `buf` (device), enqueued on the handle's stream.  Every rank receives the SAME bits:
// RCCL reduces every element along one fixed path and distributes the result; the host-staged transport adds the ranks' vectors in
// rank order on every rank.  (ncclSum = 0.)
// TEST HOOK (COSMO_HIP_COMM_CORRUPT_RANK=r): rank r's contribution to every all-reduce is scaled by 1 + 1e-3 before it leaves the rank -- a
// deliberately wrong exchange, so that the parity evidence a multi-GPU bench line carries (bench.py: comm.selftest,
// sharded_vs_single_max_rel_dev) can be shown to turn red (tests/test_gpu_sharding.py).  Never set in production.
__global__ void k_comm_corrupt(long long count, real* buf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) buf[i] = buf[i] * R(1.001);
}
static int corrupt_rank() {
  static int r = -2;
  if (r == -2) {
    const char* e = getenv("COSMO_HIP_COMM_CORRUPT_RANK"); r = e ? atoi(e) : -1;
    // never silent: a leaked variable in a job script must not produce wrong answers that still read `Solved`
    if (r >= 0) fprintf(stderr, "libcosmo_hip: WARNING: TEST HOOK COSMO_HIP_COMM_CORRUPT_RANK=%d is active -- rank %d's contribution to every all-reduce is "
                                "scaled by 1.001; results of sharded runs are WRONG on purpose\n", r, r);
  }
  return r;
}

int32_t comm_allreduce_sum(cosmo_hip_handle* h, real* buf, size_t count) {
  if (!h->comm || count == 0) return COSMO_HIP_OK;
  CommState* c = (CommState*)h->comm;
  if (c->nranks == 1) return COSMO_HIP_OK;
  c->exchanges += 1;
  c->bytes += (long long)(sizeof(real) * count);
  if (corrupt_rank() == c->rank)
    hipLaunchKernelGGL(k_comm_corrupt, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, (long long)count, buf);
  if (c->shm) {
    if ((long long)(count * (size_t)c->nranks) > c->shm->capacity) return cosmo_fail(h, COSMO_HIP_ERR_COMM, "shared segment too small for the all-reduce");
    HIPCHK(h, hipMemcpyAsync(c->shm->data + (size_t)c->rank * count, buf, sizeof(real) * count, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(shm_barrier(h, c));
    c->hsum.assign(count, R(0.0));
    for (int r = 0; r < c->nranks; ++r) {
      const real* src = c->shm->data + (size_t)r * count;
      if (r == 0) for (size_t i = 0; i < count; ++i) c->hsum[i] = src[i];
      else for (size_t i = 0; i < count; ++i) c->hsum[i] += src[i];
    }
    HIPCHK(h, hipMemcpyAsync(buf, c->hsum.data(), sizeof(real) * count, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return shm_barrier(h, c);
  }
  NCHK(h, g_rccl.AllReduce(buf, buf, count, NCCL_REAL, 0, c->comm, h->stream));
  return COSMO_HIP_OK;
}

// Synchronous all-reduce of up to 8 host doubles (op 0: sum in rank order / RCCL sum, op 1: max).  All ranks must call it at the same
// point; used by the certificates of row-sharded runs, whose scalar comparisons are chained on the host.  ncclFloat64 = 8, ncclMax = 2.
int32_t comm_allreduce_host(cosmo_hip_handle* h, double* vals, int count, int op) {
  if (!h->comm || count <= 0) return COSMO_HIP_OK;
  CommState* c = (CommState*)h->comm;
  if (c->nranks == 1) return COSMO_HIP_OK;
  if (count > 8) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_allreduce_host: at most 8 values");
  if (c->shm) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < count; ++i) c->shm->hvals[c->rank][i] = vals[i];
    CHK(shm_barrier(h, c));
    for (int i = 0; i < count; ++i) {
      double a = c->shm->hvals[0][i];
      for (int r = 1; r < c->nranks; ++r) { const double v = c->shm->hvals[r][i]; a = (op == 0) ? a + v : ((v > a || v != v) ? v : a); }
      vals[i] = a;
    }
    return shm_barrier(h, c);
  }
  if (!c->d_hv) HIPCHK(h, hipMalloc((void**)&c->d_hv, 16 * sizeof(double)));
  HIPCHK(h, hipMemcpyAsync(c->d_hv, vals, sizeof(double) * count, hipMemcpyHostToDevice, h->stream));
  NCHK(h, g_rccl.AllReduce(c->d_hv, c->d_hv + 8, (size_t)count, 8, op == 0 ? 0 : 2, c->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(vals, c->d_hv + 8, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return COSMO_HIP_OK;
}

// Infeasibility certificates in clique-sharded runs: every rank tests the SOC / PSD cones it owns; `*flag` (0 / 1, "a cone of mine
// violates the membership test") is combined with a max over the ranks.  All ranks take the same branches up to here (the
// scalars of the certificates are computed redundantly and bit-identically), so the collective is matched.  ncclInt32 = 2,
// ncclMax = 2.  Synchronous (the caller needs the decision on the host).
int32_t comm_allreduce_flag(cosmo_hip_handle* h, int* flag) {
  if (!h->comm) return COSMO_HIP_OK;
  CommState* c = (CommState*)h->comm;
  if (c->shm) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    c->shm->flag[c->rank] = *flag;
    CHK(shm_barrier(h, c));
    int v = 0;
    for (int r = 0; r < c->nranks; ++r) v = c->shm->flag[r] > v ? c->shm->flag[r] : v;
    CHK(shm_barrier(h, c));
    *flag = v;
    return COSMO_HIP_OK;
  }
  if (!c->d_flag) HIPCHK(h, hipMalloc((void**)&c->d_flag, 2 * sizeof(int)));
  HIPCHK(h, hipMemcpyAsync(c->d_flag, flag, sizeof(int), hipMemcpyHostToDevice, h->stream));
  NCHK(h, g_rccl.AllReduce(c->d_flag, c->d_flag + 1, 1, 2, 2, c->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(flag, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return COSMO_HIP_OK;
}

// self-test hook: broadcast-in-place of the whole s from rank 0 through the communicator (exercises the RCCL path even
// on a single-rank communicator, where the sharded exchange is skipped)
extern "C" int32_t cosmo_hip_comm_selftest(cosmo_hip_handle* h) {
  if (!h || !h->comm || !h->have_iterates) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_selftest: comm_init and set_iterates first");
  CommState* c = (CommState*)h->comm;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  if (h->row_shard) return cosmo_fail(h, COSMO_HIP_ERR_UNSUPPORTED, "comm_selftest: s is a row slice of rank-dependent length on a row-sharded handle (use cosmo_hip_comm_allreduce_check)");
  if (c->shm) return shm_barrier(h, c);
  NCHK(h, g_rccl.GroupStart());
  NCHK(h, g_rccl.Broadcast(h->s, h->s, (size_t)h->m, NCCL_REAL, 0, c->comm, h->stream));
  NCHK(h, g_rccl.GroupEnd());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return COSMO_HIP_OK;
}

// Known-answer test of the all-reduce the row-sharded loop relies on, through the SAME code path (comm_allreduce_sum on the handle's
// stream, `count` reals -- pass n to exercise the loop's message size).  Two vectors:
//   A. integer-valued contributions (rank + 1) * (i mod 997 + 1): the sum N (N + 1) / 2 * (i mod 997 + 1) is exact in any order
//      => out[0] = number of elements that differ from it (must be 0);
//   B. fractional contributions of mixed magnitude from a counter-based generator every rank can evaluate for every rank: the result
//      must lie within nranks * eps * sum |contribution| of the long-double sum (out[1] = elements outside), and its BITS must be the same
//      on every rank (the replicated n-side of the loop stays bit-identical only then): out[2] = FNV-1a hash of the result's bytes, to be
//      compared across the ranks by the caller.
// out[3] = transport (1 RCCL, 2 host-staged), out[4] = nranks, out[5] = RCCL version code (ncclGetVersion; 0 on the host-staged transport).
// Collective: every rank of the communicator must call it with the same count.
static inline double chk_contrib(int rank, long long i) {
  unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + (unsigned long long)(rank + 1) * 0xBF58476D1CE4E5B9ull;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
  const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;          // [-0.5, 0.5)
  static const double mag[7] = {1e-3, 1e-2, 1e-1, 1.0, 1e1, 1e2, 1e3};
  return u * mag[i % 7];
}
extern "C" int32_t cosmo_hip_comm_allreduce_check(cosmo_hip_handle* h, int64_t count, int64_t out[6]) {
  if (!h || !out || count <= 0) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_allreduce_check: bad arguments");
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (!h->comm) return cosmo_fail(h, COSMO_HIP_ERR_INVALID, "comm_allreduce_check: comm_init first");
  CommState* c = (CommState*)h->comm;
  if (hipSetDevice(h->device) != hipSuccess) return cosmo_fail(h, COSMO_HIP_ERR_HIP, "hipSetDevice failed");
  out[3] = c->shm ? 2 : 1; out[4] = c->nranks;
  if (!c->shm && g_rccl.GetVersion) { int v = 0; if (g_rccl.GetVersion(&v) == 0) out[5] = v; }
  if (c->shm && (long long)count * c->nranks > c->shm->capacity) return cosmo_fail(h, COSMO_HIP_ERR_COMM, "comm_allreduce_check: shared segment too small for %lld reals", (long long)count);
  real* d = nullptr;
  HIPCHK(h, hipMalloc((void**)&d, sizeof(real) * (size_t)count));
  std::vector<real> v((size_t)count);
  const long long ex0 = c->exchanges, by0 = c->bytes;               // the check is not part of the loop's exchange statistics
  int32_t rc = COSMO_HIP_OK;
  for (int pass = 0; pass < 2 && rc == COSMO_HIP_OK; ++pass) {
    for (long long i = 0; i < count; ++i) v[(size_t)i] = pass == 0 ? (real)((double)(c->rank + 1) * (double)(i % 997 + 1)) : (real)chk_contrib(c->rank, i);
    if (hipMemcpyAsync(d, v.data(), sizeof(real) * (size_t)count, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = cosmo_fail(h, COSMO_HIP_ERR_HIP, "comm_allreduce_check: upload failed"); break; }
    rc = comm_allreduce_sum(h, d, (size_t)count);
    if (rc != COSMO_HIP_OK) break;
    if (hipMemcpyAsync(v.data(), d, sizeof(real) * (size_t)count, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
      rc = cosmo_fail(h, COSMO_HIP_ERR_HIP, "comm_allreduce_check: download failed"); break; }
    if (pass == 0) {
      const double tri = 0.5 * c->nranks * (c->nranks + 1.0);
      for (long long i = 0; i < count; ++i) if ((double)v[(size_t)i] != tri * (double)(i % 997 + 1)) out[0] += 1;
    } else {
      unsigned long long hsh = 1469598103934665603ull;
      const unsigned char* pb = (const unsigned char*)v.data();
      for (size_t k = 0; k < sizeof(real) * (size_t)count; ++k) { hsh ^= pb[k]; hsh *= 1099511628211ull; }
      out[2] = (int64_t)hsh;
      for (long long i = 0; i < count; ++i) {
        long double sum = 0.0L, mag = 0.0L;
        for (int r = 0; r < c->nranks; ++r) { const long double t = (long double)(real)chk_contrib(r, i); sum += t; mag += t < 0 ? -t : t; }
        const long double err = (long double)v[(size_t)i] - sum;
        if ((err < 0 ? -err : err) > (long double)c->nranks * (long double)REAL_EPS * mag) out[1] += 1;
      }
    }
  }
  c->exchanges = ex0; c->bytes = by0;
  (void)hipFree(d);
  return rc;
}
