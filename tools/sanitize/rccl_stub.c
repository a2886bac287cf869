/* rccl_stub.c -- a HOST-ONLY stand-in for librccl.so.1 over the HIP stub (hip_stub.c), for the sanitizer runs of mv_comm's RCCL
 * transport: ncclCommInitAll + ONE grouped ncclAllGather per exchange (csrc/mv_comm.hip: exchange()).
 *
 * No 8-GPU node has run that path yet (1-GPU boxes: the communicator has only ever had one rank), so what a real RCCL would
 * punish with a hang or a corrupted gather is checked here and aborts with a message:
 *   - an ncclAllGather outside ncclGroupStart/End on a communicator of > 1 ranks driven from one thread (a real RCCL blocks
 *     in the first rank's call, waiting for ranks that the same thread has not posted yet),
 *   - a group that ends without exactly ONE all-gather from EVERY rank of the communicator clique, or with differing counts / types,
 *   - a rank's call on a stream, or with a send / receive buffer, of another device than the rank's,
 *   - a receive buffer too small for nranks x count (the HIP stub's allocation registry bounds-checks it),
 *   - use of a destroyed communicator.
 * The gather itself is memcpy between the eight "devices": rank r's receive buffer gets every rank's piece in RANK order.
 * Test infrastructure only (tools/sanitize/run.sh puts it in front of the real library with LD_LIBRARY_PATH). */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef void* hipStream_t;
enum { ncclSuccess = 0, ncclInvalidArgument = 4 };

int hipstub_ptr_device(const void* p, size_t n);
int hipstub_stream_device(hipStream_t s);

#define DIE(...)                                                   \
  do {                                                             \
    fprintf(stderr, "RCCL STUB VIOLATION: " __VA_ARGS__);          \
    fprintf(stderr, "\n");                                         \
    abort();                                                       \
  } while (0)

#define COMM_MAGIC 0x4e43434cu
#define MAX_RANKS 64
typedef struct clique { int nranks; long gathers; } clique_t;
typedef struct stub_comm { uint32_t magic; int rank, nranks, device; clique_t* clique; } *ncclComm_t;

static long g_gathers; /* completed grouped all-gathers (all cliques) */
long rcclstub_gathers(void) { return __atomic_load_n(&g_gathers, __ATOMIC_RELAXED); }

static size_t type_size(ncclDataType_t t) {
  switch (t) {
    case 0: case 1: return 1;          /* ncclInt8 / ncclChar, ncclUint8 */
    case 2: case 3: case 7: return 4;  /* int32, uint32, float32 */
    case 4: case 5: case 8: return 8;  /* int64, uint64, float64 */
    case 6: case 9: return 2;          /* float16, bfloat16 */
    default: DIE("unknown ncclDataType_t %d", t);
  }
  return 0;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (!comms || ndev < 1 || ndev > MAX_RANKS) return ncclInvalidArgument;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (devlist && devlist[i] == devlist[j]) DIE("ncclCommInitAll: device %d named twice (ranks %d and %d)", devlist[i], j, i);
  clique_t* cl = calloc(1, sizeof(*cl));
  cl->nranks = ndev;
  for (int i = 0; i < ndev; ++i) {
    comms[i] = calloc(1, sizeof(**comms));
    comms[i]->magic = COMM_MAGIC;
    comms[i]->rank = i;
    comms[i]->nranks = ndev;
    comms[i]->device = devlist ? devlist[i] : i;
    comms[i]->clique = cl;
  }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c || c->magic != COMM_MAGIC) DIE("ncclCommDestroy: stale communicator %p", (void*)c);
  c->magic = 0;
  if (c->rank == c->nranks - 1) free(c->clique); /* mv_comm destroys the ranks in order */
  free(c);
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "error (rccl stub)"; }

/* ---------------------------------------------------------------- groups (per thread, like the real thing) */
typedef struct { ncclComm_t comm; const void* send; void* recv; size_t count; ncclDataType_t type; } op_t;
static __thread int t_depth;
static __thread op_t t_ops[MAX_RANKS * 4];
static __thread int t_n_ops;

ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }

static void run_group(void) {
  /* every clique present in the group must be complete: one op per rank, same shape */
  int done[MAX_RANKS * 4] = {0};
  for (int i = 0; i < t_n_ops; ++i) {
    if (done[i]) continue;
    clique_t* cl = t_ops[i].comm->clique;
    op_t* by_rank[MAX_RANKS] = {0};
    for (int j = i; j < t_n_ops; ++j) {
      if (done[j] || t_ops[j].comm->clique != cl) continue;
      const int r = t_ops[j].comm->rank;
      if (by_rank[r]) {  /* a second gather of the same clique in one group: must again cover every rank -- handled as the next round */
        continue;
      }
      by_rank[r] = &t_ops[j];
      done[j] = 1;
    }
    const size_t bytes = t_ops[i].count * type_size(t_ops[i].type);
    for (int r = 0; r < cl->nranks; ++r) {
      if (!by_rank[r]) DIE("ncclGroupEnd: rank %d of a %d-rank communicator posted no ncclAllGather in this group (a real RCCL hangs)", r, cl->nranks);
      if (by_rank[r]->count != t_ops[i].count || by_rank[r]->type != t_ops[i].type)
        DIE("ncclGroupEnd: rank %d gathers %zu elements of type %d, rank %d %zu of type %d", r, by_rank[r]->count, by_rank[r]->type, t_ops[i].comm->rank, t_ops[i].count, t_ops[i].type);
    }
    for (int r = 0; r < cl->nranks; ++r)
      for (int j = 0; j < cl->nranks; ++j)
        if (bytes) memmove((char*)by_rank[r]->recv + (size_t)j * bytes, by_rank[j]->send, bytes);
    __atomic_fetch_add(&g_gathers, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&cl->gathers, 1, __ATOMIC_RELAXED);
  }
  t_n_ops = 0;
}

ncclResult_t ncclGroupEnd(void) {
  if (t_depth <= 0) DIE("ncclGroupEnd without ncclGroupStart");
  if (--t_depth == 0) run_group();
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
  if (!c || c->magic != COMM_MAGIC) DIE("ncclAllGather: stale communicator %p", (void*)c);
  const size_t bytes = count * type_size(type);
  const int sdev = hipstub_stream_device(stream);
  if (sdev != c->device) DIE("ncclAllGather: rank %d lives on device %d, its stream on device %d", c->rank, c->device, sdev);
  if (bytes) {
    const int a = hipstub_ptr_device(send, bytes), b = hipstub_ptr_device(recv, bytes * (size_t)c->nranks);
    if (a != c->device) DIE("ncclAllGather: rank %d (device %d) sends from memory of device %d", c->rank, c->device, a);
    if (b != c->device) DIE("ncclAllGather: rank %d (device %d) receives into memory of device %d", c->rank, c->device, b);
  }
  if (t_depth == 0) {
    if (c->nranks > 1) DIE("ncclAllGather of rank %d outside ncclGroupStart/End: one thread driving %d ranks must group them (a real RCCL blocks here)", c->rank, c->nranks);
    if (bytes) memmove(recv, send, bytes);
    __atomic_fetch_add(&g_gathers, 1, __ATOMIC_RELAXED);
    return ncclSuccess;
  }
  if (t_n_ops >= MAX_RANKS * 4) DIE("ncclAllGather: more than %d operations in one group", MAX_RANKS * 4);
  t_ops[t_n_ops++] = (op_t){c, send, recv, count, type};
  return ncclSuccess;
}
