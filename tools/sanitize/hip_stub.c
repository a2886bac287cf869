/* hip_stub.c -- a HOST-ONLY stand-in for libamdhip64.so.7, for the sanitizer runs of libmvmaxsim's host side.
 *
 * ThreadSanitizer / AddressSanitizer against the real ROCm runtime took a GPU node down (tools/README.md), and what those
 * runs are meant to check is OUR host code -- the append-only publish protocol, q_mu / w_mu, the per-shard streams of
 * mv_comm, buffer lifetimes -- not the runtime.  This stub gives the library EIGHT "devices" made of host memory: allocations
 * are calloc, copies are memcpy, streams and events complete immediately, kernel launches do NOTHING (scores and ids are the
 * zeros calloc left: meaningless, and irrelevant to the synchronisation being checked).
 *
 * Round 4: DEVICE AFFINITY is checked.  No 8-GPU node has run this code yet (the pool hands out 1-GPU boxes), so the multi-device
 * discipline of mv_comm -- the right current device around every launch, copy, event and stream of every shard -- is verified here:
 * every allocation, stream and event remembers the device it was created on, and
 *   - a kernel launch, async copy or memset on a stream of another device than the CURRENT one,
 *   - a copy whose device pointer lives on another device than its stream (hipMemcpyPeerAsync: than the device it names),
 *   - a copy that runs past the end of its allocation,
 *   - an event recorded on a stream of another device than the event's, elapsed time across devices, a stale handle
 * abort the process with a message naming the call.  Test infrastructure only: it is never loaded by the product path
 * (tools/sanitize/run.sh puts it in front of the real runtime with LD_LIBRARY_PATH). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int hipError_t;
typedef struct { uint32_t x, y, z; } dim3_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

#define N_DEVICES 8
#define STREAM_MAGIC 0x53545245u
#define EVENT_MAGIC 0x45564e54u

typedef struct stub_stream { uint32_t magic; int device; } *hipStream_t;
typedef struct stub_event { uint32_t magic; int device; int recorded; double t_ms; } *hipEvent_t;

static __thread int cur_dev = 0;
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

#define DIE(...)                                                      \
  do {                                                                \
    fprintf(stderr, "HIP STUB AFFINITY VIOLATION: " __VA_ARGS__);     \
    fprintf(stderr, " (current device %d)\n", cur_dev);               \
    { void* bt_[24]; int n_ = backtrace(bt_, 24); backtrace_symbols_fd(bt_, n_, 2); } /* who called: the library is built with -g */ \
    abort();                                                          \
  } while (0)

/* ---------------------------------------------------------------- allocation registry */
typedef struct { char* p; size_t n; int device; } alloc_t; /* device -1: pinned host memory, visible to every device */
static alloc_t* g_allocs;
static size_t g_n_allocs, g_cap_allocs;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static long g_checked_copies, g_checked_launches, g_checked_events;

static void reg_add(void* p, size_t n, int device) {
  pthread_mutex_lock(&g_mu);
  if (g_n_allocs == g_cap_allocs) {
    g_cap_allocs = g_cap_allocs ? 2 * g_cap_allocs : 256;
    g_allocs = realloc(g_allocs, g_cap_allocs * sizeof(alloc_t));
  }
  g_allocs[g_n_allocs++] = (alloc_t){(char*)p, n, device};
  pthread_mutex_unlock(&g_mu);
}
static int reg_remove(void* p) {
  int found = 0;
  pthread_mutex_lock(&g_mu);
  for (size_t i = 0; i < g_n_allocs; ++i)
    if (g_allocs[i].p == (char*)p) { g_allocs[i] = g_allocs[--g_n_allocs]; found = 1; break; }
  pthread_mutex_unlock(&g_mu);
  return found;
}
/* device of the allocation holding [p, p+n): -2 = not a HIP allocation (pageable host memory); aborts on an overrun */
static int reg_device(const void* p, size_t n, const char* what) {
  int dev = -2;
  pthread_mutex_lock(&g_mu);
  for (size_t i = 0; i < g_n_allocs; ++i) {
    const alloc_t a = g_allocs[i];
    if ((const char*)p >= a.p && (const char*)p < a.p + a.n) {
      if ((const char*)p + n > a.p + a.n) {
        pthread_mutex_unlock(&g_mu);
        DIE("%s: %zu bytes at offset %zu run past the end of a %zu-byte allocation of device %d", what, n, (size_t)((const char*)p - a.p), a.n, a.device);
      }
      dev = a.device;
      break;
    }
  }
  pthread_mutex_unlock(&g_mu);
  return dev;
}
/* helpers for the RCCL stand-in (tools/sanitize/rccl_stub.c) */
int hipstub_ptr_device(const void* p, size_t n) { return reg_device(p, n, "ncclAllGather buffer"); }
int hipstub_stream_device(hipStream_t s) {
  if (!s) return cur_dev;
  if (s->magic != STREAM_MAGIC) DIE("stale or foreign stream handle %p", (void*)s);
  return s->device;
}
int hipstub_current_device(void) { return cur_dev; }
void hipstub_counters(long* copies, long* launches, long* events) {
  *copies = __atomic_load_n(&g_checked_copies, __ATOMIC_RELAXED);
  *launches = __atomic_load_n(&g_checked_launches, __ATOMIC_RELAXED);
  *events = __atomic_load_n(&g_checked_events, __ATOMIC_RELAXED);
}

static int stream_dev(hipStream_t s, const char* what) {
  if (!s) return cur_dev; /* the null stream of the current device */
  if (s->magic != STREAM_MAGIC) DIE("%s: stale or foreign stream handle %p", what, (void*)s);
  if (s->device != cur_dev) DIE("%s on a stream of device %d", what, s->device);
  return s->device;
}

/* ---------------------------------------------------------------- devices */
hipError_t hipGetDeviceCount(int* n) { *n = N_DEVICES; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = cur_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= N_DEVICES) return hipErrorInvalidValue; cur_dev = d; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, int attr, int dev) { (void)attr; (void)dev; *v = 256; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (stub)" : "error (stub)"); }
hipError_t hipFuncSetAttribute(const void* f, int a, int v) { (void)f; (void)a; (void)v; return hipSuccess; }

/* ---------------------------------------------------------------- memory */
hipError_t hipMalloc(void** p, size_t n) {
  *p = calloc(n ? n : 1, 1);
  if (!*p) return hipErrorOutOfMemory;
  reg_add(*p, n ? n : 1, cur_dev);
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (p && !reg_remove(p)) DIE("hipFree(%p): not a live device allocation (double free?)", p);
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) {
  (void)flags;
  *p = calloc(n ? n : 1, 1);
  if (!*p) return hipErrorOutOfMemory;
  reg_add(*p, n ? n : 1, -1);
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  if (p && !reg_remove(p)) DIE("hipHostFree(%p): not a live pinned allocation", p);
  free(p);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  if (free_b) *free_b = (size_t)64 << 30;
  if (total_b) *total_b = (size_t)288 << 30;
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned flags) {
  (void)flags;
  if (reg_device(h, 1, "hipHostGetDevicePointer") != -1) DIE("hipHostGetDevicePointer(%p): not pinned host memory", h);
  *d = h;
  return hipSuccess;
}

/* a copy executed on `dev` (the stream's device, or the current one for the blocking form) */
static void check_copy(void* d, const void* s, size_t n, int dev, const char* what) {
  if (!n) return;
  const int dd = reg_device(d, n, what), sd = reg_device(s, n, what);
  if (dd >= 0 && dd != dev) DIE("%s: destination lives on device %d, the copy runs on device %d", what, dd, dev);
  if (sd >= 0 && sd != dev) DIE("%s: source lives on device %d, the copy runs on device %d", what, sd, dev);
  __atomic_fetch_add(&g_checked_copies, 1, __ATOMIC_RELAXED);
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) {
  (void)kind;
  check_copy(d, s, n, cur_dev, "hipMemcpy");
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) {
  (void)kind;
  check_copy(d, s, n, stream_dev(st, "hipMemcpyAsync"), "hipMemcpyAsync");
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void* d, int dd, const void* s, int sd, size_t n, hipStream_t st) {
  const int dev = stream_dev(st, "hipMemcpyPeerAsync");
  if (dd < 0 || dd >= N_DEVICES || sd < 0 || sd >= N_DEVICES) DIE("hipMemcpyPeerAsync: device %d / %d out of range", dd, sd);
  if (dev != dd && dev != sd) DIE("hipMemcpyPeerAsync between devices %d and %d on a stream of device %d", dd, sd, dev);
  if (n) {
    const int rd = reg_device(d, n, "hipMemcpyPeerAsync"), rs = reg_device(s, n, "hipMemcpyPeerAsync");
    if (rd != dd) DIE("hipMemcpyPeerAsync: destination named device %d but lives on %d", dd, rd);
    if (rs != sd) DIE("hipMemcpyPeerAsync: source named device %d but lives on %d", sd, rs);
    memmove(d, s, n);
    __atomic_fetch_add(&g_checked_copies, 1, __ATOMIC_RELAXED);
  }
  return hipSuccess;
}
static void check_set(void* d, size_t n, int dev, const char* what) {
  if (!n) return;
  const int dd = reg_device(d, n, what);
  if (dd >= 0 && dd != dev) DIE("%s: memory of device %d set from device %d", what, dd, dev);
}
hipError_t hipMemset(void* d, int v, size_t n) { check_set(d, n, cur_dev, "hipMemset"); if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  check_set(d, n, stream_dev(st, "hipMemsetAsync"), "hipMemsetAsync");
  if (n) memset(d, v, n);
  return hipSuccess;
}

/* ---------------------------------------------------------------- streams / events */
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) {
  (void)flags;
  *s = calloc(1, sizeof(**s));
  if (!*s) return hipErrorOutOfMemory;
  (*s)->magic = STREAM_MAGIC;
  (*s)->device = cur_dev;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  if (s) { if (s->magic != STREAM_MAGIC) DIE("hipStreamDestroy: stale stream %p", (void*)s); s->magic = 0; }
  free(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { if (s && s->magic != STREAM_MAGIC) DIE("hipStreamSynchronize: stale stream %p", (void*)s); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = calloc(1, sizeof(**e));
  if (!*e) return hipErrorOutOfMemory;
  (*e)->magic = EVENT_MAGIC;
  (*e)->device = cur_dev;
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags) { (void)flags; return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  if (e) { if (e->magic != EVENT_MAGIC) DIE("hipEventDestroy: stale event %p", (void*)e); e->magic = 0; }
  free(e);
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags) {
  (void)flags; /* a stream may wait for an event of ANY device: that is how the shards hand results to each other */
  if (s && s->magic != STREAM_MAGIC) DIE("hipStreamWaitEvent: stale stream %p", (void*)s);
  if (!e || e->magic != EVENT_MAGIC) DIE("hipStreamWaitEvent: stale event %p", (void*)e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (!e || e->magic != EVENT_MAGIC) DIE("hipEventRecord: stale event %p", (void*)e);
  if (s && s->magic != STREAM_MAGIC) DIE("hipEventRecord: stale stream %p", (void*)s);
  const int sdev = s ? s->device : cur_dev;
  if (sdev != e->device) DIE("hipEventRecord: event of device %d recorded on a stream of device %d", e->device, sdev);
  __atomic_store(&e->t_ms, &(double){now_ms()}, __ATOMIC_RELAXED);
  __atomic_store_n(&e->recorded, 1, __ATOMIC_RELAXED);
  __atomic_fetch_add(&g_checked_events, 1, __ATOMIC_RELAXED);
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { if (!e || e->magic != EVENT_MAGIC) DIE("hipEventSynchronize: stale event %p", (void*)e); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!a || !b || a->magic != EVENT_MAGIC || b->magic != EVENT_MAGIC) DIE("hipEventElapsedTime: stale event");
  if (a->device != b->device) DIE("hipEventElapsedTime between events of devices %d and %d", a->device, b->device);
  if (!__atomic_load_n(&a->recorded, __ATOMIC_RELAXED) || !__atomic_load_n(&b->recorded, __ATOMIC_RELAXED)) DIE("hipEventElapsedTime: event never recorded");
  double ta, tb;
  __atomic_load(&a->t_ms, &ta, __ATOMIC_RELAXED);
  __atomic_load(&b->t_ms, &tb, __ATOMIC_RELAXED);
  *ms = (float)(tb - ta);
  if (*ms <= 0.f) *ms = 1e-3f;
  return hipSuccess;
}

/* kernel launch plumbing emitted by hipcc for <<< >>>: configuration push / pop around hipLaunchKernel */
static __thread struct { dim3_t g, b; size_t shmem; hipStream_t s; } cfg;
hipError_t __hipPushCallConfiguration(dim3_t g, dim3_t b, size_t shmem, hipStream_t s) { cfg.g = g; cfg.b = b; cfg.shmem = shmem; cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3_t* g, dim3_t* b, size_t* shmem, hipStream_t* s) { *g = cfg.g; *b = cfg.b; *shmem = cfg.shmem; *s = cfg.s; return hipSuccess; }
hipError_t hipLaunchKernel(const void* f, dim3_t g, dim3_t b, void** args, size_t shmem, hipStream_t s) {
  (void)f; (void)args; (void)shmem;
  (void)stream_dev(s, "kernel launch");
  if (!g.x || !g.y || !g.z || !b.x || !b.y || !b.z) DIE("kernel launch with an empty grid / block (%u,%u,%u) x (%u,%u,%u)", g.x, g.y, g.z, b.x, b.y, b.z);
  __atomic_fetch_add(&g_checked_launches, 1, __ATOMIC_RELAXED);
  return hipSuccess;
}
void** __hipRegisterFatBinary(const void* data) { (void)data; static void* handle; return &handle; }
void __hipRegisterFunction(void** m, const void* hf, char* df, const char* dn, unsigned tl, void* tid, void* bid, void* bd, void* gd, int* ws) {
  (void)m; (void)hf; (void)df; (void)dn; (void)tl; (void)tid; (void)bid; (void)bd; (void)gd; (void)ws;
}
void __hipRegisterVar(void** m, void* v, char* a, const char* n, int e, size_t sz, int c, int g) { (void)m; (void)v; (void)a; (void)n; (void)e; (void)sz; (void)c; (void)g; }
void __hipUnregisterFatBinary(void** m) { (void)m; }
