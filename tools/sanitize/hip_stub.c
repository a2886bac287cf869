/* hip_stub.c -- a HOST-ONLY stand-in for libamdhip64.so.7, for the sanitizer runs of libmvmaxsim's host side.
 *
 * ThreadSanitizer / AddressSanitizer against the real ROCm runtime took a GPU node down (tools/README.md), and what those
 * runs are meant to check is OUR host code -- the append-only publish protocol, q_mu / w_mu, the per-shard streams of
 * mv_comm, buffer lifetimes -- not the runtime.  This stub gives the library a "device" made of host memory: allocations are
 * calloc, copies are memcpy, streams and events complete immediately, kernel launches do NOTHING (scores and ids are the
 * zeros calloc left: meaningless, and irrelevant to the synchronisation being checked).  Test infrastructure only: it is
 * never loaded by the product path (tools/sanitize/run.sh puts it in front of the real runtime with LD_LIBRARY_PATH). */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int hipError_t;
typedef struct { uint32_t x, y, z; } dim3_t;
typedef void* hipStream_t;
typedef struct stub_event { double t_ms; } *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };

static __thread int cur_dev = 0;
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

hipError_t hipGetDeviceCount(int* n) { *n = 2; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = cur_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d > 1) return hipErrorInvalidValue; cur_dev = d; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, int attr, int dev) { (void)attr; (void)dev; *v = 256; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (stub)" : "error (stub)"); }
hipError_t hipFuncSetAttribute(const void* f, int a, int v) { (void)f; (void)a; (void)v; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) { (void)flags; *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned flags) { (void)flags; *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) { (void)kind; if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) { (void)kind; (void)st; if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int dd, const void* s, int sd, size_t n, hipStream_t st) { (void)dd; (void)sd; (void)st; if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { (void)st; if (n) memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) { (void)flags; *s = malloc(8); return *s ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags) { (void)s; (void)e; (void)flags; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = calloc(1, sizeof(**e)); return *e ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags) { (void)flags; return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)s; __atomic_store(&e->t_ms, &(double){now_ms()}, __ATOMIC_RELAXED); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
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
hipError_t hipLaunchKernel(const void* f, dim3_t g, dim3_t b, void** args, size_t shmem, hipStream_t s) { (void)f; (void)g; (void)b; (void)args; (void)shmem; (void)s; return hipSuccess; }
void** __hipRegisterFatBinary(const void* data) { (void)data; static void* handle; return &handle; }
void __hipRegisterFunction(void** m, const void* hf, char* df, const char* dn, unsigned tl, void* tid, void* bid, void* bd, void* gd, int* ws) {
  (void)m; (void)hf; (void)df; (void)dn; (void)tl; (void)tid; (void)bid; (void)bd; (void)gd; (void)ws;
}
void __hipRegisterVar(void** m, void* v, char* a, const char* n, int e, size_t sz, int c, int g) { (void)m; (void)v; (void)a; (void)n; (void)e; (void)sz; (void)c; (void)g; }
void __hipUnregisterFatBinary(void** m) { (void)m; }
