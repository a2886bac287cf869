// host_stress.cpp -- concurrency stress of libmvmaxsim's HOST side through its C ABI, built with -fsanitize=thread /
// address and run against the host-only HIP stub (tools/sanitize/hip_stub.c): reader threads over every query mode and
// entry point, a writer thread (add / remove / replace / compact / save), a two-shard communicator under load, and (round 4)
// EIGHT shards on eight stub devices behind one communicator in all three transports -- RCCL (tools/sanitize/rccl_stub.c: grouped
// all-gathers with rank / device / size checks), peer copies, host -- through the single, batched, staged (FDE -> pruning -> exact
// rerank from the pinned-host tier; e4m3 scan -> exact re-score) pipelines, with a writer feeding one shard.
// Kernel launches are no-ops in the stub, so answers are meaningless; what is checked is that every call succeeds, that the
// stub sees every launch / copy / event on the device it belongs to (it aborts otherwise), and that the sanitizer sees no data
// race / lock-order inversion / heap error in the library's own code.
//   host_stress <iterations>     the stress;   host_stress violation   a deliberate cross-device memset: the stub must abort
#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mvmaxsim.h"

#define CHECK(expr)                                                                                    \
  do {                                                                                                 \
    int _rc = (expr);                                                                                  \
    if (_rc != 0) { fprintf(stderr, "FAIL %s -> %d: %s\n", #expr, _rc, mv_last_error()); std::abort(); } \
  } while (0)

static std::vector<float> rows(int n, unsigned seed) {
  std::vector<float> v((size_t)n * 128);
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
  return v;
}

static mv_index* make_index(int device, int64_t cap, int64_t id_base, int flags) {
  mv_config c{};
  c.dim = 128; c.stride_rows = 32; c.capacity_pages = cap; c.device = device; c.flags = flags; c.id_base = id_base;
  c.fde = mv_fde_config{128, 20, 5, 16, 1};
  mv_index* ix = nullptr;
  CHECK(mv_index_create(&c, &ix));
  return ix;
}

extern "C" {  // the HIP stub's own entry points (the stress drives them directly only for the self-test and the counters)
int hipSetDevice(int);
int hipMalloc(void**, size_t);
int hipMemset(void*, int, size_t);
void hipstub_counters(long* copies, long* launches, long* events);
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "violation")) {  // the affinity checker must catch a cross-device access (run.sh expects the abort)
    void* p = nullptr;
    hipSetDevice(2);
    hipMalloc(&p, 64);
    hipSetDevice(5);
    hipMemset(p, 0, 16);
    printf("NOT CAUGHT: a memset of device 2's memory from device 5 went through\n");
    return 0;
  }
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int all = MV_WITH_FLOAT | MV_WITH_BINARY | MV_WITH_FDE | MV_WITH_FP8 | MV_WITH_HOST_EXACT | MV_WITH_FDE_E4M3 | MV_WITH_FDE_FP4;  // (round 6: + the e4m3 and fp4 copies of the FDE slab)
  mv_index* ix = make_index(0, 4096, 0, all);
  CHECK(mv_index_fill_synthetic(ix, 1, 0, 512, 32, 4));
  std::atomic<bool> stop{false};
  std::atomic<long> n_queries{0}, n_writes{0};

  auto reader = [&](int tid) {
    const int modes[] = {MV_MODE_FLOAT, MV_MODE_BINARY, MV_MODE_FDE_THEN_FLOAT, MV_MODE_FDE_ONLY, MV_MODE_FLOAT_FP8, MV_MODE_FP8_THEN_FLOAT};
    std::vector<float> q = rows(20, 100 + tid), qb = rows(5 * 20, 200 + tid);
    std::vector<uint32_t> allow(64, 0x55555555u * (tid % 2 ? 1u : 3u));
    float s[64 * 5]; int64_t id[64 * 5]; int32_t n = 0, nb[5];
    std::vector<float> all_scores(4096);
    std::vector<float> qfde = rows((5 * 10240 + 127) / 128, 900 + tid);  // five query FDE vectors of 10 240 floats
    // round 5: the enqueue-only device query with deferred timings (the N > 1 step of sharded.py): ids and scores into ONE block, the block
    // merge, the timings collected one query later.  Six threads share the index, so a record may be older than "one behind" when it is
    // finished -- its timings are then another query's (documented), which a host-side check does not care about.
    void* blk = nullptr;
    float* d_os = nullptr; int64_t* d_oi = nullptr;
    hipSetDevice(0);
    if (hipMalloc(&blk, (size_t)mv_topk_block_bytes(10)) || hipMalloc((void**)&d_os, 40) || hipMalloc((void**)&d_oi, 80)) std::abort();
    mv_query_stats pend{};
    bool have_pend = false;
    for (int it = 0; it < iters && !stop.load(); ++it) {
      const int mode = modes[(it + tid) % 6];
      mv_query_stats st{};
      CHECK(mv_query_topk(ix, q.data(), MV_F32, 20, 10, mode, (it & 1) ? allow.data() : nullptr, (it & 1) ? 64 : 0, s, id, &n, (it & 2) ? &st : nullptr));
      if (it % 5 == 0) CHECK(mv_query_topk_batch(ix, qb.data(), MV_F32, 5, 20, 7, (it % 10) ? MV_MODE_FLOAT : MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, &st));
      if (it % 7 == 0) { int64_t got = 0; CHECK(mv_score_all(ix, q.data(), MV_F32, 20, MV_MODE_FLOAT, nullptr, 0, all_scores.data(), 4096, &got, nullptr)); }
      if (it % 11 == 0) { int32_t cand[8] = {0, 3, 5, 9, 100, 101, 200, 7}; CHECK(mv_score_candidates(ix, q.data(), MV_F32, 20, cand, 8, -1, s, nullptr)); }
      if (it % 6 == 3) {  // the caller's own query FDE vectors (thread-local override inside the library): single and batched
        CHECK(mv_query_topk_fde(ix, q.data(), MV_F32, 20, qfde.data(), 10, (it & 1) ? MV_MODE_FDE_THEN_FLOAT : MV_MODE_FDE_ONLY, nullptr, 0, s, id, &n, nullptr));
        CHECK(mv_query_topk_batch_fde(ix, qb.data(), MV_F32, 5, 20, qfde.data(), 7, MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, nullptr));
      }
      if (it % 4 == 1) {
        mv_query_stats now{};
        CHECK(mv_query_topk_device_async(ix, q.data(), MV_F32, 20, 10, (it & 8) ? MV_MODE_FLOAT : MV_MODE_FLOAT_FP8, nullptr, 0, (float*)((char*)blk + 80), (int64_t*)blk,
                                         nullptr /* the null stream: ordered against, not blocked on */, &now));
        CHECK(mv_merge_topk_blocks(0, blk, 1, 10, 10, d_os, d_oi, nullptr));
        if (have_pend) CHECK(mv_query_stats_finish(ix, &pend));
        pend = now;
        have_pend = true;
      }
      (void)mv_index_size(ix);
      n_queries.fetch_add(1);
    }
    if (have_pend) CHECK(mv_query_stats_finish(ix, &pend));
  };
  auto writer = [&]() {
    std::vector<float> emb = rows(8 * 24, 7);
    std::vector<float> dfde = rows((3 * 10240 + 127) / 128, 8);
    int32_t nr[8]; int32_t ords[8];
    for (int i = 0; i < 8; ++i) nr[i] = 24;
    for (int it = 0; it < iters / 2; ++it) {
      for (int i = 0; i < 8; ++i) ords[i] = 1000 + it;
      int64_t first = -1;
      if (mv_index_size(ix) + 8 <= mv_index_capacity(ix)) CHECK(mv_index_add(ix, emb.data(), MV_F32, nr, 8, ords, &first));
      if (it % 3 == 1) { int64_t gone = 0; CHECK(mv_index_remove_doc(ix, 1000 + it - 1, &gone)); }
      if (it % 9 == 4) { std::vector<uint16_t> pg(24 * 128, 0x3c00); CHECK(mv_index_replace_page(ix, 5, pg.data(), 24)); }
      if (it % 25 == 12) { int64_t m = 0; CHECK(mv_index_compact(ix, nullptr, &m)); }
      if (it % 6 == 2) CHECK(mv_index_import_fde(ix, 2, 3, dfde.data()));  // caller-supplied document FDE vectors under the readers
      if (it % 40 == 20) CHECK(mv_index_save(ix, "/tmp/mv_host_stress.idx"));
      n_writes.fetch_add(1);
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < 6; ++t) th.emplace_back(reader, t);
  th.emplace_back(writer);
  for (auto& t : th) t.join();
  th.clear();
  mv_index* back = nullptr;
  CHECK(mv_index_load("/tmp/mv_host_stress.idx", 0, &back));
  mv_index_destroy(back);

  // ---- two shards on two "devices" behind one communicator, queried from two threads while a writer feeds shard 1
  const int sflags = MV_WITH_FLOAT | MV_WITH_FDE;
  mv_index* sh[2] = {make_index(0, 1024, 0, sflags), make_index(1, 1024, 1024, sflags)};
  CHECK(mv_index_fill_synthetic(sh[0], 1, 0, 256, 32, 4));
  CHECK(mv_index_fill_synthetic(sh[1], 1, 1024, 256, 32, 4));
  for (int transport : {MV_COMM_P2P, MV_COMM_HOST}) {
    const int32_t devs[2] = {0, 1};
    mv_comm* c = nullptr;
    CHECK(mv_comm_create(2, devs, transport, &c));
    CHECK(mv_comm_attach(c, 0, sh[0]));
    CHECK(mv_comm_attach(c, 1, sh[1]));
    auto cq = [&](int tid) {
      std::vector<float> q = rows(16, 300 + tid), qb = rows(4 * 16, 400 + tid);
      std::vector<float> cqfde = rows((4 * 10240 + 127) / 128, 950 + tid);
      float s[40]; int64_t id[40]; int32_t n = 0, nb[4];
      mv_query_stats st[2];
      for (int it = 0; it < iters / 2; ++it) {
        CHECK(mv_comm_query_topk(c, q.data(), MV_F32, 16, 10, (it & 1) ? MV_MODE_FLOAT : MV_MODE_FDE_THEN_FLOAT, nullptr, 0, s, id, &n, (it & 2) ? st : nullptr));
        if (it % 4 == 0) CHECK(mv_comm_query_topk_batch(c, qb.data(), MV_F32, 4, 16, 10, MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, st));
        if (it % 5 == 1) {
          CHECK(mv_comm_query_topk_fde(c, q.data(), MV_F32, 16, cqfde.data(), 10, MV_MODE_FDE_THEN_FLOAT, nullptr, 0, s, id, &n, nullptr));
          CHECK(mv_comm_query_topk_batch_fde(c, qb.data(), MV_F32, 4, 16, cqfde.data(), 10, MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, nullptr));
        }
      }
    };
    auto cw = [&]() {
      std::vector<float> emb = rows(4 * 20, 9);
      int32_t nr[4] = {20, 20, 20, 20}, ords[4] = {7, 7, 8, 8};
      for (int it = 0; it < iters / 4; ++it)
        if (mv_index_size(sh[1]) + 4 <= mv_index_capacity(sh[1])) CHECK(mv_index_add(sh[1], emb.data(), MV_F32, nr, 4, ords, nullptr));
    };
    th.emplace_back(cq, 0);
    th.emplace_back(cq, 1);
    th.emplace_back(cw);
    for (auto& t : th) t.join();
    th.clear();
    mv_comm_destroy(c);
  }
  mv_index_destroy(sh[0]);
  mv_index_destroy(sh[1]);

  // ---- EIGHT shards on eight "devices": the node shape of BASELINE configs[2]-[4], every transport, every staged pipeline
  std::atomic<long> n_comm8{0};
  for (int shape = 0; shape < 3; ++shape) {
    // shape 0: configs[3] / [4] shard shape -- FDE + e4m3 slabs, exact rows in the pinned-host tier (pruning stage: coarse 300 > 64)
    // shape 1: bf16 + sign-bit + FDE slabs (single-stage float / binary scans, FDE pipeline reranking in "HBM")
    // shape 2: shape 0 with the exact tier SPLIT: pages [0, 70) of every shard in "HBM", the other filled pages pinned
    const int eflags = shape == 1 ? (MV_WITH_FLOAT | MV_WITH_BINARY | MV_WITH_FDE)
                                  : (MV_WITH_FDE | MV_WITH_FP8 | MV_WITH_HOST_EXACT | (shape == 2 ? MV_WITH_EXACT_SPLIT : 0));
    if (shape == 2) setenv("MV_EXACT_HBM_MAX_PAGES", "70", 1);
    mv_index* s8[8];
    int32_t devs8[8];
    for (int r = 0; r < 8; ++r) {
      devs8[r] = r;
      s8[r] = make_index(r, 512, (int64_t)r * 512, eflags);
      CHECK(mv_index_fill_synthetic(s8[r], 1, (uint64_t)r * 512, 128, 32, 4));
      CHECK(mv_index_set_option(s8[r], MV_OPT_FDE_COARSE_N, 300));
      CHECK(mv_index_set_option(s8[r], MV_OPT_RERANK_N, 64));
    }
    if (shape == 2) {  // the writers of a split tier: pages on both sides of the split, compaction across it, a checkpoint re-split on load
      if (mv_index_exact_hbm_pages(s8[7]) != 70) { fprintf(stderr, "FAIL split: %lld HBM pages, expected 70\n", (long long)mv_index_exact_hbm_pages(s8[7])); std::abort(); }
      std::vector<uint16_t> pg(32 * 128, 0x3c00), back(40 * 32 * 128);
      for (int64_t page : {3, 69, 70, 127}) {
        CHECK(mv_index_replace_page(s8[7], page, pg.data(), 20));
        CHECK(mv_index_write_rows(s8[7], page, 4, 8, pg.data()));
      }
      CHECK(mv_index_read_pages(s8[7], 50, 40, back.data()));
      for (int64_t page : {5, 40, 68, 71, 90}) CHECK(mv_index_remove_page(s8[7], page));
      int64_t new_size = 0;
      CHECK(mv_index_compact(s8[7], nullptr, &new_size));
      if (new_size != 123) { fprintf(stderr, "FAIL compact of the split tier: %lld pages left\n", (long long)new_size); std::abort(); }
      CHECK(mv_index_save(s8[7], "/tmp/mv_split_stress.idx"));
      setenv("MV_EXACT_HBM_MAX_PAGES", "33", 1);
      mv_index* re = nullptr;
      CHECK(mv_index_load("/tmp/mv_split_stress.idx", 3, &re));
      if (mv_index_exact_hbm_pages(re) != 33 || mv_index_size(re) != 123) { fprintf(stderr, "FAIL reload of the split tier\n"); std::abort(); }
      CHECK(mv_index_read_pages(re, 20, 40, back.data()));
      mv_index_destroy(re);
      remove("/tmp/mv_split_stress.idx");
      unsetenv("MV_EXACT_HBM_MAX_PAGES");
      // round 6: hot-page placement (the stub's kernels count nothing, so nothing moves: the tables, the counters' read-back and the locks are exercised)
      int64_t moved = -1, hb = -1, hh = -1;
      CHECK(mv_index_exact_tier_rebalance(s8[7], 0, &moved));
      CHECK(mv_index_exact_tier_hits(s8[7], &hb, &hh));
    }
    for (int transport : {MV_COMM_RCCL, MV_COMM_P2P, MV_COMM_HOST}) {
      mv_comm* c = nullptr;
      CHECK(mv_comm_create(8, devs8, transport, &c));
      if (mv_comm_transport(c) != transport) { fprintf(stderr, "FAIL transport %d came up as %d: %s\n", transport, mv_comm_transport(c), mv_last_error()); std::abort(); }
      for (int r = 0; r < 8; ++r) CHECK(mv_comm_attach(c, r, s8[r]));
      auto cq8 = [&](int tid) {
        const int modes0[] = {MV_MODE_FDE_THEN_FLOAT, MV_MODE_FP8_THEN_FLOAT, MV_MODE_FLOAT_FP8, MV_MODE_FDE_ONLY};
        const int modes1[] = {MV_MODE_FLOAT, MV_MODE_BINARY, MV_MODE_FDE_THEN_FLOAT, MV_MODE_FDE_ONLY};
        const int* modes = shape != 1 ? modes0 : modes1;
        std::vector<float> q = rows(16, 500 + tid), qb = rows(36 * 16, 600 + tid);
        std::vector<uint32_t> allow(8, 0xdeadbeefu), per(36 * 8, 0x77777777u);
        std::vector<float> s(36 * 10);
        std::vector<int64_t> id(36 * 10);
        int32_t n = 0, nb[36];
        mv_query_stats st[8];
        for (int it = 0; it < iters / 4; ++it) {
          const int mode = modes[(it + tid) % 4];
          CHECK(mv_comm_query_topk(c, q.data(), MV_F32, 16, 10, mode, (it & 1) ? allow.data() : nullptr, (it & 1) ? 8 : 0, s.data(), id.data(), &n, (it & 2) ? st : nullptr));
          if (it % 3 == 0) {  // batches: 5 requests, and 36 (more than one group of 32); shared and per-request filters
            const int nq = (it % 6 == 0) ? 36 : 5;
            const int bmode = shape != 1 ? ((it % 2) ? MV_MODE_FP8_THEN_FLOAT : MV_MODE_FDE_THEN_FLOAT) : ((it % 2) ? MV_MODE_FLOAT : MV_MODE_FDE_THEN_FLOAT);
            CHECK(mv_comm_query_topk_batch(c, qb.data(), MV_F32, nq, 16, 10, bmode, (it % 4 == 0) ? per.data() : nullptr, (it % 4 == 0) ? 8 : 0, (it % 4 == 0) ? 1 : 0,
                                           s.data(), id.data(), nb, st));
          }
          n_comm8.fetch_add(1);
        }
      };
      auto cw8 = [&]() {
        std::vector<float> emb = rows(4 * 20, 11);
        int32_t nr[4] = {20, 20, 20, 20}, ords[4] = {70, 70, 71, 71};
        for (int it = 0; it < iters / 8; ++it) {
          if (mv_index_size(s8[5]) + 4 <= mv_index_capacity(s8[5])) CHECK(mv_index_add(s8[5], emb.data(), MV_F32, nr, 4, ords, nullptr));
          if (it % 5 == 3) { int64_t gone = 0; CHECK(mv_index_remove_doc(s8[2], it % 32, &gone)); }
        }
      };
      th.emplace_back(cq8, 0);
      th.emplace_back(cq8, 1);
      th.emplace_back(cw8);
      for (auto& t : th) t.join();
      th.clear();
      mv_comm_destroy(c);
    }
    for (int r = 0; r < 8; ++r) mv_index_destroy(s8[r]);
  }
  // ---- round 6: the PACKED layout (row-offset table) with split-bf16 pages (hi + lo slabs) under readers and a writer of ragged pages,
  // compaction and checkpoints of it; the FDE slab's placement trial (exclusive) against running readers
  std::atomic<long> n_packed{0};
  {
    mv_config c{};
    c.dim = 128; c.stride_rows = 32; c.capacity_pages = 1024; c.capacity_rows = 1024 * 24; c.device = 1; c.id_base = 0;
    c.flags = MV_WITH_FLOAT | MV_WITH_FLOAT_LO | MV_WITH_BINARY | MV_WITH_FDE | MV_WITH_FP8 | MV_LAYOUT_PACKED | MV_WITH_FDE_E4M3;
    c.fde = mv_fde_config{128, 20, 5, 16, 1};
    mv_index* px = nullptr;
    CHECK(mv_index_create(&c, &px));
    CHECK(mv_index_fill_synthetic_ragged(px, 1, 0, 256, 5, 32, 4));
    if (mv_index_rows_used(px) <= 0 || mv_index_rows_used(px) > mv_index_capacity_rows(px)) { fprintf(stderr, "FAIL packed rows_used\n"); std::abort(); }
    auto preader = [&](int tid) {
      const int modes[] = {MV_MODE_FLOAT, MV_MODE_BINARY, MV_MODE_FDE_THEN_FLOAT, MV_MODE_FLOAT_FP8, MV_MODE_FP8_THEN_FLOAT, MV_MODE_FDE_ONLY};
      std::vector<float> q = rows(20, 300 + tid), qb = rows(5 * 20, 400 + tid);
      float s[64 * 5]; int64_t id[64 * 5]; int32_t n = 0, nb[5];
      for (int it = 0; it < iters; ++it) {
        CHECK(mv_query_topk(px, q.data(), MV_F32, 20, 10, modes[(it + tid) % 6], nullptr, 0, s, id, &n, nullptr));
        if (it % 5 == 0) CHECK(mv_query_topk_batch(px, qb.data(), MV_F32, 5, 20, 7, (it % 10) ? MV_MODE_FLOAT : MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, nullptr));
        if (it % 13 == 6 && tid == 0) { double b = 0, a = 0; int32_t m = 0; CHECK(mv_index_fde_placement_trial(px, 1, &b, &a, &m)); }
        if (it % 17 == 3) CHECK(mv_index_set_option(px, MV_OPT_FLOAT_LO_SCAN, it % 3));  // options change under running queries of the other threads
        if (it % 19 == 4) CHECK(mv_index_set_option(px, MV_OPT_FDE_COARSE_N, (it & 1) ? 40 : 0));
        if (it % 23 == 5) CHECK(mv_index_set_option(px, MV_OPT_RERANK_N, 16 + it % 32));
        if (it % 29 == 6) CHECK(mv_index_set_option(px, MV_OPT_BATCH_VARIANT, (it & 1) ? 8 : -1));
        if (it % 31 == 7) CHECK(mv_index_set_option(px, MV_OPT_FDE_COARSE_SLAB, it & 1));
        n_packed.fetch_add(1);
      }
    };
    auto pwriter = [&]() {
      std::vector<float> emb = rows(8 * 32, 17), back((size_t)4 * 32 * 128);
      std::vector<uint16_t> pg(32 * 128, 0x3c00);
      int32_t nr[8], ords[8];
      for (int it = 0; it < iters / 2; ++it) {
        int32_t total = 0;
        for (int i = 0; i < 8; ++i) { nr[i] = 3 + (it * 7 + i * 5) % 30; ords[i] = 2000 + it; total += (nr[i] + 15) / 16 * 16; }
        // emb holds 8 x 32 rows; a ragged batch is packed back to back as mv_index_add expects
        if (mv_index_size(px) + 8 <= mv_index_capacity(px) && mv_index_rows_used(px) + total <= mv_index_capacity_rows(px))
          CHECK(mv_index_add(px, emb.data(), MV_F32, nr, 8, ords, nullptr));
        if (it % 3 == 1) { int64_t gone = 0; CHECK(mv_index_remove_doc(px, 2000 + it - 1, &gone)); }
        if (it % 9 == 4) CHECK(mv_index_replace_page(px, 7, pg.data(), 5 + it % 20 > 32 ? 32 : 5 + it % 20));  // a shorter / longer page in place or at the slab's end
        if (it % 11 == 5) CHECK(mv_index_write_rows(px, 9, 1, 2, pg.data()));
        if (it % 25 == 12) { int64_t m = 0; CHECK(mv_index_compact(px, nullptr, &m)); }
        if (it % 7 == 2) CHECK(mv_index_read_pages_f32(px, 3, 4, back.data()));
        if (it % 40 == 20) CHECK(mv_index_save(px, "/tmp/mv_packed_stress.idx"));
      }
    };
    for (int t = 0; t < 3; ++t) th.emplace_back(preader, t);
    th.emplace_back(pwriter);
    for (auto& t : th) t.join();
    th.clear();
    CHECK(mv_index_save(px, "/tmp/mv_packed_stress.idx"));
    mv_index* pb = nullptr;
    CHECK(mv_index_load("/tmp/mv_packed_stress.idx", 2, &pb));
    if (mv_index_size(pb) != mv_index_size(px) || mv_index_rows_used(pb) != mv_index_rows_used(px)) { fprintf(stderr, "FAIL reload of the packed index\n"); std::abort(); }
    mv_index_destroy(pb);
    remove("/tmp/mv_packed_stress.idx");
    mv_index_destroy(px);
  }
  mv_index_destroy(ix);
  long copies = 0, launches = 0, events = 0, gathers = -1;
  hipstub_counters(&copies, &launches, &events);
  if (void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD))
    if (auto f = (long (*)())dlsym(h, "rcclstub_gathers")) gathers = f();
  printf("host_stress ok: %ld queries, %ld write rounds, %ld queries through the 8-shard communicators, %d iterations per thread; "
         "%ld queries on the packed split-bf16 index; stub checked %ld copies, %ld kernel launches, %ld event records for device affinity, %ld grouped RCCL all-gathers\n",
         n_queries.load(), n_writes.load(), n_comm8.load(), iters, n_packed.load(), copies, launches, events, gathers);
  return 0;
}
