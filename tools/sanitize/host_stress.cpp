// host_stress.cpp -- concurrency stress of libmvmaxsim's HOST side through its C ABI, built with -fsanitize=thread /
// address and run against the host-only HIP stub (tools/sanitize/hip_stub.c): reader threads over every query mode and
// entry point, a writer thread (add / remove / replace / compact / save), and a two-shard communicator under load.
// Kernel launches are no-ops in the stub, so answers are meaningless; what is checked is that every call succeeds and that
// the sanitizer sees no data race / lock-order inversion / heap error in the library's own code.
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

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int all = MV_WITH_FLOAT | MV_WITH_BINARY | MV_WITH_FDE | MV_WITH_FP8 | MV_WITH_HOST_EXACT;
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
    for (int it = 0; it < iters && !stop.load(); ++it) {
      const int mode = modes[(it + tid) % 6];
      mv_query_stats st{};
      CHECK(mv_query_topk(ix, q.data(), MV_F32, 20, 10, mode, (it & 1) ? allow.data() : nullptr, (it & 1) ? 64 : 0, s, id, &n, (it & 2) ? &st : nullptr));
      if (it % 5 == 0) CHECK(mv_query_topk_batch(ix, qb.data(), MV_F32, 5, 20, 7, (it % 10) ? MV_MODE_FLOAT : MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, &st));
      if (it % 7 == 0) { int64_t got = 0; CHECK(mv_score_all(ix, q.data(), MV_F32, 20, MV_MODE_FLOAT, nullptr, 0, all_scores.data(), 4096, &got, nullptr)); }
      if (it % 11 == 0) { int32_t cand[8] = {0, 3, 5, 9, 100, 101, 200, 7}; CHECK(mv_score_candidates(ix, q.data(), MV_F32, 20, cand, 8, -1, s, nullptr)); }
      (void)mv_index_size(ix);
      n_queries.fetch_add(1);
    }
  };
  auto writer = [&]() {
    std::vector<float> emb = rows(8 * 24, 7);
    int32_t nr[8]; int32_t ords[8];
    for (int i = 0; i < 8; ++i) nr[i] = 24;
    for (int it = 0; it < iters / 2; ++it) {
      for (int i = 0; i < 8; ++i) ords[i] = 1000 + it;
      int64_t first = -1;
      if (mv_index_size(ix) + 8 <= mv_index_capacity(ix)) CHECK(mv_index_add(ix, emb.data(), MV_F32, nr, 8, ords, &first));
      if (it % 3 == 1) { int64_t gone = 0; CHECK(mv_index_remove_doc(ix, 1000 + it - 1, &gone)); }
      if (it % 9 == 4) { std::vector<uint16_t> pg(24 * 128, 0x3c00); CHECK(mv_index_replace_page(ix, 5, pg.data(), 24)); }
      if (it % 25 == 12) { int64_t m = 0; CHECK(mv_index_compact(ix, nullptr, &m)); }
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
      float s[40]; int64_t id[40]; int32_t n = 0, nb[4];
      mv_query_stats st[2];
      for (int it = 0; it < iters / 2; ++it) {
        CHECK(mv_comm_query_topk(c, q.data(), MV_F32, 16, 10, (it & 1) ? MV_MODE_FLOAT : MV_MODE_FDE_THEN_FLOAT, nullptr, 0, s, id, &n, (it & 2) ? st : nullptr));
        if (it % 4 == 0) CHECK(mv_comm_query_topk_batch(c, qb.data(), MV_F32, 4, 16, 10, MV_MODE_FDE_THEN_FLOAT, nullptr, 0, 0, s, id, nb, st));
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
  mv_index_destroy(ix);
  printf("host_stress ok: %ld queries, %ld write rounds, %d iterations per thread\n", n_queries.load(), n_writes.load(), iters);
  return 0;
}
