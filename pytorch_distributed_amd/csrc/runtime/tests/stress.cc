// Host-side stress driver for the runtime library, meant to be built together with the runtime
// sources under -fsanitize=thread or -fsanitize=address,undefined (scripts/sanitize_runtime.sh).
// Ranks are threads of one process so the sanitizers see every access:
//   * TCP store: one server + N clients doing set/get/add/wait on shared and private keys,
//   * shm collectives: N communicators on one segment, all-reduce / broadcast / all-gather / barrier
//     with the results checked against closed forms,
//   * reducer core: bucket plan + ready marking + arrival-order rebuild,
//   * hvd engine: N engines, tensors enqueued in a different order per rank (name negotiation + fusion).
// Exit code 0 = every check passed.
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
void* pda_store_tcp_create(const char* host, int port, int is_server, long long timeout_ms);
int pda_store_tcp_port(void* h);
void pda_store_destroy(void* h);
int pda_store_set(void* h, const char* key, const void* val, int len);
long long pda_store_get(void* h, const char* key, void* out, long long cap, long long timeout_ms);
int pda_store_add(void* h, const char* key, long long delta, long long* out);
int pda_store_check(void* h, const char* key);

void* pda_shm_create(const char* name, int rank, int world, long long slot_bytes, long long timeout_ms);
void pda_shm_destroy(void* h);
long long pda_shm_all_reduce_async(void* h, void* buf, long long count, int dtype, int op);
long long pda_shm_broadcast_async(void* h, void* buf, long long bytes, int root);
long long pda_shm_all_gather_async(void* h, const void* in, void* out, long long bytes);
long long pda_shm_barrier_async(void* h);
int pda_shm_wait(void* h, long long ticket);

void* pda_reducer_create(int n, const long long* sizes_bytes, long long cap_bytes, long long first_cap_bytes, int align,
                         int pad);
void pda_reducer_destroy(void* h);
int pda_reducer_num_buckets(void* h);
int pda_reducer_bucket_of(void* h, int p);
void pda_reducer_reset(void* h);
int pda_reducer_mark_ready(void* h, int p, int* out, int max_out);
int pda_reducer_num_unready(void* h);
void pda_reducer_record_arrival(void* h, int on);
int pda_reducer_arrival(void* h, int* out, int max_out);
void pda_reducer_rebuild(void* h, const int* order, int n);

void* pda_hvd_create(void* shm, int rank, int world, long long fusion_bytes, double cycle_ms, void* exec);
void pda_hvd_shutdown(void* h);
void pda_hvd_destroy(void* h);
long long pda_hvd_enqueue(void* h, const char* name, int kind, void* ptr, long long count, int dtype, int op, int root,
                          int device, double prescale, double postscale);
int pda_hvd_wait(void* h, long long handle, char* err, int cap);
}

static std::atomic<int> g_fail{0};
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      g_fail.fetch_add(1);                                                  \
    }                                                                       \
  } while (0)

enum { F32 = 0, OP_SUM = 0 };

static void test_store(int world) {
  void* server = pda_store_tcp_create("127.0.0.1", 0, 1, 20000);
  CHECK(server != nullptr);
  if (!server) return;
  const int port = pda_store_tcp_port(server);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([=] {
      void* c = pda_store_tcp_create("127.0.0.1", port, 0, 20000);
      CHECK(c != nullptr);
      if (!c) return;
      for (int it = 0; it < 50; ++it) {
        const std::string key = "k" + std::to_string(r) + "_" + std::to_string(it);
        const std::string val = "v" + std::to_string(r * 1000 + it);
        CHECK(pda_store_set(c, key.c_str(), val.data(), (int)val.size()) == 0);
        long long tot = 0;
        CHECK(pda_store_add(c, "counter", 1, &tot) == 0);
        // read the key of the neighbour rank (blocks until that rank has written it)
        const std::string nk = "k" + std::to_string((r + 1) % world) + "_" + std::to_string(it);
        char buf[64];
        const long long n = pda_store_get(c, nk.c_str(), buf, sizeof(buf), 20000);
        const std::string want = "v" + std::to_string(((r + 1) % world) * 1000 + it);
        CHECK(n == (long long)want.size() && std::memcmp(buf, want.data(), want.size()) == 0);
      }
      pda_store_destroy(c);
    });
  for (auto& t : th) t.join();
  long long tot = 0;
  CHECK(pda_store_add(server, "counter", 0, &tot) == 0);
  CHECK(tot == 50LL * world);
  pda_store_destroy(server);
}

static void test_shm(int world) {
  const std::string name = "/pda_stress_" + std::to_string((long long)getpid());
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([=] {
      void* c = pda_shm_create(name.c_str(), r, world, 1 << 16, 30000);
      CHECK(c != nullptr);
      if (!c) return;
      for (int it = 0; it < 40; ++it) {
        const int n = 1 + (it * 977) % 40000;  // crosses the slot size: exercises chunking
        std::vector<float> v(n);
        for (int i = 0; i < n; ++i) v[i] = (float)(r + 1) * (float)((i + it) % 7);
        CHECK(pda_shm_wait(c, pda_shm_all_reduce_async(c, v.data(), n, F32, OP_SUM)) == 0);
        const float tri = (float)(world * (world + 1) / 2);
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && v[i] == tri * (float)((i + it) % 7);
        CHECK(ok);
        std::vector<int> b(257, r == it % world ? it : -1);
        CHECK(pda_shm_wait(c, pda_shm_broadcast_async(c, b.data(), (long long)b.size() * 4, it % world)) == 0);
        CHECK(b[0] == it && b[256] == it);
        std::vector<int> in(33, r * 100 + it), out(33 * world, -1);
        CHECK(pda_shm_wait(c, pda_shm_all_gather_async(c, in.data(), out.data(), 33 * 4)) == 0);
        for (int q = 0; q < world; ++q) CHECK(out[q * 33] == q * 100 + it && out[q * 33 + 32] == q * 100 + it);
        CHECK(pda_shm_wait(c, pda_shm_barrier_async(c)) == 0);
      }
      pda_shm_destroy(c);
    });
  for (auto& t : th) t.join();
}

static void test_reducer() {
  std::vector<long long> sizes;
  for (int i = 0; i < 60; ++i) sizes.push_back(4LL * (1000 + 37 * i) * (i % 5 == 0 ? 200 : 1));
  void* rd = pda_reducer_create((int)sizes.size(), sizes.data(), 1 << 20, 1 << 16, 256, 1);
  CHECK(rd != nullptr);
  const int nb = pda_reducer_num_buckets(rd);
  CHECK(nb > 1);
  pda_reducer_record_arrival(rd, 1);
  pda_reducer_reset(rd);
  int launched = 0;
  std::vector<int> out(nb + 1);
  for (int p = (int)sizes.size() - 1; p >= 0; --p) launched += pda_reducer_mark_ready(rd, p, out.data(), nb + 1);
  CHECK(launched == nb);
  CHECK(pda_reducer_num_unready(rd) == 0);
  std::vector<int> order(sizes.size());
  CHECK(pda_reducer_arrival(rd, order.data(), (int)order.size()) == (int)sizes.size());
  pda_reducer_rebuild(rd, order.data(), (int)order.size());
  CHECK(pda_reducer_num_buckets(rd) >= 1);
  CHECK(pda_reducer_bucket_of(rd, (int)sizes.size() - 1) == 0);  // first arrival sits in the first bucket
  pda_reducer_destroy(rd);
}

static void test_hvd(int world) {
  const std::string name = "/pda_stress_hvd_" + std::to_string((long long)getpid());
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([=] {
      void* c = pda_shm_create(name.c_str(), r, world, 1 << 20, 30000);
      CHECK(c != nullptr);
      if (!c) return;
      void* e = pda_hvd_create(c, r, world, 1 << 20, 1.0, nullptr);
      CHECK(e != nullptr);
      const int nt = 12;
      for (int it = 0; it < 10; ++it) {
        std::vector<std::vector<float>> ts(nt);
        std::vector<long long> hs(nt);
        for (int k = 0; k < nt; ++k) ts[k].assign(100 + 31 * k, (float)(r + 1) * (float)(k + 1));
        for (int q = 0; q < nt; ++q) {
          const int k = (q + r * 5 + it) % nt;  // rank-dependent enqueue order
          const std::string nm = "t" + std::to_string(it) + "." + std::to_string(k);
          // kind 0 = all-reduce, device 0 = host tensor (runs on the shm data plane)
          hs[k] = pda_hvd_enqueue(e, nm.c_str(), 0, ts[k].data(), (long long)ts[k].size(), F32, OP_SUM, 0, 0, 1.0,
                                  1.0 / world);
        }
        char err[256];
        for (int k = 0; k < nt; ++k) {
          CHECK(pda_hvd_wait(e, hs[k], err, sizeof(err)) == 0);
          const float want = (float)(k + 1) * (float)(world + 1) / 2.f;
          CHECK(std::fabs(ts[k][0] - want) < 1e-4f && std::fabs(ts[k].back() - want) < 1e-4f);
        }
      }
      pda_hvd_shutdown(e);
      pda_hvd_destroy(e);
      pda_shm_destroy(c);
    });
  for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? std::atoi(argv[1]) : 4;
  test_store(world);
  std::fprintf(stderr, "store ok (%d failures so far)\n", g_fail.load());
  test_shm(world);
  std::fprintf(stderr, "shm ok (%d failures so far)\n", g_fail.load());
  test_reducer();
  std::fprintf(stderr, "reducer ok (%d failures so far)\n", g_fail.load());
  test_hvd(world);
  std::fprintf(stderr, "hvd ok (%d failures so far)\n", g_fail.load());
  return g_fail.load() == 0 ? 0 : 1;
}
