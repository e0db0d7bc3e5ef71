// Horovod-style collective engine: what `hvd.init()` starts and what
// `hvd.allreduce_async_ / broadcast_async_ / synchronize / poll` talk to in the reference
// (/root/reference/5.horovod_distributed.py:92,116-125, /root/reference/5.2.horovod_pytorch_mnist.py:77,174-185).
//
// * Every rank owns a background thread.  User threads only enqueue named requests
//   and get integer handles back.
// * Each cycle the threads exchange their lists of pending request names
//   (variable-length all-gather over the shared-memory control plane).  A tensor is
//   READY once every rank has announced the same name with the same
//   kind/dtype/op/count/root; ranks may enqueue in different orders - the execution
//   order is rank 0's announcement order, identical everywhere because every rank
//   evaluates the same gathered lists.  Mismatched metadata fails the handle with a
//   message instead of hanging.
// * Consecutive ready all-reduces with equal (dtype, op, device, scales) are FUSED
//   up to the fusion threshold (64 MiB default): CPU tensors are packed into the
//   fusion buffer, reduced with one shared-memory all-reduce and unpacked with the
//   post-scale applied (average = sum * 1/size, exactly once).  Device tensors are
//   handed, group by group, to an executor callback (python: one pack kernel that
//   also casts to the wire dtype + one RCCL all-reduce on a side stream + one unpack
//   kernel that casts back and scales).
// * All-gathers (equal counts on every rank) are negotiated the same way and run one by one.
// * A shutdown request travels with the same exchange, so all threads leave together.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define PDA_EXPORT extern "C" __attribute__((visibility("default")))

extern "C" {
long long pda_shm_all_reduce_async(void* h, void* buf, long long count, int dtype, int op);
long long pda_shm_broadcast_async(void* h, void* buf, long long bytes, int root);
long long pda_shm_all_gather_async(void* h, const void* in, void* out, long long bytes);
int pda_shm_wait(void* h, long long ticket);
int pda_store_set(void* h, const char* key, const void* val, int len);
long long pda_store_get(void* h, const char* key, void* out, long long cap, long long timeout_ms);
int pda_store_delete(void* h, const char* key);
}

namespace {

enum Kind : int { K_ALLREDUCE = 0, K_BROADCAST = 1, K_ALLGATHER = 2, K_ADASUM = 3 };
enum HOp : int { H_SUM = 0, H_PROD = 1, H_MIN = 2, H_MAX = 3, H_AVG = 4 };

size_t dsize(int dt) {
  switch (dt) {
    case 0: case 5: return 4;
    case 1: case 2: return 2;
    case 3: return 1;
    default: return 8;
  }
}

// What the engine exchanges through: the negotiation lists of every cycle and the host tensors.
struct Transport {
  virtual ~Transport() {}
  virtual bool all_gather(const void* in, void* out, long long bytes) = 0;   // out: world * bytes, in rank order
  virtual bool all_reduce(void* buf, long long count, int dtype, int op) = 0;
  virtual bool broadcast(void* buf, long long bytes, int root) = 0;
  virtual const char* what() const = 0;
  // every rank's byte string, in rank order (the negotiation lists): lengths first, then the padded strings
  virtual bool gather_strings(const std::string& mine, std::vector<std::string>* all, int world) {
    long long n = (long long)mine.size();
    std::vector<long long> lens((size_t)world);
    if (!all_gather(&n, lens.data(), 8)) return false;
    long long m = 1;
    for (long long l : lens) m = std::max(m, l);
    std::string pad = mine;
    pad.resize((size_t)m, '\0');
    std::string out((size_t)m * (size_t)world, '\0');
    if (!all_gather(pad.data(), &out[0], m)) return false;
    all->clear();
    for (int r = 0; r < world; ++r) all->push_back(out.substr((size_t)r * (size_t)m, (size_t)lens[(size_t)r]));
    return true;
  }
};

// One node: the shared-memory communicator.
struct ShmTransport : Transport {
  explicit ShmTransport(void* shm) : shm_(shm) {}
  bool all_gather(const void* in, void* out, long long bytes) override {
    return pda_shm_wait(shm_, pda_shm_all_gather_async(shm_, in, out, bytes)) == 0;
  }
  bool all_reduce(void* buf, long long count, int dtype, int op) override {
    return pda_shm_wait(shm_, pda_shm_all_reduce_async(shm_, buf, count, dtype, op)) == 0;
  }
  bool broadcast(void* buf, long long bytes, int root) override {
    return pda_shm_wait(shm_, pda_shm_broadcast_async(shm_, buf, bytes, root)) == 0;
  }
  const char* what() const override { return "shared-memory"; }
  void* shm_;
};

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) u = sign;
    else {
      int e = -1;
      uint32_t m = man;
      do { ++e; m <<= 1; } while ((m & 0x400) == 0);
      u = sign | (uint32_t)(127 - 15 - e) << 23 | (m & 0x3ff) << 13;
    }
  } else if (exp == 31) u = sign | 0x7f800000u | man << 13;
  else u = sign | (exp + 112) << 23 | man << 13;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
uint16_t float_to_half(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000;
  const int32_t exp = (int32_t)((u >> 23) & 0xff) - 127 + 15;
  uint32_t man = u & 0x7fffff;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00 | (man ? 0x200 : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000;
    const int shift = 14 - exp;
    uint32_t r = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) ++r;
    return (uint16_t)(sign | r);
  }
  uint32_t r = (uint32_t)exp << 10 | man >> 13;
  const uint32_t rem = man & 0x1fff;
  if (rem > 0x1000 || (rem == 0x1000 && (r & 1))) ++r;
  return (uint16_t)(sign | r);
}
float bf16_to_float(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
uint16_t float_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffff) > 0x7f800000) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

template <typename A> A combine(A a, A b, int op) {
  switch (op) {
    case 1: return a * b;
    case 2: return b < a ? b : a;
    case 3: return b > a ? b : a;
    default: return a + b;   // sum, average
  }
}
template <typename T> void reduce_typed(T* dst, const T* all, long long count, int world, int op) {
  for (long long i = 0; i < count; ++i) {
    T acc = all[i];
    for (int r = 1; r < world; ++r) acc = combine<T>(acc, all[(size_t)r * (size_t)count + (size_t)i], op);
    if (op == 4) acc = (T)(acc / (T)world);
    dst[i] = acc;
  }
}
template <typename L, typename S> void reduce_16(uint16_t* dst, const uint16_t* all, long long count, int world, int op,
                                                 L load, S store) {
  for (long long i = 0; i < count; ++i) {
    float acc = load(all[i]);
    for (int r = 1; r < world; ++r) acc = combine<float>(acc, load(all[(size_t)r * (size_t)count + (size_t)i]), op);
    if (op == 4) acc /= (float)world;
    dst[i] = store(acc);
  }
}

// Several nodes: everything goes through the key-value store of the rendezvous (a TCP client of this engine's own,
// so that the engine thread never shares a socket with the rank's main thread).  An exchange is one `set` of the
// rank's own value and one blocking `get` per peer; a rank deletes its keys of exchange n once exchange n + 1 is
// complete, when every peer has read them.  This carries the negotiation lists (tens of bytes per cycle) and the
// occasional host tensor; device tensors travel over RCCL from the executor.
struct StoreTransport : Transport {
  StoreTransport(void* store, const char* prefix, int rank, int world, long long timeout_ms)
      : store_(store), prefix_(prefix), rank_(rank), world_(world), timeout_ms_(timeout_ms) {}
  static constexpr long long kPiece = 8 << 20;

  std::string key(long long seq, int rank, long long piece) const {
    return prefix_ + "/" + std::to_string(seq) + "/" + std::to_string(rank) + "/" + std::to_string(piece);
  }
  bool all_gather(const void* in, void* out, long long bytes) override {
    const long long seq = ++seq_;
    std::vector<std::string> mine;
    for (long long off = 0, piece = 0; off < bytes || piece == 0; off += kPiece, ++piece) {
      const long long n = std::max<long long>(0, std::min(kPiece, bytes - off));
      const std::string k = key(seq, rank_, piece);
      if (pda_store_set(store_, k.c_str(), static_cast<const char*>(in) + off, (int)n) != 0) return false;
      mine.push_back(k);
      for (int r = 0; r < world_; ++r) {
        char* dst = static_cast<char*>(out) + (size_t)r * (size_t)bytes + (size_t)off;
        if (r == rank_) {
          memcpy(dst, static_cast<const char*>(in) + off, (size_t)n);
          continue;
        }
        if (pda_store_get(store_, key(seq, r, piece).c_str(), dst, n, timeout_ms_) != n) return false;
      }
    }
    for (auto& k : prev_) pda_store_delete(store_, k.c_str());
    prev_.swap(mine);
    return true;
  }
  bool all_reduce(void* buf, long long count, int dtype, int op) override {
    const size_t es = dtype == 0 || dtype == 5 ? 4 : (dtype == 1 || dtype == 2 ? 2 : (dtype == 3 ? 1 : 8));
    // in slices, so that the gathered copy stays small
    const long long slice = std::max<long long>(1, (16 << 20) / (long long)es / world_);
    for (long long lo = 0; lo < count; lo += slice) {
      const long long n = std::min(slice, count - lo);
      char* p = static_cast<char*>(buf) + (size_t)lo * es;
      scratch_.resize((size_t)n * es * (size_t)world_);
      if (!all_gather(p, scratch_.data(), n * (long long)es)) return false;
      switch (dtype) {
        case 0: reduce_typed((float*)p, (const float*)scratch_.data(), n, world_, op); break;
        case 4: reduce_typed((double*)p, (const double*)scratch_.data(), n, world_, op); break;
        case 5: reduce_typed((int32_t*)p, (const int32_t*)scratch_.data(), n, world_, op); break;
        case 6: reduce_typed((int64_t*)p, (const int64_t*)scratch_.data(), n, world_, op); break;
        case 3: reduce_typed((uint8_t*)p, (const uint8_t*)scratch_.data(), n, world_, op); break;
        case 1: reduce_16((uint16_t*)p, (const uint16_t*)scratch_.data(), n, world_, op, half_to_float, float_to_half); break;
        case 2: reduce_16((uint16_t*)p, (const uint16_t*)scratch_.data(), n, world_, op, bf16_to_float, float_to_bf16); break;
        default: return false;
      }
    }
    return true;
  }
  bool broadcast(void* buf, long long bytes, int root) override {
    const long long seq = ++seq_;
    std::vector<std::string> mine;
    for (long long off = 0, piece = 0; off < bytes; off += kPiece, ++piece) {
      const long long n = std::min(kPiece, bytes - off);
      const std::string k = key(seq, root, piece);
      if (rank_ == root) {
        if (pda_store_set(store_, k.c_str(), static_cast<const char*>(buf) + off, (int)n) != 0) return false;
        mine.push_back(k);
      } else if (pda_store_get(store_, k.c_str(), static_cast<char*>(buf) + off, n, timeout_ms_) != n) {
        return false;
      }
    }
    // the root may only delete once every peer has read: the one-byte exchange below is that acknowledgement
    char ack = 1;
    std::vector<char> acks((size_t)world_);
    if (!all_gather(&ack, acks.data(), 1)) return false;
    for (auto& k : mine) pda_store_delete(store_, k.c_str());
    return true;
  }
  const char* what() const override { return "store"; }
  // a stored value carries its own length: one exchange instead of two
  bool gather_strings(const std::string& mine, std::vector<std::string>* all, int world) override {
    const long long seq = ++seq_;
    const std::string k = key(seq, rank_, 0);
    if (pda_store_set(store_, k.c_str(), mine.data(), (int)mine.size()) != 0) return false;
    all->assign((size_t)world, std::string());
    std::vector<char> buf(1 << 16);
    for (int r = 0; r < world; ++r) {
      if (r == rank_) {
        (*all)[(size_t)r] = mine;
        continue;
      }
      const std::string kr = key(seq, r, 0);
      long long n = pda_store_get(store_, kr.c_str(), buf.data(), (long long)buf.size(), timeout_ms_);
      if (n > (long long)buf.size()) {
        buf.resize((size_t)n);
        n = pda_store_get(store_, kr.c_str(), buf.data(), (long long)buf.size(), timeout_ms_);
      }
      if (n < 0) return false;
      (*all)[(size_t)r].assign(buf.data(), (size_t)n);
    }
    for (auto& p : prev_) pda_store_delete(store_, p.c_str());
    prev_.assign(1, k);
    return true;
  }

  void* store_;
  std::string prefix_;
  int rank_, world_;
  long long timeout_ms_;
  long long seq_ = 0;
  std::vector<std::string> prev_;
  std::vector<char> scratch_;
};

struct Request {
  long long handle;
  std::string name;
  int kind, dtype, op, root, device;
  void* ptr;
  void* ptr_out = nullptr;  // all-gather: world * count elements
  long long count;
  double prescale, postscale;
  bool announced = false;
};

struct Status {
  int state = 0;  // 0 pending, 1 done, -1 failed
  std::string error;
};

typedef int (*ExecFn)(int kind, int n, const long long* handles, int op, int root);

class Engine {
 public:
  Engine(Transport* tr, int rank, int world, long long fusion_bytes, double cycle_ms, ExecFn exec)
      : tr_(tr), rank_(rank), world_(world), fusion_bytes_(fusion_bytes), cycle_ms_(cycle_ms), exec_(exec) {
    fusion_.resize((size_t)std::max<long long>(fusion_bytes, 1 << 20));
    th_ = std::thread([this] { loop(); });
  }
  ~Engine() { shutdown(); }

  long long enqueue(const char* name, int kind, void* ptr, long long count, int dtype, int op, int root, int device,
                    double prescale, double postscale, void* ptr_out = nullptr) {
    std::lock_guard<std::mutex> g(mu_);
    Request r;
    r.handle = ++next_handle_;
    r.name = name;
    r.kind = kind; r.dtype = dtype; r.op = op; r.root = root; r.device = device;
    r.ptr = ptr; r.ptr_out = ptr_out; r.count = count; r.prescale = prescale; r.postscale = postscale;
    for (auto& q : queue_)
      if (q.name == r.name) {
        status_[r.handle].state = -1;
        status_[r.handle].error = "a request named '" + r.name + "' is already pending on this rank";
        cv_.notify_all();
        return r.handle;
      }
    status_[r.handle] = Status();
    queue_.push_back(r);
    cv_.notify_all();
    return r.handle;
  }
  int poll(long long h) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = status_.find(h);
    return it == status_.end() ? 1 : it->second.state;
  }
  int wait(long long h, char* err, int cap) {
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] {
      auto it = status_.find(h);
      return it == status_.end() || it->second.state != 0 || dead_;
    });
    auto it = status_.find(h);
    if (it == status_.end()) return 0;
    int st = it->second.state;
    if (st == 0 && dead_) {
      st = -1;
      it->second.error = "engine stopped before the request completed";
    }
    if (st < 0 && err && cap > 0) {
      strncpy(err, it->second.error.c_str(), (size_t)cap - 1);
      err[cap - 1] = 0;
    }
    status_.erase(it);
    return st < 0 ? -1 : 0;
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_requested_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  // python executor reports the fate of a device group
  void* request_ptr(long long h) { auto it = inflight_.find(h); return it == inflight_.end() ? nullptr : it->second.ptr; }
  long long request_count(long long h) { auto it = inflight_.find(h); return it == inflight_.end() ? 0 : it->second.count; }
  double request_postscale(long long h) { auto it = inflight_.find(h); return it == inflight_.end() ? 1.0 : it->second.postscale; }
  double request_prescale(long long h) { auto it = inflight_.find(h); return it == inflight_.end() ? 1.0 : it->second.prescale; }
  long long cycles() const { return cycles_.load(); }
  long long fused_groups() const { return groups_.load(); }

 private:
  static void put(std::string& s, const void* p, size_t n) { s.append(static_cast<const char*>(p), n); }

  bool gather_bytes(const std::string& mine, std::vector<std::string>* all) {
    return tr_->gather_strings(mine, all, world_);
  }

  struct Meta { int kind, dtype, op, root, device; long long count; };

  static void parse(const std::string& s, bool* stop, std::vector<std::pair<std::string, Meta>>* out) {
    size_t p = 0;
    *stop = s.size() > 0 && s[0] != 0;
    p = 1;
    while (p + 4 <= s.size()) {
      uint32_t nl;
      memcpy(&nl, s.data() + p, 4);
      p += 4;
      if (p + nl + sizeof(Meta) > s.size()) break;
      std::string name = s.substr(p, nl);
      p += nl;
      Meta m;
      memcpy(&m, s.data() + p, sizeof(Meta));
      p += sizeof(Meta);
      out->push_back({name, m});
    }
  }

  void finish(long long h, int state, const std::string& err = "") {
    std::lock_guard<std::mutex> g(mu_);
    auto it = status_.find(h);
    if (it != status_.end()) {
      it->second.state = state;
      it->second.error = err;
    }
    cv_.notify_all();
  }

  template <typename F> static void scale_buf(void* p, long long n, int dt, double s, F) {
    if (s == 1.0) return;
    if (dt == 0) { float* f = (float*)p; for (long long i = 0; i < n; ++i) f[i] = (float)(f[i] * s); }
    else if (dt == 4) { double* f = (double*)p; for (long long i = 0; i < n; ++i) f[i] *= s; }
    else if (dt == 5) { int32_t* f = (int32_t*)p; for (long long i = 0; i < n; ++i) f[i] = (int32_t)(f[i] * s); }
    else if (dt == 6) { int64_t* f = (int64_t*)p; for (long long i = 0; i < n; ++i) f[i] = (int64_t)(f[i] * s); }
  }

  void run_cpu_group(std::vector<Request>& grp) {
    const int dt = grp[0].dtype;
    const size_t es = dsize(dt);
    const int op = grp[0].op;  // H_AVG maps onto the shm backend's AVG (sum, then /world once)
    if (grp.size() == 1) {  // large or lonely tensor: reduce in place
      Request& r = grp[0];
      scale_buf(r.ptr, r.count, dt, r.prescale, 0);
      int rc = tr_->all_reduce(r.ptr, r.count, dt, op) ? 0 : -1;
      scale_buf(r.ptr, r.count, dt, r.postscale, 0);
      finish(r.handle, rc == 0 ? 1 : -1, rc == 0 ? "" : std::string(tr_->what()) + " all-reduce failed");
      return;
    }
    size_t off = 0;
    for (auto& r : grp) {
      memcpy(&fusion_[off], r.ptr, (size_t)r.count * es);
      scale_buf(&fusion_[off], r.count, dt, r.prescale, 0);
      off += (size_t)r.count * es;
    }
    int rc = tr_->all_reduce(fusion_.data(), (long long)(off / es), dt, op) ? 0 : -1;
    off = 0;
    for (auto& r : grp) {
      scale_buf(&fusion_[off], r.count, dt, r.postscale, 0);
      memcpy(r.ptr, &fusion_[off], (size_t)r.count * es);
      off += (size_t)r.count * es;
      finish(r.handle, rc == 0 ? 1 : -1, rc == 0 ? "" : std::string(tr_->what()) + " all-reduce failed");
    }
  }

  void run_device_group(std::vector<Request>& grp) {
    std::vector<long long> hs;
    for (auto& r : grp) {
      hs.push_back(r.handle);
      inflight_[r.handle] = r;
    }
    int rc = exec_ ? exec_(grp[0].kind, (int)hs.size(), hs.data(), grp[0].op, grp[0].root) : -1;
    for (auto& r : grp) {
      inflight_.erase(r.handle);
      finish(r.handle, rc == 0 ? 1 : -1, rc == 0 ? "" : "device executor failed (no executor registered?)");
    }
  }

  void execute(std::vector<Request>& ready) {
    size_t i = 0;
    while (i < ready.size()) {
      Request& r = ready[i];
      if (r.kind == K_ADASUM) {
        // device tensors only (the python side reduces host tensors itself): the executor runs the
        // halving / doubling exchange, one request at a time, in the negotiated order
        if (r.device == 0) {
          finish(r.handle, -1, "Adasum requests reach the engine for device tensors only");
        } else {
          std::vector<Request> g1{r};
          run_device_group(g1);
        }
        ++i;
        continue;
      }
      if (r.kind == K_ALLGATHER) {
        if (r.device == 0) {
          int rc = tr_->all_gather(r.ptr, r.ptr_out, r.count * (long long)dsize(r.dtype)) ? 0 : -1;
          finish(r.handle, rc == 0 ? 1 : -1, rc == 0 ? "" : std::string(tr_->what()) + " all-gather failed");
        } else {
          std::vector<Request> g1{r};
          run_device_group(g1);
        }
        ++i;
        continue;
      }
      if (r.kind == K_BROADCAST) {
        if (r.device == 0) {
          int rc = tr_->broadcast(r.ptr, r.count * (long long)dsize(r.dtype), r.root) ? 0 : -1;
          finish(r.handle, rc == 0 ? 1 : -1, rc == 0 ? "" : std::string(tr_->what()) + " broadcast failed");
        } else {
          std::vector<Request> g1{r};
          run_device_group(g1);
        }
        ++i;
        continue;
      }
      std::vector<Request> grp{r};
      long long bytes = r.count * (long long)dsize(r.dtype);
      size_t j = i + 1;
      while (j < ready.size()) {
        Request& n = ready[j];
        const long long nb = n.count * (long long)dsize(n.dtype);
        if (n.kind != K_ALLREDUCE || n.dtype != r.dtype || n.op != r.op || n.device != r.device ||
            n.prescale != r.prescale || n.postscale != r.postscale || bytes + nb > fusion_bytes_)
          break;
        grp.push_back(n);
        bytes += nb;
        ++j;
      }
      groups_.fetch_add(1);
      if (r.device == 0) run_cpu_group(grp);
      else run_device_group(grp);
      i = j;
    }
  }

  void loop() {
    for (;;) {
      {
        std::unique_lock<std::mutex> g(mu_);
        // system_clock deadline -> pthread_cond_timedwait, which ThreadSanitizer models (the
        // steady-clock wait_for maps to pthread_cond_clockwait, which older TSan runtimes do not)
        cv_.wait_until(g, std::chrono::system_clock::now() + std::chrono::microseconds((long long)(cycle_ms_ * 1000)),
                       [&] { return stop_requested_ || !queue_.empty(); });
      }
      cycles_.fetch_add(1);
      std::string mine;
      std::vector<Request> snapshot;
      {
        std::lock_guard<std::mutex> g(mu_);
        mine.push_back(stop_requested_ ? 1 : 0);
        for (auto& r : queue_) {
          uint32_t nl = (uint32_t)r.name.size();
          put(mine, &nl, 4);
          mine.append(r.name);
          Meta m;
          memset(&m, 0, sizeof(m));  // padding bytes take part in the cross-rank memcmp
          m.kind = r.kind; m.dtype = r.dtype; m.op = r.op; m.root = r.root; m.device = r.device; m.count = r.count;
          put(mine, &m, sizeof(m));
        }
        snapshot.assign(queue_.begin(), queue_.end());
      }
      std::vector<std::string> all;
      if (!gather_bytes(mine, &all)) {
        fail_everything("control-plane exchange failed (a peer died or timed out)");
        return;
      }
      bool all_stop = true, any_stop = false;
      std::vector<std::vector<std::pair<std::string, Meta>>> lists((size_t)world_);
      for (int r = 0; r < world_; ++r) {
        bool st = false;
        parse(all[(size_t)r], &st, &lists[(size_t)r]);
        all_stop = all_stop && st;
        any_stop = any_stop || st;
      }
      // ready = announced by everybody; order = rank 0's
      std::vector<Request> ready;
      for (auto& ent : lists[0]) {
        bool everywhere = true, consistent = true;
        for (int r = 1; r < world_ && everywhere; ++r) {
          bool found = false;
          for (auto& e2 : lists[(size_t)r])
            if (e2.first == ent.first) {
              found = true;
              if (memcmp(&e2.second, &ent.second, sizeof(Meta)) != 0) consistent = false;
              break;
            }
          everywhere = found;
        }
        if (!everywhere) continue;
        for (auto& rq : snapshot)
          if (rq.name == ent.first) {
            if (!consistent) {
              remove_from_queue(rq.handle);
              finish(rq.handle, -1, "tensor '" + rq.name + "' was submitted with different type/shape/op on different ranks");
            } else {
              ready.push_back(rq);
              remove_from_queue(rq.handle);
            }
            break;
          }
      }
      if (!ready.empty()) execute(ready);
      if (all_stop || (any_stop && ready.empty() && all_queues_empty(lists))) {
        std::lock_guard<std::mutex> g(mu_);
        dead_ = true;
        cv_.notify_all();
        return;
      }
    }
  }
  static bool all_queues_empty(const std::vector<std::vector<std::pair<std::string, Meta>>>& l) {
    for (auto& v : l)
      if (!v.empty()) return false;
    return true;
  }
  void remove_from_queue(long long h) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto it = queue_.begin(); it != queue_.end(); ++it)
      if (it->handle == h) {
        queue_.erase(it);
        return;
      }
  }
  void fail_everything(const std::string& why) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& r : queue_) {
      status_[r.handle].state = -1;
      status_[r.handle].error = why;
    }
    queue_.clear();
    dead_ = true;
    cv_.notify_all();
  }

  std::unique_ptr<Transport> tr_;
  int rank_, world_;
  long long fusion_bytes_;
  double cycle_ms_;
  ExecFn exec_;
  std::vector<char> fusion_;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Request> queue_;
  std::map<long long, Status> status_;
  std::map<long long, Request> inflight_;
  long long next_handle_ = 0;
  bool stop_requested_ = false, dead_ = false;
  std::atomic<long long> cycles_{0}, groups_{0};
};

}  // namespace

PDA_EXPORT void* pda_hvd_create(void* shm, int rank, int world, long long fusion_bytes, double cycle_ms, void* exec) {
  return new Engine(new ShmTransport(shm), rank, world, fusion_bytes, cycle_ms, reinterpret_cast<ExecFn>(exec));
}
// The same engine for ranks on several nodes: negotiation and host tensors through the key-value store `store` (a
// client handle of store.cc that the engine thread has to itself), every key under `prefix`.
PDA_EXPORT void* pda_hvd_create_store(void* store, const char* prefix, int rank, int world, long long fusion_bytes,
                                      double cycle_ms, void* exec, long long timeout_ms) {
  return new Engine(new StoreTransport(store, prefix, rank, world, timeout_ms), rank, world, fusion_bytes, cycle_ms,
                    reinterpret_cast<ExecFn>(exec));
}
PDA_EXPORT void pda_hvd_shutdown(void* h) { static_cast<Engine*>(h)->shutdown(); }
PDA_EXPORT void pda_hvd_destroy(void* h) { delete static_cast<Engine*>(h); }
PDA_EXPORT long long pda_hvd_enqueue(void* h, const char* name, int kind, void* ptr, long long count, int dtype, int op,
                                     int root, int device, double prescale, double postscale) {
  return static_cast<Engine*>(h)->enqueue(name, kind, ptr, count, dtype, op, root, device, prescale, postscale);
}
PDA_EXPORT long long pda_hvd_enqueue2(void* h, const char* name, int kind, void* ptr, void* ptr_out, long long count,
                                      int dtype, int op, int root, int device, double prescale, double postscale) {
  return static_cast<Engine*>(h)->enqueue(name, kind, ptr, count, dtype, op, root, device, prescale, postscale, ptr_out);
}
PDA_EXPORT int pda_hvd_poll(void* h, long long handle) { return static_cast<Engine*>(h)->poll(handle); }
PDA_EXPORT int pda_hvd_wait(void* h, long long handle, char* err, int cap) {
  return static_cast<Engine*>(h)->wait(handle, err, cap);
}
PDA_EXPORT void* pda_hvd_request_ptr(void* h, long long handle) { return static_cast<Engine*>(h)->request_ptr(handle); }
PDA_EXPORT long long pda_hvd_request_count(void* h, long long handle) { return static_cast<Engine*>(h)->request_count(handle); }
PDA_EXPORT double pda_hvd_request_prescale(void* h, long long handle) { return static_cast<Engine*>(h)->request_prescale(handle); }
PDA_EXPORT double pda_hvd_request_postscale(void* h, long long handle) { return static_cast<Engine*>(h)->request_postscale(handle); }
PDA_EXPORT long long pda_hvd_cycles(void* h) { return static_cast<Engine*>(h)->cycles(); }
PDA_EXPORT long long pda_hvd_fused_groups(void* h) { return static_cast<Engine*>(h)->fused_groups(); }
