// Shared-memory CPU collective backend: all-reduce / broadcast /
// all-gather / barrier between the processes of one host through a POSIX shm
// segment.  It is the data plane of the Horovod-style API on CPU tensors
// (`--no-cuda`, /root/reference/5.2.horovod_pytorch_mnist.py:22,75-78,112), the
// control plane of its negotiation, and the "fake backend" of the CPU test-suite.
//
// Segment layout:  Header | result area [slot_bytes] | W rank slots [slot_bytes each]
// all_reduce per chunk: every rank copies its chunk into its slot -> barrier ->
// rank r reduces the r-th 1/W of the chunk over all W slots into the result area
// (a direct reduce-scatter: all ranks work in parallel on disjoint ranges) ->
// barrier -> everyone copies the result out (the all-gather half) -> barrier.
// Barriers are sense-reversing on C++11 atomics in the segment, with a timeout so
// a dead peer turns into an error instead of a hang.
// Every communicator owns a worker thread with a FIFO queue: *_async calls return a
// ticket, pda_shm_wait(ticket) blocks; sync calls are enqueue+wait.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define PDA_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

enum DType : int { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_U8 = 3, DT_F64 = 4, DT_I32 = 5, DT_I64 = 6 };
enum RedOp : int { OP_SUM = 0, OP_PROD = 1, OP_MIN = 2, OP_MAX = 3, OP_AVG = 4 };

size_t dtype_size(int dt) {
  switch (dt) {
    case DT_F32: case DT_I32: return 4;
    case DT_F16: case DT_BF16: return 2;
    case DT_U8: return 1;
    default: return 8;
  }
}

inline float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, out;
  if (exp == 0) {
    if (man == 0) out = sign;
    else {
      exp = 127 - 15 + 1;
      while (!(man & 0x400)) { man <<= 1; --exp; }
      man &= 0x3ff;
      out = sign | (exp << 23) | (man << 13);
    }
  } else if (exp == 31) out = sign | 0x7f800000u | (man << 13);
  else out = sign | ((exp + 127 - 15) << 23) | (man << 13);
  float f;
  memcpy(&f, &out, 4);
  return f;
}
inline uint16_t float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000;
  int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffff;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00 | (man ? 0x200 : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000;
    const uint32_t shift = (uint32_t)(14 - exp);
    uint32_t h = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
  const uint32_t rem = man & 0x1fff;
  if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}
inline float bf16_to_float(uint16_t b) {
  uint32_t x = (uint32_t)b << 16;
  float f;
  memcpy(&f, &x, 4);
  return f;
}
inline uint16_t float_to_bf16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffff) > 0x7f800000) return (uint16_t)((x >> 16) | 0x40);  // keep NaN a NaN
  x += 0x7fff + ((x >> 16) & 1);
  return (uint16_t)(x >> 16);
}

template <typename T> inline T red(T a, T b, int op) {
  switch (op) {
    case OP_PROD: return a * b;
    case OP_MIN: return b < a ? b : a;
    case OP_MAX: return b > a ? b : a;
    default: return a + b;
  }
}

struct Header {
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> attached;
  std::atomic<uint32_t> bar_count;
  std::atomic<uint32_t> bar_sense;
  std::atomic<uint32_t> abort_flag;
  uint32_t world;
  uint64_t slot_bytes;
};
constexpr size_t kHeaderBytes = 256;
constexpr uint32_t kMagic = 0x70646173;  // "pdas"

struct Job {
  std::function<int()> fn;
  long long ticket;
};

class ShmComm {
 public:
  ShmComm(const std::string& name, int rank, int world, size_t slot_bytes, long long timeout_ms)
      : name_(name), rank_(rank), world_(world), slot_(slot_bytes), timeout_ms_(timeout_ms) {}
  ~ShmComm() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
    if (base_) ::munmap(base_, total_);
    if (rank_ == 0) ::shm_unlink(name_.c_str());
  }

  bool open() {
    total_ = kHeaderBytes + slot_ * (size_t)(world_ + 1);
    int fd = -1;
    if (rank_ == 0) {
      ::shm_unlink(name_.c_str());
      fd = ::shm_open(name_.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) return false;
      if (::ftruncate(fd, (off_t)total_) != 0) { ::close(fd); return false; }
    } else {
      auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms_);
      while (std::chrono::steady_clock::now() < deadline) {
        fd = ::shm_open(name_.c_str(), O_RDWR, 0600);
        if (fd >= 0) {
          struct stat st{};
          if (::fstat(fd, &st) == 0 && (size_t)st.st_size >= total_) break;
          ::close(fd);
          fd = -1;
        }
        ::usleep(1000);
      }
      if (fd < 0) return false;
    }
    base_ = static_cast<char*>(::mmap(nullptr, total_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    ::close(fd);
    if (base_ == MAP_FAILED) { base_ = nullptr; return false; }
    hdr_ = reinterpret_cast<Header*>(base_);
    if (rank_ == 0) {
      hdr_->world = (uint32_t)world_;
      hdr_->slot_bytes = slot_;
      hdr_->bar_count.store(0);
      hdr_->bar_sense.store(0);
      hdr_->abort_flag.store(0);
      hdr_->attached.store(0);
      hdr_->magic.store(kMagic, std::memory_order_release);
    } else {
      auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms_);
      while (hdr_->magic.load(std::memory_order_acquire) != kMagic) {
        if (std::chrono::steady_clock::now() > deadline) return false;
        ::usleep(200);
      }
      if ((int)hdr_->world != world_ || hdr_->slot_bytes != slot_) return false;
    }
    hdr_->attached.fetch_add(1);
    worker_ = std::thread([this] { run(); });
    return true;
  }

  char* result() { return base_ + kHeaderBytes; }
  char* slot(int r) { return base_ + kHeaderBytes + slot_ * (size_t)(r + 1); }

  // 0 ok, -1 timeout / aborted
  int barrier() {
    if (world_ == 1) return 0;
    const uint32_t my_sense = local_sense_ ^= 1u;
    if (hdr_->bar_count.fetch_add(1, std::memory_order_acq_rel) == (uint32_t)world_ - 1) {
      hdr_->bar_count.store(0, std::memory_order_relaxed);
      hdr_->bar_sense.store(my_sense, std::memory_order_release);
      return 0;
    }
    auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms_);
    int spins = 0;
    while (hdr_->bar_sense.load(std::memory_order_acquire) != my_sense) {
      if (hdr_->abort_flag.load(std::memory_order_relaxed)) return -1;
      if (++spins > 2000) {
        ::usleep(50);
        if (std::chrono::steady_clock::now() > deadline) {
          hdr_->abort_flag.store(1);
          return -1;
        }
      } else if (spins > 200) {
        std::this_thread::yield();
      }
    }
    return 0;
  }

  void reduce_range(int dt, int op, size_t lo, size_t hi) {
    // result[lo:hi] = reduce over ranks of slot[r][lo:hi]   (element indices)
    auto loop = [&](auto load, auto store, auto zero) {
      using A = decltype(zero);
      for (size_t i = lo; i < hi; ++i) {
        A acc = load(slot(0), i);
        for (int r = 1; r < world_; ++r) acc = red<A>(acc, load(slot(r), i), op);
        if (op == OP_AVG) acc = (A)(acc / (A)world_);
        store(result(), i, acc);
      }
    };
    switch (dt) {
      case DT_F32: loop([](char* b, size_t i) { return reinterpret_cast<float*>(b)[i]; },
                        [](char* b, size_t i, float v) { reinterpret_cast<float*>(b)[i] = v; }, 0.0f); break;
      case DT_F64: loop([](char* b, size_t i) { return reinterpret_cast<double*>(b)[i]; },
                        [](char* b, size_t i, double v) { reinterpret_cast<double*>(b)[i] = v; }, 0.0); break;
      case DT_I32: loop([](char* b, size_t i) { return (int64_t) reinterpret_cast<int32_t*>(b)[i]; },
                        [](char* b, size_t i, int64_t v) { reinterpret_cast<int32_t*>(b)[i] = (int32_t)v; }, (int64_t)0); break;
      case DT_I64: loop([](char* b, size_t i) { return reinterpret_cast<int64_t*>(b)[i]; },
                        [](char* b, size_t i, int64_t v) { reinterpret_cast<int64_t*>(b)[i] = v; }, (int64_t)0); break;
      case DT_U8: loop([](char* b, size_t i) { return (int64_t) reinterpret_cast<uint8_t*>(b)[i]; },
                       [](char* b, size_t i, int64_t v) { reinterpret_cast<uint8_t*>(b)[i] = (uint8_t)v; }, (int64_t)0); break;
      case DT_F16: loop([](char* b, size_t i) { return half_to_float(reinterpret_cast<uint16_t*>(b)[i]); },
                        [](char* b, size_t i, float v) { reinterpret_cast<uint16_t*>(b)[i] = float_to_half(v); }, 0.0f); break;
      case DT_BF16: loop([](char* b, size_t i) { return bf16_to_float(reinterpret_cast<uint16_t*>(b)[i]); },
                         [](char* b, size_t i, float v) { reinterpret_cast<uint16_t*>(b)[i] = float_to_bf16(v); }, 0.0f); break;
    }
  }

  int all_reduce(void* buf, long long count, int dt, int op) {
    const size_t es = dtype_size(dt);
    if (world_ == 1) return 0;
    const size_t per = slot_ / es;
    char* p = static_cast<char*>(buf);
    for (long long done = 0; done < count;) {
      const size_t n = (size_t)std::min<long long>((long long)per, count - done);
      memcpy(slot(rank_), p + (size_t)done * es, n * es);
      if (barrier()) return -1;
      const size_t lo = n * (size_t)rank_ / (size_t)world_, hi = n * (size_t)(rank_ + 1) / (size_t)world_;
      reduce_range(dt, op, lo, hi);
      if (barrier()) return -1;
      memcpy(p + (size_t)done * es, result(), n * es);
      if (barrier()) return -1;
      done += (long long)n;
    }
    return 0;
  }

  int broadcast(void* buf, long long bytes, int root) {
    if (world_ == 1) return 0;
    char* p = static_cast<char*>(buf);
    for (long long done = 0; done < bytes;) {
      const size_t n = (size_t)std::min<long long>((long long)slot_, bytes - done);
      if (rank_ == root) memcpy(result(), p + done, n);
      if (barrier()) return -1;
      if (rank_ != root) memcpy(p + done, result(), n);
      if (barrier()) return -1;
      done += (long long)n;
    }
    return 0;
  }

  // out = concat over ranks of in (bytes each)
  int all_gather(const void* in, void* out, long long bytes) {
    const char* pi = static_cast<const char*>(in);
    char* po = static_cast<char*>(out);
    if (world_ == 1) {
      memcpy(po, pi, (size_t)bytes);
      return 0;
    }
    for (long long done = 0; done < bytes || (bytes == 0 && done == 0);) {
      const size_t n = (size_t)std::min<long long>((long long)slot_, bytes - done);
      memcpy(slot(rank_), pi + done, n);
      if (barrier()) return -1;
      for (int r = 0; r < world_; ++r) memcpy(po + (size_t)r * (size_t)bytes + (size_t)done, slot(r), n);
      if (barrier()) return -1;
      done += (long long)n;
      if (bytes == 0) break;
    }
    return 0;
  }

  long long enqueue(std::function<int()> fn) {
    std::lock_guard<std::mutex> g(mu_);
    const long long t = ++next_ticket_;
    q_.push_back({std::move(fn), t});
    cv_.notify_all();
    return t;
  }
  int wait(long long ticket) {
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] { return done_ticket_ >= ticket || stop_; });
    auto it = errors_.find(ticket);
    if (it != errors_.end()) {
      int e = it->second;
      errors_.erase(it);
      return e;
    }
    return 0;
  }
  int poll(long long ticket) {
    std::lock_guard<std::mutex> g(mu_);
    return done_ticket_ >= ticket ? 1 : 0;
  }
  void abort() {
    if (hdr_) hdr_->abort_flag.store(1);
  }
  int rank() const { return rank_; }
  int world() const { return world_; }

 private:
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        j = std::move(q_.front());
        q_.pop_front();
      }
      const int rc = j.fn();
      {
        std::lock_guard<std::mutex> g(mu_);
        if (rc != 0) errors_[j.ticket] = rc;
        done_ticket_ = j.ticket;
      }
      cv_.notify_all();
    }
  }

  std::string name_;
  int rank_, world_;
  size_t slot_;
  long long timeout_ms_;
  size_t total_ = 0;
  char* base_ = nullptr;
  Header* hdr_ = nullptr;
  uint32_t local_sense_ = 0;
  std::thread worker_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Job> q_;
  long long next_ticket_ = 0, done_ticket_ = 0;
  std::map<long long, int> errors_;
  bool stop_ = false;
};

}  // namespace

PDA_EXPORT void* pda_shm_create(const char* name, int rank, int world, long long slot_bytes, long long timeout_ms) {
  auto* c = new ShmComm(name, rank, world, (size_t)slot_bytes, timeout_ms);
  if (!c->open()) {
    delete c;
    return nullptr;
  }
  return c;
}
PDA_EXPORT void pda_shm_destroy(void* h) { delete static_cast<ShmComm*>(h); }
PDA_EXPORT long long pda_shm_all_reduce_async(void* h, void* buf, long long count, int dtype, int op) {
  auto* c = static_cast<ShmComm*>(h);
  return c->enqueue([=] { return c->all_reduce(buf, count, dtype, op); });
}
PDA_EXPORT long long pda_shm_broadcast_async(void* h, void* buf, long long bytes, int root) {
  auto* c = static_cast<ShmComm*>(h);
  return c->enqueue([=] { return c->broadcast(buf, bytes, root); });
}
PDA_EXPORT long long pda_shm_all_gather_async(void* h, const void* in, void* out, long long bytes) {
  auto* c = static_cast<ShmComm*>(h);
  return c->enqueue([=] { return c->all_gather(in, out, bytes); });
}
PDA_EXPORT long long pda_shm_barrier_async(void* h) {
  auto* c = static_cast<ShmComm*>(h);
  return c->enqueue([=] { return c->barrier(); });
}
PDA_EXPORT int pda_shm_wait(void* h, long long ticket) { return static_cast<ShmComm*>(h)->wait(ticket); }
PDA_EXPORT int pda_shm_poll(void* h, long long ticket) { return static_cast<ShmComm*>(h)->poll(ticket); }
PDA_EXPORT void pda_shm_abort(void* h) { static_cast<ShmComm*>(h)->abort(); }
