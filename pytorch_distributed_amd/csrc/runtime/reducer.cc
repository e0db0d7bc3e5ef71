// Gradient-bucket reducer core: the bookkeeping half of
// DistributedDataParallel, i.e. what torch's C++ Reducer does behind
// `DistributedDataParallel(model, device_ids=[rank])` and `loss.backward()`
// (/root/reference/2.distributed.py:114,231), and apex DDP's arrival-order flat
// buckets (/root/reference/4.apex_distributed2.py:178).
//
//  * plan: parameters are assigned to buckets in a given order (reverse
//    registration order at first, gradient-arrival order after a rebuild) until a
//    byte cap is reached; the first bucket has its own smaller cap so
//    communication starts early; every parameter offset is aligned so multi-tensor
//    kernels and RCCL see 16-byte aligned sub-buffers, and every bucket is padded
//    to a multiple of 8 ranks x 1 KiB so its 8 shards stay vector-aligned on a
//    fully-connected 8-GPU xGMI node.
//  * run: mark_ready(param) decrements the bucket's pending count; buckets become
//    launchable strictly in bucket order (all ranks must issue collectives in the
//    same order), so a bucket that fills early waits for its predecessors.
//  * arrival order of the first backward is recorded for the rebuild.
// The collectives themselves are issued by python (parallel/distributed.py) on the
// RCCL process group; this object only decides what is ready when.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#define PDA_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct Bucket {
  std::vector<int> params;
  long long bytes = 0;   // padded size
  int pending = 0;
  bool launched = false;
};

class Reducer {
 public:
  Reducer(int n, const long long* sizes, long long cap, long long first_cap, int align, int pad)
      : sizes_(sizes, sizes + n), cap_(cap), first_cap_(first_cap), align_(align > 0 ? align : 16),
        pad_(pad > 0 ? pad : 16) {
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = n - 1 - i;  // reverse registration order
    plan(order);
  }

  void plan(const std::vector<int>& order) {
    std::lock_guard<std::mutex> g(mu_);
    buckets_.clear();
    bucket_of_.assign(sizes_.size(), -1);
    offset_of_.assign(sizes_.size(), 0);
    Bucket cur;
    auto close = [&] {
      if (cur.params.empty()) return;
      cur.bytes = round_up(cur.bytes, pad_);
      buckets_.push_back(cur);
      cur = Bucket();
    };
    for (int p : order) {
      const long long sz = round_up(sizes_[p], align_);
      bucket_of_[p] = (int)buckets_.size();
      offset_of_[p] = cur.bytes;
      cur.params.push_back(p);
      cur.bytes += sz;
      const long long cap = buckets_.empty() ? first_cap_ : cap_;
      if (cur.bytes >= cap) close();  // a bucket closes once it reaches/exceeds its cap
    }
    close();
    reset_locked();
    arrival_.clear();
  }

  void reset() {
    std::lock_guard<std::mutex> g(mu_);
    reset_locked();
  }

  // Marks param ready; writes the ids of buckets that became launchable (in order) to out.
  int mark_ready(int p, int* out, int max_out) {
    std::lock_guard<std::mutex> g(mu_);
    if (p < 0 || p >= (int)sizes_.size() || ready_[p]) return 0;
    ready_[p] = 1;
    if (record_arrival_) arrival_.push_back(p);
    Bucket& b = buckets_[bucket_of_[p]];
    --b.pending;
    int n = 0;
    while (next_launch_ < (int)buckets_.size() && buckets_[next_launch_].pending == 0 && n < max_out) {
      buckets_[next_launch_].launched = true;
      out[n++] = next_launch_++;
    }
    return n;
  }

  // Parameters that never produced a gradient this pass (unused parameters): mark them ready so
  // the remaining buckets can launch; returns launchable buckets like mark_ready.
  int flush_unready(int* out, int max_out) {
    std::lock_guard<std::mutex> g(mu_);
    for (size_t p = 0; p < ready_.size(); ++p)
      if (!ready_[p]) {
        ready_[p] = 1;
        --buckets_[bucket_of_[p]].pending;
      }
    int n = 0;
    while (next_launch_ < (int)buckets_.size() && buckets_[next_launch_].pending == 0 && n < max_out) {
      buckets_[next_launch_].launched = true;
      out[n++] = next_launch_++;
    }
    return n;
  }

  int num_unready() {
    std::lock_guard<std::mutex> g(mu_);
    int n = 0;
    for (char r : ready_) n += r ? 0 : 1;
    return n;
  }

  int num_buckets() const { return (int)buckets_.size(); }
  long long bucket_bytes(int b) const { return buckets_[b].bytes; }
  int bucket_num_params(int b) const { return (int)buckets_[b].params.size(); }
  int bucket_params(int b, int* out, int max_out) const {
    const int n = std::min<int>(max_out, (int)buckets_[b].params.size());
    memcpy(out, buckets_[b].params.data(), sizeof(int) * (size_t)n);
    return n;
  }
  int bucket_of(int p) const { return bucket_of_[p]; }
  long long offset_of(int p) const { return offset_of_[p]; }
  void set_record_arrival(bool on) { record_arrival_ = on; }
  int arrival(int* out, int max_out) {
    std::lock_guard<std::mutex> g(mu_);
    const int n = std::min<int>(max_out, (int)arrival_.size());
    memcpy(out, arrival_.data(), sizeof(int) * (size_t)n);
    return n;
  }
  // Re-plan in the given order (missing params are appended in reverse registration order).
  void rebuild(const int* order, int n) {
    std::vector<int> ord(order, order + n);
    std::vector<char> seen(sizes_.size(), 0);
    std::vector<int> clean;
    for (int p : ord)
      if (p >= 0 && p < (int)sizes_.size() && !seen[p]) {
        seen[p] = 1;
        clean.push_back(p);
      }
    for (int p = (int)sizes_.size() - 1; p >= 0; --p)
      if (!seen[p]) clean.push_back(p);
    plan(clean);
  }

 private:
  static long long round_up(long long v, long long a) { return (v + a - 1) / a * a; }
  void reset_locked() {
    ready_.assign(sizes_.size(), 0);
    for (auto& b : buckets_) {
      b.pending = (int)b.params.size();
      b.launched = false;
    }
    next_launch_ = 0;
  }

  std::vector<long long> sizes_;
  long long cap_, first_cap_;
  int align_, pad_;
  std::vector<Bucket> buckets_;
  std::vector<int> bucket_of_;
  std::vector<long long> offset_of_;
  std::vector<char> ready_;
  std::vector<int> arrival_;
  bool record_arrival_ = true;
  int next_launch_ = 0;
  std::mutex mu_;
};

}  // namespace

PDA_EXPORT void* pda_reducer_create(int n, const long long* sizes_bytes, long long cap_bytes, long long first_cap_bytes,
                                    int align, int pad) {
  return new Reducer(n, sizes_bytes, cap_bytes, first_cap_bytes, align, pad);
}
PDA_EXPORT void pda_reducer_destroy(void* h) { delete static_cast<Reducer*>(h); }
PDA_EXPORT int pda_reducer_num_buckets(void* h) { return static_cast<Reducer*>(h)->num_buckets(); }
PDA_EXPORT long long pda_reducer_bucket_bytes(void* h, int b) { return static_cast<Reducer*>(h)->bucket_bytes(b); }
PDA_EXPORT int pda_reducer_bucket_params(void* h, int b, int* out, int max_out) {
  return static_cast<Reducer*>(h)->bucket_params(b, out, max_out);
}
PDA_EXPORT int pda_reducer_bucket_num_params(void* h, int b) { return static_cast<Reducer*>(h)->bucket_num_params(b); }
PDA_EXPORT int pda_reducer_bucket_of(void* h, int p) { return static_cast<Reducer*>(h)->bucket_of(p); }
PDA_EXPORT long long pda_reducer_offset_of(void* h, int p) { return static_cast<Reducer*>(h)->offset_of(p); }
PDA_EXPORT void pda_reducer_reset(void* h) { static_cast<Reducer*>(h)->reset(); }
PDA_EXPORT int pda_reducer_mark_ready(void* h, int p, int* out, int max_out) {
  return static_cast<Reducer*>(h)->mark_ready(p, out, max_out);
}
PDA_EXPORT int pda_reducer_flush_unready(void* h, int* out, int max_out) {
  return static_cast<Reducer*>(h)->flush_unready(out, max_out);
}
PDA_EXPORT int pda_reducer_num_unready(void* h) { return static_cast<Reducer*>(h)->num_unready(); }
PDA_EXPORT void pda_reducer_record_arrival(void* h, int on) { static_cast<Reducer*>(h)->set_record_arrival(on != 0); }
PDA_EXPORT int pda_reducer_arrival(void* h, int* out, int max_out) { return static_cast<Reducer*>(h)->arrival(out, max_out); }
PDA_EXPORT void pda_reducer_rebuild(void* h, const int* order, int n) { static_cast<Reducer*>(h)->rebuild(order, n); }
